"""Report of a rocprofv3 --hsa-trace --hip-trace database: every hsa_queue_create (and scratch / memory-pool allocation of 8 MiB
and more is not visible here -- only the API calls), its duration, the thread, and the HIP API call that encloses it on that thread,
relative to the first k_permtest_bits_blk launch of each step.  usage: queue_trace_report.py <db>"""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
cur = db.cursor()
rows = cur.execute("select name, category, tid, start, end from regions where name like 'hsa_queue_create%' or name like 'hsa_amd_queue%' "
                   "or name like 'hsa_queue_destroy%' order by start").fetchall()
t0 = cur.execute("select min(start) from regions").fetchone()[0]
print('%d queue calls; trace starts at t0' % len(rows))
steps = [r[0] for r in cur.execute("select start from regions where name = 'hipLaunchKernel' or name = 'hipModuleLaunchKernel' order by start limit 1")]
# the launches of the headline kernel: hipLaunchKernel calls are too many to tell apart by name; use the kernel dispatch table
k = cur.execute("select start from kernels where name like '%k_bits_observed%' order by start").fetchall() if cur.execute(
    "select count(*) from sqlite_master where name = 'kernels'").fetchone()[0] else []
step_starts = [r[0] for r in k]
for name, cat, tid, s, e in rows:
    hip = cur.execute("select name, start, end from regions where tid = ? and category like 'HIP%' and start <= ? and end >= ? order by start desc limit 1",
                      (tid, s, e)).fetchone()
    step = sum(1 for x in step_starts if x <= s)
    print('%-28s tid %d at %9.3f ms (%7.3f ms long) inside %s; after %d steps began' %
          (name, tid, 1e-6 * (s - t0), 1e-6 * (e - s), hip[0] if hip else '-', step))
print('steps seen (k_bits_observed dispatches):', len(step_starts))
# long HIP calls of the launching thread during the steps
if step_starts:
    longc = cur.execute("select name, tid, start, end from regions where category like 'HIP%' and start >= ? and (end - start) > 2000000 order by start",
                        (step_starts[0],)).fetchall()
    for name, tid, s, e in longc[:20]:
        print('long HIP call %-32s tid %d at %9.3f ms: %.3f ms' % (name, tid, 1e-6 * (s - t0), 1e-6 * (e - s)))
