#!/bin/bash
# What is the 10-15 ms step that one 5 + 20 run in three to ten contains?  Per run: the slowest step, its minor faults, the change of the
# resident set inside it, the thread that took the faults, and (SAFE_BENCH_SMAPS=2) the largest mappings after the timed region.
echo "numa_balancing: $(cat /proc/sys/kernel/numa_balancing 2>/dev/null)  thp: $(cat /sys/kernel/mm/transparent_hugepage/enabled 2>/dev/null)"
python3 - <<'PY'
import ctypes, os
libc = ctypes.CDLL(None, use_errno=True)
mask = ctypes.c_ulong(0)
mode = ctypes.c_int(0)
r = libc.syscall(239, ctypes.byref(mode), ctypes.byref(mask), 64, None, 0)     # get_mempolicy
print('get_mempolicy rc', r, 'errno', ctypes.get_errno(), 'mode', mode.value, 'mask', hex(mask.value))
PY
vm() { grep -E "^(numa_pte_updates|numa_hint_faults|numa_pages_migrated|pgmigrate_success|thp_fault_alloc|compact_stall|pgfault) " /proc/vmstat | tr '\n' ' '; echo; }
for i in 1 2 3 4 5; do
  echo "before: $(vm)"
  SAFE_BENCH_SMAPS=2 timeout 300 python3 bench.py --gpus 1 --steps 20 --warmup 5 --extras 0 --cpu-perms 0 2>/dev/null | python3 -c "
import sys, json
for ln in sys.stdin:
    if ln.startswith('{'):
        d = json.loads(ln); s = d['step_probe']['slowest_steps'][0]; print('mean %.3f' % d['ms_per_step'], 'slowest: step', s['step'], '%.2f ms' % s['ms'], 'faults', s['minor_faults'], 'tables_enq %.2f' % s['tables_enqueued_ms'], 'rss change', s.get('resident_pages_change'), d['step_probe'].get('minor_faults_by_thread'), 'total rss change', d['step_probe']['totals_over_timed_steps'].get('resident_pages_change'), [(m['rss_kb'] // 1024, m['mapping'].split()[1][-28:]) for m in d['step_probe'].get('mappings_largest_kb', [])])
"
  echo "after:  $(vm)"
done
