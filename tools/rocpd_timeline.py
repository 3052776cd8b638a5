#!/usr/bin/env python3
"""GPU timeline of the LAST headline step in a rocprofv3 kernel trace (rocpd database): every kernel between the last
k_attr_stats-like start of a step and its k_counts_finalize, start / end relative to the step's first kernel, in microseconds.
usage: rocpd_timeline.py <results.db> [first-kernel-substring [last-kernel-substring]]"""
import sqlite3
import sys


def main(path, first='k_bits_prep', last='k_counts_finalize'):
    con = sqlite3.connect(path)
    cols = [r[1] for r in con.execute("pragma table_info(kernels)")]
    name_col = 'name' if 'name' in cols else 'kernel_name'
    rows = con.execute("select %s, start, end from kernels order by start" % name_col).fetchall()
    starts = [i for i, r in enumerate(rows) if first in r[0]]
    if not starts:
        print('no kernel matching', first)
        return
    i0 = starts[-1]
    t0 = rows[i0][1]
    busy_end = t0
    for name, s, e in rows[i0:]:
        gap = (s - busy_end) / 1e3
        print('%9.1f %9.1f  %8.1f us  gap %7.1f  %s' % ((s - t0) / 1e3, (e - t0) / 1e3, (e - s) / 1e3, gap, name[:70]))
        busy_end = max(busy_end, e)
        if last in name:
            break


if __name__ == '__main__':
    main(*sys.argv[1:])
