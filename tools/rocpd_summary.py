#!/usr/bin/env python3
"""Summarise a rocprofv3 rocpd database (ROCm 7.2 default output) as a per-kernel table:
calls, total / average / min / max duration.  Used to produce profiles/*.txt."""
import sqlite3
import sys


def main(path):
    con = sqlite3.connect(path)
    cols = [r[1] for r in con.execute("pragma table_info(kernels)")]
    name_col = 'name' if 'name' in cols else 'kernel_name'
    rows = con.execute("select %s, start, end from kernels" % name_col).fetchall()
    agg = {}
    for name, start, end in rows:
        a = agg.setdefault(name, [0, 0, None, 0])
        d = end - start
        a[0] += 1
        a[1] += d
        a[2] = d if a[2] is None else min(a[2], d)
        a[3] = max(a[3], d)
    total = sum(a[1] for a in agg.values()) or 1
    print('%-72s %7s %14s %12s %12s %12s %7s' % ('kernel', 'calls', 'total_ns', 'avg_ns', 'min_ns', 'max_ns', '%'))
    for name, a in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        print('%-72s %7d %14d %12d %12d %12d %6.2f' % (name[:72], a[0], a[1], a[1] // a[0], a[2], a[3], 100.0 * a[1] / total))


if __name__ == '__main__':
    main(sys.argv[1])
