#!/usr/bin/env python3
"""Per-kernel PMC counter sums from a rocprofv3 rocpd database (counter collection run)."""
import sqlite3
import sys


def main(path, pattern=''):
    con = sqlite3.connect(path)
    cols = [r[1] for r in con.execute("pragma table_info(counters_collection)")]
    rows = con.execute("select * from counters_collection").fetchall()
    ix = {c: i for i, c in enumerate(cols)}
    name_col = 'kernel_name' if 'kernel_name' in ix else 'name'
    agg = {}
    for r in rows:
        kn = r[ix[name_col]]
        if pattern and pattern not in kn:
            continue
        key = (kn[:60], r[ix['counter_name']])
        a = agg.setdefault(key, [0.0, 0])
        a[0] += r[ix['value']]
        a[1] += 1
    for (kn, cn), (v, c) in sorted(agg.items()):
        print('%-60s %-28s sum=%.4g  n=%d' % (kn, cn, v, c))


if __name__ == '__main__':
    main(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else '')
