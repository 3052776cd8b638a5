#!/usr/bin/env python3
"""Print the kernel/copy timeline (ms, relative) of a rocprofv3 rocpd database."""
import sqlite3, sys
con = sqlite3.connect(sys.argv[1])
skip = int(sys.argv[2]) if len(sys.argv) > 2 else 0
rows = con.execute("select name, start, end from kernels order by start").fetchall()
try:
    cp = con.execute("select name, start, end from memory_copies order by start").fetchall()
except Exception:
    cp = []
ev = sorted([(s, e, n[:48]) for n, s, e in rows] + [(s, e, 'COPY ' + str(n)[:40]) for n, s, e in cp])
ev = ev[skip:]
t0 = ev[0][0]
for s, e, n in ev:
    print('%9.3f %9.3f  %7.3f  %s' % ((s - t0) / 1e6, (e - t0) / 1e6, (e - s) / 1e6, n))
