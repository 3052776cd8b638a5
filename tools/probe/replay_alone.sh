#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/${1:-replay_alone}; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
python3 $R/tools/probe/replay_alone.py 1000
timeout 300 rocprofv3 --kernel-trace --stats -d $O/tr -o r -- python3 $R/tools/probe/replay_alone.py 1000 > $O/log.txt 2>&1
python3 $R/tools/rocpd_summary.py $O/tr/r_results.db | head -12
python3 - <<PY
import sqlite3,glob
con=sqlite3.connect(glob.glob('$O/tr/*.db')[0])
cols=[r[1] for r in con.execute("pragma table_info(kernels)")]
nc='name' if 'name' in cols else 'kernel_name'
rows=con.execute("select %s,start,end from kernels order by start"%nc).fetchall()
rp=[(e-s)/1e3 for nm,s,e in rows if 'replay' in nm]
print('replay durations us:', [round(x,1) for x in rp[-10:]])
PY
rm -rf $O/tr
