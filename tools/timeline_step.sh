#!/bin/bash
# GPU timeline of one seeded headline step (rocprofv3 kernel trace of tools/trace_step.py); usage: timeline_step.sh [tag]
export GPU_MAX_HW_QUEUES=8
R=$GRAFT_REPO_ROOT; TAG=${1:-tl}; O=$R/gpurun_out/$TAG; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace -d $O/tr -o r -- python3 $R/tools/trace_step.py > $O/trace.log 2>&1
python3 $R/tools/rocpd_timeline.py $(ls $O/tr/*.db $O/tr/*/*.db 2>/dev/null | head -1) k_attr_stats > $O/timeline.txt 2>&1 || python3 $R/tools/rocpd_timeline.py $(ls $O/tr/*.db $O/tr/*/*.db 2>/dev/null | head -1) > $O/timeline.txt 2>&1
rm -rf $O/tr
tail -2 $O/trace.log; head -60 $O/timeline.txt
