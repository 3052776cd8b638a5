#!/bin/bash
# GPU timeline of the last configs[3] hypergeometric call of tools/bench_big.py hyper 10000; usage: timeline_hyper.sh [tag]
export GPU_MAX_HW_QUEUES=8
R=$GRAFT_REPO_ROOT; TAG=${1:-tlh}; O=$R/gpurun_out/$TAG; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace -d $O/tr -o r -- python3 $R/tools/bench_big.py hyper 10000 > $O/trace.log 2>&1
python3 $R/tools/rocpd_timeline.py $(ls $O/tr/*.db $O/tr/*/*.db 2>/dev/null | head -1) k_mfma_planes01 k_u32_to_f64 > $O/timeline.txt 2>&1
rm -rf $O/tr
tail -3 $O/trace.log; cat $O/timeline.txt
