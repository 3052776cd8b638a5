#!/bin/bash
# kernel-trace summary of the configs[3] hypergeometric call (bench_big.py hyper M); usage: prof_hyper.sh [M] [tag]
# the benched configuration (bench.py / run_batch.py set it for themselves; under rocprofv3 the runtime is initialised
# before Python runs, so it must come from the shell)
export GPU_MAX_HW_QUEUES=8
M=${1:-10000}; TAG=${2:-hyp}
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $ROOT/gpurun_out/$TAG -o r -- python3 $ROOT/tools/bench_big.py hyper $M > $ROOT/gpurun_out/$TAG.log 2>&1
python3 $ROOT/tools/rocpd_summary.py $ROOT/gpurun_out/$TAG/r_results.db | head -14
tail -3 $ROOT/gpurun_out/$TAG.log
