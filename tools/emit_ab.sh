#!/bin/bash
# A/B of k_hyp_emit variants on one box (configs[3]: 20 000 x 10 000): usage emit_ab.sh <out file> "<ENV=VAL ...>" ...
export GPU_MAX_HW_QUEUES=8 BIG_ITERS=${BIG_ITERS:-12}
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/${1:-emit_ab.txt}; shift
: > $OUT
for rep in 1 2; do
  for v in "$@"; do
    echo "== $v" >> $OUT
    env $v timeout 300 python3 $R/tools/bench_big.py hyper 10000 2>&1 | tail -1 >> $OUT
  done
done
cat $OUT
