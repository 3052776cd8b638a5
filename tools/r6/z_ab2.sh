#!/bin/bash
# round 6: two builds of the library side by side (z-score and 'sum' matrix-core kernels), one box
cd $GRAFT_REPO_ROOT; O=gpurun_out/${1:-r6z2}; mkdir -p $O; S=$O/summary.txt; : > $S
export GPU_MAX_HW_QUEUES=8
cp safepy_amd/libsafe_hip.so /tmp/keep.so
for rep in 1 2; do
for lib in /tmp/keep.so safepy_amd/$2; do
  cp $lib safepy_amd/libsafe_hip.so
  echo "== lib=$(basename $lib) z" >> $S
  timeout 300 python tools/bench_big.py quant 2048 200 z-score 2>&1 | tail -1 >> $S
  echo "== lib=$(basename $lib) sum" >> $S
  timeout 300 python tools/bench_big.py quant 2048 200 2>&1 | tail -1 >> $S
done; done
cp /tmp/keep.so safepy_amd/libsafe_hip.so
cat $S
