#!/bin/bash
# round 6: what a SHORT launch of the bit-sliced kernel costs (kernels-only calls of P permutations in one launch, tables ready; the
# per-task clock trace SAFE_HIP_BITS_DBG=128 shows the tasks' durations)
cd $GRAFT_REPO_ROOT; O=gpurun_out/${1:-r6short}; mkdir -p $O; S=$O/summary.txt; : > $S
for P in ${2:-10 16 24 32 64 128}; do
  echo "== P=$P" >> $S
  SAFE_HIP_STAGES=$P SAFE_HIP_BITS_DBG=128 timeout 120 python tools/bits_ablate.py --one $P 2>&1 | tail -8 | sed -n '3,$p' >> $S
done
cat $S
