#!/bin/bash
# round 6: filtered z-scores -- k_permtest_mfma_gz against the general kernel's FM = 2 on one box
cd $GRAFT_REPO_ROOT; O=gpurun_out/${1:-r6z}; mkdir -p $O; S=$O/summary.txt; : > $S
export GPU_MAX_HW_QUEUES=8
timeout 1200 python -m pytest tests/test_gpu_mfma.py -x -q -m gpu -k "zscore or z_score or filtered" > $O/pytest_z.log 2>&1; echo "pytest z rc=$?" >> $S; tail -3 $O/pytest_z.log >> $S
for form in ${2:-gz general gz general}; do
  echo "== form=$form" >> $S
  F=$form; [ $form = gz ] && F=""
  SAFE_HIP_MFMA_FORM=$F timeout 300 python tools/bench_big.py quant 2048 200 z-score 2>&1 | tail -1 >> $S
done
cat $S
