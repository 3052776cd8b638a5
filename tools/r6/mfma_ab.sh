#!/bin/bash
# round 6: k_permtest_mfma_g (LDS-DMA gather) against round 5's k_permtest_mfma_f on one box; then the diagnostic builds
# (needs safepy_amd/libsafe_hip_diag.so = make DIAG=1; the shipped library is put back at the end)
# usage (through gpurun): tools/r6/mfma_ab.sh <out tag> "<form:dbg> ..." "<pytest files>"
cd $GRAFT_REPO_ROOT
O=gpurun_out/${1:-r6a}; mkdir -p $O
export GPU_MAX_HW_QUEUES=8
S=$O/summary.txt; : > $S
for t in ${3:-tests/test_gpu_mfma.py}; do
  timeout 1200 python -m pytest $t -x -q -m gpu > $O/pytest_$(basename $t .py).log 2>&1; echo "pytest $t rc=$?" >> $S; tail -3 $O/pytest_$(basename $t .py).log >> $S
done
for form in ${4:-g f g f}; do
  echo "== form=$form" >> $S
  SAFE_HIP_MFMA_FORM=$form timeout 300 python tools/bench_big.py quant 2048 200 2>&1 | tail -1 >> $S
done
if [ -f safepy_amd/libsafe_hip_diag.so ]; then
  cp safepy_amd/libsafe_hip.so /tmp/keep.so; cp safepy_amd/libsafe_hip_diag.so safepy_amd/libsafe_hip.so
  for spec in ${2:-g:0 g:1024 g:2 g:8 g:64 g:4 g:512}; do
    form=${spec%%:*}; dbg=${spec##*:}
    echo "== DIAG form=$form dbg=$dbg" >> $S
    SAFE_HIP_MFMA_FORM=$form SAFE_HIP_MFMA_DBG=$dbg timeout 300 python tools/bench_big.py quant 2048 200 ${5:-sum} 2>&1 | grep -v "^define\|DIAGNOSTIC\|diagnostic\|amdgpu.ids" | tail -5 | grep -v "call 1[0-9][0-9]\.\|call [6-9][0-9]\." >> $S
  done
  cp /tmp/keep.so safepy_amd/libsafe_hip.so
fi
cat $S
