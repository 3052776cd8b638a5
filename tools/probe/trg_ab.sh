#!/bin/bash
# A/B of the general matrix-core kernel's LDS form (lib_trg1.so: rows as they come + transposing reads; lib_trg0.so: register
# transposes): tests once per build, then the configs[4] rank share with z-scores, the general kernel's filtered form and six slices
for lib in lib_trg1 lib_trg0; do
  cp safepy_amd/$lib.so safepy_amd/libsafe_hip.so
  echo "== $lib"; python3 -m pytest tests/test_gpu_mfma.py -x -q 2>&1 | tail -2
done
for i in 1 2; do
  for lib in lib_trg1 lib_trg0; do
    cp safepy_amd/$lib.so safepy_amd/libsafe_hip.so
    echo "== $lib z-score"; VARIANTS=own,own python3 tools/probe/mfma_share.py 6250 1000 z-score 2>&1 | grep "^filter" | tail -1 | cut -c1-120
    echo "== $lib sum general,six"; VARIANTS=general,six python3 tools/probe/mfma_share.py 6250 1000 2>&1 | grep "^filter" | cut -c1-120
  done
done
cp safepy_amd/lib_trg1.so safepy_amd/libsafe_hip.so
