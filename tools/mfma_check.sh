#!/bin/bash
# quick GPU check of the MFMA kernel: parity tests + config-5 rank-share probe + config-2-shaped quantitative probe
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_mfma.py -x -q 2>&1 | tail -3
timeout 600 python tools/bench_big.py quant 1024 128 2>&1 | grep -v amdgpu.ids | tail -1
timeout 300 python tools/bench_quant.py 1024 200 2>&1 | grep "mfma"
