#!/bin/bash
cd $GRAFT_REPO_ROOT; O=gpurun_out/${1:-r6flush}; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_golden2.py tests/test_gpu_fullsize.py tests/test_gpu_u8.py -x -q -m gpu > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest.log
bash tools/r6/short_launch.sh $1_trace "16 128" "0,4,64" | grep -v "^\[gpurun"
bash tools/r6/step_ab.sh $1_steps 3 "base=SAFE_HIP_BITS_SHORT=0,4,64" "short8=SAFE_HIP_BITS_SHORT=48,8,128"
