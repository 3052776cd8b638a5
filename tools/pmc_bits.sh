#!/bin/bash
# PMC passes for the headline bench step (k_permtest_bits_pre); run on the GPU box through gpurun
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/${1:-pb}; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
CMD="python3 $R/bench.py --steps 1 --warmup 1 --cpu-perms 0 --extras 0"
timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD --kernel-trace -d $O/pmc1 -o r -- $CMD > $O/pmc1.log 2>&1
timeout 300 rocprofv3 --pmc SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_WAVES SQ_INSTS_VMEM_WR --kernel-trace -d $O/pmc2 -o r -- $CMD > $O/pmc2.log 2>&1
timeout 300 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum GRBM_GUI_ACTIVE --kernel-trace -d $O/pmc3 -o r -- $CMD > $O/pmc3.log 2>&1
tail -1 $O/pmc3.log
