#!/bin/bash
# the benched configuration (bench.py / run_batch.py set it for themselves; under rocprofv3 the runtime is initialised
# before Python runs, so it must come from the shell)
export GPU_MAX_HW_QUEUES=8
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/${1:-h}; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
CMD="python3 $R/tools/bench_big.py hyper 10000"
timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD --kernel-trace -d $O/pmc1 -o r -- $CMD > $O/pmc1.log 2>&1
timeout 300 rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_WAVES SQ_INSTS_VMEM_WR SQ_INST_CYCLES_VMEM_RD SQ_THREAD_CYCLES_VALU GRBM_GUI_ACTIVE --kernel-trace -d $O/pmc2 -o r -- $CMD > $O/pmc2.log 2>&1
tail -1 $O/pmc2.log
