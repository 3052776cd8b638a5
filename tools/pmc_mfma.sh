#!/bin/bash
# PMC passes for k_permtest_mfma at one rank's config-5 shape (run on the GPU box through gpurun)
# usage: pmc_mfma.sh <out tag> [sum|z-score]
# the benched configuration (bench.py / run_batch.py set it for themselves; under rocprofv3 the runtime is initialised
# before Python runs, so it must come from the shell)
export GPU_MAX_HW_QUEUES=8
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/${1:-d}; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
CMD="python3 $R/tools/bench_big.py quant 1024 128 ${2:-sum}"
timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD --kernel-trace -d $O/pmc1 -o r -- $CMD > $O/pmc1.log 2>&1
timeout 300 rocprofv3 --pmc SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VMEM --kernel-trace -d $O/pmc2 -o r -- $CMD > $O/pmc2.log 2>&1
timeout 300 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum GRBM_GUI_ACTIVE --kernel-trace -d $O/pmc3 -o r -- $CMD > $O/pmc3.log 2>&1
timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $O/pmc4 -o r -- $CMD > $O/pmc4.log 2>&1
for i in 1 2 3 4; do python3 $R/tools/rocpd_counters.py $(ls $O/pmc$i/*/*.db $O/pmc$i/*.db 2>/dev/null | head -1) k_permtest_mfma >> $O/pmc_summary.txt 2>&1; done
rm -rf $O/pmc1 $O/pmc2 $O/pmc3 $O/pmc4
tail -2 $O/pmc4.log
