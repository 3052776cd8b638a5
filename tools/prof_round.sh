#!/bin/bash
# Round profile of the bench command on the GPU box (run through gpurun): kernel-trace statistics of the default bench run,
# the SQ / TCC counter passes of the headline step (tools/pmc_bits.sh: separate --pmc passes, --kernel-trace only), and
# FETCH_SIZE / WRITE_SIZE of the whole bench (extras included) in two more passes (MI355X_MICROARCH.md, HBM section).
# Writes text summaries under gpurun_out/<tag>/; tools/update_profiles.py <tag> turns them into profiles/rNN_* (run it
# in the build container right after, at the commit that was profiled).
# usage: prof_round.sh [tag]
# the benched configuration (bench.py / run_batch.py set it for themselves; under rocprofv3 the runtime is initialised
# before Python runs, so it must come from the shell)
export GPU_MAX_HW_QUEUES=8
TAG=${1:-prof}
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
CMD="python3 $ROOT/bench.py --steps 3 --warmup 1 --cpu-perms 0"
timeout 900 rocprofv3 --kernel-trace --stats -d $OUT/trace -o r -- $CMD > $OUT/trace.log 2>&1
python3 $ROOT/tools/rocpd_summary.py $OUT/trace/r_results.db > $OUT/kernel_stats.txt
grep '^{' $OUT/trace.log | tail -1 > $OUT/bench_line.json
# the headline alone, as the driver runs it (5 + 20 steps): k_permtest_bits_blk's average here = the line's roofline.kernel_ms
CMDH="python3 $ROOT/bench.py --steps 20 --warmup 5 --extras 0 --cpu-perms 0"
timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/trace_h -o r -- $CMDH > $OUT/trace_h.log 2>&1
python3 $ROOT/tools/rocpd_summary.py $OUT/trace_h/r_results.db > $OUT/kernel_stats_headline.txt
grep '^{' $OUT/trace_h.log | tail -1 > $OUT/bench_line_headline.json
rm -rf $OUT/trace_h
# GPU timeline of one seeded step
timeout 300 rocprofv3 --kernel-trace -d $OUT/tr -o r -- python3 $ROOT/tools/trace_step.py > $OUT/trace_step.log 2>&1
python3 $ROOT/tools/rocpd_timeline.py $(ls $OUT/tr/*.db $OUT/tr/*/*.db 2>/dev/null | head -1) > $OUT/timeline.txt 2>&1
rm -rf $OUT/tr
CMD1="python3 $ROOT/bench.py --steps 1 --warmup 1 --cpu-perms 0"
timeout 900 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $OUT/fetch -o r -- $CMD1 > $OUT/fetch.log 2>&1
timeout 900 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $OUT/write -o r -- $CMD1 > $OUT/write.log 2>&1
python3 $ROOT/tools/rocpd_counters.py $OUT/fetch/r_results.db > $OUT/pmc_traffic.txt
python3 $ROOT/tools/rocpd_counters.py $OUT/write/r_results.db >> $OUT/pmc_traffic.txt
rm -rf $OUT/trace $OUT/fetch $OUT/write
bash $ROOT/tools/pmc_mfma.sh $TAG/mfma_sum sum > /dev/null 2>&1
bash $ROOT/tools/pmc_mfma.sh $TAG/mfma_z z-score > /dev/null 2>&1
bash $ROOT/tools/pmc_bits.sh $TAG/bits
rm -rf $OUT/bits/pmc1 $OUT/bits/pmc2 $OUT/bits/pmc3 $OUT/bits/pmc4 $OUT/bits/pmc5
head -30 $OUT/kernel_stats.txt
