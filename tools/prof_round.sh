#!/bin/bash
# Round profile of the bench command on the GPU box: kernel-trace statistics, then FETCH_SIZE and WRITE_SIZE in
# separate counter passes (MI355X_MICROARCH.md, HBM section).  Writes text summaries under gpurun_out/<tag>/.
# usage: prof_round.sh [tag]   (copy the *.txt it prints into profiles/)
TAG=${1:-prof}
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
CMD="python3 $ROOT/bench.py --steps 3 --warmup 1 --cpu-perms 0"
timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/trace -o r -- $CMD > $OUT/trace.log 2>&1
python3 $ROOT/tools/rocpd_summary.py $OUT/trace/r_results.db > $OUT/kernel_stats.txt
grep '^{' $OUT/trace.log | tail -1 > $OUT/bench_line.json
CMD1="python3 $ROOT/bench.py --steps 1 --warmup 1 --cpu-perms 0"
timeout 600 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $OUT/fetch -o r -- $CMD1 > $OUT/fetch.log 2>&1
timeout 600 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $OUT/write -o r -- $CMD1 > $OUT/write.log 2>&1
python3 $ROOT/tools/rocpd_counters.py $OUT/fetch/r_results.db > $OUT/pmc_traffic.txt
python3 $ROOT/tools/rocpd_counters.py $OUT/write/r_results.db >> $OUT/pmc_traffic.txt
rm -rf $OUT/trace $OUT/fetch $OUT/write
head -30 $OUT/kernel_stats.txt
