#!/bin/bash
# FETCH_SIZE / WRITE_SIZE calibration on this box (tools/ubench/fetch_calib.hip): two separate --pmc passes, --kernel-trace only.
# usage (through gpurun): bash tools/pmc_calib.sh [tag]   -> gpurun_out/<tag>/calib.txt ; tools/update_profiles.py reads it
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/${1:-calib}; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -o /tmp/fetch_calib $R/tools/ubench/fetch_calib.hip || exit 1
timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $O/f -o r -- /tmp/fetch_calib > $O/f.log 2>&1
timeout 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $O/w -o r -- /tmp/fetch_calib > $O/w.log 2>&1
: > $O/calib.txt
for d in f w; do python3 $R/tools/rocpd_counters.py $(ls $O/$d/*/*.db $O/$d/*.db 2>/dev/null | head -1) >> $O/calib.txt; done
rm -rf $O/f $O/w
cat $O/calib.txt
