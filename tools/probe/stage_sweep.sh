#!/bin/bash
# seeded headline step against the pipeline's stage sizes (SAFE_HIP_CHUNK, SAFE_HIP_STAGES, SAFE_HIP_TAIL_STAGE); through gpurun
cd $GRAFT_REPO_ROOT
run() { python3 bench.py --steps ${STEPS:-100} --warmup 30 --extras 0 --cpu-perms 0 2>/dev/null | python3 -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); r=d['roofline']
        print('  mean %.3f median %.3f  busy %.3f  launches %d' % (d['ms_per_step'], d['step_ms_min_median_max'][1], r.get('kernel_busy_ms_per_step',0), r.get('launches_per_step',0)))
"; }
for v in "" "SAFE_HIP_CHUNK=160" "SAFE_HIP_CHUNK=192" "SAFE_HIP_CHUNK=255" "SAFE_HIP_CHUNK=192 SAFE_HIP_STAGES=16,64,160" "SAFE_HIP_CHUNK=255 SAFE_HIP_STAGES=16,64,192" "SAFE_HIP_TAIL_STAGE=0" "SAFE_HIP_TAIL_STAGE=2" ""; do
  echo "[$v]"; env $v bash -c "$(declare -f run); run"
done
