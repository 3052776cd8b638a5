# seeded headline step under different first pipeline stages (SAFE_HIP_STAGES) -- A/B of the pipeline fill
for cfg in "SAFE_HIP_STAGES=64" "SAFE_HIP_STAGES=48" "SAFE_HIP_STAGES=96" "SAFE_HIP_STAGES=128" "SAFE_HIP_STAGES=64,192" "SAFE_HIP_STAGES=64 SAFE_HIP_TAIL_STAGE=0" "SAFE_HIP_STAGES=64 SAFE_HIP_BITS_MERGE=2" "SAFE_HIP_STAGES=64,192 SAFE_HIP_BITS_MERGE=2"; do
  for rep in 1 2; do
  env $cfg python bench.py --steps 30 --extras 0 --cpu-perms 0 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
r=d['per_rank'][0]
print('$cfg', 'ms', round(d['ms_per_step'],3), 'med', round(d['step_ms_min_median_max'][1],3), 'stream', round(r['host_stream_ms'],2), 'draw', round(r['draw_busy_ms'],2), 'kbusy', round(r['gpu_kernel_busy_ms'],2), 'cpu', round(d['host_cpu_ms_per_step'],1), 'launches', d['roofline']['launches_per_step'])
"
  done
done
