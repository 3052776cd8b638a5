for cfg in "" "SAFE_HIP_BLOCKING_SYNC=1" "SAFE_BENCH_NO_PIN=1" "SAFE_HIP_BLOCKING_SYNC=1 SAFE_BENCH_NO_PIN=1"; do
  for rep in 1 2; do
  env $cfg python bench.py --steps 30 --extras 0 --cpu-perms 0 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
r=d['per_rank'][0]
print('$cfg', 'ms', round(d['ms_per_step'],3), 'med', round(d['step_ms_min_median_max'][1],3), 'stream', round(r['host_stream_ms'],2), 'draw', round(r['draw_busy_ms'],2), 'kbusy', round(r['gpu_kernel_busy_ms'],2), 'cpu', round(d['host_cpu_ms_per_step'],1))
"
  done
done
