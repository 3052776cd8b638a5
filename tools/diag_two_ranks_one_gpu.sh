#!/bin/bash
# Two (or $RANKS) ranks of bench.py on ONE GPU (SAFE_BENCH_SHARE_DEVICE=1: gloo-staged exchange): the host side of an N > 1 run --
# shared permutation stream vs one stream per rank, all CPUs vs two CPUs per rank (taskset) -- on the single-GPU test box.
export SAFE_BENCH_SHARE_DEVICE=1
RANKS=${RANKS:-2}
for mode in shared own; do
  for cpus in all limited; do
    if [ $mode = own ]; then export SAFE_HIP_SHARED_STREAM=0; else unset SAFE_HIP_SHARED_STREAM; fi
    if [ $cpus = all ]; then pre=""; else pre="taskset -c 0-$((2 * RANKS - 1))"; fi
    $pre python bench.py --gpus $RANKS --steps 10 --warmup 2 --cpu-perms 0 --multi-extras ${EXTRAS:-0} 2>gpurun_out/diag_${mode}_${cpus}.err | tail -1 > gpurun_out/diag_${mode}_${cpus}.json
    python - <<PY
import json
d=json.load(open("gpurun_out/diag_${mode}_${cpus}.json"))
x=d["exchange"]
print("${mode} ${cpus}: step %.2f ms, without exchange %.2f ms; host CPU ms per step without exchange, per rank: %s; %s" % (d["ms_per_step"], x["no_exchange_ms_per_step"], [round(v,2) for v in x["host_cpu_ms_per_step_no_exchange_per_rank"]], d["host"]))
print("    (role, tables enqueued at ms, waited for producer ms, kernels ms):", [(r["role"], round(r["host_stream_ms"],2), round(r["waited_for_producer_ms"],2), round(r["gpu_kernel_ms"],2)) for r in d["per_rank"]])
for k,v in d.get("multi_gpu_configs",{}).items(): print("   ",k, round(v["ms_per_step"],2), [ (r["role"], round(r["host_stream_ms"],2), round(r["waited_for_producer_ms"],2), round(r["gpu_kernel_ms"],2), round(r["exchange_ms"],2)) for r in v["per_rank"]])
PY
  done
done
