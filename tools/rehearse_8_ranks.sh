#!/bin/bash
# Rehearsal of what the driver's `bench.py --gpus 8` does, on ONE GPU (SAFE_BENCH_SHARE_DEVICE=1: every rank on device 0, the
# exchange staged through gloo): 8 ranks, the node-shared permutation stream with 7 consumers, the multi_gpu_configs extras
# (configs[2] strong scaling seeded + unseeded, configs[4] rank share), under the bench hosts' 16-CPU quota (taskset 0-15).
# The GPU is time-shared by the 8 ranks, so `value` means nothing; what is checked: it runs to the end inside the driver's
# time limit, no ring timeout, the line is complete, and the producer rank's host CPU per step.
# usage: rehearse_8_ranks.sh [ranks] [steps]
export SAFE_BENCH_SHARE_DEVICE=1
RANKS=${1:-8}; STEPS=${2:-10}
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/rehearse8; mkdir -p $O
t0=$(date +%s.%N)
taskset -c 0-15 python $R/bench.py --gpus $RANKS --steps $STEPS --warmup 2 --cpu-perms 0 2> $O/err.txt | tail -1 > $O/line.json
rc=$?
t1=$(date +%s.%N)
echo "exit $rc, wall $(echo "$t1 - $t0" | bc) s"
grep -i -E "timeout|error|Traceback|not published" $O/err.txt | head -5
python3 - <<PY
import json
d = json.load(open("$O/line.json"))
print("ranks %d: step %.2f ms (%d steps), host CPU per step %.2f ms, host %s" % (d["n_gpus"], d["ms_per_step"], d["steps"], d["host_cpu_ms_per_step"], d["host"]))
print("  per rank (role, tables enqueued ms, waited for producer ms, kernels busy ms, host cpu ms):")
for r in d["per_rank"]:
    print("   ", r["role"], round(r["host_stream_ms"], 2), round(r["waited_for_producer_ms"], 2), round(r["gpu_kernel_busy_ms"], 2), round(r["host_cpu_ms_per_step"], 2))
for k, v in d.get("multi_gpu_configs", {}).items():
    print("  %s: %.1f ms per step; roles %s" % (k, v["ms_per_step"], [r["role"] for r in v["per_rank"]]))
PY
