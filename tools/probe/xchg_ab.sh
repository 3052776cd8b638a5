#!/bin/bash
# A/B of the overlapped exchange at ONE rank with a real (single-rank) RCCL group: what the column-chunked tail costs the step and
# the kernels, what is left of the exchange after them.  usage: tools/probe/xchg_ab.sh [rounds] [steps] ["<overlap> <tail>" ...]
# (tail: a fraction of the permutations, or - for the library's default; SAFE_BENCH_SEED=none in the environment: unseeded calls)
rounds=${1:-3}; steps=${2:-40}; shift 2
cfgs=("$@"); [ ${#cfgs[@]} -eq 0 ] && cfgs=("0 -" "1 -" "1 0.4")
for i in $(seq $rounds); do
  for cfg in "${cfgs[@]}"; do
    set -- $cfg
    if [ "$2" = "-" ]; then unset SAFE_HIP_XCHG_TAIL; else export SAFE_HIP_XCHG_TAIL=$2; fi
    SAFE_HIP_XCHG_OVERLAP=$1 SAFE_BENCH_FORCE_DIST=1 python3 bench.py --steps $steps --warmup 5 --extras 0 --cpu-perms 0 2>/dev/null | python3 -c "
import sys, json
for ln in sys.stdin:
    if ln.startswith('{'):
        d = json.loads(ln); r = d['per_rank'][0]
        print('overlap=$1 tail=$2: step %.3f ms  kernels busy %.3f  exchange after the kernels %.3f  draw busy %.3f  no-exchange step %.3f' % (d['ms_per_step'], r['gpu_kernel_busy_ms'], r['exchange_ms'], r['draw_busy_ms'], d['exchange']['no_exchange_ms_per_step']))
"
  done
done
