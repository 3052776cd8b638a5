#!/bin/bash
# Who opens a hardware queue in the middle of a run?  HSA + HIP API trace of driver-style runs (no counters), kept for the first
# run whose bench line shows a ~4100-fault step.  usage: queue_trace.sh [attempts]
export GPU_MAX_HW_QUEUES=8
cd /tmp && export TMPDIR=/tmp
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
for i in $(seq ${1:-12}); do
  rm -rf /tmp/qt
  timeout 300 rocprofv3 --hsa-trace --hip-trace -d /tmp/qt -o q -- python3 $ROOT/bench.py --gpus 1 --steps 20 --warmup 5 --extras 0 --cpu-perms 0 > /tmp/qt.log 2>&1
  line=$(grep '^{' /tmp/qt.log | tail -1)
  verdict=$(echo "$line" | python3 -c "
import sys, json
d = json.loads(sys.stdin.read()); s = d['step_probe']['slowest_steps'][0]
print('mean %.3f slowest step %d %.2f ms faults %d' % (d['ms_per_step'], s['step'], s['ms'], s['minor_faults']))" 2>/dev/null)
  echo "attempt $i: $verdict"
  if [ "$KEEP_ALWAYS" = "1" ] || echo "$verdict" | grep -q "faults 4[0-9][0-9][0-9]"; then
    mkdir -p $ROOT/gpurun_out/queue_trace
    python3 $ROOT/tools/probe/queue_trace_report.py $(find /tmp/qt -name "*.db" | head -1) > $ROOT/gpurun_out/queue_trace/report.txt 2>&1
    echo "$line" > $ROOT/gpurun_out/queue_trace/bench_line.json
    ls -la $ROOT/gpurun_out/queue_trace/
    break
  fi
done
