#!/bin/bash
# kernels-only call (tables ready) of the bit-sliced kernel: diagnostic builds and launch spans side by side
# usage: ab_stream_span.sh "<dbg values>" "<spans>"
for span in ${2:-100 200}; do for d in ${1:-0 256}; do
  echo -n "span=$span dbg=$d: "; SAFE_HIP_BITS_SPAN=$span SAFE_HIP_BITS_DBG=$d timeout 120 python tools/bits_ablate.py --one 1000 2>/dev/null | tail -1
done; done
