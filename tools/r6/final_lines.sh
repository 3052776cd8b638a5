#!/bin/bash
# the round's bench lines: the default run (50 + 200 steps, extras) and three driver-style runs (5 + 20 steps, extras)
cd $GRAFT_REPO_ROOT; O=gpurun_out/${1:-r6lines}; mkdir -p $O
python bench.py > $O/default.json 2> $O/default.err; echo "default rc=$?"
for i in 1 2 3; do python bench.py --steps 20 --warmup 5 > $O/driver_$i.json 2> $O/driver_$i.err; echo "driver $i rc=$?"; done
python - <<PY
import json, glob
for p in sorted(glob.glob("$O/*.json")):
    d = json.loads(open(p).read().strip().splitlines()[-1])
    m = d.get("mfma_bound_kernels", {})
    print(p.split("/")[-1], "ms/step %.3f" % d["ms_per_step"], "min/med/max", [round(x, 3) for x in d["step_ms_min_median_max"]], "value %.3e" % d["value"],
          "chain us/perm %.2f" % d["draw_chain"]["us_per_permutation"], "kbusy %.2f" % d["roofline"]["kernel_busy_ms_per_step"],
          "unseeded %.3f" % d["unseeded_device_stream"]["1000_permutations"]["ms_per_step"],
          "mfma share s", [round(v["config5_rank_share_seconds"], 3) for v in m.values()])
PY
