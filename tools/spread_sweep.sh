#!/bin/bash
# walk_spread sweep (S lanes per target, 64/S targets per wave) on the tree-only BASELINE configs and the clustered TreePM probe:
#   tools/spread_sweep.sh TAG      -> gpurun_out/TAG_spread_*.json
TAG=${1:-r04}
for c in c2 c1; do
  for s in 1 2 4 8; do
    t=""; [ $s -gt 1 ] && t="--tune walk_spread=$s"
    timeout -k 10 300 python bench.py --config $c --steps 5 --warmup 1 --no-cpu-baseline $t > gpurun_out/${TAG}_spread_${c}_s$s.json 2> gpurun_out/${TAG}_spread_${c}_s$s.err
    python - <<PY
import json
d = json.loads(open("gpurun_out/${TAG}_spread_${c}_s$s.json").read().strip().splitlines()[-1])
a = d.get("accuracy") or {}
print("$c S=$s", round(d["value"] / 1e6, 2), "M/s", round(d["ms_per_step"], 2), "ms", {k: round(v, 2) for k, v in d["config"]["phases_ms"].items()}, "ia", round(d["config"]["ia_per_particle"], 1), "rms", a.get("rms"), "split", d["roofline"].get("split_walk"))
PY
  done
done
timeout -k 10 300 python tools/clustered_probe.py 20 > gpurun_out/${TAG}_clustered_probe.json 2> gpurun_out/${TAG}_clustered_probe.err
python - <<PY
import json
d = json.load(open("gpurun_out/${TAG}_clustered_probe.json"))
for k, v in d.items():
    print(k, v)
PY
