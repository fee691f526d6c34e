#!/bin/bash
# the other BASELINE.json configurations on one GPU (bench.py --config): JSON lines into gpurun_out/<tag>_bench_<cfg>.json
TAG=${1:-r02}
for c in c1 c2 c3 c5; do
  timeout -k 10 500 python bench.py --config $c --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/${TAG}_bench_$c.json 2> gpurun_out/${TAG}_bench_$c.err
  python - <<PY
import json
d = json.load(open("gpurun_out/${TAG}_bench_$c.json"))
a = d.get("accuracy") or {}
print("$c", round(d["value"] / 1e6, 2), "M/s", round(d["ms_per_step"], 2), "ms", {k: round(v, 2) for k, v in d["config"]["phases_ms"].items()}, "ia", round(d["config"]["ia_per_particle"], 1), "rms", a.get("rms"))
PY
done
