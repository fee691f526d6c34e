#!/bin/bash
# every BASELINE.json configuration on one GPU (bench.py --config), each line with roofline, accuracy and cpu_baseline:
#   tools/bench_configs.sh TAG ["c1 c2 c3 c4 c5"]   -> gpurun_out/TAG_bench_<cfg>.json
TAG=${1:-r04}
CFGS=${2:-c1 c2 c3 c4 c5}
for c in $CFGS; do
  timeout -k 10 900 python bench.py --config $c --steps 5 --warmup 1 > gpurun_out/${TAG}_bench_$c.json 2> gpurun_out/${TAG}_bench_$c.err
  python - <<PY
import json
d = json.loads(open("gpurun_out/${TAG}_bench_$c.json").read().strip().splitlines()[-1])
a = d.get("accuracy") or {}
b = d.get("cpu_baseline") or {}
print("$c", round(d["value"] / 1e6, 2), "M/s", round(d["ms_per_step"], 2), "ms", {k: round(v, 2) for k, v in d["config"]["phases_ms"].items()}, "ia", round(d["config"]["ia_per_particle"], 1), "rms", a.get("rms"), "cpu", b.get("value"), b.get("cores"))
PY
done
