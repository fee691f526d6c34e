#!/bin/bash
# Calibration of FETCH_SIZE for the evaluation kernel's access shapes (run through gpurun from the repo root):
#   tools/ubench/fetch_calib.sh   ->  gpurun_out/fetch_calib/fetch_calib.json  (copy to profiles/r03_fetch_calibration.json)
set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/fetch_calib
mkdir -p $O
/opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -o $O/fetch_calib $R/tools/ubench/fetch_calib.hip
rocprofv3 --output-format csv --pmc FETCH_SIZE -d $O/pmc -o p -- $O/fetch_calib > $O/known.jsonl 2> $O/err.log
python3 - "$O" <<'PY'
import csv, glob, json, os, sys
O = sys.argv[1]
known = [json.loads(l) for l in open(os.path.join(O, "known.jsonl")) if l.startswith("{")]
rows = []
for fn in glob.glob(os.path.join(O, "pmc", "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(fn)):
        if r["Counter_Name"] == "FETCH_SIZE":
            rows.append((int(r["Dispatch_Id"]), r["Kernel_Name"].split("(")[0], float(r["Counter_Value"]) * 1024.0,
                         (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-6))
rows.sort()
by = {}
for d, k, v, ms in rows:
    by.setdefault(d, [k, 0.0, ms])[1] += v
disp = [by[d] for d in sorted(by) if "k_" in by[d][0]]
out = []
for kn, (k, v, ms) in zip(known, disp):
    total = kn["bytes"] + kn.get("index_bytes", 0)
    out.append({"kernel": kn["kernel"], "order": kn.get("order"), "bytes_read_exactly_once": kn["bytes"], "index_bytes_streamed": kn.get("index_bytes", 0),
                "FETCH_SIZE_bytes": v, "FETCH_SIZE_over_bytes": v / total, "ms": ms, "GBps": total / ms / 1e6})
json.dump({"note": "FETCH_SIZE reported by rocprofv3 --pmc against bytes each kernel reads exactly once from 2 GiB buffers (8 x the Infinity Cache); "
                   "gather kernels also stream their 4-byte indices (calibrated shape, counted at 1/2)", "results": out}, open(os.path.join(O, "fetch_calib.json"), "w"), indent=1)
print(json.dumps(out, indent=1))
PY
rm -rf $O/pmc $O/fetch_calib
