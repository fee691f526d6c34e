#!/usr/bin/env python3
"""Condense rocprofv3 --pmc counter_collection CSVs (one directory per pass) into the per-kernel summary kept under
profiles/, and the per-launch HBM traffic of the walk's evaluation kernel into a small json that bench.py reads.

usage: summarize_pmc.py OUT.txt TRAFFIC.json PASSDIR [PASSDIR ...]
"""
import csv
import glob
import json
import os
import re
import sys
from collections import defaultdict


def short(name):
    name = re.sub(r"\(.*", "", name)
    return name.replace("void ", "").strip()


def main():
    out_txt, out_json, dirs = sys.argv[1], sys.argv[2], sys.argv[3:]
    lines = []
    traffic = {}
    for d in dirs:
        files = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
        if not files:
            continue
        per_kernel = defaultdict(lambda: defaultdict(float))
        ndisp = defaultdict(set)
        per_disp = defaultdict(lambda: defaultdict(float))
        dur = {}
        for fn in files:
            with open(fn) as f:
                for row in csv.DictReader(f):
                    k = short(row["Kernel_Name"])
                    per_kernel[k][row["Counter_Name"]] += float(row["Counter_Value"])
                    ndisp[k].add(row["Dispatch_Id"])
                    if "k_walk" in k or "k_eval" in k:
                        per_disp[(k, int(row["Dispatch_Id"]))][row["Counter_Name"]] += float(row["Counter_Value"])
                        dur[(k, int(row["Dispatch_Id"]))] = (int(row["End_Timestamp"]) - int(row["Start_Timestamp"])) * 1e-6
        lines.append("== " + os.path.basename(d.rstrip("/")))
        for k in sorted(per_kernel):
            if k.startswith("rocprim") or k.startswith("hipcub") or "fft" in k.lower() or k.startswith("__amd"):
                continue
            lines.append("%-60s dispatches=%d %s" % (k[:60], len(ndisp[k]), dict(per_kernel[k])))
        for (k, i) in sorted(per_disp, key=lambda t: t[1]):
            v = dict(per_disp[(k, i)])
            v["dur_ms"] = round(dur[(k, i)], 3)
            lines.append("   %s dispatch %d: %s" % (k[:40], i, v))
            if k.endswith("2>") or ", 2>" in k or "k_eval_ring" in k:
                for c in ("FETCH_SIZE", "WRITE_SIZE"):
                    if c in v:
                        traffic.setdefault(c, []).append(v[c] * 1024.0)
    with open(out_txt, "w") as f:
        f.write("\n".join(lines) + "\n")
    if traffic:
        # the second half of the evaluation launches belongs to the timed relative-criterion step (steps=1, warmup=0:
        # first force computation = theta pass, second = relative pass)
        def timed_mean(v):
            h = v[len(v) // 2:]
            return sum(h) / len(h)
        fb, wb = timed_mean(traffic.get("FETCH_SIZE", [0])), timed_mean(traffic.get("WRITE_SIZE", [0]))
        with open(out_json, "w") as f:
            import hashlib
            csrc = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gadget-2.0.7-ngravs_amd", "csrc")
            h = hashlib.sha256()
            for name in ("kernels_walk.hip", "kernels_eval.hip", "eval_asm.inc", "walk_device.hpp"):   # = bench.py WALK_SOURCES
                h.update(open(os.path.join(csrc, name), "rb").read())
            sha = h.hexdigest()[:16]
            kern = sorted(k for (k, i) in per_disp if "k_eval_ring" in k or k.endswith("2>") or ", 2>" in k)
            json.dump({"kernel": (kern[-1] if kern else "?") + " (evaluation)", "workload": "C4 64M", "walk_source_sha16": sha,
                       "launches_per_step": len(traffic.get("FETCH_SIZE", [0])) - len(traffic.get("FETCH_SIZE", [0])) // 2,
                       "fetch_bytes_reported": fb, "write_bytes": wb, "traffic_bytes_per_launch": fb + wb,
                       "note": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes), mean over the evaluation-kernel "
                               "launches of the timed relative-criterion step; FETCH_SIZE as reported: the kernel's reads are 16-byte list "
                               "quads and 32-byte record gathers, for which FETCH_SIZE counts the fabric bytes 1:1 (64-byte requests; "
                               "calibrated by tools/ubench/fetch_calib.hip, profiles/r03_fetch_calibration.json -- only wide coalesced "
                               "streaming reads are reported at 1/2)"}, f, indent=1)


if __name__ == "__main__":
    main()
