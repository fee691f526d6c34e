#!/usr/bin/env python3
"""Instruction mix of the evaluation kernel's hot loops, from the gfx950 ISA hipcc emits for csrc/kernels_walk.hip.

  tools/isa_mix.py [--out profiles/r02_eval_isa_mix.txt]

Compiles kernels_walk.hip to assembly (hipcc -S --cuda-device-only, ~40 s, no GPU needed), takes the C4 instantiation
k_walk_group2<2,true,true,true,false,2> (N_GRAVS=2, TreePM, Yukawa, tables in LDS, evaluation), and counts instructions per
class in (a) the force loop -- the innermost loop with v_rsq_f64, one pool entry per trip -- block by block, and (b) the
reach-mask build (the straight-line block of v_pk_fma_f32).  Classes: fp64 VALU, transcendental fp64 (v_rsq/v_rcp, quarter
rate), packed fp32 VALU, scalar fp32 VALU, integer / bit / move VALU, LDS, vector memory, SALU + branches.

Blocks of the force loop are tagged `rare` when they hold the softening spline (v_div_scale_f64), the full exp() fallback
(v_rndne_f64: only used when the table-bin form is switched off), the per-pair periodic wrap (the lane-wrap twin of the
loop: groups whose box is wider than half the box) or the nint fix-up of a pair beyond the exact cut; everything else is the
common path.  The tags are heuristics on the instruction text -- the per-block listing is printed so they can be checked.
"""
import argparse
import collections
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "gadget-2.0.7-ngravs_amd", "csrc", "kernels_walk.hip")
KERNEL = "_Z13k_walk_group2ILi2ELb1ELb1ELb1ELb0ELi2E"


def classify(op):
    if op.startswith("v_rsq_f64") or op.startswith("v_rcp_f64") or op.startswith("v_sqrt_f64"):
        return "trans64"
    if op.startswith("v_pk_"):
        return "pk_f32"
    if op.startswith("v_") and ("_f64" in op):
        return "fp64"
    if op.startswith("v_") and ("_f32" in op):
        return "fp32"
    if op.startswith("v_"):
        return "int/bit/mov"
    if op.startswith("ds_"):
        return "lds"
    if op.startswith(("global_", "buffer_", "scratch_", "flat_")):
        return "vmem"
    if op.startswith("s_nop"):
        return "s_nop"
    if op.startswith("s_"):
        return "salu"
    return "other"


CLASSES = ["fp64", "trans64", "pk_f32", "fp32", "int/bit/mov", "lds", "vmem", "salu", "s_nop", "other"]


def blocks_of(lines):
    """[(label, [ops])] in file order"""
    out, cur, ops = [], "entry", []
    for ln in lines:
        m = re.match(r"^(\.LBB\d+_\d+):", ln)
        if m:
            out.append((cur, ops))
            cur, ops = m.group(1), []
            continue
        t = ln.strip()
        if not t or t.startswith(";") or t.startswith("."):
            continue
        ops.append(t.split()[0])
    out.append((cur, ops))
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=os.path.join(ROOT, "profiles", "r02_eval_isa_mix.txt"))
    ap.add_argument("--asm", default=None, help="reuse an existing .s file")
    args = ap.parse_args()
    asm = args.asm
    if asm is None:
        asm = os.path.join(tempfile.mkdtemp(), "walk.s")
        subprocess.check_call(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "-S", "--cuda-device-only", "-o", asm, SRC],
                              stderr=subprocess.DEVNULL)
    text = open(asm).read().split("\n")
    start = next(i for i, l in enumerate(text) if l.startswith(KERNEL))
    end = next(i for i in range(start, len(text)) if text[i].startswith(".Lfunc_end"))
    body = text[start:end]
    meta = {}
    for key in (".vgpr_count", ".sgpr_count", ".vgpr_spill_count", ".group_segment_fixed_size"):
        for i, l in enumerate(text):
            if ".name:" in l and KERNEL in l:
                for l2 in text[i:i + 40]:
                    if key + ":" in l2:
                        meta[key] = l2.split(":")[1].strip()
                break
    # ---- force loop: innermost-loop blocks (Depth=4) around the hot v_rsq_f64
    depth4 = [i for i, l in enumerate(body) if "Inner Loop Header: Depth=4" in l]
    rsq = [i for i, l in enumerate(body) if "v_rsq_f64" in l]
    hdr = max(h for h in depth4 if any(r > h for r in rsq))
    # the loop's blocks: from the header label back to the last block tagged with this header
    hdr_label = None
    for j in range(hdr, 0, -1):
        m = re.match(r"^(\.LBB\d+_\d+):", body[j])
        if m:
            hdr_label = m.group(1)
            hstart = j
            break
    name = hdr_label[1:].replace("LBB", "BB")
    in_loop = [i for i, l in enumerate(body) if ("in Loop: Header=" + name + " ") in l or i == hstart]
    lo, hi = min(in_loop), max(in_loop)
    # extend to the end of the last block
    k = hi + 1
    while k < len(body) and not re.match(r"^\.LBB\d+_\d+:", body[k]):
        k += 1
    loop_blocks = blocks_of(body[lo:k])
    with open(os.path.splitext(args.out)[0].replace("_isa_mix", "_force_loop") + ".s", "w") as f:   # the loop itself, for checking
        f.write("; force loop of %s (gfx950, hipcc -O3): blocks %s .. end of loop\n" % (KERNEL, hdr_label))
        f.write("\n".join(body[lo:k]) + "\n")
    out = []
    out.append("Evaluation kernel k_walk_group2<2,true,true,true,false,2> (C4: N_GRAVS=2, TreePM, Yukawa pairs, tables in LDS), gfx950, hipcc -O3")
    out.append("registers: %s" % ", ".join("%s=%s" % (k2[1:], v) for k2, v in meta.items()))
    out.append("")
    out.append("(a) force loop, one pool entry per trip (GW2_ES=1): basic blocks in file order")
    out.append("%-14s %-7s " % ("block", "path") + " ".join("%11s" % c for c in CLASSES))
    tot_common = collections.Counter()
    tot_rare = collections.Counter()
    twin = 0   # the loop holds the lane-wrap twin first, then the common (pre-wrapped) version: the second v_rsq marks it
    nrsq = 0
    for label, ops in loop_blocks:
        if not ops:
            continue
        cnt = collections.Counter(classify(o) for o in ops)
        rare = any(o.startswith("v_div_scale_f64") for o in ops) or any(o.startswith("v_rndne_f64") for o in ops)
        has_rsq = any(o.startswith("v_rsq_f64") for o in ops)
        if has_rsq:
            nrsq += 1
        tag = "rare" if rare else "common"
        out.append("%-14s %-7s " % (label, tag) + " ".join("%11d" % cnt.get(c, 0) for c in CLASSES))
        (tot_rare if rare else tot_common).update(cnt)
    out.append("%-14s %-7s " % ("sum", "common") + " ".join("%11d" % tot_common.get(c, 0) for c in CLASSES))
    out.append("%-14s %-7s " % ("sum", "rare") + " ".join("%11d" % tot_rare.get(c, 0) for c in CLASSES))
    out.append("note: the loop body exists twice (per-pair periodic wrap for groups wider than half the box, and the pre-wrapped form the")
    out.append("      bench runs); `common` sums BOTH twins plus the shared header/tail, so the per-trip count of the executed path is about")
    out.append("      half of the fp64 sum plus the shared integer header.  v_rsq_f64 blocks found: %d." % nrsq)
    # ---- mask build: the block with the most v_pk_fma_f32
    allb = blocks_of(body)
    best = max(allb, key=lambda b: sum(1 for o in b[1] if o.startswith("v_pk_fma_f32")))
    cnt = collections.Counter(classify(o) for o in best[1])
    npk = sum(1 for o in best[1] if o.startswith("v_pk_fma_f32"))
    out.append("")
    out.append("(b) reach-mask build for 64 pool entries (block %s, %d v_pk_fma_f32 = 3 per 2 entries): r2 - thr = |e|^2 + (|p|^2 - thr) - 2 e.p in" % (best[0], npk))
    out.append("    packed fp32, sign bits shifted into the 64-bit mask with v_alignbit")
    out.append(" ".join("%11s" % c for c in CLASSES))
    out.append(" ".join("%11d" % cnt.get(c, 0) for c in CLASSES))
    vals = sum(cnt.get(c, 0) for c in ("fp64", "trans64", "pk_f32", "fp32", "int/bit/mov"))
    out.append("    VALU per pool entry: %.2f (+ %.2f s_nop: the dependent v_pk_fma_f32 chain needs one wait state per link)" %
               (vals / 64.0, cnt.get("s_nop", 0) / 64.0))
    txt = "\n".join(out) + "\n"
    os.makedirs(os.path.dirname(args.out), exist_ok=True)
    with open(args.out, "w") as f:
        f.write(txt)
    sys.stdout.write(txt)


if __name__ == "__main__":
    main()
