#!/usr/bin/env python3
"""Instruction mix of the evaluation kernel's hot loops, from the gfx950 ISA hipcc emits for csrc/kernels_walk.hip.

  tools/isa_mix.py [--out profiles/r02_eval_isa_mix.txt]

Compiles kernels_walk.hip to assembly (hipcc -S --cuda-device-only, ~40 s, no GPU needed), takes the C4 instantiation
k_walk_group2<2,true,true,true,false,2> (N_GRAVS=2, TreePM, Yukawa, tables in LDS, evaluation) and counts instructions per
class in
  (a) the force loop -- one pool entry per trip.  The source instantiates it four times (per-pair periodic wrap yes/no x
      Yukawa factor through the table bins yes/no), chosen by scalar branches outside the loop; the bench runs the variant
      WITHOUT per-pair wrap and WITH the table-bin factor: the innermost loop that holds v_fract_f64 and no wrap arithmetic
      (the shortest of the loops with v_fract_f64).  Blocks are tagged `rare` from the source's own asm markers of its two
      wave-level rare branches ("; beyond the exact cut", "; softened pair");
  (b) the reach-mask build for 64 pool entries: the block with the v_mfma_f32_32x32x2_f32 instructions;
  (c) the chunk loop around them (fetch, cull, compaction of one batch of 64 list items), without (a) and (b).
Classes: fp64 VALU, transcendental fp64 (v_rsq/v_rcp, quarter rate), packed fp32 VALU, scalar fp32 VALU, integer / bit / move
VALU, MFMA, LDS, vector memory, SALU + branches, s_nop.  The tags are heuristics on the instruction text -- the loop itself is
written next to the table (…_force_loop.s) so that they can be checked.
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
    if op.startswith("#"):
        return "marker"
    if op.startswith("v_mfma"):
        return "mfma"
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


CLASSES = ["fp64", "trans64", "pk_f32", "fp32", "int/bit/mov", "mfma", "lds", "vmem", "salu", "s_nop", "other"]
VALU = ("fp64", "trans64", "pk_f32", "fp32", "int/bit/mov")


MARKS = ("beyond the exact cut", "softened pair")


def blocks_of(lines):
    """[(label, depth, header, [ops])] in file order; `; %bb.N:` sub-blocks are kept apart.  The source's asm markers of the
    rare branches ("; beyond the exact cut", "; softened pair") are kept as pseudo-ops "#<marker>"."""
    out, cur, depth, hdr, ops = [], "entry", 0, "", []
    for ln in lines:
        m = re.match(r"^(\.LBB\d+_\d+):(.*)", ln) or re.match(r"^; %(bb\.\d+):(.*)", ln)
        for mk in MARKS:
            if ln.strip() == "; " + mk:
                ops.append("#" + mk)
        if m:
            out.append((cur, depth, hdr, ops))
            cur, ops, rest = m.group(1), [], m.group(2)
            d = re.search(r"Depth=(\d+)", rest)
            if d:
                depth = int(d.group(1))
            h = re.search(r"Header=(BB\d+_\d+)", rest)
            hdr = h.group(1) if h else ("" if d is None else hdr)
            continue
        if "Loop Header: Depth=" in ln:
            depth = int(re.search(r"Depth=(\d+)", ln).group(1))
            hdr = cur[1:].replace("LBB", "BB") if cur.startswith(".LBB") else hdr
        t = ln.strip()
        if not t or t.startswith(";") or t.startswith("."):
            continue
        ops.append(t.split()[0])
    out.append((cur, depth, hdr, ops))
    return out


def row(label, tag, cnt):
    return "%-14s %-7s " % (label, tag) + " ".join("%11d" % cnt.get(c, 0) for c in CLASSES)


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
    for i, l in enumerate(text):
        if ".name:" in l and KERNEL in l:
            seg = text[max(0, i - 40):i + 40]
            for key in (".vgpr_count", ".sgpr_count", ".vgpr_spill_count", ".private_segment_fixed_size"):
                for l2 in seg:
                    if key + ":" in l2:
                        meta[key] = l2.split(":")[1].strip()
            break
    blocks = blocks_of(body)
    # ---- (a) the force-loop variants: innermost loops (depth 4) holding a v_rsq_f64
    loops = collections.OrderedDict()
    for label, depth, hdr, ops in blocks:
        if depth == 4 and hdr:
            loops.setdefault(hdr, []).append((label, ops))
    cand = []
    for hdr, bl in loops.items():
        allops = [o for _, ops in bl for o in ops]
        if any(o.startswith("v_rsq_f64") for o in allops):
            cand.append((hdr, bl, any(o.startswith("v_fract_f64") for o in allops), len(allops)))
    withfract = [c for c in cand if c[2]]
    hdr, bl, _, _ = min(withfract, key=lambda c: c[3])
    # raw text of that loop, for checking
    lab = "." + hdr.replace("BB", "LBB")
    idx = [i for i, l in enumerate(body) if ("Header=" + hdr + " ") in l or l.startswith(lab + ":")]
    lo, hi = min(idx), max(idx)
    k = hi + 1
    while k < len(body) and not re.match(r"^\.LBB\d+_\d+:", body[k]):
        k += 1
    with open(os.path.splitext(args.out)[0].replace("_isa_mix", "_force_loop") + ".s", "w") as f:
        f.write("; force loop of %s (gfx950, hipcc -O3), the variant the C4 bench runs (no per-pair wrap, table-bin exp): loop %s\n" % (KERNEL, hdr))
        f.write("\n".join(body[lo:k]) + "\n")
    out = []
    out.append("Evaluation kernel k_walk_group2<2,true,true,true,false,2> (C4: N_GRAVS=2, TreePM, Yukawa pairs, tables in LDS), gfx950, hipcc -O3")
    out.append("registers: %s" % ", ".join("%s=%s" % (k2[1:], v) for k2, v in meta.items()))
    out.append("force-loop variants found (innermost loops with v_rsq_f64): %s" %
               ", ".join("%s: %d instructions%s" % (c[0], c[3], " [table-bin exp]" if c[2] else "") for c in cand))
    out.append("")
    out.append("(a) force loop, one pool entry per trip; variant %s (no per-pair wrap, table-bin exp): basic blocks in file order" % hdr)
    out.append("%-14s %-7s " % ("block", "path") + " ".join("%11s" % c for c in CLASSES))
    tot_common, tot_rare = collections.Counter(), collections.Counter()
    in_soft = False
    for label, ops in bl:
        if not ops:
            continue
        cnt = collections.Counter(classify(o) for o in ops)
        cnt.pop("marker", None)
        # rare: the block with the exact-cut fix-up, and everything from the "softened pair" marker on except the loop's tail
        # (the block that clears the mask bit with v_lshl_add_u64 and accumulates)
        if "#softened pair" in ops:
            in_soft = True
        is_tail = any(o.startswith("v_lshl_add_u64") for o in ops)   # the layout puts parts of the spline behind the tail
        rare = (in_soft and not is_tail) or ("#beyond the exact cut" in ops)
        out.append(row(label, "rare" if rare else "common", cnt))
        (tot_rare if rare else tot_common).update(cnt)
    out.append(row("sum", "common", tot_common))
    out.append(row("sum", "rare", tot_rare))
    valu = sum(tot_common.get(c, 0) for c in VALU)
    out.append("common path: %d VALU per trip (the v_rsq_f64 issues at a quarter of the rate: + 3 issue slots), %d LDS reads, %d scalar" %
               (valu, tot_common.get("lds", 0), tot_common.get("salu", 0)))
    # ---- (b) mask build: the block with the MFMAs
    best = max(blocks, key=lambda b: sum(1 for o in b[3] if o.startswith("v_mfma")))
    cnt = collections.Counter(classify(o) for o in best[3])
    out.append("")
    out.append("(b) reach masks for 64 pool entries (block %s): D[entry][target] = |e|^2 + (ex,ey,ez,1).(-2tx,-2ty,-2tz,|t|^2-thr) as %d" %
               (best[0], cnt.get("mfma", 0)))
    out.append("    v_mfma_f32_32x32x2_f32 (2 entry blocks x 2 target blocks x K=4), sign bits packed with v_alignbit, v_permlane32_swap to the owners")
    out.append(" ".join("%11s" % c for c in CLASSES))
    out.append(" ".join("%11d" % cnt.get(c, 0) for c in CLASSES))
    vals = sum(cnt.get(c, 0) for c in VALU)
    out.append("    VALU per pool entry: %.2f (+ %d MFMA per 64 entries; round 2 start: 3.09 VALU per entry, no MFMA)" % (vals / 64.0, cnt.get("mfma", 0)))
    # ---- (c) chunk loop: the depth-3 blocks of the loop that contains the MFMA block, in file order: [fetch, cull, compaction]
    #      [MFMA masks] [fp64 exact-mask ladder: tuning walk_exact_reach only] [force-loop variants] [remainder move, latch]
    chdr = best[2]
    ibest = next(i for i, b_ in enumerate(blocks) if b_[0] == best[0] and b_[2] == chdr)
    d4 = [i for i, b_ in enumerate(blocks) if b_[1] == 4 and i > ibest]
    ilast4 = max(d4) if d4 else ibest
    tot = collections.Counter()
    nblk = 0
    for i, (label, depth, h, ops) in enumerate(blocks):
        if depth == 3 and h == chdr and (i < ibest or i > ilast4):
            tot.update(collections.Counter(classify(o) for o in ops))
            nblk += 1
    tot.pop("marker", None)
    out.append("")
    out.append("(c) chunk loop %s around (a) and (b): fetch of 64 list items, cull against the group's box, compaction into the pool," % chdr)
    out.append("    remainder move -- static sum over its %d blocks outside (a), (b) and the fp64 exact-mask fallback; both sides of its" % nblk)
    out.append("    branches (periodic wrap, type fetch, quad refill every 4th trip), so an upper bound per batch of 64 items")
    out.append(" ".join("%11s" % c for c in CLASSES))
    out.append(" ".join("%11d" % tot.get(c, 0) for c in CLASSES))
    txt = "\n".join(out) + "\n"
    os.makedirs(os.path.dirname(args.out), exist_ok=True)
    with open(args.out, "w") as f:
        f.write(txt)
    sys.stdout.write(txt)


if __name__ == "__main__":
    main()
