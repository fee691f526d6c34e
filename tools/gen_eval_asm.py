#!/usr/bin/env python3
"""Writes gadget-2.0.7-ngravs_amd/csrc/eval_asm.inc: the gfx950 assembly blocks of the ring-pool evaluation kernel
(kernels_eval.hip) as C string macros, from templates with NAMED registers.

  tools/gen_eval_asm.py            (no arguments; the output is committed, the build does not run this script)

Registers.  The blocks use v104 .. v127 and s90 .. s95 as temporaries and say so in their clobber lists; everything that lives
from one block to the next is an operand.  (Tried and dropped: keeping the records of the chunk in flight in registers above an
amdgpu_waves_per_eu cap, out of the compiler's sight -- the cap does not hold where the register pressure is highest, and the
request written in assembly was not faster than the compiler's.)  A VALU instruction of this ISA takes ONE scalar operand
(SGPR, VCC or literal).
"""
import os
import re
import struct

OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gadget-2.0.7-ngravs_amd", "csrc", "eval_asm.inc")


def pair(n):
    return "v[%d:%d]" % (n, n + 1)


REGS = {
    # trip loop
    "E0": "v[104:107]", "E1": "v[108:111]",          # entry as loaded: (x, y), (z, mass)
    "DX": pair(104), "DY": pair(106), "DZ": pair(108), "MW": pair(110), "MWLO": "v110", "MWHI": "v111",
    "R2": pair(112), "TT": pair(112),                # r^2, later the short-range table value
    "RI": pair(114), "RR": pair(116),                # 1/r; r^2 + tiny, then r
    "T1": pair(118), "T1LO": "v118", "T2": pair(120), "T2LO": "v120", "T2HI": "v121",
    "T3": pair(122), "T3LO": "v122", "T3HI": "v123",  # fac
    "TE": pair(124), "TELO": "v124", "TEHI": "v125",
    "I1": "v112", "I2": "v113",                      # table offsets / addresses (R2 is dead by then ... see the loop)
    "J": "v126", "A": "v127",
    # cull
    "EX": pair(104), "EY": pair(106), "EXY": "v[104:107]", "EZ": pair(108), "B0": pair(110), "B1": pair(112), "B2": pair(114),
    "L": "v116", "SQ": "v117", "EA": "v118", "SQ2": "v119",
}


def lit(x):
    lo, hi = struct.unpack("<II", struct.pack("<d", x))
    return "0x%08x" % lo, "0x%08x" % hi


def smov(pair_lo, x):
    lo, hi = lit(x)
    return ["s_mov_b32 s%d, %s" % (pair_lo, lo), "s_mov_b32 s%d, %s" % (pair_lo + 1, hi)]


def subst(lines):
    out = []
    for l in lines:
        code, _, comment = l.partition(";;")
        code = re.sub(r"\{(\w+)\}", lambda m: REGS[m.group(1)], code.rstrip())
        out.append((code, comment.strip()))
    return out


def macro(name, lines, params=""):
    body = subst(lines)
    txt = ["#define %s%s \\" % (name, params)]
    for i, (code, comment) in enumerate(body):
        if code.startswith("@"):           # a macro parameter spliced in
            s = "  %s" % code[1:]
        else:
            s = '  "%s\\n"' % code
        if comment:
            s = s.ljust(78) + "/* " + comment + " */"
        txt.append(s + (" \\" if i + 1 < len(body) else ""))
    return "\n".join(txt) + "\n"


def clobbers(name, vregs, sregs):
    items = ['"v%d"' % v for v in vregs] + ['"s%d"' % s for s in sregs] + ['"vcc"', '"scc"', '"memory"']
    return "#define %s %s\n" % (name, ", ".join(items))


# ---- trip loop ---------------------------------------------------------------------------------------------------------------
YUK_ET = [
    "v_fract_f64_e32 {T1}, {T1}                      ;; fb: position inside the table bin",
    "v_mul_f64 {T3}, {T1}, %[ec3]",
    "v_lshl_add_u32 {I2}, {I1}, 3, %[etab]",
    "v_add_f64 {T3}, {T3}, -%[ec2]",
    "ds_read_b64 {TE}, {I2}                          ;; E[bin] = exp(-ym r_bin)",
    "v_lshl_add_u32 {I1}, {I1}, 3, %[trow]",
    "v_fma_f64 {T3}, {T3}, {T1}, %[ec1]",
    "ds_read_b64 {TT}, {I1}                          ;; short-range table",
    "v_fma_f64 {T3}, {T3}, {T1}, -%[ec0]",
    "v_mul_f64 {T2}, {RI}, {RI}                      ;; 1/r^2",
    "v_fma_f64 {T3}, {T3}, {T1}, 1.0                 ;; exp(-ym (r - r_bin)), degree 4",
    "s_waitcnt lgkmcnt(1)",
    "v_mul_f64 {T3}, {TE}, {T3}                      ;; exp(-ym r)",
    "v_mul_f64 {T3}, %[cY], {T3}",
    "v_fma_f64 {TE}, %[ym], {RI}, {T2}               ;; ym/r + 1/r^2",
    "v_mul_f64 {T3}, {TE}, {T3}",
    "v_fmac_f64_e32 {T3}, %[cN], {T2}                ;; + cN/r^2",
]
NOYUK = [
    "v_lshl_add_u32 {I1}, {I1}, 3, %[trow]",
    "ds_read_b64 {TT}, {I1}",
    "v_mul_f64 {T2}, {RI}, {RI}",
    "v_mul_f64 {T3}, %[cN], {T2}",
]
TRIP = [
    "s_mov_b32 %[ntr], 0",
    "L_er_top_%=:",
    "v_cmp_eq_u32_e32 vcc, %[tail], %[q]",
    "s_cbranch_vccz L_er_done_%=",
    "s_add_u32 %[ntr], %[ntr], 1",
    "v_ffbl_b32_e32 {J}, %[m]                        ;; -1 for an empty mask: the NULL entry",
    "v_add_co_u32_e64 {I1}, s[90:91], %[m], -1       ;; carry <=> m != 0: the lanes with a real entry",
    "v_lshl_add_u32 {A}, {J}, 4, %[q]",
    "v_and_b32_e32 %[m], {I1}, %[m]",
    "v_cmp_eq_u32_e32 vcc, 0, %[m]",
    "s_and_saveexec_b64 s[92:93], vcc                ;; lanes whose block is used up follow the link",
    "v_add_u32_e32 {I1}, %[q], %[lane4]",
    "ds_read_b32 %[q], %[q] offset:256",
    "ds_read_b32 %[m], {I1}",
    "s_mov_b64 exec, s[92:93]",
    "ds_read_b128 {E0}, {A} offset:320",
    "ds_read_b128 {E1}, {A} offset:848",
    "s_waitcnt lgkmcnt(0)",
    "v_add_f64 {DX}, {DX}, -%[tpx]",
    "v_add_f64 {DY}, {DY}, -%[tpy]",
    "v_mul_f64 {R2}, {DY}, {DY}",
    "v_add_f64 {DZ}, {DZ}, -%[tpz]",
    "v_fmac_f64_e32 {R2}, {DX}, {DX}",
    "v_fmac_f64_e32 {R2}, {DZ}, {DZ}",
    "v_cmp_ngt_f64_e32 vcc, %[reach2], {R2}          ;; !(r2 < reach2)",
    "s_and_b64 s[94:95], vcc, s[90:91]              ;; (SCC = the result is not zero)",
    "s_cbranch_scc0 L_er_incut_%=",
    "v_cndmask_b32_e64 {MWHI}, {MWHI}, 0, s[94:95]   ;; rare: beyond the exact cut -- no mass, not counted",
    "v_cndmask_b32_e64 {MWLO}, {MWLO}, 0, s[94:95]",
    "v_cndmask_b32_e64 {T1LO}, 0, 1, s[94:95]",
    "v_sub_u32_e32 %[nint], %[nint], {T1LO}",
    "L_er_incut_%=:",
    "v_add_f64 {RR}, {R2}, %[tiny]                   ;; self / coincident pairs stay finite",
    "v_rsq_f64_e32 {RI}, {RR}",
    "v_cmp_gt_f64_e32 vcc, %[h2max], {R2}            ;; closer than the largest softening length?  (vcc lives until the branch below)",
    "v_mul_f64 {T1}, {RR}, {RI}                      ;; one Newton step: y += y/2 (1 - x y^2)",
    "v_fma_f64 {T1}, -{T1}, {RI}, 1.0",
    "v_mul_f64 {T2}, {RI}, 0.5",
    "v_fma_f64 {RI}, {T2}, {T1}, {RI}                ;; 1/r",
    "v_mul_f64 {RR}, {RR}, {RI}                      ;; r",
    "v_mul_f64 {T1}, {RR}, %[asmthfac]",
    "v_cvt_i32_f64_e32 {I1}, {T1}                    ;; table bin (saturating conversion, then clamped); r2 is no longer needed",
    "v_min_i32_e32 {I1}, 0x7ff, {I1}",
    "@YUKSEG",
    "s_waitcnt lgkmcnt(0)",
    "v_fma_f64 {T3}, -%[utor2wpi], {TT}, {T3}        ;; - long-range part",
    "v_mul_f64 {T3}, {MW}, {T3}",
    "v_mul_f64 {T3}, {RI}, {T3}                      ;; fac = f m / r",
    "s_cbranch_vccnz L_er_soft_%=",
    "L_er_acc_%=:",
    "v_fmac_f64_e32 %[ax], {DX}, {T3}",
    "v_fmac_f64_e32 %[ay], {DY}, {T3}",
    "v_fmac_f64_e32 %[az], {DZ}, {T3}",
    "s_branch L_er_top_%=",
    # ---- rare: a pair possibly inside the softening radius (forcetree.c:1415-1417, ngravs.c:420-434)
    "L_er_soft_%=:",
    "v_mul_i32_i24_e32 {T1LO}, 0xfffffff1, {J}       ;; type byte of entry j: [slot] + ER_TYPE + j = A - 15 j + 272",
    "v_add_u32_e32 {T1LO}, {A}, {T1LO}",
    "ds_read_u8 {T1LO}, {T1LO} offset:272",
    "s_waitcnt lgkmcnt(0)",
    "v_lshl_add_u32 {T1LO}, {T1LO}, 3, %[etab]",
    "@\"ds_read_b64 {TE}, {T1LO} offset:\" FSTOFF \"\\n\"".replace("{TE}", REGS["TE"]).replace("{T1LO}", REGS["T1LO"]) + " ;; softening length of the source's type",
    "s_waitcnt lgkmcnt(0)",
    "v_max_f64 {TE}, {TE}, %[hT]                     ;; h = max(target, source)",
    "v_rcp_f64_e32 {TT}, {TE}",
    "v_cmp_lt_f64_e64 s[94:95], {RR}, {TE}           ;; soft = r < h",
    "v_fma_f64 {T1}, -{TE}, {TT}, 1.0                ;; 1/h: two Newton steps",
    "v_fma_f64 {TT}, {TT}, {T1}, {TT}",
    "v_fma_f64 {T1}, -{TE}, {TT}, 1.0",
    "v_fma_f64 {TT}, {TT}, {T1}, {TT}                ;; h_inv",
    "v_mul_f64 {RI}, {TE}, {RI}                      ;; 1/u = h/r",
    "v_mul_f64 {T1}, {RR}, {TT}                      ;; u = r/h",
    "v_mul_f64 {T2}, {T1}, {T1}                      ;; u^2",
    "v_ldexp_f64 {TE}, {T1}, 5                       ;; 32 u",
] + smov(90, 38.4) + [
    "v_add_f64 {TE}, {TE}, -s[90:91]",
] + smov(92, 10.666666666667) + [
    "v_fma_f64 {TE}, {T2}, {TE}, s[92:93]            ;; u < 1/2: 10.67 + u^2 (32 u - 38.4)",
    "v_mul_f64 {RR}, {T2}, {T1}                      ;; u^3 (r is no longer needed)",
    "v_mul_f64 {T2}, {T2}, s[90:91]                  ;; 38.4 u^2",
] + smov(90, 21.333333333333) + [
    "v_add_f64 {T2}, {T2}, s[90:91]                  ;; 21.33 + 38.4 u^2",
] + smov(90, -48.0) + [
    "v_fma_f64 {T2}, {T1}, s[90:91], {T2}            ;; - 48 u",
    "v_fma_f64 {T2}, -{RR}, s[92:93], {T2}           ;; - 10.67 u^3",
    "v_mul_f64 {RR}, {RI}, {RI}",
    "v_mul_f64 {RR}, {RR}, {RI}                      ;; 1/u^3",
] + smov(90, 0.066666666667) + [
    "v_fma_f64 {T2}, -{RR}, s[90:91], {T2}           ;; - 0.0667 / u^3",
    "v_cmp_gt_f64_e32 vcc, 0.5, {T1}                 ;; u < 1/2",
    "v_cndmask_b32_e32 {T2LO}, {T2LO}, {TELO}, vcc",
    "v_cndmask_b32_e32 {T2HI}, {T2HI}, {TEHI}, vcc",
    "v_mul_f64 {TE}, %[cS], {MW}                     ;; cS m h_inv^3 v",
    "v_mul_f64 {TE}, {TE}, {TT}",
    "v_mul_f64 {TE}, {TE}, {TT}",
    "v_mul_f64 {TE}, {TE}, {TT}",
    "v_mul_f64 {TE}, {TE}, {T2}",
    "v_cndmask_b32_e64 {T3LO}, {T3LO}, {TELO}, s[94:95]",
    "v_cndmask_b32_e64 {T3HI}, {T3HI}, {TEHI}, s[94:95]",
    "s_branch L_er_acc_%=",
    "L_er_done_%=:",
]

# ---- trip loop, two entries per trip (ER_TRIP2_ASM) --------------------------------------------------------------------------------
# A wave at 4 waves per SIMD spends most of its time waiting on its own dependent chain (LDS round trips, v_rsq_f64, ~35 dependent
# fp64 instructions per entry): measured 69.6 / 77.8 / 98.5 ms with 16 / 12 / 8 waves per CU.  Two entries per trip as two
# interleaved instruction streams (a, b) halve that chain per entry.  Stream b takes the lane's NEXT bit after stream a's cursor
# update, so it may come from the next block: no slot is lost to pairing.
def stream_regs(s):
    base = 104 if s == "a" else 80
    r = {}
    names = {"DX": 0, "DY": 2, "DZ": 4, "MW": 6, "R2": 8, "TT": 8, "RI": 10, "RR": 12, "T1": 14, "T2": 16, "T3": 18, "TE": 20}
    for k, o in names.items():
        r[k] = pair(base + o)
        r[k + "LO"] = "v%d" % (base + o)
        r[k + "HI"] = "v%d" % (base + o + 1)
    r["E0"] = "v[%d:%d]" % (base, base + 3)
    r["E1"] = "v[%d:%d]" % (base + 4, base + 7)
    r["I1"], r["I2"] = "v%d" % (base + 8), "v%d" % (base + 9)
    r["J"], r["A"] = "v%d" % (base + 22), "v%d" % (base + 23)
    r["ACT"] = "s[90:91]" if s == "a" else "s[88:89]"
    r["SOFT"] = "s[94:95]" if s == "a" else "s[86:87]"
    r["S"] = s
    return r


def fill(lines, r):
    out = []
    for l in lines:
        code, sep, comment = l.partition(";;")
        code = re.sub(r"\{(\w+)\}", lambda m: r[m.group(1)], code)
        out.append(code + (sep + comment if sep else ""))
    return out


def zipl(a, b):
    out = []
    for i in range(max(len(a), len(b))):
        if i < len(a):
            out.append(a[i])
        if i < len(b):
            out.append(b[i])
    return out


T2_FETCH = [
    "v_ffbl_b32_e32 {J}, %[m]                        ;; -1 for an empty mask: the NULL entry",
    "v_add_co_u32_e64 {I1}, {ACT}, %[m], -1          ;; carry <=> m != 0: the lanes with a real entry",
    "v_lshl_add_u32 {A}, {J}, 4, %[q]",
    "ds_read_b128 {E0}, {A} offset:320",
    "ds_read_b128 {E1}, {A} offset:848",
    "v_and_b32_e32 %[m], {I1}, %[m]",
    "v_cmp_eq_u32_e32 vcc, 0, %[m]",
    "s_and_saveexec_b64 s[92:93], vcc                ;; lanes whose block is used up follow the link",
    "v_add_u32_e32 {I1}, %[q], %[lane4]",
    "ds_read_b32 %[q], %[q] offset:256",
    "ds_read_b32 %[m], {I1}",
    "s_mov_b64 exec, s[92:93]",
]
T2_S1 = [
    "v_add_f64 {DX}, {DX}, -%[tpx]",
    "v_add_f64 {DY}, {DY}, -%[tpy]",
    "v_add_f64 {DZ}, {DZ}, -%[tpz]",
    "v_mul_f64 {R2}, {DY}, {DY}",
    "v_fmac_f64_e32 {R2}, {DX}, {DX}",
    "v_fmac_f64_e32 {R2}, {DZ}, {DZ}",
    "v_cmp_ngt_f64_e32 vcc, %[reach2], {R2}          ;; !(r2 < reach2)",
    "s_and_b64 s[92:93], vcc, {ACT}",
    "s_cmp_eq_u64 s[92:93], 0",
    "s_cbranch_scc1 L_er_incut{S}_%=",
    "v_cndmask_b32_e64 {MWHI}, {MWHI}, 0, s[92:93]   ;; rare: beyond the exact cut -- no mass, not counted",
    "v_cndmask_b32_e64 {MWLO}, {MWLO}, 0, s[92:93]",
    "v_cndmask_b32_e64 {T1LO}, 0, 1, s[92:93]",
    "v_sub_u32_e32 %[nint], %[nint], {T1LO}",
    "L_er_incut{S}_%=:",
]
T2_S2 = [
    "v_add_f64 {RR}, {R2}, %[tiny]                   ;; self / coincident pairs stay finite",
    "v_rsq_f64_e32 {RI}, {RR}",
    "v_cmp_lt_f64_e64 {SOFT}, {R2}, %[h2max]         ;; closer than the largest softening length?",
    "v_mul_f64 {T1}, {RR}, {RI}                      ;; one Newton step: y += y/2 (1 - x y^2)",
    "v_fma_f64 {T1}, -{T1}, {RI}, 1.0",
    "v_mul_f64 {T2}, {RI}, 0.5",
    "v_fma_f64 {RI}, {T2}, {T1}, {RI}                ;; 1/r",
    "v_mul_f64 {RR}, {RR}, {RI}                      ;; r",
    "v_mul_f64 {T1}, {RR}, %[asmthfac]",
    "v_cvt_i32_f64_e32 {I1}, {T1}                    ;; table bin (saturating conversion, then clamped); r2 is no longer needed",
    "v_min_i32_e32 {I1}, 0x7ff, {I1}",
]
T2_S3_YUK = [
    "v_fract_f64_e32 {T1}, {T1}                      ;; fb: position inside the table bin",
    "v_lshl_add_u32 {I2}, {I1}, 3, %[etab]",
    "ds_read_b64 {TE}, {I2}                          ;; E[bin] = exp(-ym r_bin)",
    "v_lshl_add_u32 {I1}, {I1}, 3, %[trow]",
    "ds_read_b64 {TT}, {I1}                          ;; short-range table",
    "v_mul_f64 {T3}, {T1}, %[ec3]",
    "v_add_f64 {T3}, {T3}, -%[ec2]",
    "v_fma_f64 {T3}, {T3}, {T1}, %[ec1]",
    "v_fma_f64 {T3}, {T3}, {T1}, -%[ec0]",
    "v_mul_f64 {T2}, {RI}, {RI}                      ;; 1/r^2",
    "v_fma_f64 {T3}, {T3}, {T1}, 1.0                 ;; exp(-ym (r - r_bin)), degree 4",
]
T2_S4_YUK = [
    "v_mul_f64 {T3}, {TE}, {T3}                      ;; exp(-ym r)",
    "v_mul_f64 {T3}, %[cY], {T3}",
    "v_fma_f64 {TE}, %[ym], {RI}, {T2}               ;; ym/r + 1/r^2",
    "v_mul_f64 {T3}, {TE}, {T3}",
    "v_fmac_f64_e32 {T3}, %[cN], {T2}                ;; + cN/r^2",
    "v_fma_f64 {T3}, -%[utor2wpi], {TT}, {T3}        ;; - long-range part",
    "v_mul_f64 {T3}, {MW}, {T3}",
    "v_mul_f64 {T3}, {RI}, {T3}                      ;; fac = f m / r",
]
T2_S3_NOYUK = [
    "v_lshl_add_u32 {I1}, {I1}, 3, %[trow]",
    "ds_read_b64 {TT}, {I1}",
    "v_mul_f64 {T2}, {RI}, {RI}",
    "v_mul_f64 {T3}, %[cN], {T2}",
]
T2_S4_NOYUK = [
    "v_fma_f64 {T3}, -%[utor2wpi], {TT}, {T3}        ;; - long-range part",
    "v_mul_f64 {T3}, {MW}, {T3}",
    "v_mul_f64 {T3}, {RI}, {T3}                      ;; fac = f m / r",
]
T2_S5 = [
    "s_cmp_lg_u64 {SOFT}, 0",
    "s_cbranch_scc1 L_er_soft{S}_%=",
    "L_er_acc{S}_%=:",
    "v_fmac_f64_e32 %[ax], {DX}, {T3}",
    "v_fmac_f64_e32 %[ay], {DY}, {T3}",
    "v_fmac_f64_e32 %[az], {DZ}, {T3}",
]
T2_SOFT = [
    "L_er_soft{S}_%=:",
    "v_mul_i32_i24_e32 {T1LO}, 0xfffffff1, {J}       ;; type byte of entry j: [slot] + ER_TYPE + j = A - 15 j + 272",
    "v_add_u32_e32 {T1LO}, {A}, {T1LO}",
    "ds_read_u8 {T1LO}, {T1LO} offset:272",
    "s_waitcnt lgkmcnt(0)",
    "v_lshl_add_u32 {T1LO}, {T1LO}, 3, %[etab]",
    "@FST {TE}, {T1LO} ;; softening length of the source's type",
    "s_waitcnt lgkmcnt(0)",
    "v_max_f64 {TE}, {TE}, %[hT]                     ;; h = max(target, source)",
    "v_rcp_f64_e32 {TT}, {TE}",
    "v_cmp_lt_f64_e64 {SOFT}, {RR}, {TE}             ;; soft = r < h",
    "v_fma_f64 {T1}, -{TE}, {TT}, 1.0                ;; 1/h: two Newton steps",
    "v_fma_f64 {TT}, {TT}, {T1}, {TT}",
    "v_fma_f64 {T1}, -{TE}, {TT}, 1.0",
    "v_fma_f64 {TT}, {TT}, {T1}, {TT}                ;; h_inv",
    "v_mul_f64 {RI}, {TE}, {RI}                      ;; 1/u = h/r",
    "v_mul_f64 {T1}, {RR}, {TT}                      ;; u = r/h",
    "v_mul_f64 {T2}, {T1}, {T1}                      ;; u^2",
    "v_ldexp_f64 {TE}, {T1}, 5                       ;; 32 u",
] + smov(90, 38.4) + [
    "v_add_f64 {TE}, {TE}, -s[90:91]",
] + smov(92, 10.666666666667) + [
    "v_fma_f64 {TE}, {T2}, {TE}, s[92:93]            ;; u < 1/2: 10.67 + u^2 (32 u - 38.4)",
    "v_mul_f64 {RR}, {T2}, {T1}                      ;; u^3 (r is no longer needed)",
    "v_mul_f64 {T2}, {T2}, s[90:91]                  ;; 38.4 u^2",
] + smov(90, 21.333333333333) + [
    "v_add_f64 {T2}, {T2}, s[90:91]                  ;; 21.33 + 38.4 u^2",
] + smov(90, -48.0) + [
    "v_fma_f64 {T2}, {T1}, s[90:91], {T2}            ;; - 48 u",
    "v_fma_f64 {T2}, -{RR}, s[92:93], {T2}           ;; - 10.67 u^3",
    "v_mul_f64 {RR}, {RI}, {RI}",
    "v_mul_f64 {RR}, {RR}, {RI}                      ;; 1/u^3",
] + smov(90, 0.066666666667) + [
    "v_fma_f64 {T2}, -{RR}, s[90:91], {T2}           ;; - 0.0667 / u^3",
    "v_cmp_gt_f64_e32 vcc, 0.5, {T1}                 ;; u < 1/2",
    "v_cndmask_b32_e32 {T2LO}, {T2LO}, {TELO}, vcc",
    "v_cndmask_b32_e32 {T2HI}, {T2HI}, {TEHI}, vcc",
    "v_mul_f64 {TE}, %[cS], {MW}                     ;; cS m h_inv^3 v",
    "v_mul_f64 {TE}, {TE}, {TT}",
    "v_mul_f64 {TE}, {TE}, {TT}",
    "v_mul_f64 {TE}, {TE}, {TT}",
    "v_mul_f64 {TE}, {TE}, {T2}",
    "v_cndmask_b32_e64 {T3LO}, {T3LO}, {TELO}, {SOFT}",
    "v_cndmask_b32_e64 {T3HI}, {T3HI}, {TEHI}, {SOFT}",
    "s_branch L_er_acc{S}_%=",
]


def trip2(yuk):
    ra, rb = stream_regs("a"), stream_regs("b")
    s3, s4 = (T2_S3_YUK, T2_S4_YUK) if yuk else (T2_S3_NOYUK, T2_S4_NOYUK)
    seq = ["s_mov_b32 %[ntr], 0",
           "L_er_top_%=:",
           "v_cmp_eq_u32_e32 vcc, %[tail], %[q]",
           "s_cbranch_vccz L_er_done_%=",
           "s_add_u32 %[ntr], %[ntr], 2"]
    seq += fill(T2_FETCH, ra)
    seq += ["s_waitcnt lgkmcnt(0)                            ;; stream b takes the lane's NEXT bit: after the cursor update"]
    seq += fill(T2_FETCH, rb)
    seq += fill(T2_S1, ra)
    seq += ["s_waitcnt lgkmcnt(0)"]
    seq += fill(T2_S1, rb)
    seq += zipl(fill(T2_S2, ra), fill(T2_S2, rb))
    seq += zipl(fill(s3, ra), fill(s3, rb))
    seq += ["s_waitcnt lgkmcnt(0)"]
    seq += zipl(fill(s4, ra), fill(s4, rb))
    seq += fill(T2_S5, ra) + fill(T2_S5, rb)
    seq += ["s_branch L_er_top_%="]
    seq += fill(T2_SOFT, ra) + fill(T2_SOFT, rb)
    seq += ["L_er_done_%=:"]
    # the FST splice: ds_read_b64 with the offset as a macro parameter
    out = []
    for l in seq:
        if l.startswith("@FST"):
            code, _, comment = l.partition(";;")
            regs_ = code[len("@FST"):].strip()
            out.append("@\"ds_read_b64 %s offset:\" FSTOFF \"\\n\" ;;%s" % (regs_, comment))
        else:
            out.append(l)
    return out



# ---- trip loop with the NEXT entry in flight (ER_TRIP3_*_ASM) ------------------------------------------------------------------------
# ER_TRIP_ASM starts every trip with an LDS round trip nobody hides: cursor -> ds_read_b128 x 2 -> s_waitcnt -> first use.  Here the
# loop is unrolled twice over two register sets (x: v104-v111 + v126/v127, y: v96-v103 + v94/v95): as soon as the entry of trip t has
# arrived (and with it, in order, the link words of the lanes that changed slots), the cursor step of trip t+1 is taken and ITS entry
# requested, then trip t is evaluated.  The exit test of trip t+1 comes before that step -- a trip that will not run takes no bit.
def role_regs(s):
    r = dict(REGS)
    if s == "y":
        r.update({"E0": "v[96:99]", "E1": "v[100:103]", "DX": pair(96), "DY": pair(98), "DZ": pair(100), "MW": pair(102), "MWLO": "v102",
                  "MWHI": "v103", "J": "v94", "A": "v95"})
    r["ACT"] = "s[90:91]" if s == "x" else "s[88:89]"
    r["S"] = s
    return r


P_CURSOR = [
    "v_ffbl_b32_e32 {J}, %[m]                        ;; -1 for an empty mask: the NULL entry",
    "v_add_co_u32_e64 {T3LO}, {ACT}, %[m], -1        ;; carry <=> m != 0: the lanes with a real entry",
    "v_lshl_add_u32 {A}, {J}, 4, %[q]",
    "v_and_b32_e32 %[m], {T3LO}, %[m]",
    "v_cmp_eq_u32_e32 vcc, 0, %[m]",
    "s_and_saveexec_b64 s[92:93], vcc                ;; lanes whose block is used up follow the link",
    "v_add_u32_e32 {T3LO}, %[q], %[lane4]",
    "ds_read_b32 %[q], %[q] offset:256",
    "ds_read_b32 %[m], {T3LO}",
    "s_mov_b64 exec, s[92:93]",
    "ds_read_b128 {E0}, {A} offset:320",
    "ds_read_b128 {E1}, {A} offset:848",
]
P_HEAD = [
    "v_add_f64 {DX}, {DX}, -%[tpx]",
    "v_add_f64 {DY}, {DY}, -%[tpy]",
    "v_mul_f64 {R2}, {DY}, {DY}",
    "v_add_f64 {DZ}, {DZ}, -%[tpz]",
    "v_fmac_f64_e32 {R2}, {DX}, {DX}",
    "v_fmac_f64_e32 {R2}, {DZ}, {DZ}",
    "v_cmp_ngt_f64_e32 vcc, %[reach2], {R2}          ;; !(r2 < reach2)",
    "s_and_b64 s[94:95], vcc, {ACT}",
    "s_cmp_eq_u64 s[94:95], 0",
    "s_cbranch_scc1 L_er_incut{S}_%=",
    "v_cndmask_b32_e64 {MWHI}, {MWHI}, 0, s[94:95]   ;; rare: beyond the exact cut -- no mass, not counted",
    "v_cndmask_b32_e64 {MWLO}, {MWLO}, 0, s[94:95]",
    "v_cndmask_b32_e64 {T1LO}, 0, 1, s[94:95]",
    "v_sub_u32_e32 %[nint], %[nint], {T1LO}",
    "L_er_incut{S}_%=:",
    "v_add_f64 {RR}, {R2}, %[tiny]                   ;; self / coincident pairs stay finite",
    "v_rsq_f64_e32 {RI}, {RR}",
    "v_cmp_lt_f64_e64 s[94:95], {R2}, %[h2max]       ;; closer than the largest softening length?",
    "v_mul_f64 {T1}, {RR}, {RI}                      ;; one Newton step: y += y/2 (1 - x y^2)",
    "v_fma_f64 {T1}, -{T1}, {RI}, 1.0",
    "v_mul_f64 {T2}, {RI}, 0.5",
    "v_fma_f64 {RI}, {T2}, {T1}, {RI}                ;; 1/r",
    "v_mul_f64 {RR}, {RR}, {RI}                      ;; r",
    "v_mul_f64 {T1}, {RR}, %[asmthfac]",
    "v_cvt_i32_f64_e32 {I1}, {T1}                    ;; table bin (saturating conversion, then clamped); r2 is no longer needed",
    "v_min_i32_e32 {I1}, 0x7ff, {I1}",
]
P_TAIL = [
    "s_waitcnt lgkmcnt(0)",
    "v_fma_f64 {T3}, -%[utor2wpi], {TT}, {T3}        ;; - long-range part",
    "v_mul_f64 {T3}, {MW}, {T3}",
    "v_mul_f64 {T3}, {RI}, {T3}                      ;; fac = f m / r",
    "s_cmp_lg_u64 s[94:95], 0",
    "s_cbranch_scc1 L_er_soft{S}_%=",
    "L_er_acc{S}_%=:",
    "v_fmac_f64_e32 %[ax], {DX}, {T3}",
    "v_fmac_f64_e32 %[ay], {DY}, {T3}",
    "v_fmac_f64_e32 %[az], {DZ}, {T3}",
]


def trip3(yuk):
    i0 = TRIP.index("L_er_soft_%=:")
    soft = TRIP[i0 + 1:TRIP.index("L_er_done_%=:")]      # the softened-pair path of ER_TRIP_ASM, relabelled per register set below
    rx, ry = role_regs("x"), role_regs("y")
    seg = YUK_ET if yuk else NOYUK
    seq = ["s_mov_b32 %[ntr], 0",
           "s_mov_b32 s87, 0                                ;; 1: the trip being evaluated is the last one",
           "v_cmp_eq_u32_e32 vcc, %[tail], %[q]",
           "s_cbranch_vccz L_er_done_%="]
    seq += fill(P_CURSOR, rx)
    for cur, nxt in ((rx, ry), (ry, rx)):
        S = cur["S"]
        seq += ["L_er_top%s_%%=:" % S,
                "s_waitcnt lgkmcnt(0)                            ;; this trip's entry (requested a trip ago) and, before it, the link words",
                "s_add_u32 %[ntr], %[ntr], 1",
                "v_cmp_eq_u32_e32 vcc, %[tail], %[q]             ;; will there be another trip?",
                "s_cbranch_vccnz L_er_pf%s_%%=" % S,
                "s_mov_b32 s87, 1",
                "s_branch L_er_cmp%s_%%=" % S,
                "L_er_pf%s_%%=:" % S]
        seq += fill(P_CURSOR, nxt)
        seq += ["L_er_cmp%s_%%=:" % S]
        seq += fill(P_HEAD, cur) + fill(seg, cur) + fill(P_TAIL, cur)
        seq += ["s_cmp_lg_u32 s87, 0",
                "s_cbranch_scc1 L_er_done_%="]
    seq += ["s_branch L_er_topx_%="]
    for r in (rx, ry):
        S = r["S"]
        seq += ["L_er_soft%s_%%=:" % S]
        for l in soft:
            if l.startswith("@"):
                # the FST splice of ER_TRIP_ASM names REGS' registers: TE and T1LO are shared temporaries, the same in both sets
                seq.append(l)
            else:
                l = l.replace("L_er_acc_%=", "L_er_acc%s_%%=" % S)
                if S == "y":   # the path's scratch pair is the set's own (by now dead) mask of real entries: s[90:91] holds set x's NEXT one
                    l = l.replace("s[90:91]", "s[88:89]").replace("s_mov_b32 s90,", "s_mov_b32 s88,").replace("s_mov_b32 s91,", "s_mov_b32 s89,")
                seq.append(fill([l], r)[0])
    seq += ["L_er_done_%=:"]
    return seq


# ---- tree-only force loop (ER_DIRECT_ASM; k_walk_group2 with PM = false, one lane per target) ---------------------------------------
# Every pool entry interacts with every target: no masks, all lanes read the same entry (LDS broadcast).  The compiler's version of
# this loop carried its three accumulators through six register copies per entry and left the entry's LDS round trip exposed
# (33 issue slots per entry, 204 cycles per entry and SIMD measured on the 4 M Plummer sphere).  Here: unrolled twice over two
# register sets, the next entry requested before the current one is evaluated, 21 VALU per entry.  Pool entries are 32-byte
# records (x, y | z, m) from %[ptr] on; s89 counts the entries; a lane without a target is handed cN = cS = 0.
D_COMPUTE = [
    "v_add_f64 {DX}, {DX}, -%[tpx]",
    "v_add_f64 {DY}, {DY}, -%[tpy]",
    "v_mul_f64 {R2}, {DY}, {DY}",
    "v_add_f64 {DZ}, {DZ}, -%[tpz]",
    "v_fmac_f64_e32 {R2}, {DX}, {DX}",
    "v_fmac_f64_e32 {R2}, {DZ}, {DZ}",
    "v_add_f64 {RR}, {R2}, %[tiny]                   ;; self / coincident pairs stay finite",
    "v_rsq_f64_e32 {RI}, {RR}",
    "v_cmp_gt_f64_e32 vcc, %[h2max], {R2}            ;; closer than the largest softening length?  (vcc lives until the branch below)",
    "v_mul_f64 {T1}, {RR}, {RI}                      ;; one Newton step: y += y/2 (1 - x y^2)",
    "v_fma_f64 {T1}, -{T1}, {RI}, 1.0",
    "v_mul_f64 {T2}, {RI}, 0.5",
    "v_fma_f64 {RI}, {T2}, {T1}, {RI}                ;; 1/r",
    "v_mul_f64 {T2}, {RI}, {RI}",
    "v_mul_f64 {T3}, %[cN], {T2}                     ;; cN / r^2",
    "v_mul_f64 {T3}, {MW}, {T3}",
    "v_mul_f64 {T3}, {RI}, {T3}                      ;; fac = f m / r",
    "s_cbranch_vccnz L_ed_soft{S}_%=",
    "L_ed_acc{S}_%=:",
    "v_fmac_f64_e32 %[ax], {DX}, {T3}",
    "v_fmac_f64_e32 %[ay], {DY}, {T3}",
    "v_fmac_f64_e32 %[az], {DZ}, {T3}",
]
D_SOFT_HEAD = [
    "L_ed_soft{S}_%=:",
    "v_mul_f64 {RR}, {RR}, {RI}                      ;; r",
    "v_add_u32_e32 {T1LO}, s89, %[tyb]               ;; type byte of the entry",
    "ds_read_u8 {T1LO}, {T1LO}",
    "s_waitcnt lgkmcnt(0)",
    "v_lshl_add_u32 {T1LO}, {T1LO}, 3, %[fst]",
    "ds_read_b64 {TE}, {T1LO}                        ;; softening length of the source's type",
]


def direct():
    i0 = TRIP.index("L_er_soft_%=:")
    soft = TRIP[i0 + 7:TRIP.index("L_er_done_%=:")]      # from the wait for the softening length on: shared with ER_TRIP_ASM
    rx, ry = role_regs("x"), role_regs("y")
    rs = role_regs("x")
    rs["S"] = "s"
    seq = ["s_mov_b32 s89, 0",
           "ds_read_b128 %s, %%[ptr]" % rx["E0"],
           "ds_read_b128 %s, %%[ptr] offset:16" % rx["E1"],
           "s_bitcmp1_b32 %[n], 0                           ;; an odd count: one entry alone first, then pairs (one loop test per pair)",
           "s_cbranch_scc0 L_ed_pairs_%=",
           "s_waitcnt lgkmcnt(0)"]
    seq += fill(D_COMPUTE, rs)
    seq += ["s_mov_b32 s89, 1",
            "v_add_u32_e32 %[ptr], 32, %[ptr]",
            "ds_read_b128 %s, %%[ptr]                        ;; (with n = 1: one entry past the end, inside the pool, not used)" % rx["E0"],
            "ds_read_b128 %s, %%[ptr] offset:16" % rx["E1"],
            "L_ed_pairs_%=:",
            "s_cmp_lt_u32 s89, %[n]",
            "s_cbranch_scc0 L_ed_done_%=",
            "L_ed_topx_%=:",
            "s_waitcnt lgkmcnt(0)",
            "ds_read_b128 %s, %%[ptr] offset:32" % ry["E0"],
            "ds_read_b128 %s, %%[ptr] offset:48" % ry["E1"]]
    seq += fill(D_COMPUTE, rx)
    seq += ["s_waitcnt lgkmcnt(0)",
            "v_add_u32_e32 %[ptr], 64, %[ptr]",
            "ds_read_b128 %s, %%[ptr]                        ;; the next pair's first entry (past the end after the last pair: not used)" % rx["E0"],
            "ds_read_b128 %s, %%[ptr] offset:16" % rx["E1"]]
    seq += fill(D_COMPUTE, ry)
    seq += ["s_add_u32 s89, s89, 2",
            "s_cmp_lt_u32 s89, %[n]",
            "s_cbranch_scc1 L_ed_topx_%=",
            "s_branch L_ed_done_%="]
    for r in (rs, rx, ry):
        S = r["S"]
        head = list(D_SOFT_HEAD)
        if S == "y":   # the second entry of the pair
            head = [head[0], head[1], "s_add_u32 s88, s89, 1", head[2].replace("s89", "s88")] + head[3:]
        seq += fill(head, r)
        for l in soft:
            l = l.replace("L_er_acc_%=", "L_ed_acc%s_%%=" % S)
            seq.append(fill([l], r)[0])
    seq += ["L_ed_done_%=:",
            "s_waitcnt lgkmcnt(0)                            ;; the request that ran ahead must have landed before its registers are anybody else's"]
    return seq

# ---- tree-only force loop, two entries per trip as two interleaved instruction streams (ER_DIRECT2_ASM; -DGW_DIRECT=2) ---------------
def d2_regs(sfx):
    r = role_regs("x" if sfx in ("x", "s") else "y")
    if sfx == "y":
        base = 82
        names = {"R2": 0, "TT": 0, "RI": 2, "RR": 4, "T1": 6, "T2": 8, "T3": 10, "TE": 12}
        for k, o in names.items():
            r[k] = pair(base + o)
            r[k + "LO"] = "v%d" % (base + o)
            r[k + "HI"] = "v%d" % (base + o + 1)
    r["SOFT"] = "s[88:89]" if sfx == "y" else "s[94:95]"
    r["S"] = sfx + "2"
    return r


D2_NOBR = [l.replace("v_cmp_gt_f64_e32 vcc, %[h2max], {R2}", "v_cmp_gt_f64_e64 {SOFT}, %[h2max], {R2}") for l in D_COMPUTE[:D_COMPUTE.index("s_cbranch_vccnz L_ed_soft{S}_%=")]]
D2_ACC = ["s_cmp_lg_u64 {SOFT}, 0",
          "s_cbranch_scc1 L_ed_soft{S}_%=",
          "L_ed_acc{S}_%=:",
          "v_fmac_f64_e32 %[ax], {DX}, {T3}",
          "v_fmac_f64_e32 %[ay], {DY}, {T3}",
          "v_fmac_f64_e32 %[az], {DZ}, {T3}"]


def direct2():
    i0 = TRIP.index("L_er_soft_%=:")
    soft = TRIP[i0 + 7:TRIP.index("L_er_done_%=:")]
    rs, rx, ry = d2_regs("s"), d2_regs("x"), d2_regs("y")
    seq = ["s_mov_b32 s86, 0",
           "s_bitcmp1_b32 %[n], 0                           ;; an odd count: one entry alone first",
           "s_cbranch_scc0 L_ed_pairs_%=",
           "ds_read_b128 %s, %%[ptr]" % rs["E0"],
           "ds_read_b128 %s, %%[ptr] offset:16" % rs["E1"],
           "v_add_u32_e32 %[ptr], 32, %[ptr]",
           "s_waitcnt lgkmcnt(0)"]
    seq += fill(D2_NOBR, rs) + fill(D2_ACC, rs)
    seq += ["s_mov_b32 s86, 1",
            "L_ed_pairs_%=:",
            "s_cmp_lt_u32 s86, %[n]",
            "s_cbranch_scc0 L_ed_done_%=",
            "L_ed_top_%=:",
            "ds_read_b128 %s, %%[ptr]" % rx["E0"],
            "ds_read_b128 %s, %%[ptr] offset:16" % rx["E1"],
            "ds_read_b128 %s, %%[ptr] offset:32" % ry["E0"],
            "ds_read_b128 %s, %%[ptr] offset:48" % ry["E1"],
            "v_add_u32_e32 %[ptr], 64, %[ptr]",
            "s_waitcnt lgkmcnt(0)"]
    seq += zipl(fill(D2_NOBR, rx), fill(D2_NOBR, ry))
    seq += fill(D2_ACC, rx) + fill(D2_ACC, ry)
    seq += ["s_add_u32 s86, s86, 2",
            "s_cmp_lt_u32 s86, %[n]",
            "s_cbranch_scc1 L_ed_top_%=",
            "s_branch L_ed_done_%="]
    for r in (rs, rx, ry):
        S = r["S"]
        head = list(D_SOFT_HEAD)
        if S == "y2":   # the second entry of the pair
            head = [head[0], head[1], "s_add_u32 s87, s86, 1", head[2].replace("s89", "s87")] + head[3:]
        else:
            head = [h.replace("s89", "s86") for h in head]   # (s[88:89] is the second stream's softening flag: the counter lives in s86)
        seq += fill(head, r)
        for l in soft:
            l = l.replace("L_er_acc_%=", "L_ed_acc%s_%%=" % S)
            seq.append(fill([l], r)[0])
    seq += ["L_ed_done_%=:"]
    return seq


# ---- trip loop, taken branches off the common path (ER_TRIP4_ASM; -DER_ES=4) ---------------------------------------------------------
# ER_TRIP_ASM takes two branches on every trip: the jump over the four instructions of "beyond the exact cut" and the jump back to the
# top.  Here the rare block is out of line (the common case falls through) and the loop is unrolled twice (one jump back per two trips).
def trip4():
    top = TRIP.index("L_er_top_%=:")
    cut = TRIP.index("s_cbranch_scc0 L_er_incut_%=")
    incut = TRIP.index("L_er_incut_%=:")
    back = TRIP.index("s_branch L_er_top_%=")
    soft0 = TRIP.index("L_er_soft_%=:")
    done = TRIP.index("L_er_done_%=:")
    rare = TRIP[cut + 1:incut]
    seq = [TRIP[0]]

    def relabel(l, k):
        for name in ("top", "incut", "acc", "soft", "rare"):
            l = l.replace("L_er_%s_%%=" % name, "L_er_%s%s_%%=" % (name, k))
        return l
    for k in ("a", "b"):
        body = TRIP[top:cut] + ["s_cbranch_scc1 L_er_rare_%=", "L_er_incut_%=:"] + TRIP[incut + 1:back]
        body = [l for l in body if not l.startswith("s_add_u32 %[ntr]")]      # trips are counted per pair
        if k == "b":
            # the exit test ("no lane is on the oldest slot") once per pair of trips: a trip taken after the slot has emptied is an ordinary
            # trip of the lanes on the newer slots (one idle trip at the end of a list, where all lanes are parked)
            body = [l for l in body if not l.startswith("v_cmp_eq_u32_e32 vcc, %[tail], %[q]") and not l.startswith("s_cbranch_vccz L_er_done_%=")]
        seq += [relabel(l, k) for l in body]
    seq += ["s_add_u32 %[ntr], %[ntr], 2", "s_branch L_er_topa_%="]
    for k in ("a", "b"):
        seq += ["L_er_rare%s_%%=:" % k] + rare + ["s_branch L_er_incut%s_%%=" % k]
        seq += [relabel(l, k) for l in TRIP[soft0:done]]
    seq += ["L_er_done_%=:"]
    return seq

# ---- cull --------------------------------------------------------------------------------------------------------------------
CULL_WRAP = [
    "v_mul_f64 {B0}, {EX}, %[invbox]                 ;; nearest image: x - box rint(x / box)",
    "v_mul_f64 {B1}, {EY}, %[invbox]",
    "v_mul_f64 {B2}, {EZ}, %[invbox]",
    "v_rndne_f64_e32 {B0}, {B0}",
    "v_rndne_f64_e32 {B1}, {B1}",
    "v_rndne_f64_e32 {B2}, {B2}",
    "v_fma_f64 {EX}, -{B0}, %[box], {EX}",
    "v_fma_f64 {EY}, -{B1}, %[box], {EY}",
    "v_fma_f64 {EZ}, -{B2}, %[box], {EZ}",
]
CULL = [
    "v_add_f64 {EX}, %[rx], -%[bcx]",
    "v_add_f64 {EY}, %[ry], -%[bcy]",
    "v_add_f64 {EZ}, %[rz], -%[bcz]",
    "@WRAPSEG",
    "v_add_f64 {B0}, |{EX}|, -%[bhx]                 ;; a source farther than the cut from the whole box ...",
    "v_add_f64 {B1}, |{EY}|, -%[bhy]",
    "v_add_f64 {B2}, |{EZ}|, -%[bhz]",
    "v_max_f64 {B0}, {B0}, 0",
    "v_max_f64 {B1}, {B1}, 0",
    "v_max_f64 {B2}, {B2}, 0",
    "v_mul_f64 {B0}, {B0}, {B0}",
    "v_fma_f64 {B0}, {B1}, {B1}, {B0}",
    "v_fma_f64 {B0}, {B2}, {B2}, {B0}",
    "v_cmp_gt_f64_e32 vcc, %[reach2], {B0}           ;; ... contributes to no target",
    "v_cmp_neq_f64_e64 s[90:91], 0, %[rw]            ;; nor does a massless one (empty species of a node)",
    "s_and_b64 s[90:91], s[90:91], vcc",
    "s_bcnt1_i32_b64 %[cnt], s[90:91]",
    "v_mbcnt_lo_u32_b32 {L}, s90, 0",
    "v_mbcnt_hi_u32_b32 {L}, s91, {L}",
    "v_add_u32_e32 {L}, %[wrpos], {L}                ;; position among the waiting entries",
    "s_and_saveexec_b64 s[92:93], s[90:91]",
    "v_cmp_gt_u32_e32 vcc, 32, {L}                   ;; (vcc is a scalar operand too: the slots go through VGPRs)",
    "v_mov_b32_e32 {SQ}, %[q0]",
    "v_mov_b32_e32 {SQ2}, %[q1]",
    "v_cndmask_b32_e32 {SQ}, {SQ2}, {SQ}, vcc",
    "v_cmp_gt_u32_e32 vcc, 64, {L}",
    "v_mov_b32_e32 {SQ2}, %[q2]",
    "v_cndmask_b32_e32 {SQ}, {SQ2}, {SQ}, vcc",
    "v_and_b32_e32 {EA}, 31, {L}",
    "v_lshl_add_u32 {EA}, {EA}, 4, {SQ}",
    "ds_write_b128 {EA}, {EXY} offset:320",
    "ds_write_b64 {EA}, {EZ} offset:848",
    "ds_write_b64 {EA}, %[rw] offset:856",
    "s_mov_b64 exec, s[92:93]",
]


def main():
    hdr = ("// eval_asm.inc -- generated by tools/gen_eval_asm.py (named registers -> numbers); do not edit by hand.\n"
           "// The gfx950 assembly blocks of k_eval_ring (kernels_eval.hip): temporaries v[104:127], s[90:95] (clobbers).\n\n")
    out = [hdr]
    out.append("#define ER_FST_OFF_ET \"16640\"   /* fsT behind the exp(-ym r_bin) table: NTAB * 8 + 32 * 8 */\n#define ER_FST_OFF_NOET \"256\"\n")
    out.append(macro("ER_YUK_ET", YUK_ET))
    out.append(macro("ER_NOYUK", NOYUK))
    out.append(macro("ER_TRIP_ASM", TRIP, "(YUKSEG, FSTOFF)"))
    out.append(clobbers("ER_TRIP_CLOBBERS", range(104, 128), range(90, 96)))
    out.append(macro("ER_TRIP2_YUK_ASM", trip2(True), "(FSTOFF)"))
    out.append(macro("ER_TRIP2_NOYUK_ASM", trip2(False), "(FSTOFF)"))
    out.append(clobbers("ER_TRIP2_CLOBBERS", range(80, 128), range(86, 96)))
    out.append(macro("ER_TRIP3_YUK_ASM", trip3(True), "(FSTOFF)"))
    out.append(macro("ER_TRIP3_NOYUK_ASM", trip3(False), "(FSTOFF)"))
    out.append(clobbers("ER_TRIP3_CLOBBERS", range(94, 128), range(86, 96)))
    out.append(macro("ER_DIRECT_ASM", direct()))
    out.append(clobbers("ER_DIRECT_CLOBBERS", range(96, 128), range(88, 96)))
    out.append(macro("ER_DIRECT2_ASM", direct2()))
    out.append(clobbers("ER_DIRECT2_CLOBBERS", range(82, 128), range(86, 96)))
    out.append(macro("ER_TRIP4_ASM", trip4(), "(YUKSEG, FSTOFF)"))
    out.append(macro("ER_CULL_WRAP", CULL_WRAP))
    out.append(macro("ER_CULL_ASM", CULL, "(WRAPSEG)"))
    out.append(clobbers("ER_CULL_CLOBBERS", range(104, 120), range(90, 94)))
    with open(OUT, "w") as f:
        f.write("\n".join(out))
    print("wrote", OUT)


main()
