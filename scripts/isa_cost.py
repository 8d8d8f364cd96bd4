#!/usr/bin/env python
"""Static VALU-issue cost of a kernel's hot loop, priced with the per-instruction costs measured by
scripts/valu_calibrate.hip on the MI355X box (profiles/r02/valu_calibration.md).

    python scripts/isa_cost.py <file.s> <kernel-name-substring> [--loop N] [--list]

Finds the kernel in hipcc's `-save-temps` assembly, takes its loop with the most FMAs (or the N-th largest) and
prices every VALU instruction of it:

    2 cycles   plain fp32 / logic / add-sub / right-shift ops with VGPR, inline-constant or literal sources
    4 cycles   * any VALU op with an SGPR (or vcc / exec) SOURCE operand
               * v_cmp*, v_cndmask*, v_cvt*, v_floor / v_fract / v_trunc / v_ceil / v_rndne, v_min* / v_max* / v_med3*,
                 v_lshlrev_b32, every VOP3-only integer op (v_lshl_add_u32, v_add_lshl_u32, v_mad_*, v_mul_lo_u32,
                 v_mul_u32_u24, v_bfe, v_bfi, v_and_or, v_add3, v_add_co / v_addc_co), 64-bit integer ops,
                 v_readfirstlane, DPP forms, v_pk_*_f32
               * v_fmac_f32 / v_fma_f32 whose destination is one of its sources while the other two sources are
                 VGPRs of the same parity (or the same register)
    8 cycles   transcendentals (v_rcp, v_rsq, v_sqrt, v_exp, v_log, v_sin, v_cos)

This is an issue-slot model (what SQ_ACTIVE_INST_VALU - SQ_ACTIVE_INST_VALU2 counts in quad-cycles), not a
latency model.  Development tool: gives the relative cost of two builds without a GPU run."""
from __future__ import annotations

import collections
import re
import sys

FOUR = ("v_cmp", "v_cndmask", "v_cvt", "v_floor", "v_fract", "v_trunc", "v_ceil", "v_rndne", "v_min", "v_max", "v_med3", "v_lshlrev_b32",
        "v_lshl_add", "v_add_lshl", "v_mad_", "v_mul_lo", "v_mul_hi", "v_mul_u32_u24", "v_mul_i32_i24", "v_bfe", "v_bfi", "v_and_or", "v_or3", "v_add3", "v_add_co",
        "v_addc_co", "v_sub_co", "v_subb_co", "v_readfirstlane", "v_readlane", "v_writelane", "v_pk_", "v_lshlrev_b64", "v_lshrrev_b64", "v_xad", "v_perm", "v_alignbit", "v_mbcnt",
        "v_ldexp", "v_frexp", "v_sad", "v_lerp", "v_cubeid", "v_div_", "v_mov_b64", "v_fma_f64", "v_mul_f64", "v_add_f64", "v_cvt_f64")
EIGHT = ("v_rcp", "v_rsq", "v_sqrt", "v_exp", "v_log", "v_sin", "v_cos")
SREG = re.compile(r"^[-|]*(s\d+|s\[\d+:\d+\]|vcc|exec|vcc_lo|vcc_hi|exec_lo|exec_hi|m0|ttmp\d+)\|?$")
VREG = re.compile(r"^[-|]*v(\d+)\|?$")


def operands(line):
    op, _, rest = line.partition(" ")
    args = [a.strip() for a in rest.split(",")] if rest.strip() else []
    # drop modifiers that follow the last operand ("v1 clamp", "v2 row_shr:1 ...")
    cleaned = []
    for a in args:
        cleaned.append(a.split()[0] if a else a)
    mods = " ".join(" ".join(a.split()[1:]) for a in args)
    return op, cleaned, mods


def cost(line):
    op, args, mods = operands(line)
    base = op.replace("_e32", "").replace("_e64", "").replace("_dpp", "").replace("_sdwa", "")
    if any(base.startswith(p) for p in EIGHT):
        return 8, "trans"
    if op.endswith("_dpp") or "quad_perm" in mods or "row_" in mods:
        return 4, "dpp"
    if any(base.startswith(p) for p in FOUR):
        return 4, "half-rate op"
    n_dst = 2 if (base.startswith("v_add_co") or base.startswith("v_addc") or base.startswith("v_mad_u64") or base.startswith("v_div_scale")) else 1
    srcs = args[n_dst:]
    if any(SREG.match(a) for a in srcs):
        return 4, "sgpr source"
    if base.startswith("v_fmac_f32") or base.startswith("v_fma_f32") or base.startswith("v_mac_f32") or base.startswith("v_mad_f32"):
        dst = args[0]
        s = list(srcs) + ([dst] if base.startswith("v_fmac") or base.startswith("v_mac") else [])
        regs = [VREG.match(a) for a in s]
        names = [m.group(1) if m else None for m in regs]
        d = VREG.match(dst).group(1)
        if d in names and len(s) == 3:
            others = [n for n, a in zip(names, s) if n != d]
            if len(others) == 2 and all(o is not None for o in others) and (int(others[0]) % 2 == int(others[1]) % 2):
                return 4, "fma operand-pair parity"
        return 2, "fma"
    return 2, "full rate"


def loops_of(body):
    """(start, end) line indices of backward-branch loops."""
    labels = {}
    for i, l in enumerate(body):
        m = re.match(r"^(\.LBB\d+_\d+):", l)
        if m:
            labels[m.group(1)] = i
    out = []
    for i, l in enumerate(body):
        m = re.match(r"^\s*s_c?branch\w*\s+(\.LBB\d+_\d+)", l)
        if m and m.group(1) in labels and labels[m.group(1)] < i:
            out.append((labels[m.group(1)], i))
    return out


def main():
    path, kern = sys.argv[1], sys.argv[2]
    which = int(sys.argv[sys.argv.index("--loop") + 1]) if "--loop" in sys.argv else 0
    lines = open(path).read().split("\n")
    start = next(i for i, l in enumerate(lines) if re.match(r"^_Z\w*" + re.escape(kern) + r"\w*:", l))
    end = next(i for i in range(start, len(lines)) if lines[i].startswith(".Lfunc_end"))        # a kernel can hold several s_endpgm
    body = [l.split(";")[0].rstrip() for l in lines[start:end]]
    cands = []
    for a, b in loops_of(body):
        ins = [l.strip() for l in body[a:b + 1] if l.strip() and not l.strip().startswith(".") and not l.strip().endswith(":")]
        nf = sum(1 for l in ins if l.startswith("v_fma"))
        cands.append((nf, a, b, ins))
    cands.sort(key=lambda c: -c[0])
    nf, a, b, ins = cands[which]
    valu = [l for l in ins if l.startswith("v_")]
    tot = 0
    why = collections.Counter()
    whyc = collections.Counter()
    for l in valu:
        c, w = cost(l)
        tot += c
        why[w] += 1
        whyc[w] += c
        if "--list" in sys.argv:
            print(f"{c}  {w:26s} {l}")
    other = collections.Counter(l.split()[0].rstrip("_e32").split("_")[0] + "_" + l.split()[0].split("_")[1] for l in ins if not l.startswith("v_"))
    print(f"kernel {kern}: loop lines {a}..{b}: {len(valu)} VALU instructions, {tot} issue cycles (model), {tot / max(len(valu), 1):.2f} cycles/instruction")
    for w, n in why.most_common():
        print(f"   {w:28s} {n:4d} instructions {whyc[w]:5d} cycles")
    print("   non-VALU:", dict(other))


if __name__ == "__main__":
    main()
