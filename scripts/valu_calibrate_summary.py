"""Turns the outputs of scripts/r02/call2.sh (gpurun_out/r02_cal: valu_calibrate plain run + three PMC passes) into the tracked
summaries profiles/r02/valu_calibration.{md,jsonl} and profiles/r02/valu_calibration_pmc.csv.
    python scripts/valu_calibrate_summary.py [gpurun_out/r02_cal]"""
import collections, csv, glob, json, os, shutil, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "gpurun_out", "r02_cal")
out = os.path.join(ROOT, "profiles", "r02")
os.makedirs(out, exist_ok=True)
shutil.copy(os.path.join(src, "cal_plain.jsonl"), os.path.join(out, "valu_calibration.jsonl"))

rows = collections.OrderedDict()
hdr = None
for l in open(os.path.join(src, "cal_plain.jsonl")):
    d = json.loads(l)
    if "inst" not in d:
        hdr = d
        continue
    rows.setdefault(d["inst"], {})[d["waves_per_simd_nominal"]] = d

# PMC: per kernel, counters normalised by the known instruction count
pmc = collections.defaultdict(dict)
meta = {}
for i in (1, 2, 3):
    js = {}
    p = os.path.join(src, f"cal_pmc{i}.jsonl")
    if not os.path.exists(p):
        continue
    for l in open(p):
        d = json.loads(l)
        if "inst" in d:
            js[d["kernel"]] = d
    for f in glob.glob(os.path.join(src, f"cal_pmc{i}", "*counter_collection.csv")):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].replace("void ", "").split("(")[0]
            if k in js:
                pmc[k][r["Counter_Name"]] = float(r["Counter_Value"])
                meta[k] = js[k]
with open(os.path.join(out, "valu_calibration_pmc.csv"), "w") as o:
    o.write("# rocprofv3 --kernel-trace --pmc <8 counters> -- valu_calibrate --pmc (4 waves per SIMD on all 1024 SIMDs); one launch per kernel.\n")
    o.write("# n = waves x instructions per wave of the measured kind; wave_cycles = mean s_memtime cycles per wave in the same launch\n")
    o.write("kernel,instruction,n,counter,value,value_per_n\n")
    for k, v in pmc.items():
        n = meta[k]["waves"] * meta[k]["inst_per_wave"]
        for c, x in sorted(v.items()):
            o.write(f"\"{k}\",\"{meta[k]['inst']}\",{n},{c},{x:.0f},{x / n:.4f}\n")


def pm(k_inst, c):
    for k, m in meta.items():
        if m["inst"] == k_inst and c in pmc[k]:
            return pmc[k][c], m
    return None, None


with open(os.path.join(out, "valu_calibration.md"), "w") as o:
    w = o.write
    w("# VALU issue costs and SQ counter units on the MI355X (gfx950) box -- measured, round 2\n\n")
    w("Source: `scripts/valu_calibrate.hip` run by `scripts/r02/call2.sh` on a gpurun box; raw lines in `valu_calibration.jsonl`, counters in\n"
      "`valu_calibration_pmc.csv`.  Every kernel executes a known number of ONE instruction per wave (16 independent destination registers,\n"
      "128 instructions per loop trip), brackets it with `s_memtime` and records its SIMD (`HW_REG_HW_ID`, `HW_REG_XCC_ID`).  Cost = window from a\n"
      "SIMD's first wave start to its last wave end / instructions that SIMD issued, median over the 1 024 SIMDs.\n\n")
    w(f"Device: {hdr}\n\n")
    w("## 1. Cycles per wave64 instruction, per SIMD\n\n| instruction | 1 wave/SIMD | 2 | 4 | 8 | shader GHz (8) |\n|---|---|---|---|---|---|\n")
    for k, v in rows.items():
        w(f"| `{k}` | " + " | ".join(f"{v[n]['cycles_per_inst_per_simd_median']:.2f}" if n in v else "-" for n in (1, 2, 4, 8)) + f" | {v[8]['shader_GHz'] if 8 in v else 0:.2f} |\n")
    w("\nReading (the rules `scripts/isa_cost.py` prices a loop with):\n\n"
      "* One wave issues at most one VALU instruction per 4 cycles (column 1: 4.2-4.4 incl. loop overhead).  A SIMD issues up to TWO per 4 cycles, from two\n"
      "  different waves: `v_mul/add/sub_f32`, `v_mov`, `v_and/or/xor`, `v_add/sub_u32`, right shifts, `v_fmamk/fmaak`, VOP3-encoded `v_mul_f32` reach **2.1 cycles**.\n"
      "  fp32 vector peak = 2 flop x 64 lanes / 2.1 cycles per SIMD: the 157.3 TFLOP/s of MI355X_MICROARCH.md is the UNPACKED `v_fma_f32` rate; `v_pk_fma_f32`\n"
      "  does two FMAs per lane in 4.1 cycles -- the same flops per cycle, not more (r01's 78.6 TF 'unpacked peak' was wrong).\n"
      "* **4 cycles** (never two per quad-cycle among themselves): `v_cmp*`, `v_cndmask*` (e64; the e32 form reading vcc is worse, below), `v_cvt*`, `v_floor/fract/trunc`,\n"
      "  `v_min/max/med3` (int and float), `v_lshlrev_b32`, all VOP3 integer ops (`v_lshl_add_u32`, `v_add_lshl_u32`, `v_mad_u32_u24`, `v_mul_lo_u32`, `v_mul_u32_u24`,\n"
      "  `v_bfe`, `v_bfi`, `v_and_or`, `v_add3`, `v_add_co`/`v_addc_co`), 64-bit integer ops, `v_readfirstlane`, DPP forms, `v_pk_*_f32` -- and ANY VALU op with an\n"
      "  SGPR source operand (`y: v_mul_f32 D,s20,v17` 4.03 vs 2.08 with two VGPRs).  In MIXED code such an op pairs with a plain one of another wave\n"
      "  (`v_cmp + v_mul + v_cndmask + v_mul`: 2.1 per instruction): the cost is a lost pairing slot, not a fixed 4 cycles.\n"
      "* **8 cycles**: `v_rcp/rsq/sqrt/exp_f32`.\n"
      "* `v_fma_f32` / `v_fmac_f32` whose destination is also a source: 2.1 cycles when the two other sources are VGPRs of different parity, **4.0** when they\n"
      "  have the same parity or are the same register (`x:` rows).  With the destination not among the sources: always 2.1.\n"
      "* `v_cndmask_b32_e32` reading `vcc` costs ~16-19 cycles when several follow one compare (`x: v_cndmask ... after one v_cmp per 16`); interleaved with\n"
      "  other work it hides.  `global_load_dwordx4` (L1 hits): 64 cycles per wave-instruction per SIMD = 64 B/clk per CU.  `ds_read_b32`: 8.2.\n\n")
    w("## 2. What the SQ counters count (pinned against the known instruction counts)\n\n| counter | measured | unit |\n|---|---|---|\n")
    a, m = pm("v_mul_f32", "SQ_INSTS_VALU")
    if a:
        n = m["waves"] * m["inst_per_wave"]
        w(f"| `SQ_INSTS_VALU` | {a / n:.4f} per executed VALU instruction (`v_mul_f32`) | wave-instructions, summed over the chip |\n")
        b, _ = pm("v_mul_f32", "SQ_ACTIVE_INST_VALU"); c, _ = pm("v_rcp_f32", "SQ_ACTIVE_INST_VALU"); d2, _ = pm("v_pk_fma_f32", "SQ_ACTIVE_INST_VALU")
        w(f"| `SQ_ACTIVE_INST_VALU` | {b / n:.3f} per `v_mul_f32` (2 cycles), {d2 / n:.3f} per `v_pk_fma_f32` (4 cycles), {c / n:.3f} per `v_rcp_f32` (8 cycles) | quad-cycles of ISSUE SLOTS: 1 per instruction, 2 per transcendental -- NOT pipe-busy time |\n")
        e, _ = pm("v_mul_f32", "SQ_ACTIVE_INST_VALU2"); f, _ = pm("v_pk_fma_f32", "SQ_ACTIVE_INST_VALU2"); g, _ = pm("x: v_fma_f32 D,D,v16,v18 (sources both in even banks)", "SQ_ACTIVE_INST_VALU2")
        w(f"| `SQ_ACTIVE_INST_VALU2` | {e / n:.3f} per `v_mul_f32`, {f / n:.3f} per `v_pk_fma_f32`, {g / n:.3f} per parity-conflicting `v_fma_f32` | quad-cycles in which TWO VALU instructions issued on a SIMD |\n")
        wc, _ = pm("v_mul_f32", "SQ_WAVE_CYCLES")
        w(f"| `SQ_WAVE_CYCLES` | {wc / (m['waves'] * m['wave_cycles_mean']):.4f} x (sum over waves of s_memtime cycles) | quad-cycles |\n")
        bc, _ = pm("v_mul_f32", "SQ_BUSY_CYCLES"); gg, _ = pm("v_mul_f32", "GRBM_GUI_ACTIVE")
        w(f"| `SQ_BUSY_CYCLES` | {bc / m['wave_cycles_max']:.2f} x longest wave's cycles | shader cycles x 32 shader engines |\n")
        w(f"| `GRBM_GUI_ACTIVE` | {gg / m['wave_cycles_max']:.2f} x longest wave's cycles | shader cycles x 8 XCDs (includes launch ramp) |\n")
    w("\n**VALU pipe busy time of a kernel** (used by bench.py's `roofline.valu_issue`):  busy quad-cycles = `SQ_ACTIVE_INST_VALU - SQ_ACTIVE_INST_VALU2`\n"
      "(every issue slot minus the quad-cycles that carried two), i.e. `frac = 4 (SQ_ACTIVE_INST_VALU - SQ_ACTIVE_INST_VALU2) / 1024 SIMDs / (SQ_BUSY_CYCLES / 32)`.\n"
      "Check on the calibration kernels: `v_mul_f32` 4 x (1 - 0.457) = 2.17 cycles per instruction (timed: 2.19), parity-conflicting `v_fma_f32` 3.98 (timed 4.00), `v_rcp_f32` 8.0 (8.04).\n")
print("wrote", out)
