"""Turn the rocprofv3 outputs of scripts/profile_bench.sh (gpurun_out/prof_<tag>) into the small tracked summaries under
profiles/<round>/ (BTBA_PROFILE_ROUND, default r06) and add / replace that workload's record in profiles/sweep_counters.json -- what bench.py quotes as roofline.traffic /
roofline.valu_issue.  A record is keyed by a hash of the kernel sources AND by the whole workload (config, instances, distinct
instances, mask, cache and correspondence layout): bench.py quotes it only for a line measured on exactly that.
    python scripts/summarize_profiles.py <tag> <version> [--masked] [--config c3] [--instances 32] [--distinct 32] [--entryj]"""
import collections, csv, glob, json, os, shutil, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
ROUND = os.environ.get("BTBA_PROFILE_ROUND", "r06")


def agg(path):
    a = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(path)):
        a[r["Kernel_Name"].split("(")[0].replace("void ", "")][r["Counter_Name"]].append(float(r["Counter_Value"]))
    return a


def arg(name, default):
    return sys.argv[sys.argv.index(name) + 1] if name in sys.argv else default


def main():
    import bench
    tag, ver = sys.argv[1], sys.argv[2]
    masked, entryj = "--masked" in sys.argv, "--entryj" in sys.argv
    config, B, distinct = arg("--config", "c3"), int(arg("--instances", "32")), int(arg("--distinct", "32"))
    dominant = "k_sparse_sweep" if bench.CONFIGS[config]["w_dense"] == 0.0 else "k_fused_sweeps"
    prof, out = os.path.join(ROOT, "gpurun_out", "prof_" + tag), os.path.join(ROOT, "profiles", ROUND)
    os.makedirs(out, exist_ok=True)
    sfx = f"{config}x{B}_{ver}" + ("_masked" if masked else "") + ("_entryj" if entryj else "")
    cmdline = "python bench.py " + open(os.path.join(prof, "args.txt")).read().replace("bench args:", "").strip()
    shutil.copy(os.path.join(prof, "stats", "bench_kernel_stats.csv"), os.path.join(out, f"bench_{sfx}_kernel_stats.csv"))
    f = agg(glob.glob(os.path.join(prof, "fetch", "*counter_collection.csv"))[0])
    w = agg(glob.glob(os.path.join(prof, "write", "*counter_collection.csv"))[0])
    dom = None
    hbm_csv = f"profiles/{ROUND}/bench_{sfx}_pmc_fetch_write.csv"
    with open(os.path.join(ROOT, hbm_csv), "w") as o:
        o.write("# rocprofv3 --kernel-trace --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, scripts/profile_bench.sh); HBM bytes per launch =\n"
                "# FETCH_SIZE[KiB] x 1024 x 2 (gfx950 correction) + WRITE_SIZE[KiB] x 1024, MI355X_MICROARCH.md HBM section\n"
                f"# workload: {cmdline} --steps 3 --warmup 1 --no-cpu-baseline --no-kernel-timing  (distinct instances: {distinct})\n")
        o.write("kernel,launches,FETCH_SIZE_KiB_mean,WRITE_SIZE_KiB_mean,hbm_bytes_per_launch_corrected\n")
        for k in f:
            fs, ws = f[k]["FETCH_SIZE"], w[k]["WRITE_SIZE"]
            fm, wm = sum(fs) / len(fs), sum(ws) / len(ws)
            b = int(fm * 1024 * 2 + wm * 1024)
            o.write(f"\"{k}\",{len(fs)},{fm:.3f},{wm:.3f},{b}\n")
            if dominant in k and (dom is None or b > dom[1]):      # (bench.py's tracker_call leg launches the masked instantiation of the same kernel for ONE window: the benched one moves the most bytes)
                dom = (k, b)
    sq = agg(glob.glob(os.path.join(prof, "sq", "*counter_collection.csv"))[0])
    sq_csv = f"profiles/{ROUND}/bench_{sfx}_sq_counters.csv"
    summary = {}
    with open(os.path.join(ROOT, sq_csv), "w") as o:
        o.write("# rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VALU2 SQ_BUSY_CYCLES SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE\n"
                f"# -- {cmdline} --steps 3 --warmup 1 --no-cpu-baseline --no-kernel-timing  (distinct instances: {distinct}; scripts/profile_bench.sh); mean per launch, summed over the chip.\n"
                "# Units (profiles/r02/valu_calibration.md): *_INST_* and SQ_WAVE_CYCLES in quad-cycles, SQ_BUSY_CYCLES = shader cycles x 32 SEs, GRBM_GUI_ACTIVE = cycles x 8 XCDs.\n"
                "# derived: kernel_cycles = SQ_BUSY_CYCLES / 32; valu_busy_frac = 4 (SQ_ACTIVE_INST_VALU - SQ_ACTIVE_INST_VALU2) / 1024 / kernel_cycles\n")
        o.write("kernel,counter,launches,mean_per_launch\n")
        for k, v in sq.items():
            if "btba" not in k:
                continue
            m = {c: sum(x) / len(x) for c, x in v.items()}
            for c, x in sorted(v.items()):
                o.write(f"\"{k}\",{c},{len(x)},{m[c]:.1f}\n")
            cyc = m["SQ_BUSY_CYCLES"] / 32.0
            busy = 4.0 * (m["SQ_ACTIVE_INST_VALU"] - m["SQ_ACTIVE_INST_VALU2"]) / 1024.0
            d = {"kernel_cycles": cyc, "valu_busy_frac": busy / cyc, "valu_cycles_per_inst": busy / max(m["SQ_INSTS_VALU"] / 1024.0, 1e-9),
                 "valu_dual_issued_frac": 2.0 * m["SQ_ACTIVE_INST_VALU2"] / max(m["SQ_INSTS_VALU"], 1e-9), "waves_per_simd": 4.0 * m["SQ_WAVE_CYCLES"] / 1024.0 / cyc}
            for c, x in d.items():
                o.write(f"\"{k}\",derived_{c},{len(v['SQ_INSTS_VALU'])},{x:.4f}\n")
            if dom and k == dom[0]:
                summary = d
    if dom:
        rec = {"config": config, "instances": B, "distinct": distinct, "kernel": dom[0], "fused": dominant == "k_fused_sweeps", "masked": masked, "float4_cache": False,
               "entryj": entryj, "kernel_source_hash": bench.kernel_source_hash(), "hbm_bytes_per_launch": dom[1],
               "valu_busy_frac": round(summary.get("valu_busy_frac", 0), 4) if summary else None,
               "valu_cycles_per_inst": round(summary.get("valu_cycles_per_inst", 0), 3) if summary else None,
               "valu_dual_issued_frac": round(summary.get("valu_dual_issued_frac", 0), 3) if summary else None,
               "waves_per_simd": round(summary.get("waves_per_simd", 0), 2) if summary else None,
               "source_hbm": hbm_csv, "source_sq": sq_csv, "source_stats": f"profiles/{ROUND}/bench_{sfx}_kernel_stats.csv"}
        path = os.path.join(ROOT, "profiles", "sweep_counters.json")
        try:
            old = json.load(open(path))
            recs = old.get("records", []) if isinstance(old, dict) else []
        except Exception:
            recs = []
        same = lambda r: all(r.get(k) == rec[k] for k in ("config", "instances", "distinct", "masked", "float4_cache", "entryj", "fused"))
        recs = [r for r in recs if not same(r) and r.get("kernel_source_hash") == rec["kernel_source_hash"]] + [rec]      # records of other kernel sources are dropped
        json.dump({"records": recs}, open(path, "w"), indent=1)
    print("dominant kernel", dom, summary)


if __name__ == "__main__":
    main()
