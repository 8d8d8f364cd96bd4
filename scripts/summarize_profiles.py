"""Turn the rocprofv3 outputs of scripts/profile_bench.sh / scripts/pmc_sq.sh (gpurun_out/prof_<tag>, gpurun_out/pmc_<tag>)
into the small tracked summaries under profiles/r01/ and refresh profiles/dense_sweep_traffic.json (read by bench.py).
    python scripts/summarize_profiles.py <tag> <version> [--fused] [--masked]"""
import collections, csv, glob, json, os, shutil, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def agg(path):
    a = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(path)):
        a[r["Kernel_Name"].split("(")[0].replace("void ", "")][r["Counter_Name"]].append(float(r["Counter_Value"]))
    return a


def main():
    tag, ver = sys.argv[1], sys.argv[2]
    fused, masked = "--fused" in sys.argv, "--masked" in sys.argv
    prof, out = os.path.join(ROOT, "gpurun_out", "prof_" + tag), os.path.join(ROOT, "profiles", "r01")
    sfx = f"{ver}_{'fused' if fused else 'separate'}"
    shutil.copy(os.path.join(prof, "stats", "bench_kernel_stats.csv"), os.path.join(out, f"bench_c3x32_kernel_stats_{sfx}.csv"))
    f = agg(glob.glob(os.path.join(prof, "fetch", "*counter_collection.csv"))[0])
    w = agg(glob.glob(os.path.join(prof, "write", "*counter_collection.csv"))[0])
    dom = None
    with open(os.path.join(out, f"bench_c3x32_pmc_fetch_write_{sfx}.csv"), "w") as o:
        o.write("kernel,launches,FETCH_SIZE_KiB_mean,WRITE_SIZE_KiB_mean,hbm_bytes_per_launch_corrected\n")
        for k in f:
            fs, ws = f[k]["FETCH_SIZE"], w[k]["WRITE_SIZE"]
            fm, wm = sum(fs) / len(fs), sum(ws) / len(ws)
            b = int(fm * 1024 * 2 + wm * 1024)            # MI355X_MICROARCH.md HBM section: KiB units, FETCH x2 on gfx950
            o.write(f"\"{k}\",{len(fs)},{fm:.3f},{wm:.3f},{b}\n")
            if ("k_fused_sweeps" in k) if fused else ("k_dense_sweep" in k):
                dom = (k, b)
    pmc = os.path.join(ROOT, "gpurun_out", "pmc_" + tag)
    if os.path.isdir(pmc):
        with open(os.path.join(out, f"sq_counters_c3x32_{sfx}.csv"), "w") as o:
            o.write("# rocprofv3 --kernel-trace --pmc <group> -- python bench.py --steps 2 --warmup 1 --distinct 2 --no-cpu-baseline --no-kernel-timing "
                    "(scripts/pmc_sq.sh); mean per launch; SQ_* cycle counters are in quad-cycles, summed over the chip\n")
            o.write("kernel,counter,launches,mean_per_launch\n")
            for p in sorted(glob.glob(os.path.join(pmc, "*_p*", "*counter_collection.csv"))):
                for k, v in agg(p).items():
                    if "btba" in k:
                        for c, x in sorted(v.items()):
                            o.write(f"\"{k}\",{c},{len(x)},{sum(x) / len(x):.1f}\n")
    if dom:
        json.dump({"config": "c3", "instances": 32, "kernel": dom[0], "hbm_bytes_per_launch": dom[1], "fused": fused, "masked": masked,
                   "float4_cache": False,
                   "source": f"profiles/r01/bench_c3x32_pmc_fetch_write_{sfx}.csv (rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE, separate passes; "
                             "FETCH_SIZE[KiB] x 1024 x 2 + WRITE_SIZE[KiB] x 1024, MI355X_MICROARCH.md HBM section)"},
                  open(os.path.join(ROOT, "profiles", "dense_sweep_traffic.json"), "w"), indent=1)
    print("dominant kernel", dom)


if __name__ == "__main__":
    main()
