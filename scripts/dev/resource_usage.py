"""Per-kernel register / LDS / scratch usage of a libbtba build (hipcc -Rpass-analysis=kernel-resource-usage), no GPU needed.
usage: python scripts/dev/resource_usage.py [filter-substring] [-- extra hipcc flags]"""
import re
import subprocess
import sys

args = sys.argv[1:]
extra = args[args.index("--") + 1:] if "--" in args else []
flt = [a for a in (args[:args.index("--")] if "--" in args else args)]
cmd = ["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fno-slp-vectorize", "-Wno-pass-failed", "-fPIC", "-shared", "-fvisibility=hidden",
       "-Rpass-analysis=kernel-resource-usage", "-o", "/tmp/btba_ru.so", "bundletrack_amd/csrc/btba_api.hip"] + extra
out = subprocess.run(cmd, capture_output=True, text=True).stderr
cur, rows = None, {}
for line in out.splitlines():
    m = re.search(r"Function Name: (\S+)", line)
    if m:
        cur = subprocess.run(["c++filt", m.group(1)], capture_output=True, text=True).stdout.strip().split("(")[0]
        rows[cur] = {}
        continue
    m = re.search(r"remark:\s+([A-Za-z ]+?)(?: \[[^\]]*\])?: (\d+)", line)
    if m and cur:
        rows[cur][m.group(1).strip()] = int(m.group(2))
    elif " error" in line:
        print(line)
for k, v in rows.items():
    if flt and not any(f in k for f in flt):
        continue
    print(f"{k[:64]:64s} VGPR {v.get('VGPRs', -1):3d}  SGPR {v.get('TotalSGPRs', -1):3d}  spill S {v.get('SGPRs Spill', -1):3d} V {v.get('VGPRs Spill', -1):3d}"
          f"  scratch {v.get('ScratchSize', -1):4d}  waves/SIMD {v.get('Occupancy', -1):2d}  LDS {v.get('LDS Size', -1):6d}")
