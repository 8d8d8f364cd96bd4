#!/bin/bash
# round 3, GPU call 46: k_system_solve assembly phase split by stamps (debug build: slots 6 / 7 moved behind the off-diagonal and the diagonal blocks)
set -u
REPO=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
O=$REPO/gpurun_out/r03_46
mkdir -p "$O"
export TMPDIR=/tmp
cd "$REPO"
BTBA_LIB_PATH=$REPO/build/ab/solve_stamps.so timeout 300 python - > "$O/stamps.txt" 2> "$O/err.txt" <<'PY'
import os, sys
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from bundletrack_amd.optimizer import BatchSolver, Workspace
import bench
os.environ["BTBA_BENCH_NPROC"] = "1"
cfg = bench.CONFIGS["c3"]
inst = bench.generate_instances(cfg, [0, 1])
dev = torch.device("cuda:0")
ws = Workspace()
for B in (1, 32):
    pick = [inst[b % 2] for b in range(B)]
    bs = BatchSolver(ws)
    corr, offs, mx = bs.pack_correspondences([p["corr"] for p in pick], 15)
    cam_d = torch.from_numpy(np.stack([p["campos"] for p in pick])).to(dev); nrm_d = torch.from_numpy(np.stack([p["normals"] for p in pick])).to(dev)
    corr_d = torch.from_numpy(corr.view(np.uint8).reshape(B, -1, 32)).to(dev); offs_d = torch.from_numpy(offs.astype(np.int32)).to(dev)
    poses_d = torch.from_numpy(np.stack([p["poses"] for p in pick])).to(dev)
    tv = bs.trace_view(bs.solve(cam_d, nrm_d, pick[0]["intr"], corr_d, offs_d, mx, poses_d, trace=True))
    c = tv.clk[0].mean(0)
    print(f"B={B}: reduce done {c[0]:.0f} | congruence dump {c[1]-c[0]:.0f} | off-diagonal blocks {c[6]-c[1]:.0f} | diagonal blocks {c[7]-c[6]:.0f} | rhs + preconditioner + barrier {c[2]-c[7]:.0f} | PCG {c[3]-c[5]:.0f} | update {c[4]-c[3]:.0f}")
PY
cat "$O/stamps.txt"; tail -3 "$O/err.txt"
