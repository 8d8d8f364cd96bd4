import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, ctypes as C
from bundletrack_amd.optimizer import Workspace, DEPTH_PROCESSING_DEFAULTS
from bundletrack_amd._lib import lib, check
from oracle import oracle as O
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from test_depth_processing import noisy_depth


def run(ws, d, fill, **kw):
    p = dict(DEPTH_PROCESSING_DEFAULTS); p.update(kw)
    H, W = d.shape
    dg = torch.from_numpy(d).cuda()
    out = torch.full_like(dg, fill)
    check(lib().btba_process_depth(ws.handle, H, W, dg.data_ptr(), out.data_ptr(), int(p["erode_radius"]), float(p["erode_diff"]), float(p["erode_ratio"]),
                                   int(p["bf_radius"]), float(p["sigma_d"]), float(p["sigma_r"])), "pd")
    torch.cuda.synchronize()
    return out.cpu().numpy()


def main():
    ws = Workspace()
    for shape in ((37, 53), (96, 128), (480, 640)):
        d, _ = noisy_depth(2, *shape)
        for kw in ({}, dict(bf_radius=0), dict(erode_radius=0, erode_ratio=2.0), dict(erode_radius=2, erode_ratio=0.5, bf_radius=3, sigma_d=1.5, sigma_r=0.01)):
            p = dict(DEPTH_PROCESSING_DEFAULTS); p.update(kw)
            ref = O.process_depth(d, p["erode_radius"], p["erode_diff"], p["erode_ratio"], p["bf_radius"], p["sigma_d"], p["sigma_r"])
            a = run(ws, d, -7.0, **kw); b = run(ws, d, -9.0, **kw)
            print(shape, kw, "unwritten", int((a == -7.0).sum()), "run-to-run equal", np.array_equal(a, b), "zero-pattern mismatch", int(((a == 0) != (ref == 0)).sum()),
                  "maxdiff", float(np.abs(a - ref).max()))
            mm = (a == 0) != (ref == 0)
            if mm.any():
                ys, xs = np.where(mm)
                print("   first mismatches (y,x,got,ref):", [(int(y), int(x), float(a[y, x]), float(ref[y, x])) for y, x in list(zip(ys, xs))[:6]], "rows", sorted(set(ys.tolist()))[:12], "cols", sorted(set(xs.tolist()))[-12:])


if __name__ == "__main__":
    main()
