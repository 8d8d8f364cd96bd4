"""CPU-side checks of the product: the C-ABI library loads and exports every symbol include/btba.h
declares, host helpers behave, and the caller-side logic (marshalling, keyframe memory) matches the
reference's rules.  No compute kernel is launched here."""
import ctypes as C

import numpy as np
import pytest

from bundletrack_amd import _lib, bundler, synthetic as S


def test_library_exports_every_declared_symbol():
    declared = _lib.declared_symbols()
    assert set(declared) == set(_lib.EXPORTED_SYMBOLS)
    L = _lib.lib()
    for name in declared:
        assert hasattr(L, name), name
    assert L.btba_version() == 105


def test_struct_sizes_match_header():
    assert C.sizeof(_lib.Params) == 16 * 4 + 2 * 8 + 8    # 15 words, 4 bytes of padding, two host pointers (weights_*_per_iter), their length + 4 bytes of tail padding
    assert _lib.ENTRYJ_DTYPE.itemsize == 32          # struct EntryJ, SIFTImageManager.h:44-59
    L = _lib.TraceLayout()
    _lib.lib().btba_trace_layout_get(15, 105, 5, C.byref(L))
    assert L.record_floats == 15 * 6 * 4 + 15 * 16 + 5 * 4 + 105 * 28 + 90 * 90 + 8
    assert L.off_A + 90 * 90 == L.off_clk and L.off_clk + 8 == L.record_floats


def test_default_params_are_the_shipping_config():
    p = _lib.default_params()
    assert (p.n_gn_iters, p.n_pcg_iters) == (7, 5)                      # config_ycbineoat.yml:24-25
    assert abs(p.robust_delta - 0.005) < 1e-9 and abs(p.dense_dist_thresh - 0.02) < 1e-9
    assert abs(p.dense_normal_thresh - np.float32(np.cos(np.pi / 4))) < 1e-7
    assert (p.depth_min, p.depth_max) == (np.float32(0.1), 9999.0)      # CUDASolverBundling.cpp:97-98
    assert (p.weight_sparse, p.weight_dense_depth, p.image_downscale) == (1.0, 1.0, 4.0)
    assert p.pair_policy == _lib.PAIRS_TARGET_LOWER


def test_strerror_and_status_codes():
    L = _lib.lib()
    assert L.btba_strerror(0) == b"ok"
    for code in (1, 2, 3, 4, 99):
        assert len(L.btba_strerror(code)) > 0


def test_bucket_correspondences_is_stable_and_validates():
    rng = np.random.default_rng(0)
    n = 5
    pb_pairs = [(i, j) for i in range(n) for j in range(i + 1, n)]
    corr = np.zeros(300, _lib.ENTRYJ_DTYPE)
    pick = rng.integers(0, len(pb_pairs), 300)
    corr["imgIdx_i"] = [pb_pairs[k][0] for k in pick]
    corr["imgIdx_j"] = [pb_pairs[k][1] for k in pick]
    corr["pos_i"] = rng.normal(size=(300, 3))
    corr["pos_j"] = np.arange(300)[:, None]                 # tag = original position
    corr["imgIdx_i"][::11] = 0xFFFFFFFF                     # invalid entries are dropped
    out, off = _lib.bucket_correspondences(corr, n)
    valid = corr[corr["imgIdx_i"] != 0xFFFFFFFF]
    assert off[-1] == len(valid) == len(out)
    for p, (i, j) in enumerate(pb_pairs):
        seg = out[off[p]:off[p + 1]]
        assert np.all(seg["imgIdx_i"] == i) and np.all(seg["imgIdx_j"] == j)
        assert np.all(np.diff(seg["pos_j"][:, 0]) > 0)      # stable: caller's order kept inside a pair
    # already pair-major input (what Bundler::optimizeGPU produces) passes through unchanged
    pb = S.make_problem(3, 20, seed=1, full_res=False)
    out2, off2 = _lib.bucket_correspondences(pb.corr, 3)
    assert np.array_equal(out2.view(np.uint8), pb.corr.view(np.uint8)) and list(off2) == [0, 20, 40, 60]
    bad = pb.corr.copy(); bad["imgIdx_j"][0] = 7
    with pytest.raises(_lib.BtbaError):
        _lib.bucket_correspondences(bad, 3)
    bad = pb.corr.copy(); bad["imgIdx_i"][0], bad["imgIdx_j"][0] = 2, 1     # i must be < j (Bundler.cpp:311-316)
    with pytest.raises(_lib.BtbaError):
        _lib.bucket_correspondences(bad, 3)
    empty, off3 = _lib.bucket_correspondences(pb.corr[:0], 3)
    assert len(empty) == 0 and list(off3) == [0, 0, 0, 0]


def test_rotation_geodesic_distance():
    R = S.so3_exp(np.array([0.0, 0.3, 0.0]))
    assert abs(bundler.rotation_geodesic_distance(R, np.eye(3)) - 0.3) < 1e-4
    assert bundler.rotation_geodesic_distance(np.eye(3), np.eye(3)) == 0.0


def _orbit_frames(n, step_deg):
    return [bundler.FrameRef(id=k, pose_in_model=S.orbit_pose(np.deg2rad(step_deg * k)).astype(np.float32), n_keypts=100) for k in range(n)]


def test_keyframe_pool_needs_min_rot_from_every_keyframe():
    mem = bundler.KeyframeMemory(min_rot_deg=10.0)
    frames = _orbit_frames(40, 4.0)
    added = [f.id for f in frames if mem.check_and_add_keyframe(f)]
    assert added[0] == 0
    for a, b in zip(added, added[1:]):
        assert (b - a) * 4.0 >= 10.0 - 1e-3          # 4 deg steps -> every third frame
    assert added[:4] == [0, 3, 6, 9]
    failed = bundler.FrameRef(id=99, pose_in_model=S.orbit_pose(3.0).astype(np.float32), status="FAIL")
    assert not mem.check_and_add_keyframe(failed)


def test_select_keyframes_greedy_rot():
    mem = bundler.KeyframeMemory(min_rot_deg=10.0, max_BA_frames=6)
    frames = _orbit_frames(30, 11.0)
    for f in frames[:20]:
        assert mem.check_and_add_keyframe(f)
    new = frames[20]
    chosen = mem.select_keyframes_for_ba(new)
    ids = [f.id for f in chosen]
    assert len(ids) == 6 and ids == sorted(ids) and 0 in ids and 20 in ids        # Bundler.cpp:237,286
    # greedy min-sum rotation distance to {new, kf0}: on a monotone orbit the sum is constant between them
    # for in-between frames, so the first keyframe in pool order wins ties (strict '<', Bundler.cpp:254)
    assert ids[1] == 1
    small = bundler.KeyframeMemory(max_BA_frames=15)
    for f in frames[:5]:
        small.check_and_add_keyframe(f)
    assert [f.id for f in small.select_keyframes_for_ba(frames[7])] == [0, 1, 2, 3, 4, 7]   # pool fits: take all (:227-235)


def test_marshal_window_layout_and_gate():
    frames = _orbit_frames(3, 11.0)[::-1]                       # unsorted on purpose
    rng = np.random.default_rng(0)
    m = {(1, 0): (rng.normal(size=(4, 3)), rng.normal(size=(4, 3))),
         (2, 0): (rng.normal(size=(2, 3)), rng.normal(size=(2, 3))),
         (2, 1): (rng.normal(size=(3, 3)), rng.normal(size=(3, 3)))}
    w = bundler.marshal_window(frames, m, newframe=frames[0], min_fm_edges_newframe=4)
    assert [f.id for f in w.frames] == [0, 1, 2]
    assert list(w.n_match_per_pair) == [4, 2, 3] and len(w.corr) == 9
    assert list(w.corr["imgIdx_i"]) == [0] * 6 + [1] * 3 and list(w.corr["imgIdx_j"]) == [1] * 4 + [2] * 5
    assert np.allclose(w.corr["pos_j"][:4], m[(1, 0)][0]) and np.allclose(w.corr["pos_i"][:4], m[(1, 0)][1])   # pos_i = ptB, pos_j = ptA
    assert w.n_edges_newframe == 5 and w.run_ba                      # new frame id 2 touches pairs (0,2),(1,2)
    w2 = bundler.marshal_window(frames, m, newframe=frames[0], min_fm_edges_newframe=5)
    assert not w2.run_ba and frames[0].status == "NO_BA"             # `<=` gate, Bundler.cpp:343-347


def test_tensor_handover_is_validated():
    """The C ABI sees plain pointers: the Python host layer refuses anything but dense row-major CUDA tensors (a strided
    view -- e.g. torch.from_numpy of a fancy-indexed numpy array -- would otherwise be read as garbage without an error)."""
    import torch
    from bundletrack_amd.optimizer import _dev_ptr
    assert _dev_ptr(None) is None
    with pytest.raises(ValueError, match="CUDA"):
        _dev_ptr(torch.zeros(4), "zn")


def test_python_constants_match_the_header():
    """Status codes, pair policies, tuning flags and the window limits of include/btba.h against the Python mirror
    (bundletrack_amd/_lib.py): the two are maintained by hand, so a drift must fail here."""
    import os, re
    from bundletrack_amd import _lib
    hdr = open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "include", "btba.h")).read()
    enums = {k: int(v) for k, v in re.findall(r"\b(BTBA_[A-Z0-9_]+)\s*=\s*(\d+)", hdr)}
    defines = {k: int(v) for k, v in re.findall(r"#define\s+(BTBA_[A-Z0-9_]+)\s+(\d+)\b", hdr)}
    for name in ("BTBA_OK", "BTBA_EINVAL", "BTBA_EHIP", "BTBA_ENUMERIC", "BTBA_ENOMEM", "BTBA_ESCHED"):
        assert enums[name] == getattr(_lib, name), name
    for name in ("TARGET_LOWER", "TARGET_MORE_VALID", "EXPLICIT"):
        assert enums["BTBA_PAIRS_" + name] == getattr(_lib, "PAIRS_" + name), name
    flags = {k[len("BTBA_FLAG_"):]: v for k, v in enums.items() if k.startswith("BTBA_FLAG_")}
    assert len(set(flags.values())) == len(flags) and all(v & (v - 1) == 0 for v in flags.values())     # distinct single bits
    for name, v in flags.items():
        assert getattr(_lib, "FLAG_" + name) == v, name
    assert defines["BTBA_MAX_FRAMES"] == 85 and defines["BTBA_MAX_FRAMES_LDS"] == 31     # the reference's MAX_NUM_IMAGES; the LDS-resident limit
    opts = {k[len("BTBA_OPT_"):]: v for k, v in enums.items() if k.startswith("BTBA_OPT_")}
    assert len(opts) == 16 and all(getattr(_lib, "OPT_" + name) == v for name, v in opts.items())
    assert enums["BTBA_REDUCE_DETERMINISTIC"] == _lib.REDUCE_DETERMINISTIC and enums["BTBA_REDUCE_ATOMIC"] == _lib.REDUCE_ATOMIC
    assert 128 not in flags.values()                    # the bit that was BTBA_FLAG_FUSE stays without a meaning
    assert C.sizeof(_lib.Stats) == 104      # (chain_iterations took the struct's tail padding)


def test_counter_file_matches_the_committed_pmc_summaries():
    """profiles/sweep_counters.json (what bench.py quotes as roofline.traffic / roofline.valu_issue) must carry the numbers of the PMC
    summaries it names -- both are written by scripts/summarize_profiles.py -- and bench.py must ignore a record that was taken on other
    kernel sources (kernel_source_hash) or another workload: config, instances, DISTINCT instances, mask, cache / correspondence layout."""
    import csv, json, os
    import bench
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    path = os.path.join(root, "profiles", "sweep_counters.json")
    if not os.path.exists(path):
        pytest.skip("no counter summary committed")
    recs = json.load(open(path)).get("records", [])
    for tj in recs:
        rows = [r for r in csv.DictReader(l for l in open(os.path.join(root, tj["source_hbm"])) if not l.startswith("#")) if r["kernel"] == tj["kernel"]]
        assert len(rows) == 1 and int(rows[0]["hbm_bytes_per_launch_corrected"]) == tj["hbm_bytes_per_launch"]
        fetch_kib, write_kib = float(rows[0]["FETCH_SIZE_KiB_mean"]), float(rows[0]["WRITE_SIZE_KiB_mean"])
        assert abs(int(fetch_kib * 1024 * 2 + write_kib * 1024) - tj["hbm_bytes_per_launch"]) <= 2048      # the guide's gfx950 correction
        head = [l for l in open(os.path.join(root, tj["source_sq"])) if l.startswith("#")]
        assert any(f"distinct instances: {tj['distinct']}" in l for l in head)                          # the pass ran the workload the record names
        sq = {r["counter"]: float(r["mean_per_launch"]) for r in csv.DictReader(l for l in open(os.path.join(root, tj["source_sq"])) if not l.startswith("#")) if r["kernel"] == tj["kernel"]}
        busy = 4.0 * (sq["SQ_ACTIVE_INST_VALU"] - sq["SQ_ACTIVE_INST_VALU2"]) / 1024.0 / (sq["SQ_BUSY_CYCLES"] / 32.0)
        assert abs(busy - tj["valu_busy_frac"]) < 2e-3 and 0.0 <= busy <= 1.0
        args = (tj["config"], tj["instances"], tj["masked"], tj["float4_cache"], tj["fused"])
        got = bench.profiled_counters(*args, tj["distinct"], tj["entryj"])
        assert (got is not None) == (tj["kernel_source_hash"] == bench.kernel_source_hash())            # stale sources -> not quoted
        assert bench.profiled_counters(tj["config"], tj["instances"] + 1, *args[2:], tj["distinct"], tj["entryj"]) is None
        assert bench.profiled_counters(*args, max(1, tj["distinct"] // 8), tj["entryj"]) is None          # a pass on 4 tiled instances does not speak for 32
        assert bench.profiled_counters(*args, tj["distinct"], not tj["entryj"]) is None
