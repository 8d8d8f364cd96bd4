"""The chained launch (btba_kernels.hpp: k_chain -- all Gauss-Newton iterations of a batch in ONE launch, every instance's system solve handed
over inside the launch) against the plain two-launches-per-iteration schedule, bit for bit, and the exact configuration bench.py times
(32 distinct c3 instances, 24-byte correspondences, prebuilt block ranges) against the reference's own solveBundlingStub
(/root/reference/src/cuda/Solver/SolverBundling.cu:931-1003 compiled for the CPU: oracle/_ref, prebuilt -- nothing under /root/reference is read here)."""
import os

import numpy as np
import pytest

from bundletrack_amd import _lib, synthetic as S

pytestmark = pytest.mark.gpu


class Batch:
    """One batch resident on the GPU the way bench.py holds it."""

    def __init__(self, pbs, masked=False):
        import torch
        from bundletrack_amd.optimizer import BatchSolver
        self.torch, self.dev = torch, torch.device("cuda:0")
        self.pbs, self.N, self.B = pbs, pbs[0].n_frames, len(pbs)
        corr, offs, self.mx = BatchSolver.pack_correspondences([pb.corr for pb in pbs], self.N)
        self.corr_d = torch.from_numpy(corr.view(np.uint8).reshape(self.B, -1, 32)).to(self.dev)
        self.offs_d = torch.from_numpy(offs.astype(np.int32)).to(self.dev)
        self.zn_d = torch.from_numpy(np.stack([S.compact_cache(pb) for pb in pbs])).to(self.dev)
        self.poses0 = torch.from_numpy(np.stack([pb.poses_init for pb in pbs])).to(self.dev)
        self.masked = masked

    def solve(self, chain, corr24=False, aux=False, tiles=0, period=0, timeout_ms=None, relayout=True, corr_nt=None, legacy_solve=False):
        """-> (poses [B, N, 4, 4], stats).  chain: BTBA_OPT_CHAIN (-1 library's choice, 0 plain schedule, 1 chained wherever supported).
        legacy_solve: the plain schedule's system solves by k_system_solve (BTBA_OPT_SOLVE_SMALL = 0) -- the sums, in the order, that the chained
        launch's in-launch solve items reproduce bit for bit (round 5's k_solve_small forms the same sums in another order)."""
        from bundletrack_amd.optimizer import BatchSolver, Workspace
        ws = Workspace()
        ws.set_option(_lib.OPT_CHAIN, chain)
        ws.set_option(_lib.OPT_SOLVE_SMALL, 0 if legacy_solve else 1)
        ws.set_option(_lib.OPT_CHAIN_SPARSE_PERIOD, period)
        ws.set_option(_lib.OPT_RELAYOUT, 1 if relayout else 0)
        if timeout_ms is not None:
            ws.set_option(_lib.OPT_CHAIN_TIMEOUT_MS, timeout_ms)
        if corr_nt is not None:
            ws.set_option(_lib.OPT_CORR_NONTEMPORAL, corr_nt)      # 1 all instances, 0 none, -1 the library's choice
        bs = BatchSolver(ws)
        bs.params.dense_tiles = tiles
        if self.masked:
            bs.params.flags |= _lib.FLAG_COMPACTION
        aux_d = bs.cache_aux(self.zn_d, valid_lists=self.masked) if (aux or corr24) else None
        if corr24:
            aux_d["corr24"], flag = bs.pack_correspondences24(self.corr_d, self.offs_d, self.mx, self.N, check_order=True)
            self.torch.cuda.synchronize()
            assert int(flag.cpu()[0]) == 0
        poses_d = self.poses0.clone()
        pb = self.pbs[0]
        bs.solve_zn(self.zn_d, pb.H, pb.W, pb.K, None if corr24 else self.corr_d, self.offs_d, self.mx, poses_d, aux=aux_d, corr_stride=self.corr_d.shape[1])
        ws.sync()
        st = ws.collect_stats()
        out = poses_d.cpu().numpy()
        ws.close()
        return out, st


@pytest.mark.parametrize("name,B,K,m,masked,tiles,period", [
    ("16 x K=5, full frames, two tiles", 16, 5, 300, False, 2, 0),
    ("16 x K=5, sparse items interleaved", 16, 5, 300, False, 2, 3),
    ("9 instances (not a multiple of 8)", 9, 4, 200, False, 2, 0),
    ("16 x K=6, object-masked frames (valid-pixel lists)", 16, 6, 250, True, 1, 0),
    ("3 instances, one tile", 3, 7, 400, False, 1, 2),
])
def test_chained_launch_has_the_bits_of_the_plain_schedule(name, B, K, m, masked, tiles, period):
    pbs = [S.make_problem(K, m, 900 + 17 * b, background=not masked, full_res=False) for b in range(B)]
    bt = Batch(pbs, masked)
    plain, st0 = bt.solve(0, tiles=tiles, legacy_solve=True)
    chained, st1 = bt.solve(1, tiles=tiles, period=period, legacy_solve=True)      # (the last iteration's solve is a launch of its own: the same kernel as the plain schedule's)
    assert st0["chain_iterations"] == 0 and st1["chain_iterations"] == 7, (st0, st1)
    assert np.isfinite(chained).all()
    assert np.array_equal(plain, chained), f"{name}: worst difference {np.abs(plain - chained).max():.3e}"
    # ... and the poses moved (the comparison above is not between two copies of the input)
    assert np.abs(chained - np.stack([pb.poses_init for pb in pbs])).max() > 1e-4


def test_chained_launch_under_default_options_stays_inside_the_pose_bar():
    """With the library's defaults (BTBA_OPT_SOLVE_SMALL = 1) the plain schedule's system solves run in k_solve_small while the chained launch's in-launch
    solve items still reproduce k_system_solve's sums: BTBA_OPT_CHAIN = 1 is then NOT bit-identical to the plain schedule (include/btba.h says so) -- the same
    sums in another order.  Bounded here: both schedules under default options, final poses within the 1e-4 rad / m bar on well-conditioned windows."""
    pbs = [S.make_problem(6, 250, 900 + 17 * b, background=False, full_res=False) for b in range(16)]
    bt = Batch(pbs, masked=True)
    plain, st0 = bt.solve(0, tiles=1)
    chained, st1 = bt.solve(1, tiles=1)
    assert st0["chain_iterations"] == 0 and st1["chain_iterations"] == 7, (st0, st1)
    worst = max(max(S.pose_error(plain[b, k], chained[b, k])) for b in range(16) for k in range(6))
    print(f"chained vs plain under default options (k_solve_small in the plain schedule): worst final pose difference {worst:.2e}")
    assert np.isfinite(chained).all() and worst < 1e-4, worst


@pytest.fixture(scope="module")
def c3x32():
    return Batch([S.make_problem(15, 2000, S.config_seed(5, b), background=True, full_res=False) for b in range(32)])


def test_the_benched_path_is_pinned(c3x32):
    """What `python bench.py` times: btba_solve_batch_zn_aux with B = 32 DISTINCT c3 instances, aux.corr24 (24-byte correspondences packed
    from EntryJ), prebuilt block ranges, the library's own schedule (the fused launch with ONE tile per pair, sparse items closing it) --
      (a) bit-identical to the EntryJ / no-aux call on the plain schedule, the configuration the other parity tests reach;
      (b) instances 0, 13 and 31 against the reference's own solveBundlingStub, < 1e-4 rad / m."""
    bt = c3x32
    benched, st = bt.solve(-1, corr24=True, aux=True)
    assert st["chain_iterations"] == 0 and st["dense_tiles"] == 1 and st["sparse_chunks"] == 1 and st["fused_sweeps"] == 1, st
    plain, st0 = bt.solve(0)                                     # EntryJ in, re-laid out to 24-byte records by the first iteration's sweep (BTBA_OPT_RELAYOUT = 1)
    assert st0["chain_iterations"] == 0
    assert np.array_equal(benched, plain), f"worst difference {np.abs(benched - plain).max():.3e}"
    wire, _ = bt.solve(0, relayout=False)                        # ... and with every iteration reading the 32-byte wire format (the default; what bench.py's value_incl_pack times)
    assert np.array_equal(benched, wire)
    legacy, _ = bt.solve(-1, corr24=True, aux=True, legacy_solve=True)      # the same batch with rounds 1-4's system solve (k_system_solve) ...
    chained, st1 = bt.solve(1, corr24=True, aux=True, legacy_solve=True)    # ... whose sums the chained launch of the same batch reproduces bit for bit
    assert st1["chain_iterations"] == 7 and np.array_equal(legacy, chained)
    worst = max(max(S.pose_error(benched[b, k], legacy[b, k])) for b in range(bt.B) for k in range(15))
    print(f"k_solve_small vs k_system_solve on the benched batch: worst final pose difference {worst:.2e}")
    assert worst < 1e-4
    from oracle import reference as R
    if not os.path.exists(R.SO_SOLVER):
        pytest.skip("oracle/_ref/libbtba_ref_solver.so not built")
    for b in (0, 13, 31):
        pb = bt.pbs[b]
        campos, normals, intr = S.analytic_cache(pb)
        ref, _ = R.solve(campos, normals, intr, pb.corr, pb.poses_init, weight_dense=1.0)
        worst = max(max(S.pose_error(benched[b, k], ref[k])) for k in range(15))
        print(f"benched path, instance {b}: worst pose difference against the reference's solver {worst:.2e}")
        assert worst < 1e-4, (b, worst)


def test_nontemporal_correspondence_stream_has_the_same_bits(c3x32):
    """Once a batch's frames + correspondences exceed the memory-side cache (c3 x 32: 147 + 161 MB against 256 MB) the sparse items read the
    correspondences of the instances that no longer fit with non-temporal loads (btba_api.hip: corr_nt_auto) -- a cache policy, not arithmetic:
    all / none / the library's choice give the same poses, with 24-byte records and with EntryJ."""
    bt = c3x32
    auto, _ = bt.solve(-1, corr24=True, aux=True)
    for nt in (0, 1):
        out, _ = bt.solve(-1, corr24=True, aux=True, corr_nt=nt)
        assert np.array_equal(auto, out), (nt, np.abs(auto - out).max())
    wire_nt, _ = bt.solve(0, relayout=False, corr_nt=1)
    wire, _ = bt.solve(0, relayout=False, corr_nt=0)
    assert np.array_equal(wire_nt, wire) and np.array_equal(wire, auto)


def test_chained_launch_is_reproducible_under_load(c3x32):
    """The hand-offs inside the launch (partial records -> solve item -> next iterate's poses) must never deliver stale data: 24 chained
    solves of the full c3 x 32 batch -- 1 536 resident workgroups, every compute unit's L1 and scalar cache warm with the previous
    iterations' lines -- all carry the bits of the first one (which test_the_benched_path_is_pinned holds against the plain schedule)."""
    bt = c3x32
    first, _ = bt.solve(1, corr24=True, aux=True, legacy_solve=True)
    for rep in range(24):
        again, st = bt.solve(1, corr24=True, aux=True, period=(3 if rep % 2 else 0), legacy_solve=True)
        assert st["chain_iterations"] == 7
        assert np.array_equal(first, again), f"run {rep}: worst difference {np.abs(first - again).max():.3e}"


def test_c3_masked_batch_chained_equals_plain():
    pbs = [S.make_problem(15, 2000, S.config_seed(5, b), background=False, full_res=False) for b in range(32)]
    bt = Batch(pbs, masked=True)
    plain, st0 = bt.solve(0, aux=True, legacy_solve=True)
    auto, st2 = bt.solve(-1, aux=True, legacy_solve=True)              # the library's own choice for object-masked frames is the plain schedule (their sweeps are shorter than an in-launch solve)
    assert st2["chain_iterations"] == 0 and np.array_equal(plain, auto)
    chained, st1 = bt.solve(1, aux=True, legacy_solve=True)
    assert st0["chain_iterations"] == 0 and st1["chain_iterations"] == 7 and st1["dense_tiles"] == 1, (st0, st1)
    assert np.array_equal(plain, chained), f"worst difference {np.abs(plain - chained).max():.3e}"


def test_watchdog_reports_a_stuck_launch_and_the_workspace_falls_back():
    """A wait inside the chained launch that is never satisfied (here: solve items that do not publish their iterate -- a developer switch) must
    end in the watchdog, not in a hung GPU: the launch runs out, the next host synchronisation reports BTBA_ESCHED, and the workspace solves
    with the plain schedule from then on -- correctly."""
    import torch
    from bundletrack_amd.optimizer import BatchSolver, Workspace
    pbs = [S.make_problem(5, 300, 300 + b, background=True, full_res=False) for b in range(16)]
    bt = Batch(pbs)
    plain, _ = bt.solve(0, tiles=2)
    ws = Workspace()
    ws.set_option(_lib.OPT_CHAIN, 1)
    ws.set_option(_lib.OPT_CHAIN_TIMEOUT_MS, 2)
    ws.set_option(1000, 64)                          # developer switch (not part of the ABI): solve items do not publish
    bs = BatchSolver(ws)
    bs.params.dense_tiles = 2
    poses_d = bt.poses0.clone()
    bs.solve_zn(bt.zn_d, pbs[0].H, pbs[0].W, pbs[0].K, bt.corr_d, bt.offs_d, bt.mx, poses_d)
    with pytest.raises(_lib.BtbaError) as err:
        ws.sync()
    assert err.value.status == _lib.BTBA_ESCHED
    ws.set_option(1000, 0)
    poses_d = bt.poses0.clone()
    bs.solve_zn(bt.zn_d, pbs[0].H, pbs[0].W, pbs[0].K, bt.corr_d, bt.offs_d, bt.mx, poses_d)
    ws.sync()
    st = ws.collect_stats()
    assert st["chain_iterations"] == 0                # chaining stays off on a workspace whose watchdog fired
    assert np.array_equal(poses_d.cpu().numpy(), plain)
    ws.close()
