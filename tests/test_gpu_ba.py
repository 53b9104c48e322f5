"""GPU parity: a10-a13 -- local bundle adjustment (LM + Huber + Schur, FP64 with the reduced camera system
formed on the FP64 matrix core).  Tolerances: poses 1e-8, point parameters 1e-7 (inverse depth) / 1e-6 (XYZ),
costs 1e-8 relative; iteration / accepted-step counts, depth flags and the chi2 > 5.9915 outlier
classification must match exactly."""
import numpy as np
import pytest

from alvaar_amd import synth
from oracles import Orc, Ref, ref_available
from test_oracle_vs_ref import ba_compare, xyz_problem

pytestmark = pytest.mark.gpu


def _checkers(full=False):
    return [("orc", Orc)] + ([("ref", Ref)] if ref_available() else [])


@pytest.mark.parametrize("nkf,npt,seed,iters,ftol", [(6, 200, 1, 5, 0.0), (20, 600, 42, 5, 0.0), (8, 300, 2, 5, 1e-3), (5, 80, 3, 2, 0.0),
                                                     (3, 10, 4, 5, 0.0)])
def test_local_ba_invdepth(ctx, nkf, npt, seed, iters, ftol):
    pb = synth.make_ba_problem(nkf, npt, seed)
    g = ctx.local_ba(pb, iters, ftol)
    for name, O in _checkers():
        ba_compare(g, O.local_ba(pb, iters, ftol))


@pytest.mark.parametrize("nkf,npt,seed", [(6, 150, 5), (12, 400, 6)])
def test_local_ba_xyz(ctx, nkf, npt, seed):
    pb = xyz_problem(nkf, npt, seed)
    g = ctx.local_ba(pb, 5, 0.0, inv_depth=False)
    for name, O in _checkers():
        ba_compare(g, O.local_ba(pb, 5, 0.0, inv_depth=False), pt_tol=1e-6)


def test_local_ba_full_size(ctx):
    """BASELINE config 4: 20 KF x 3000 pts, 5 LM iterations.  Compared against the compiled reference
    (Ceres) when present, else the restatement; plus size-independent properties: the cost decreases
    monotonically over accepted steps and the result is invariant to the order observations are listed in."""
    pb = synth.make_ba_problem(20, 3000, 42)
    g = ctx.local_ba(pb, 5, 0.0)
    O = Ref if ref_available() else Orc
    ba_compare(g, O.local_ba(pb, 5, 0.0))
    assert g["info"][2] < 0.1 * g["info"][1]
    perm = np.random.RandomState(0).permutation(len(pb["obs_kf"]))
    pb2 = dict(pb, obs_kf=pb["obs_kf"][perm], obs_pt=pb["obs_pt"][perm], obs_uv=pb["obs_uv"][perm])
    g2 = ctx.local_ba(pb2, 5, 0.0)
    assert np.abs(g2["poses"] - g["poses"]).max() < 1e-10
    assert np.allclose(g2["chi2"], g["chi2"][perm], rtol=1e-9, atol=1e-12)


@pytest.mark.parametrize("nkf,npt,seed", [(28, 900, 9), (40, 600, 10)])
def test_local_ba_more_cameras_than_fit_in_lds(ctx, nkf, npt, seed):
    """6 x (free cameras) > 136: the reduced camera system no longer fits the LDS and the blocked Cholesky runs on the
    copy in device memory (k_solve<false>); same tolerances."""
    pb = synth.make_ba_problem(nkf, npt, seed)
    assert 6 * int((pb["kf_const"] == 0).sum()) > 136
    g = ctx.local_ba(pb, 5, 0.0)
    ba_compare(g, Orc.local_ba(pb, 5, 0.0))


@pytest.mark.parametrize("sizes", [[(20, 3000, 42)], [(20, 3000, 42), (8, 400, 1), (12, 1500, 2), (5, 120, 3)], [(6 + (b % 9), 200 + 37 * b, 10 + b) for b in range(64)]])
def test_local_ba_batch_bitwise_equal_to_single_solves(ctx, sizes):
    """alva_local_ba_batch: B ragged problems with one set of launches per LM iteration; every problem's poses, points, chi2 / depth flags
    and LM bookkeeping (summaries, accepted steps, costs) are BIT-IDENTICAL to its own alva_local_ba call.  Problems stop at different
    iterations (function tolerance 1e-3, as Optimizer::localBA sets it)."""
    from alvaar_amd import synth
    # alternating noise levels: problems converge after different numbers of iterations
    pbs = [synth.make_ba_problem(k, n, s, pose_noise=(0.0002 if i % 3 == 0 else 0.02), invdepth_noise=(0.001 if i % 3 == 0 else 0.05)) for i, (k, n, s) in enumerate(sizes)]
    for ftol, iters in ((1e-3, 5), (0.0, 5)):
        single = [ctx.local_ba(pb, iters, ftol) for pb in pbs]
        batch = ctx.local_ba_batch(pbs, iters, ftol)
        stops = set()
        for a, b in zip(single, batch):
            assert a["ok"] == b["ok"]
            assert np.array_equal(a["info"][:4], b["info"])
            assert np.array_equal(a["poses"].view(np.uint64), b["poses"].view(np.uint64))
            assert np.array_equal(a["pts"].view(np.uint64), b["pts"].view(np.uint64))
            assert np.array_equal(a["chi2"].view(np.uint64), b["chi2"].view(np.uint64)) and np.array_equal(a["depth"], b["depth"])
            stops.add(int(a["info"][0]))
        if len(sizes) > 8 and ftol > 0:
            assert len(stops) > 1     # the batch really is ragged in iterations too


@pytest.mark.parametrize("nkf,npt,seed,iters,ftol", [(6, 200, 1, 5, 0.0), (20, 3000, 42, 5, 0.0), (14, 1500, 9, 5, 1e-3), (3, 10, 4, 5, 0.0), (31, 900, 12, 3, 0.0)])
def test_local_ba_csr_bitwise_equal_to_the_host_structured_solve(ctx, nkf, npt, seed, iters, ftol):
    """alva_local_ba_csr -- observations grouped by point, the (observing, anchor) pair grouping built by the device's stable counting
    sort, the outlier sweep's test returned as one bit per residual block -- against alva_local_ba on the same problem: poses and point
    parameters BIT-IDENTICAL, iteration counts equal, the bits == (chi2 > threshold or point behind the camera) of the host-structured
    solve."""
    pb = dict(synth.make_ba_problem(nkf, npt, seed))
    rng = np.random.RandomState(seed)
    # outliers, so that the sweep has something to flag
    bad_obs = rng.choice(len(pb["obs_kf"]), max(2, len(pb["obs_kf"]) // 200), replace=False)
    uv = pb["obs_uv"].copy()
    uv[bad_obs] += rng.uniform(4, 9, (len(bad_obs), 2))
    pb["obs_uv"] = uv
    a = ctx.local_ba(pb, iters, ftol)
    b = ctx.local_ba_csr(pb, iters, ftol, chi2_threshold=5.9915)
    assert a["ok"] == b["ok"] and list(a["info"][:4]) == list(b["info"][:4])
    assert np.array_equal(a["poses"].view(np.uint64), b["poses"].view(np.uint64))
    assert np.array_equal(a["pts"].view(np.uint64), b["pts"].view(np.uint64))
    want = (a["chi2"] > 5.9915) | (a["depth"] == 0)
    assert np.array_equal(b["bad"], want[b["order"]]) and b["n_bad"] == int(want.sum()) and b["n_bad"] >= 1
