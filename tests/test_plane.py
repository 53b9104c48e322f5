"""f3: plane under the map points.

As shipped, System::processPlane (src/slam/src/system.cpp:177-342) has no defined behaviour.  The second implementation
the HIP path is compared with is the reference's OWN function compiled from its own source with exactly its four defects repaired
(oracle/ref_shim_plane.cpp + oracle/ref_plane_patch.sed: ref_find_plane_patched, in oracle/_ref/libalva_ref.so) -- float OpenCV SVDs
and all.  The CPU restatement (oracle/alva_oracle_plane.c) is pinned to it here; alva_find_plane is compared with both.  Tolerance: the
repaired reference takes its plane parameters from float SVDs, the restatement and the HIP path from cross products and a double
eigen-decomposition, so parameters agree to ~1e-6; poses are compared at 1e-5 on scenes where no decision (best hypothesis, inlier
threshold) hangs on less than 1e-3 of its value -- orc_find_plane_margins says which -- and most scenes must qualify."""
import ctypes as C

import numpy as np
import pytest

import oracles as O


def _scene(n_plane, n_clutter, seed, tilt=0.02, depth=4.0):
    """points on a plane that faces the world z axis (the reference's orientation test accepts only those) + clutter"""
    rng = np.random.RandomState(seed)
    nrm = np.array([tilt, -0.5 * tilt, 1.0])
    nrm /= np.linalg.norm(nrm)
    xy = rng.uniform(-2, 2, (n_plane, 2))
    z = depth - (nrm[0] * xy[:, 0] + nrm[1] * xy[:, 1]) / nrm[2]
    P = np.column_stack([xy, z]) + 0.004 * rng.randn(n_plane, 3)
    Q = np.column_stack([rng.uniform(-2, 2, (n_clutter, 2)), rng.uniform(1.0, 8.0, n_clutter)])
    pts = np.ascontiguousarray(np.vstack([P, Q]))
    rng.shuffle(pts)
    pose = np.array([0.1, -0.05, 0.2, 0.0, 0.0, 0.0, 1.0])
    return pts, pose, nrm


def _samples(n, iters, seed):
    rng = np.random.RandomState(seed)
    return np.ascontiguousarray(np.sort(np.array([rng.choice(n, 3, replace=False) for _ in range(iters)]), axis=1).astype(np.int32))


def orc_find_plane(pts, pose, samples):
    out = np.zeros(16, np.float32)
    f = O.orc_lib().orc_find_plane(pts.ctypes.data_as(C.c_void_p), len(pts), pose.ctypes.data_as(C.c_void_p),
                                   samples.ctypes.data_as(C.c_void_p), len(samples), out.ctypes.data_as(C.c_void_p))
    return out if f else None


def ref_find_plane(pts, pose, samples):
    """the reference's processPlane, repaired (oracle/ref_shim_plane.cpp)"""
    L = O.ref_lib()
    L.ref_find_plane_patched.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]
    out = np.zeros(16, np.float32)
    f = L.ref_find_plane_patched(pts.ctypes.data_as(C.c_void_p), len(pts), pose.ctypes.data_as(C.c_void_p), samples.ctypes.data_as(C.c_void_p),
                                 len(samples), out.ctypes.data_as(C.c_void_p))
    return out if f else None


def margins(pts, samples):
    """(accepted hypotheses, relative gap between the best score and the runner-up, smallest relative distance of a point from the inlier
    threshold): below ~1e-5 a float SVD and a cross product may decide differently"""
    m = np.zeros(3, np.float32)
    acc = O.orc_lib().orc_find_plane_margins(pts.ctypes.data_as(C.c_void_p), len(pts), samples.ctypes.data_as(C.c_void_p), len(samples),
                                             m.ctypes.data_as(C.c_void_p))
    return acc, float((m[1] - m[0]) / m[0]) if acc else 0.0, float(m[2])


def _normal_of(pose16):
    R = pose16.reshape(4, 4)[:3, :3].T          # toPoseArray(cv::Mat) stores the rotation's columns
    Rx = np.array([[1, 0, 0], [0, np.cos(1.0), -np.sin(1.0)], [0, np.sin(1.0), np.cos(1.0)]])
    return (R @ Rx.T)[:, 0]                     # R = R1 Rodrigues((1,0,0)); R1 maps (1,0,0) onto the plane normal


def test_restatement_finds_the_plane():
    pts, pose, nrm = _scene(300, 120, 3)
    out = orc_find_plane(pts, pose, _samples(len(pts), 250, 1))
    assert out is not None and out[15] == 1.0 and out[3] == out[7] == out[11] == 0.0
    n = _normal_of(out)
    assert abs(abs(n @ nrm) - 1.0) < 2e-3 and n[2] > 0        # flipped away from Oc - origin, Oc = -R t (utils.cpp:54-79)
    on_plane = abs((out[12:15] - np.array([0, 0, 4.0])) @ nrm)
    assert on_plane < 0.02                                    # the origin is the mean of the inliers
    R = out.reshape(4, 4)[:3, :3]
    assert np.abs(R @ R.T - np.eye(3)).max() < 1e-5


def test_restatement_conventions():
    pts, pose, _ = _scene(20, 5, 4)
    assert orc_find_plane(pts, pose, _samples(len(pts), 50, 2)) is None          # < 32 points (:181)
    rng = np.random.RandomState(5)
    cloud = np.ascontiguousarray(rng.uniform(-1, 1, (200, 3)) * [2, 2, 0.02] @ np.array([[1, 0, 0], [0, 0, 1], [0, -1, 0.0]]) + [0, 0, 4])
    # a wall seen edge-on (normal along y): every iteration fails the orientation test, the stored distances stay 0 and ALL points
    # pass "distance < 1.4 * 1e10" (:192-193, :249-259): the refit then runs on the whole cloud
    out = orc_find_plane(cloud, pose, _samples(200, 100, 3))
    assert out is not None and abs(abs(_normal_of(out)[1]) - 1.0) < 1e-3


SCENES = [(300, 120, s, 250) for s in range(24)] + [(40, 10, 6, 60), (1500, 900, 7, 250), (5000, 3000, 8, 100), (800, 50, 21, 250), (200, 600, 22, 250)]


@pytest.mark.ref
def test_restatement_equals_the_repaired_reference():
    """oracle/alva_oracle_plane.c against System::processPlane compiled from the reference's source with its four defects repaired: same
    found / not found on every scene, poses to 1e-5 on every scene without a float-sensitive decision (at least half of the scenes
    qualify; measured: <= 1.2e-7 on all of them) and to 5e-2 on the others (one borderline inlier more or less moves the mean of ~300
    inliers by up to ~2e-2)"""
    compared = 0
    for n_plane, n_clutter, seed, iters in SCENES:
        pts, pose, _ = _scene(n_plane, n_clutter, seed)
        S = _samples(len(pts), iters, seed + 1)
        a, b = orc_find_plane(pts, pose, S), ref_find_plane(pts, pose, S)
        assert (a is None) == (b is None), (n_plane, n_clutter, seed)
        acc, gap, edge = margins(pts, S)
        if a is not None and acc and gap > 1e-3 and edge > 1e-3:
            assert np.abs(a - b).max() <= 1e-5, (seed, np.abs(a - b).max())
            compared += 1
        elif a is not None:
            assert np.abs(a - b).max() <= 5e-2, (seed, np.abs(a - b).max())
    assert compared >= len(SCENES) // 2, compared
    pts, pose, _ = _scene(20, 5, 4)
    assert ref_find_plane(pts, pose, _samples(len(pts), 50, 2)) is None           # < 32 points (:181)
    pts, pose, _ = _scene(30, 400, 5)                                               # a plane too thin to collect 32 inliers among clutter
    S = _samples(len(pts), 30, 2)
    assert (orc_find_plane(pts, pose, S) is None) == (ref_find_plane(pts, pose, S) is None)


@pytest.fixture(scope="module")
def ctx():
    import alvaar_amd
    c = alvaar_amd.Context(0)
    yield c
    c.close()


@pytest.mark.gpu
@pytest.mark.parametrize("n_plane,n_clutter,seed,iters", [(300, 120, 3, 250), (40, 10, 6, 60), (1500, 900, 7, 250), (5000, 3000, 8, 100)])
def test_gpu_equals_restatement(ctx, n_plane, n_clutter, seed, iters):
    import torch
    pts, pose, nrm = _scene(n_plane, n_clutter, seed)
    S = _samples(len(pts), iters, seed)
    ref = orc_find_plane(pts, pose, S)
    out = ctx.find_plane(torch.from_numpy(pts).cuda(), pose, samples3=S)
    assert (ref is None) == (out is None)
    if ref is not None:
        assert np.abs(out - ref).max() < 2e-4, np.abs(out - ref).max()      # float sums in a different order, float SVD vs eigenvector


@pytest.mark.gpu
@pytest.mark.ref
def test_gpu_equals_the_repaired_reference(ctx):
    """alva_find_plane against the reference's own processPlane with its four defects repaired (ref_find_plane_patched), same sample
    indices: found / not found on every scene; poses to 1e-4 (the HIP path sums its 4x4 moment matrix and the inlier mean in another
    order: a few float ulps of a coordinate ~4) on every scene without a float-sensitive decision"""
    import torch
    compared = 0
    for n_plane, n_clutter, seed, iters in SCENES:
        pts, pose, _ = _scene(n_plane, n_clutter, seed)
        S = _samples(len(pts), iters, seed + 1)
        ref = ref_find_plane(pts, pose, S)
        out = ctx.find_plane(torch.from_numpy(pts).cuda(), pose, samples3=S)
        assert (ref is None) == (out is None), (n_plane, n_clutter, seed)
        acc, gap, edge = margins(pts, S)
        if ref is not None and acc and gap > 1e-3 and edge > 1e-3:
            assert np.abs(out - ref).max() <= 1e-4, (seed, np.abs(out - ref).max())
            compared += 1
        elif ref is not None:
            assert np.abs(out - ref).max() <= 5e-2, (seed, np.abs(out - ref).max())
    assert compared >= len(SCENES) // 2, compared


@pytest.mark.gpu
def test_gpu_conventions(ctx):
    import torch
    pts, pose, _ = _scene(20, 5, 4)
    assert ctx.find_plane(torch.from_numpy(pts).cuda(), pose) is None
    rng = np.random.RandomState(5)
    cloud = np.ascontiguousarray(rng.uniform(-1, 1, (200, 3)) * [2, 2, 0.02] @ np.array([[1, 0, 0], [0, 0, 1], [0, -1, 0.0]]) + [0, 0, 4])
    S = _samples(200, 100, 3)
    ref, out = orc_find_plane(cloud, pose, S), ctx.find_plane(torch.from_numpy(cloud).cuda(), pose, samples3=S)
    assert ref is not None and out is not None and np.abs(out - ref).max() < 2e-4
    pts, pose, nrm = _scene(400, 100, 9)
    out = ctx.find_plane(torch.from_numpy(pts).cuda(), pose)            # own sample stream
    assert out is not None and abs(abs(_normal_of(out) @ nrm) - 1.0) < 5e-3
