"""f3: plane under the map points.  PARITY UNPINNED -- System::processPlane (src/slam/src/system.cpp:177-342) has
no defined behaviour to pin against.  Both the CPU restatement (oracle/alva_oracle_plane.c) and alva_find_plane
follow what the statements of that function say they want to compute; the tests check them against each other and against the
geometry they were given."""
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
