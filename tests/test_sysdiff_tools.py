"""CPU: the trajectory alignment the unhooked System differential uses (tests/sysdiff.py: sim3_aligned_diff)."""
import numpy as np
from scipy.spatial.transform import Rotation

import sysdiff


def _traj(n, seed, straight):
    rng = np.random.RandomState(seed)
    out = []
    for k in range(n):
        q = Rotation.from_rotvec(rng.normal(size=3) * 0.05).as_quat()
        t = np.array([0.02 * k, 0.01 * k, 0.0]) + (0 if straight else rng.normal(size=3) * 0.05)
        out.append(np.concatenate([t, q]))
    return out


def test_alignment_recovers_a_similarity_exactly_even_on_a_straight_path():
    """a camera moving along a LINE (the synthetic streams do): the centres alone leave the rotation about that line free, the
    orientations fix it"""
    for straight in (True, False):
        ref = _traj(60, 1, straight)
        G, c, tt = Rotation.from_rotvec([0.3, -0.2, 0.1]), 1.7, np.array([1.0, 2.0, 3.0])
        got = [np.concatenate([c * G.apply(p[:3]) + tt, (G * Rotation.from_quat(p[3:])).as_quat()]) for p in ref]
        scale, dpos, drot = sysdiff.sim3_aligned_diff(ref, got)
        assert abs(scale - 1 / c) < 1e-12 and dpos < 1e-12 and drot < 1e-7


def test_alignment_reports_what_is_not_gauge():
    ref = _traj(60, 2, False)
    got = [p.copy() for p in ref]
    got[30][:3] += np.array([0.0, 0.01, 0.0])                       # one camera centre off by 1 cm
    r = Rotation.from_rotvec([0.0, 0.0, 2e-3]) * Rotation.from_quat(got[10][3:])
    got[10][3:] = r.as_quat()                                        # one orientation off by 2 mrad
    scale, dpos, drot = sysdiff.sim3_aligned_diff(ref, got)
    extent = np.linalg.norm(np.array([p[:3] for p in ref]) - np.mean([p[:3] for p in ref], 0), axis=1).max()
    assert 0.5 * 0.01 / extent < dpos < 1.5 * 0.01 / extent and 1.5e-3 < drot < 2.5e-3
