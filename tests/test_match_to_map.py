"""f1: Mapper::matchToMap.  CPU: the oracle restatement against the reference's OWN Mapper::matchToMap run
on a map built with the reference's own classes (oracle/ref_shim_map.cpp) and against committed golden problems; GPU:
alva_match_to_map against the oracle on the golden problems (which carry what only the reference's containers determine:
the keypoint grid order, the unordered_set iteration order, T_cw as Sophus holds it).  Discrete output: exact."""
import numpy as np
import pytest
from pathlib import Path

from alvaar_amd import synth
from oracles import ref_match_to_map, orc_match_to_map, flatten_match_to_map, py_match_to_map_aux, ref_available

G = Path(__file__).resolve().parent / "golden"
CASES = [dict(n=400, seed=1), dict(n=800, seed=2, dist=(-0.2, 0.05, 0.001, -0.001)), dict(n=300, seed=3, kp3=10),
         dict(n=700, seed=11, px_noise=1.5, max_flips=70), dict(n=600, seed=15, twin_frac=0.6), dict(n=1500, seed=4)]


def _problem(c):
    kw = {k: v for k, v in c.items() if k not in ("n", "seed", "kp3")}
    return synth.make_match_to_map_problem(c["n"], c["seed"], **kw), c.get("kp3")


@pytest.mark.ref
@pytest.mark.parametrize("case", CASES)
def test_oracle_equals_reference(case):
    if not ref_available():
        pytest.skip("compiled reference not present")
    pb, kp3 = _problem(case)
    ref, aux = ref_match_to_map(pb, num_kp3d=kp3)
    assert orc_match_to_map(pb, aux, num_kp3d=kp3) == ref
    assert len(ref) > 20 and len(ref) < len(pb["frame_kp_order"])       # merges happen, and not everything merges


def _golden():
    z = np.load(G / "match_to_map.npz", allow_pickle=False)
    out = []
    for i in range(int(z["count"])):
        pb = {k[len(f"p{i}_"):]: z[k] for k in z.files if k.startswith(f"p{i}_") and not k.startswith(f"p{i}_aux_") and not k.startswith(f"p{i}_exp")}
        pb["cell_size"], pb["num_kp3d"] = int(pb["cell_size"]), int(pb["num_kp3d"])
        aux = {k[len(f"p{i}_aux_"):]: z[k] for k in z.files if k.startswith(f"p{i}_aux_")}
        aux["grid_cells"], aux["num_cells_w"] = int(aux["grid_cells"]), int(aux["num_cells_w"])
        exp = {int(a): int(b) for a, b in z[f"p{i}_exp"]}
        out.append((pb, aux, exp))
    return out


def test_oracle_equals_golden():
    for pb, aux, exp in _golden():
        assert orc_match_to_map(pb, aux) == exp


@pytest.mark.gpu
def test_hip_equals_oracle_and_golden(ctx):
    import torch
    for pb, aux, exp in _golden():
        cell_mp, local = flatten_match_to_map(pb, aux)
        d = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
        out = ctx.match_to_map(pb["calib"], pb["cell_size"], aux["num_cells_w"], aux["grid_cells"], d(aux["cell_ptr"].astype(np.int32)),
                               d(cell_mp), d(aux["kf_q"]), d(aux["kf_t"]), d(pb["mp_wpt"]), d(pb["mp_is3d"]), d(pb["obs_ptr"]), d(pb["obs_kf"]),
                               d(pb["obs_px"]), d(pb["obs_desc"]), len(pb["kf_id"]) - 1, pb["num_kp3d"], d(local)).cpu().numpy()
        got = {int(pb["mp_id"][m]): int(pb["mp_id"][out[m]]) for m in range(len(out)) if out[m] >= 0}
        assert got == exp == orc_match_to_map(pb, aux)
    # empty local map: nothing matches
    pb, aux, exp = _golden()[0]
    cell_mp, local = flatten_match_to_map(pb, aux)
    d = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
    out = ctx.match_to_map(pb["calib"], pb["cell_size"], aux["num_cells_w"], aux["grid_cells"], d(aux["cell_ptr"].astype(np.int32)), d(cell_mp),
                           d(aux["kf_q"]), d(aux["kf_t"]), d(pb["mp_wpt"]), d(pb["mp_is3d"]), d(pb["obs_ptr"]), d(pb["obs_kf"]), d(pb["obs_px"]),
                           d(pb["obs_desc"]), len(pb["kf_id"]) - 1, pb["num_kp3d"], d(local[:0]))
    assert int((out >= 0).sum()) == 0


@pytest.mark.gpu
@pytest.mark.parametrize("n,seed,kw", [(4000, 7, {}), (2500, 8, dict(px_noise=1.2, max_flips=60, twin_frac=0.5, dist=(-0.15, 0.03, 0.0005, 0.0004)))])
def test_hip_equals_oracle_full_size(ctx, n, seed, kw):
    """thousands of local map points / keypoints, inputs ordered by the numpy stand-in"""
    import torch
    pb = synth.make_match_to_map_problem(n, seed, **kw)
    aux = py_match_to_map_aux(pb)
    exp = orc_match_to_map(pb, aux)
    cell_mp, local = flatten_match_to_map(pb, aux)
    d = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
    out = ctx.match_to_map(pb["calib"], pb["cell_size"], aux["num_cells_w"], aux["grid_cells"], d(aux["cell_ptr"]), d(cell_mp), d(aux["kf_q"]),
                           d(aux["kf_t"]), d(pb["mp_wpt"]), d(pb["mp_is3d"]), d(pb["obs_ptr"]), d(pb["obs_kf"]), d(pb["obs_px"]), d(pb["obs_desc"]),
                           len(pb["kf_id"]) - 1, pb["num_kp3d"], d(local)).cpu().numpy()
    got = {int(pb["mp_id"][m]): int(pb["mp_id"][out[m]]) for m in range(len(out)) if out[m] >= 0}
    assert got == exp and len(exp) > 500
