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


def _records_of(pb, shuffle_seed):
    """The flat synthetic map as map-point RECORDS (csrc/slam/mp_rec.hpp) in pinned chunks + the operations that fill the descriptor
    tables: row m -> a record slot chosen by a permutation (slots are recycled in the product: rows and slots do not coincide), an entry
    per observation {keyframe id, observed | holds-the-keypoint | has-a-descriptor, px}, plus -- like a live map has them -- entries the
    gather must drop: an observer keyframe that is not in the keyframe table, and an entry without the holds-the-keypoint flag."""
    import torch
    from alvaar_amd.capi import Context
    dt = Context.mp_record_dtype()
    n_mp = len(pb["mp_id"])
    rng = np.random.RandomState(shuffle_seed)
    n_slots = n_mp + 500
    slot_of = rng.permutation(n_slots)[:n_mp].astype(np.int32)
    n_chunks = (n_slots + 4095) // 4096
    chunks = [torch.zeros(4096 * dt.itemsize, dtype=torch.uint8).pin_memory() for _ in range(n_chunks)]
    views = [c.numpy().view(dt) for c in chunks]
    ops = []
    kf_id = pb["kf_id"]
    for m in range(n_mp):
        s = int(slot_of[m])
        r = views[s >> 12][s & 4095]
        r["X"], r["id"], r["is3d"], r["observed"], r["dev_slot"], r["inv_depth"] = pb["mp_wpt"][m], pb["mp_id"][m], pb["mp_is3d"][m], 1, s, -1.0
        a, b = int(pb["obs_ptr"][m]), int(pb["obs_ptr"][m + 1])
        ents = [(int(kf_id[pb["obs_kf"][o]]), 7, pb["obs_px"][o]) for o in range(a, b)]
        if m % 7 == 0:
            ents.append((3, 1 | 2 | 4, np.array([5.0, 5.0], np.float32)))          # keyframe 3 is not in the table (ids start at 10)
        if m % 11 == 0:   # observed-by without a keypoint, in a keyframe of the table that does not observe the point otherwise
            free = [int(k) for k in kf_id[:-1] if int(k) not in {e[0] for e in ents}]
            if free:
                ents.append((free[0], 1, np.array([9.0, 9.0], np.float32)))
        ents.sort(key=lambda e: e[0])
        r["n_ent"], r["n_obs"], r["has_desc"] = len(ents), len(ents), 1 if b > a else 0
        for i, (kf, fl, px) in enumerate(ents):
            r["ent"][i]["kf"], r["ent"][i]["flags"], r["ent"][i]["px"] = kf, fl, px
        ops.append((s, 3, -1, None, 0))
        for o in range(a, b):
            ops.append((s, 0, int(kf_id[pb["obs_kf"][o]]), pb["obs_desc"][o], 0))
    table = torch.tensor([c.data_ptr() for c in chunks], dtype=torch.int64).cuda()
    return chunks, table, slot_of, ops, n_slots


@pytest.mark.gpu
def test_hip_record_form_equals_oracle_and_golden(ctx):
    """alva_match_to_map_records -- the form the System uses since round 5: records gathered out of pinned host memory, descriptors read
    from the device-resident tables -- on the golden problems and on a full-size one: the same matches as the oracle / the reference."""
    import torch
    from alvaar_amd.capi import MedoidStore
    problems = [(pb, aux, exp) for pb, aux, exp in _golden()]
    big = synth.make_match_to_map_problem(4000, 7)
    aux_big = py_match_to_map_aux(big)
    problems.append((big, aux_big, orc_match_to_map(big, aux_big)))
    for k, (pb, aux, exp) in enumerate(problems):
        cell_mp, local = flatten_match_to_map(pb, aux)
        chunks, table, slot_of, ops, n_slots = _records_of(pb, 100 + k)
        store = MedoidStore(ctx)
        store.replay(ops, n_slots)
        d = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
        out = ctx.match_to_map_records(pb["calib"], pb["cell_size"], aux["num_cells_w"], aux["grid_cells"], d(np.asarray(aux["cell_ptr"], np.int32)), d(cell_mp),
                                       d(np.asarray(pb["kf_id"], np.int32)), d(aux["kf_q"]), d(aux["kf_t"]), len(pb["kf_id"]) - 1, d(slot_of), table, store,
                                       pb["num_kp3d"], d(local)).cpu().numpy()
        got = {int(pb["mp_id"][m]): int(pb["mp_id"][out[m]]) for m in range(len(out)) if out[m] >= 0}
        assert got == exp, f"problem {k}"
        store.close()
        del chunks
