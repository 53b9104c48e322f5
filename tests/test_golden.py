"""The restatement oracle (CPU, always) and the HIP path (GPU) against the committed golden fixtures -- outputs of the
compiled reference (tests/golden/make_golden.py).  This is what pins parity on machines without oracle/_ref."""
from pathlib import Path

import numpy as np
import pytest

from alvaar_amd import synth
from oracles import Orc
from test_oracle_vs_ref import orb_key, xyz_problem

G = Path(__file__).resolve().parent / "golden"


@pytest.fixture(scope="module")
def img():
    return np.load(G / "image_stages.npz")


@pytest.fixture(scope="module")
def pose():
    return np.load(G / "pose_ba.npz")


def bits(a):
    return np.ascontiguousarray(a).view(np.uint32)


# ------------------------------------------------------------------------------------------ CPU: oracle vs golden
def test_orc_image_stages(img):
    assert np.array_equal(Orc.rgba2gray(img["rgba"]), img["gray_of_rgba"])
    gp, dp = Orc.build_pyramid(img["g"], 9, 3)
    for l in range(4):
        assert np.array_equal(gp[l], img[f"pyr_gray{l}"]) and np.array_equal(dp[l], img[f"pyr_deriv{l}"])
    assert np.array_equal(Orc.orb_blur(img["g"]), img["blur"])
    d, v = Orc.describe(img["g"], img["pts"])
    assert np.array_equal(d, img["desc"]) and np.array_equal(v, img["valid"])
    i, dd = Orc.bf_match(img["bf_q"], img["bf_t"])
    assert np.array_equal(i, img["bf_idx"]) and np.array_equal(dd, img["bf_dist"])


def test_orc_klt(img):
    nx, st, er = Orc.lk(img["klt_prev"], img["klt_curr"], img["klt_pts"], img["klt_init"], 3)
    assert np.array_equal(st, img["lk_status"]) and np.array_equal(bits(nx), bits(img["lk_next"]))
    ok = st.astype(bool)
    assert np.array_equal(bits(er[ok]), bits(img["lk_err"][ok]))
    pr, s2 = Orc.fbklt(img["klt_prev"], img["klt_curr"], img["klt_pts"], img["klt_init"], 3)
    assert np.array_equal(s2, img["fb_status"]) and np.array_equal(bits(pr), bits(img["fb_prior"]))


def test_orc_detectors(img):
    p, q = Orc.detect_grid(img["g"], 12)
    assert q == float(img["det_q"]) and np.array_equal(bits(p), bits(img["det_pts"]))
    xy, sc = Orc.fast(img["g"], 20)
    assert np.array_equal(xy, img["fast_xy"]) and np.array_equal(sc, img["fast_score"])
    kp, d = Orc.orb(img["orb_img"], 300)
    a, b = orb_key(kp), orb_key(img["orb_kp"])
    assert np.array_equal(bits(kp[a]), bits(img["orb_kp"][b])) and np.array_equal(d[a], img["orb_desc"][b])


def test_orc_pose_and_ba(pose):
    pb = synth.make_pnp_problem(150, 7, outlier_frac=0.2, pose_noise=0.02)
    ok, R, t, o = Orc.p3p_lmeds(pb["bv"], pb["wpt"])
    assert ok == bool(pose["p3p_ok"]) and np.abs(R - pose["p3p_R"]).max() < 1e-8 and np.abs(t - pose["p3p_t"]).max() < 1e-8
    assert np.array_equal(o, pose["p3p_outliers"])
    ok, p, o, info = Orc.pnp_refine(pb["uv"], pb["wpt"], pb["pose_init"], pb["K"])
    assert ok == bool(pose["pnp_ok"]) and np.abs(p - pose["pnp_pose"]).max() < 1e-9 and np.array_equal(o, pose["pnp_outliers"])
    assert info[0] == pose["pnp_info"][0] and info[4] == pose["pnp_info"][4]
    r = Orc.local_ba(synth.make_ba_problem(6, 120, 9), 5, 0.0)
    assert np.abs(r["poses"] - pose["ba_poses"]).max() < 1e-8 and np.abs(r["pts"] - pose["ba_pts"]).max() < 1e-7
    assert np.array_equal(r["info"][[0, 3]], pose["ba_info"][[0, 3]]) and np.array_equal(r["depth"], pose["ba_depth"])
    rx = Orc.local_ba(xyz_problem(5, 80, 10), 5, 0.0, inv_depth=False)
    assert np.abs(rx["poses"] - pose["bax_poses"]).max() < 1e-8 and np.abs(rx["pts"] - pose["bax_pts"]).max() < 1e-6


# ------------------------------------------------------------------------------------------ GPU: HIP path vs golden
@pytest.mark.gpu
def test_hip_image_stages(ctx, img):
    import torch
    import alvaar_amd
    cu = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
    assert np.array_equal(ctx.rgba2gray(cu(img["rgba"])).cpu().numpy(), img["gray_of_rgba"])
    g = cu(img["g"])
    pyr = alvaar_amd.Pyramid(ctx, 160, 120, 9, 3)
    pyr.build_from_gray(g)
    for l in range(4):
        hg, hd = pyr.download_level(l)
        assert np.array_equal(hg, img[f"pyr_gray{l}"]) and np.array_equal(hd, img[f"pyr_deriv{l}"])
    d, v = ctx.describe(g, cu(img["pts"]))
    assert np.array_equal(d.cpu().numpy(), img["desc"]) and np.array_equal(v.cpu().numpy(), img["valid"])
    i, dd = ctx.bf_match_hamming(cu(img["bf_q"]), cu(img["bf_t"]))
    assert np.array_equal(i.cpu().numpy(), img["bf_idx"]) and np.array_equal(dd.cpu().numpy(), img["bf_dist"])
    pp, cp = alvaar_amd.Pyramid(ctx, 160, 120, 9, 3), alvaar_amd.Pyramid(ctx, 160, 120, 9, 3)
    pp.build_from_gray(cu(img["klt_prev"]))
    cp.build_from_gray(cu(img["klt_curr"]))
    pr, st = ctx.fbklt_track(pp, cp, cu(img["klt_pts"]), cu(img["klt_init"]), 3)
    assert np.array_equal(st.cpu().numpy(), img["fb_status"]) and np.array_equal(bits(pr.cpu().numpy()), bits(img["fb_prior"]))
    p, q = ctx.detect_grid(g, 12)
    assert q == float(img["det_q"]) and np.array_equal(bits(p.cpu().numpy()), bits(img["det_pts"]))
    xy, sc = ctx.fast(g, 20)
    assert np.array_equal(xy.cpu().numpy(), img["fast_xy"]) and np.array_equal(sc.cpu().numpy(), img["fast_score"])
    orb = alvaar_amd.Orb(ctx, 320, 240, 300)
    kp, de = orb.detect_and_compute(cu(img["orb_img"]))
    b = orb_key(img["orb_kp"])
    assert np.array_equal(bits(kp.cpu().numpy()), bits(img["orb_kp"][b])) and np.array_equal(de.cpu().numpy(), img["orb_desc"][b])


@pytest.mark.gpu
def test_hip_pose_and_ba(ctx, pose):
    import torch
    pb = synth.make_pnp_problem(150, 7, outlier_frac=0.2, pose_noise=0.02)
    cu = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
    ok, R, t, o = ctx.p3p_lmeds(cu(pb["bv"]), cu(pb["wpt"]))
    assert ok == bool(pose["p3p_ok"]) and np.abs(R - pose["p3p_R"]).max() < 1e-8 and np.array_equal(o, pose["p3p_outliers"])
    ok, p, o, info = ctx.pnp_refine(cu(pb["uv"]), cu(pb["wpt"]), pb["pose_init"], pb["K"])
    assert ok == bool(pose["pnp_ok"]) and np.abs(p - pose["pnp_pose"]).max() < 1e-9 and np.array_equal(o, pose["pnp_outliers"])
    r = ctx.local_ba(synth.make_ba_problem(6, 120, 9), 5, 0.0)
    assert np.abs(r["poses"] - pose["ba_poses"]).max() < 1e-8 and np.abs(r["pts"] - pose["ba_pts"]).max() < 1e-7
    assert np.array_equal(r["info"][[0, 3]], pose["ba_info"][[0, 3]])
