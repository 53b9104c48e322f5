#!/usr/bin/env python3
"""Generates tests/golden/*.npz from the COMPILED REFERENCE (oracle/_ref/libalva_ref.so = AlvaAR's slam sources +
vendored OpenCV 4.5.5 / Ceres 2.0.0 / OpenGV, built by oracle/build_ref_shim.sh in the container that has
/root/reference).  The reference has no golden vectors of its own; these are outputs of the
reference itself on small seeded inputs, committed so that the restatement oracle and the HIP path stay pinned on
machines where the reference library is absent.  Re-run: `python tests/golden/make_golden.py`."""
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests"))
from alvaar_amd import synth  # noqa: E402
from oracles import Ref, ref_lib  # noqa: E402
from test_oracle_vs_ref import klt_case, _test_points, xyz_problem  # noqa: E402

OUT = Path(__file__).resolve().parent


def img(w, h, seed, k=2, noise=True):
    return synth.frame_gray(synth.texture_canvas(w, h, seed), k, w, h, noise_seed=seed if noise else None)


def main():
    info = ref_lib().ref_build_info().decode()
    base = [l.strip() for l in info.splitlines() if "Baseline" in l or "Version control" in l]
    # image stages, 160x120
    w, h = 160, 120
    rgba = synth.random_rgba(w, h, 1)
    g = img(w, h, 2)
    gp, dp = Ref.build_pyramid(g, 9, 3)
    pts = _test_points(w, h, 64, 3)
    desc, valid = Ref.describe(g, pts)
    prev, curr, kpts, kinit = klt_case(w, h, 120, 4)
    lk_next, lk_st, lk_err = Ref.lk(prev, curr, kpts, kinit, 3)
    fb_prior, fb_st = Ref.fbklt(prev, curr, kpts, kinit, 3)
    det_pts, det_q = Ref.detect_grid(g, 12)
    fxy, fsc = Ref.fast(g, 20)
    g2 = img(320, 240, 5, noise=False)
    okp, odesc = Ref.orb(g2, 300)
    rng = np.random.RandomState(6)
    q = rng.randint(0, 256, (40, 32)).astype(np.uint8)
    t = rng.randint(0, 256, (50, 32)).astype(np.uint8)
    t[25] = t[3]
    q[7] = t[3]
    bf_idx, bf_dist = Ref.bf_match(q, t)
    np.savez_compressed(OUT / "image_stages.npz", rgba=rgba, gray_of_rgba=Ref.rgba2gray(rgba), g=g,
                        pyr_gray0=gp[0], pyr_gray1=gp[1], pyr_gray2=gp[2], pyr_gray3=gp[3],
                        pyr_deriv0=dp[0], pyr_deriv1=dp[1], pyr_deriv2=dp[2], pyr_deriv3=dp[3],
                        blur=Ref.orb_blur(g), pts=pts, desc=desc, valid=valid,
                        klt_prev=prev, klt_curr=curr, klt_pts=kpts, klt_init=kinit, lk_next=lk_next, lk_status=lk_st, lk_err=lk_err,
                        fb_prior=fb_prior, fb_status=fb_st, det_pts=det_pts, det_q=np.float64(det_q), fast_xy=fxy, fast_score=fsc,
                        orb_img=g2, orb_kp=okp, orb_desc=odesc, bf_q=q, bf_t=t, bf_idx=bf_idx, bf_dist=bf_dist,
                        ref_build=np.array(base))
    # pose + BA
    pb = synth.make_pnp_problem(150, 7, outlier_frac=0.2, pose_noise=0.02)
    ok, R, tt, outl = Ref.p3p_lmeds(pb["bv"], pb["wpt"])
    ok2, pose, outl2, pinfo = Ref.pnp_refine(pb["uv"], pb["wpt"], pb["pose_init"], pb["K"])
    ba = synth.make_ba_problem(6, 120, 9)
    rb = Ref.local_ba(ba, 5, 0.0)
    bx = xyz_problem(5, 80, 10)
    rx = Ref.local_ba(bx, 5, 0.0, inv_depth=False)
    np.savez_compressed(OUT / "pose_ba.npz", p3p_ok=ok, p3p_R=R, p3p_t=tt, p3p_outliers=outl, pnp_ok=ok2, pnp_pose=pose, pnp_outliers=outl2,
                        pnp_info=pinfo, ba_poses=rb["poses"], ba_pts=rb["pts"], ba_chi2=rb["chi2"], ba_depth=rb["depth"], ba_info=rb["info"][:4],
                        bax_poses=rx["poses"], bax_pts=rx["pts"], bax_info=rx["info"][:4])
    print("wrote", [p.name for p in OUT.glob("*.npz")])


def triangulation():
    from oracles import ref_triangulate
    pb = synth.make_triangulation_problem(200, 3, 11)
    ref, T = ref_triangulate(pb["pose_kf"], pb["pose_new"], pb["group"], pb["bvl"], pb["bvr"], pb["unpxl"], pb["unpxr"], pb["K"])
    np.savez_compressed(OUT / "triangulate.npz", T=T, group=pb["group"], bvl=pb["bvl"], bvr=pb["bvr"], unpxl=pb["unpxl"], unpxr=pb["unpxr"],
                        K=np.array(pb["K"]), pose_kf=pb["pose_kf"], pose_new=pb["pose_new"], **{"out_" + k: v for k, v in ref.items()})


def clahe():
    from oracles import ref_clahe
    g = img(160, 120, 21)
    np.savez_compressed(OUT / "clahe.npz", g=g, clip=np.float64(3.0), tiles=np.array([3, 2]), out=ref_clahe(g, 3.0, (3, 2)))


def distortion():
    from oracles import ref_undistort_points, ref_project_dist
    rng = np.random.RandomState(31)
    K, dist = np.array([520.0, 515.0, 318.5, 241.25]), np.array([-0.28, 0.07, 0.0002, -0.0003])
    px = rng.uniform(-40, [680, 520], (500, 2)).astype(np.float32)
    P = np.stack([rng.uniform(-3, 3, 500), rng.uniform(-2, 2, 500), rng.uniform(0.5, 9, 500)], 1)
    np.savez_compressed(OUT / "distortion.npz", K=K, dist=dist, px=px, P=P, und=ref_undistort_points(px, K, dist),
                        proj=ref_project_dist(P, K, dist))


def match_to_map():
    from oracles import ref_match_to_map
    cases = [dict(n=250, seed=41), dict(n=300, seed=42, dist=(-0.2, 0.05, 0.001, -0.001), px_noise=0.8, max_flips=40, twin_frac=0.3),
             dict(n=200, seed=43, px_noise=1.5, max_flips=70, twin_frac=0.6)]
    out = {"count": np.int32(len(cases))}
    for i, c in enumerate(cases):
        kw = {k: v for k, v in c.items() if k not in ("n", "seed")}
        pb = synth.make_match_to_map_problem(c["n"], c["seed"], **kw)
        ref, aux = ref_match_to_map(pb)
        for k, v in pb.items():
            out[f"p{i}_{k}"] = np.asarray(v)
        for k, v in aux.items():
            out[f"p{i}_aux_{k}"] = np.asarray(v)
        out[f"p{i}_exp"] = np.array(sorted(ref.items()), np.int32).reshape(-1, 2)
    np.savez_compressed(OUT / "match_to_map.npz", **out)


def relpose():
    import oracles as O
    """f2b: the compiled reference's RANSAC stage (model, inlier set, iteration count) and refined pose on four two-view problems."""
    out = {"n_cases": 4}
    for k, (n, seed, of) in enumerate([(150, 31, 0.25), (400, 32, 0.4), (64, 33, 0.1), (9, 34, 0.0)]):
        p = synth.make_relpose_problem(n, seed, of)
        ok, R, t, mask, iters = O.relpose_ransac(p["bv1"], p["bv2"], which="ref")
        ok2, Ropt, topt, outl = O.compute_5pt(p["bv1"], p["bv2"], which="ref")
        assert ok == ok2
        out.update({f"bv1_{k}": p["bv1"], f"bv2_{k}": p["bv2"], f"ok_{k}": ok, f"R_{k}": R, f"t_{k}": t, f"mask_{k}": mask, f"iters_{k}": iters,
                    f"Ropt_{k}": Ropt, f"topt_{k}": topt})
    np.savez_compressed(OUT / "relpose.npz", **out)


if __name__ == "__main__":
    match_to_map()
    triangulation()
    clahe()
    distortion()
    relpose()
    main()
