import sys, time
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import numpy as np, torch
import alvaar_amd, oracles as O
from alvaar_amd import synth
ctx = alvaar_amd.Context(0)
dev = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
thr = 2.0 * (1.0 - np.cos(np.arctan(np.float64(np.float32(3.0) / np.float32(579.4)))))
alld = []
for n, seed, of in [(200, 1, 0.2), (60, 4, 0.2), (1000, 3, 0.1), (500, 2, 0.3)]:
    p = synth.make_relpose_problem(n, seed, of)
    S = O.relpose_draw_samples(n, 256)
    b1, b2 = dev(p["bv1"]), dev(p["bv2"])
    models, counts = ctx.relpose_hypotheses(b1, b2, S)
    t0 = time.time(); models, counts = ctx.relpose_hypotheses(b1, b2, S); t1 = time.time()
    ds = []; cm = 0; okm = 0
    for k in range(len(S)):
        ok, R, t = O.relpose_model(p["bv1"], p["bv2"], S[k])
        if ok != (counts[k] >= 0): okm += 1; continue
        if not ok: continue
        d = max(np.abs(models[k, :9].reshape(3, 3) - R).max(), np.abs(models[k, 9:] - t).max()); ds.append(d)
        c = int((O.relpose_scores(p["bv1"], p["bv2"], R, t) < thr).sum())
        if d < 1e-9 and c != counts[k]: cm += 1
    ds = np.array(ds); alld += list(ds)
    print(n, "hyp: ok mismatch", okm, "quantiles 50/90/99/max", np.quantile(ds, [.5, .9, .99, 1.0]), ">1e-9:", (ds > 1e-9).sum(), "count mismatches", cm, "time 256 hyp %.3f ms" % ((t1 - t0) * 1e3))
alld = np.array(alld); print("all", len(alld), (alld > 1e-9).mean(), (alld > 1e-7).mean())
for n, seed, of in [(200, 1, 0.2), (500, 2, 0.3), (1000, 3, 0.1), (60, 4, 0.2), (300, 5, 0.5), (120, 6, 0.35), (2000, 8, 0.25), (40, 9, 0.1)]:
    p = synth.make_relpose_problem(n, seed, of)
    b1, b2 = dev(p["bv1"]), dev(p["bv2"])
    ok, R, t, mask, info = ctx.compute_5pt_essential(b1, b2)
    ts = []
    for _ in range(5):
        t0 = time.time(); ok, R, t, mask, info = ctx.compute_5pt_essential(b1, b2); ts.append(time.time() - t0)
    t0 = time.time(); ctx.compute_5pt_essential(b1, b2, optimize=False); tr = time.time() - t0
    oko, Ro, to, masko, iters = O.relpose_ransac(p["bv1"], p["bv2"])
    t0 = time.time(); ok2, R2, t2, out2 = O.compute_5pt(p["bv1"], p["bv2"]); tc = time.time() - t0
    m = np.array(info.ransac_model)
    print(n, "ok", ok, oko, "iters", info.iterations, iters, "inl", info.n_inliers, masko.sum(), "mask eq", np.array_equal(mask, masko),
          "ransac model diff %.1e" % max(np.abs(m[:9].reshape(3, 3) - Ro).max(), np.abs(m[9:] - to).max()),
          "| refined dR %.1e dtdir %.1e" % (np.abs(R - R2).max(), np.abs(t / np.linalg.norm(t) - t2 / np.linalg.norm(t2)).max()),
          "lm it/status/nfev", info.lm_iterations, info.lm_status, info.lm_nfev, "| gpu %.2f ms (ransac only %.2f) cpu oracle %.2f ms" % (min(ts) * 1e3, tr * 1e3, tc * 1e3))
