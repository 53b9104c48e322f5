"""Repeats the 560-frame cell-12 differential until a frame's keypoint pixels differ from the reference's; prints the first differing
keypoints (ours, the reference's, ours on the previous frame).  env: RUNS (6)"""
import os, sys
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import numpy as np
from alvaar_amd import synth
import sysdiff
w, h, n = 640, 480, 200
canvas = synth.texture_canvas(w, h, 7)
base = [synth.gray_to_rgba(synth.frame_gray(canvas, k, w, h, noise_seed=11)) for k in range(n)]
period = 2 * (n - 1)
frames = [base[(k % period) if (k % period) < n else period - (k % period)] for k in range(int(os.environ.get("FRAMES", "560")))]
ref = sysdiff.RefSystem(w, h, 12)
rec = []
for k, f in enumerate(frames):
    st, p7, _ = ref.step(f, 33.0 * k)
    rec.append((st, p7.copy(), ref.frame_keypoints(), list(ref.state()), ref.map_points()))
for run in range(int(os.environ.get("RUNS", "6"))):
    gpu = sysdiff.GpuSystem(w, h, 12)
    prev = None
    bad = False
    for k, f in enumerate(frames):
        r = rec[k]
        if r[0] == 1 and (k == 0 or rec[k - 1][0] != 1):
            gpu.set_init_pose(r[1])
        st, p7, _ = gpu.step(f, 33.0 * k)
        ids, px, un, i3, hd = gpu.frame_keypoints()
        rids, rpx, run_, ri3, rhd = r[2]
        if st != r[0] or list(gpu.state()) != r[3] or not np.array_equal(ids, rids):
            print(f"run {run} frame {k}: status/state/ids differ", st, r[0]); bad = True; break
        d = np.abs(px - rpx).max(axis=1) if len(ids) else np.zeros(0)
        if (d > 0).any():
            j = np.nonzero(d > 0)[0]
            print(f"run {run} frame {k}: {len(j)} of {len(ids)} keypoints differ in pixels; pose diff {sysdiff.pose_diff(r[1], p7):.2e}")
            for q in j[:6]:
                pp = prev.get(int(ids[q])) if prev else None
                print("   id", int(ids[q]), "3d", int(i3[q]), "ours", px[q], "ref", rpx[q], "ours prev frame", pp, "unpx ours", un[q], "ref", run_[q])
            bad = True
            break
        mi, mx, mf, minv, md = gpu.map_points()
        ri, rx, rf, rinv, rd = r[4]
        if not (np.array_equal(mi, ri) and np.array_equal(mf, rf)):
            print(f"run {run} frame {k}: map point table differs"); bad = True; break
        if not np.array_equal(md, rd):
            j = np.nonzero((md != rd).any(axis=1))[0]
            print(f"run {run} frame {k}: {len(j)} descriptor medoids differ; keyframes {list(gpu.keyframe_ids())}")
            for q in j[:8]:
                bits = int(np.unpackbits(md[q] ^ rd[q]).sum())
                print("   map point", int(mi[q]), "flags", mf[q], "differing bits", bits, "in frame now:", int(mi[q]) in set(int(v) for v in ids))
            bad = True
            break
        prev = {int(i): p.copy() for i, p in zip(ids, px)}
    print(f"run {run}: {'MISMATCH' if bad else 'ok'}", flush=True)
    gpu.close()
    if bad:
        break
