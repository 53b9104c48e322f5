import os, sys
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import numpy as np
from alvaar_amd import synth
import sysdiff
w, h, n = 640, 480, 200
canvas = synth.texture_canvas(w, h, 7)
base = [synth.gray_to_rgba(synth.frame_gray(canvas, k, w, h, noise_seed=11)) for k in range(n)]
period = 2 * (n - 1)
frames = [base[(k % period) if (k % period) < n else period - (k % period)] for k in range(560)]
first = None
for run in range(int(os.environ.get("RUNS", "5"))):
    ref = sysdiff.RefSystem(w, h, 12)
    rec = []
    for k, f in enumerate(frames):
        st, p7, _ = ref.step(f, 33.0 * k)
        mi, mx, mf, minv, md = ref.map_points()
        ids, px, un, i3, hd = ref.frame_keypoints()
        rec.append((st, p7.copy(), list(ref.state()), md.copy(), px.copy(), ids.copy()))
    ref.close()
    if first is None:
        first = rec
        print("run 0 recorded", flush=True)
        continue
    ok = True
    for k, (a, b) in enumerate(zip(first, rec)):
        if a[0] != b[0] or a[2] != b[2] or not np.array_equal(a[5], b[5]):
            print(f"run {run} frame {k}: status/state/ids differ"); ok = False; break
        if not np.array_equal(a[4], b[4]):
            print(f"run {run} frame {k}: pixels differ"); ok = False; break
        if not np.array_equal(a[3], b[3]):
            print(f"run {run} frame {k}: medoids differ ({int((a[3] != b[3]).any(axis=1).sum())} points)"); ok = False; break
        if not np.array_equal(a[1], b[1]):
            print(f"run {run} frame {k}: pose differs by {np.abs(a[1]-b[1]).max():.2e}"); ok = False; break
    print(f"run {run}: {'identical' if ok else 'DIFFERENT'}", flush=True)
