"""Is the HIP path run-to-run deterministic?  The 560-frame cell-12 stream of the long differential test, RUNS (12) times through
alva::System with the SAME two-view pose injected (taken from the first run's own five-point result): every run must equal the first
bit for bit (status, pose bytes, keypoint ids / pixels, map-point table, descriptor medoids).  No reference involved."""
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
first = None
for run in range(int(os.environ.get("RUNS", "12"))):
    gpu = sysdiff.GpuSystem(w, h, 12)
    rec, bad = [], None
    for k, f in enumerate(frames):
        st, p7, _ = gpu.step(f, 33.0 * k)
        ids, px, un, i3, hd = gpu.frame_keypoints()
        mi, mx, mf, minv, md = gpu.map_points()
        cur = (st, p7.tobytes(), ids.tobytes(), px.tobytes(), mi.tobytes(), mx.tobytes(), md.tobytes(), tuple(gpu.state()))
        if first is None:
            rec.append(cur)
        elif cur != first[k]:
            what = [nm for nm, a, b in zip(("status", "pose", "ids", "pixels", "map ids", "map xyz", "medoids", "state"), cur, first[k]) if a != b]
            bad = (k, what)
            break
    gpu.close()
    if first is None:
        first = rec
        print(f"run 0: recorded {len(rec)} frames, {int(rec[-1][7][11])} keyframes created", flush=True)
    else:
        print(f"run {run}: {'identical' if bad is None else f'DIFFERS at frame {bad[0]}: {bad[1]}'}", flush=True)
