"""driven by tools/asan_host_logic.sh: the GPU-less harness's sanitizer build over the bench stream, check mode, a reset half way
env: CELL (12), CLAHE (0), DIST (unset: no lens distortion)"""
import os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
sys.path.insert(0, os.path.join(R, "tests"))
import ctypes as C
import oracles
oracles._ref = C.CDLL(os.environ["ALVA_ASAN_LIB"])
oracles._ref.ref_build_info.restype = C.c_char_p
import sysdiff
from alvaar_amd import synth

w, h, NF = 640, 480, 200
canvas = synth.texture_canvas(w, h, 7)
frames = [synth.gray_to_rgba(synth.frame_gray(canvas, k, w, h, noise_seed=11)) for k in range(NF)]
period = 2 * (NF - 1)
idx = lambda k: (k % period) if (k % period) < NF else period - (k % period)
os.environ["ALVA_CHECK_OBS_MIRROR"] = "1"
cell, clahe = int(os.environ.get("CELL", "12")), bool(int(os.environ.get("CLAHE", "0")))
dist = (0.1, -0.05, 0.001, 0.0005) if os.environ.get("DIST") else (0.0, 0.0, 0.0, 0.0)
s = sysdiff.CpuSystem(w, h, cell, clahe=clahe, dist=dist)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 500
for k in range(n):
    s.step(frames[idx(k)], 33.0 * k)
    if k == n // 2:
        s.reset()
    if k % 100 == 99:
        s.map_points()   # the inspection path too
print("sanitizer run finished:", n, "frames, state", [int(v) for v in s.state()])
