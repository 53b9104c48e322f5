"""GPU vs reference, frame by frame, without asserting: prints status / state equality / pose difference per frame and writes the two
stage-call logs (GPU stages; the same map layer over the reference's L1 stages) for tools/stage_trace_diff.py.
env: CELL, FRAMES, STREAM=crop|cropnoise|plane, INJECT=1"""
import os
import sys
sys.path.insert(0, ".")
sys.path.insert(0, "tests")
out = os.environ.get("OUT", "gpurun_out")
os.makedirs(out, exist_ok=True)
os.environ["ALVA_STAGE_TRACE"] = f"{out}/gpu.trace"
os.environ["ALVA_STAGE_TRACE_CPU"] = f"{out}/cpu.trace"
import numpy as np  # noqa: E402
from alvaar_amd import synth  # noqa: E402
import sysdiff  # noqa: E402

w, h = 640, 480
cell, n = int(os.environ.get("CELL", "40")), int(os.environ.get("FRAMES", "60"))
stream, inject = os.environ.get("STREAM", "crop"), os.environ.get("INJECT", "1") == "1"
canvas = synth.texture_canvas(w, h, 7 if stream.startswith("crop") else 5)
f = sysdiff.intrinsics(w, h)[0]


def frame(k):
    if stream == "crop":
        return synth.gray_to_rgba(synth.frame_gray(canvas, k, w, h))
    if stream == "cropnoise":
        return synth.gray_to_rgba(synth.frame_gray(canvas, k, w, h, noise_seed=11))
    return synth.plane_stream_frame(canvas, k, w, h, f, noise_seed=100)


ref, cpu, gpu = sysdiff.RefSystem(w, h, cell), sysdiff.CpuSystem(w, h, cell), sysdiff.GpuSystem(w, h, cell)
init = None
for k in range(n):
    fr = frame(k)
    s1, p1, _ = ref.step(fr, 33.0 * k)
    if init is None and s1 == 1 and inject:
        init = p1.copy()
        cpu.set_init_pose(init)
        gpu.set_init_pose(init)
    s2, p2, _ = cpu.step(fr, 33.0 * k)
    s3, p3, _ = gpu.step(fr, 33.0 * k)
    eq = s1 == s3 and list(ref.state()) == list(gpu.state())
    ka, kb = ref.frame_keypoints(), gpu.frame_keypoints()
    ids_eq = np.array_equal(ka[0], kb[0])
    dpx = float(np.abs(ka[1] - kb[1]).max()) if ids_eq and len(ka[0]) else -1
    print(f"{k:3d} status {s1} {s3}  state_eq {eq}  ids_eq {ids_eq}  dpx {dpx:.2e}  dpose gpu-ref {sysdiff.pose_diff(p1, p3):.2e}  cpu-ref {sysdiff.pose_diff(p1, p2):.2e}")
    if not eq:
        print("  ref", list(ref.state()))
        print("  gpu", list(gpu.state()))
        break
