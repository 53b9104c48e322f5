"""The host-fed surface (AlvaAR.findCameraPose = memImg.write + alva_system_find_camera_pose, src/system.js:175-177) against the resident-frame
loop on the same stream: frames/s of both and the caller's copy.  ALVA_NO_BAR_FRAME=1: the page-locked host buffer read over PCIe by the
image kernel (rounds 3 - 4) instead of the frame buffer in host-writable device memory.  env: FRAMES (1200), WINDOW (600)"""
import os, sys, time
sys.path.insert(0, ".")
import numpy as np
import bench_detail as bench

n, win = int(os.environ.get("FRAMES", "1200")), int(os.environ.get("WINDOW", "600"))
job = bench.SystemJob(0, 7, host_copy=True)
for k in range(n - win):
    job.step_host()
t0 = time.perf_counter()
for k in range(win):
    job.step_host()
dt_host = time.perf_counter() - t0
t0 = time.perf_counter()
for k in range(win):
    job.step()
dt_dev = time.perf_counter() - t0
t0 = time.perf_counter()
for i in range(100):
    np.copyto(job.ar.mem_img, job.host_frames[i])
copy_us = (time.perf_counter() - t0) / 100 * 1e6
print(f"host-fed {win / dt_host:.0f} frames/s | resident {win / dt_dev:.0f} frames/s | ratio {dt_dev / dt_host:.3f} | caller copy {copy_us:.1f} us | "
      f"frame buffer in device memory: {getattr(job.ar, '_bar_frame', False)} | status {job.status_hist}")
