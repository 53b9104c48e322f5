"""TEST INFRASTRUCTURE: one REPRODUCIBLE run of the reference's own System per call -- oracle/_ref/ref_run (oracle/ref_run.cpp), a
process that holds nothing but the reference, started without address-space randomisation.

The reference's result depends on its heap layout (Ceres orders parameter blocks by address): inside the test's
own process two runs on the same frames differ from the first local BA on and, on the long streams, about one run in five ends up on
another discrete path.  ref_run's heap holds the reference's allocations only, from a fixed base, so its records are a function of the
frames alone: the long-stream differentials of tests/test_gpu_system.py compare the HIP path with ONE such run, no second attempt."""
from __future__ import annotations

import os
import struct
import subprocess
import tempfile
from pathlib import Path

import numpy as np

import sysdiff

ROOT = Path(__file__).resolve().parent.parent
REF_RUN = ROOT / "oracle" / "_ref" / "ref_run"
MAGIC = 0x314A5241


def run_reference(base_gray, index, w, h, cell, clahe=False, dist=(0.0, 0.0, 0.0, 0.0), reset_at=(), timeout=1800):
    """base_gray: [n_base][h][w] uint8; index[k] = the base frame of step k (frame = (g, g, g, 255), timestamp 33 k).
    Returns (records, init_pose, final): per step the dict test_gpu_system._reference_run builds, and the final map's keyframes, from ONE run
    of oracle/_ref/ref_run."""
    if not REF_RUN.exists():
        raise FileNotFoundError(f"{REF_RUN} is missing: build it with oracle/build_ref_shim.sh (needs /root/reference)")
    base_gray = np.ascontiguousarray(base_gray, np.uint8)
    index = np.ascontiguousarray(index, np.int32)
    n = len(index)
    assert base_gray.shape[1:] == (h, w)
    fx, fy, cx, cy = sysdiff.intrinsics(w, h)
    rst = np.zeros(n, np.int32)
    rst[list(reset_at)] = 1
    with tempfile.TemporaryDirectory(dir="/tmp") as d:
        job, out = os.path.join(d, "job.bin"), os.path.join(d, "records.bin")
        with open(job, "wb") as f:
            f.write(struct.pack("<7i8d", MAGIC, w, h, cell, int(clahe), len(base_gray), n, fx, fy, cx, cy, *map(float, dist)))
            f.write(index.tobytes())
            f.write(rst.tobytes())
            f.write((33.0 * np.arange(n)).astype(np.float64).tobytes())
            f.write(base_gray.tobytes())
        # a fixed, minimal environment: the records must not depend on what the caller's shell exports
        subprocess.run([str(REF_RUN), job, out], check=True, timeout=timeout, env={"PATH": "/usr/bin:/bin", "LC_ALL": "C"}, cwd="/tmp",
                       stdout=subprocess.DEVNULL)
        buf = np.fromfile(out, np.uint8)
    rec, pos, init_pose = [], 0, None

    def take(dtype, count):
        nonlocal pos
        nbytes = np.dtype(dtype).itemsize * count
        a = buf[pos:pos + nbytes].view(dtype).copy()
        pos += nbytes
        return a
    for _ in range(n):
        st = int(take(np.int32, 1)[0])
        p7, p16, state = take(np.float64, 7), take(np.float32, 16), take(np.int32, 16)
        nk = int(take(np.int32, 1)[0])
        kps = (take(np.int32, nk), take(np.float32, 2 * nk).reshape(nk, 2), take(np.float32, 2 * nk).reshape(nk, 2), take(np.uint8, nk), take(np.uint8, nk))
        nf = int(take(np.int32, 1)[0])
        kfs = take(np.int32, nf)
        nm = int(take(np.int32, 1)[0])
        mps = (take(np.int32, nm), take(np.float64, 3 * nm).reshape(nm, 3), take(np.int32, 5 * nm).reshape(nm, 5), take(np.float64, nm),
               take(np.uint8, 32 * nm).reshape(nm, 32))
        if init_pose is None and st == 1:
            init_pose = p7.copy()
        rec.append(dict(status=st, pose7=p7, pose16=p16, state=state, kps=kps, kfs=kfs, mps=mps))
    final = RecordedRef()
    for _ in range(int(take(np.int32, 1)[0])):
        kf = int(take(np.int32, 1)[0])
        pose, info = take(np.float64, 7), take(np.int32, 6)
        nk = int(take(np.int32, 1)[0])
        ids, px, i3 = take(np.int32, nk), take(np.float32, 2 * nk).reshape(nk, 2), take(np.uint8, nk)
        nc = int(take(np.int32, 1)[0])
        final.kf[kf] = (pose, info, ids, px, i3, take(np.int32, 2 * nc).reshape(nc, 2))
    assert pos == len(buf), "ref_run's records were not consumed exactly"
    return rec, init_pose, final


class RecordedRef:
    """the final map's keyframes of a ref_run, behind the three methods sysdiff.compare_keyframes calls"""

    def __init__(self):
        self.kf = {}

    def keyframe_ids(self):
        return np.array(sorted(self.kf), np.int32)

    def keyframe(self, kfid):
        return self.kf[int(kfid)][:5]

    def covisibility(self, kfid):
        return self.kf[int(kfid)][5]

    def close(self):
        pass
