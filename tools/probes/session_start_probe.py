"""Do the first frames of a fresh session depend on anything but their input?  Creates N sessions one after the other (each: configure +
warm-up, then K frames of the same resident stream) and compares the per-frame digests of keypoint ids + pixels with the first session's.
env: N (60), K (5), CELL (40), CHURN (1: allocate / free torch tensors of random sizes between sessions)"""
import os, sys, hashlib
sys.path.insert(0, ".")
import numpy as np
import torch
from alvaar_amd import synth
from alvaar_amd.system import AlvaAR

N, K, cell, churn = int(os.environ.get("N", "60")), int(os.environ.get("K", "5")), int(os.environ.get("CELL", "40")), os.environ.get("CHURN", "1") == "1"
W, H = 640, 480
canvas = synth.texture_canvas(W, H, 7)
fr = torch.from_numpy(np.stack([synth.gray_to_rgba(synth.frame_gray(canvas, k, W, H)) for k in range(K)])).cuda()
rng = np.random.RandomState(0)
ref, bad, junk = None, [], []
for s in range(N):
    ar = AlvaAR(W, H, cell_size=cell, random_sampling=False)
    dig = []
    for k in range(K):
        ar.find_camera_pose_device(int(fr[k].data_ptr()), 33.0 * k)
        ids, px, i3 = ar.keypoints()
        dig.append(hashlib.sha1(ids.tobytes() + px.tobytes()).hexdigest()[:12])
    ar.close()
    if ref is None:
        ref = dig
    elif dig != ref:
        bad.append((s, [k for k in range(K) if dig[k] != ref[k]]))
    if churn:
        junk.append(torch.empty(int(rng.randint(1, 64)) << 18, dtype=torch.uint8, device="cuda").fill_(1))
        if len(junk) > 3:
            junk.pop(int(rng.randint(0, len(junk))))
print(f"{N} sessions x {K} frames (cell {cell}): {len(bad)} differ from the first session's digests: {bad[:10]}")
