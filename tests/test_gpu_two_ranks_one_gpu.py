"""GPU: the cross-rank shared-map merge with TWO ranks on the ONE GPU a test box has (the fuse -> apply path had only ever
run with world size 1 on nccl, or on gloo with hand-made records).  Two processes, ranks 0 / 1, both on device 0, process group on gloo
(RCCL refuses two ranks on one device; test_gpu_rccl_merge.py keeps the nccl run for 2-GPU boxes).  Each rank runs a REAL alva::System
session of the same scene -- rank 1 starts 8 frames later, so it initialises its own map in its own gauge (origin, scale) -- packs its map
ON THE DEVICE (alva_system_pack_map_records), and goes through multi.map_merge_round(register=True): all_gather, descriptor-based
Sim(3) registration, fuse on the GPU, apply.  Asserted: the fused set is identical on both ranks and non-trivial, rank 1's absorbed points
carry rank 0's ids as shared ids, rank 0 applies nothing, and each session then keeps tracking exactly like a twin that never merged.
Parity unpinned (the reference has one map; semantics from MapManager::mergeMapPoints, map_manager.cpp:428-513)."""
import hashlib
import os
import socket
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))
W, H = 640, 480


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    sys.path.insert(0, HERE)
    sys.path.insert(0, os.path.dirname(HERE))
    import torch
    import torch.distributed as dist
    import alvaar_amd
    from alvaar_amd import multi, synth
    from alvaar_amd.system import AlvaAR
    torch.cuda.set_device(0)
    sh = multi.Shard(rank, world, 0)
    assert multi.init_process_group(sh, "gloo")
    canvas = synth.texture_canvas(W, H, 7)
    frames = torch.from_numpy(np.stack([synth.gray_to_rgba(synth.frame_gray(canvas, k, W, H)) for k in range(104)])).cuda()
    ar, twin = (AlvaAR(W, H, cell_size=24, random_sampling=False) for _ in range(2))
    off = 8 * rank
    st = 3
    for k in range(off, 90):
        st = ar.find_camera_pose_device(int(frames[k].data_ptr()), 33.0 * k)
        twin.find_camera_pose_device(int(frames[k].data_ptr()), 33.0 * k)
    assert st == 1
    ctx = alvaar_amd.Context(0)
    # the device-side pack equals the host-side export as a set of records
    blk, n = multi.system_map_records(ar, rank, 8192)
    os.environ["ALVA_HOST_MAP_PACK"] = "1"
    blk_h, n_h = multi.system_map_records(ar, rank, 8192)
    del os.environ["ALVA_HOST_MAP_PACK"]
    a, b = (np.sort(x.cpu().numpy().reshape(-1, multi.RECORD_BYTES)[:m].copy().view("V64").ravel()) for x, m in ((blk, n), (blk_h, n_h)))
    pack_equal = n == n_h and n > 100 and np.array_equal(a, b) and int((blk.cpu().numpy().reshape(-1, 64)[n:, 4:8].view(np.int32) == -1).all())
    dist.barrier()
    res = multi.map_merge_round(ar, ctx, sh, capacity=8192, register=True)
    loc, sst, sid = ar.shared_ids()
    # the fused set, as every rank derives it: digest of (stream, id, keep, absorbed_by) recomputed from the same gathered block
    allrec = multi.all_gather_map(multi.system_map_records(ar, rank, 8192)[0])
    reg = multi.register_streams(allrec, ctx)
    s2, i2, keep, absorbed = multi.fuse_duplicates(multi.apply_registration(allrec, reg), ctx)
    digest = hashlib.sha256(b"".join(np.ascontiguousarray(v.cpu().numpy()).tobytes() for v in (s2, i2, keep.to(torch.uint8), absorbed))).hexdigest()
    # tracking after the merge == the twin that never merged
    same = True
    for k in range(90, 104):
        s_a = ar.find_camera_pose_device(int(frames[k].data_ptr()), 33.0 * k)
        s_t = twin.find_camera_pose_device(int(frames[k].data_ptr()), 33.0 * k)
        same = same and s_a == s_t == 1 and np.array_equal(ar.pose7()[0].view(np.uint64), twin.pose7()[0].view(np.uint64))
    q.put(dict(rank=rank, pack_equal=bool(pack_equal), n=n, fused=res["fused"], applied=res["applied"], registered=res["registered"], gathered=res["records_gathered"],
               backend=res["backend"], shared=len(loc), shared_streams=sorted(set(int(x) for x in sst)), digest=digest, same=bool(same),
               pack_us=res["pack_us"], inliers=(res.get("stream_frames") or {}).get(1, {}).get("inliers", 0)))
    dist.barrier()
    dist.destroy_process_group()
    ar.close()
    twin.close()


def test_two_ranks_on_one_gpu_merge_real_session_maps():
    import torch.multiprocessing as mp
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted((q.get(timeout=400) for _ in range(world)), key=lambda d: d["rank"])
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    r0, r1 = res
    print(r0, r1)
    assert r0["backend"] == r1["backend"] == "gloo" and r0["gathered"] == r1["gathered"] == r0["n"] + r1["n"]
    assert r0["pack_equal"] and r1["pack_equal"]                       # device-side pack == host-side export
    assert r0["digest"] == r1["digest"]                                 # identical fused set on both ranks
    assert r0["fused"] == r1["fused"] >= 30 and r1["inliers"] >= 30     # the two gauges were registered and the maps fused
    assert r0["applied"] == 0 and r1["applied"] == r1["fused"] and r0["registered"] and r1["registered"]
    assert r1["shared"] == r1["fused"] and r1["shared_streams"] == [0] and r0["shared"] == 0
    assert r0["same"] and r1["same"]                                    # tracking continues bit-identically
    assert max(r0["pack_us"], r1["pack_us"]) < 5000
