"""GPU: the optional shared-map merge on the REAL collective backend ("nccl" = RCCL on ROCm; north_star: "RCCL over xGMI only for the
optional shared-map merge", semantics from MapManager::mergeMapPoints, /root/reference/src/slam/src/map_manager.cpp:428-513 -- the
older point absorbs the newer; parity unpinned: the reference has one map).  A process group is initialised on nccl with world_size 1
(and 2 when two devices are visible), the records are packed in DEVICE memory, moved by ONE all_gather_into_tensor and fused by
alva_fuse_map_points; the result must equal the sequential statement of the rule (tests/map_merge_ref.py)."""
import os
import socket
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                      HSA_ENABLE_IPC_MODE_LEGACY="0")
    sys.path.insert(0, HERE)
    sys.path.insert(0, os.path.dirname(HERE))
    import torch
    import torch.distributed as dist
    import alvaar_amd
    from alvaar_amd import multi
    from map_merge_ref import fuse_duplicates_sequential
    from test_gpu_map_merge import _records
    torch.cuda.set_device(rank)
    sh = multi.shard_from_env()
    assert multi.init_process_group(sh, "nccl", force=True)
    assert dist.get_backend() == "nccl" and dist.get_world_size() == world
    # every rank owns the records of `per` streams of one synthetic 8-stream map (so world_size 1 still fuses across streams)
    st, ids, xyz, desc = _records(8, 400, 5)
    mine = [s for s in range(8) if s % world == rank]
    cap = 8 * 420 // world
    m = np.isin(st, mine)
    # pack_records writes ONE stream id per block; several streams per rank => pack per stream and concatenate into the rank's block
    blocks = [multi.pack_records(s, ids[st == s], xyz[st == s], desc[st == s], cap // len(mine)) for s in mine]
    block = torch.cat(blocks, 0)
    assert block.is_cuda and m.sum() > 0
    allrec = multi.all_gather_map(block)                        # RCCL all_gather_into_tensor of device memory
    assert allrec.is_cuda and allrec.shape[0] == world * block.shape[0]
    ctx = alvaar_amd.Context(rank)
    s2, i2, keep, absorbed = multi.fuse_duplicates(allrec, ctx)
    o = np.lexsort((ids, st))
    rk, ra = fuse_duplicates_sequential(st[o], ids[o], xyz[o], desc[o])
    ok = (np.array_equal(s2.cpu().numpy(), st[o]) and np.array_equal(i2.cpu().numpy(), ids[o]) and np.array_equal(keep.cpu().numpy(), rk)
          and np.array_equal(absorbed.cpu().numpy().astype(np.int64), ra))
    q.put((rank, bool(ok), int((~rk).sum()), dist.get_backend()))
    dist.barrier()
    dist.destroy_process_group()


def _run(world):
    import torch.multiprocessing as mp
    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=300) for _ in range(world))
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    for rank, ok, fused, backend in res:
        assert ok and fused > 0 and backend == "nccl", res


def test_map_merge_on_rccl_world_size_1():
    _run(1)


def test_map_merge_on_rccl_world_size_2():
    import torch
    if torch.cuda.device_count() < 2:
        pytest.skip("needs two GPUs (the driver's test box has one); the world_size-2 path is covered on gloo in test_multi_gloo.py")
    _run(2)


def test_system_map_goes_through_the_exchange():
    """the records of a live alva::System session (3-D map points + descriptor medoids) packed, gathered and fused: with one stream
    nothing may be fused (the rule only fuses across streams) and every record comes back"""
    import torch
    import alvaar_amd
    from alvaar_amd import multi, synth
    from alvaar_amd.system import AlvaAR
    w, h = 640, 480
    canvas = synth.texture_canvas(w, h, 7)
    ar = AlvaAR(w, h, cell_size=40, random_sampling=False)
    st = 3
    for k in range(60):
        _, st = ar.findCameraPose(synth.gray_to_rgba(synth.frame_gray(canvas, k, w, h)), 33.0 * k)
    assert st == 1
    ctx = alvaar_amd.Context(0)
    r = multi.map_merge_round(ar, ctx, multi.Shard(0, 1, 0), capacity=4096)
    assert r["records_this_rank"] > 50 and r["records_gathered"] == r["records_this_rank"] and r["fused"] == 0
    ar.close()
