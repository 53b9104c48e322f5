"""world_size-2 gloo tests (CPU) of the N>1 path: stream sharding, the max-over-ranks timing rule of bench.py and the
optional all_gather map exchange.  The data path itself has no collective (independent streams)."""
import os
import socket

import numpy as np
import torch
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from alvaar_amd import multi
    sh = multi.shard_from_env()
    assert multi.init_process_group(sh, "gloo")
    # each rank "processes" 100 frames; rank 1 is slower -> job time = slowest rank
    local = 0.5 if rank == 0 else 2.0
    rate = multi.aggregate_rate(100, local)
    # map exchange: 3 + rank points per stream, one duplicated across streams
    rng = np.random.RandomState(0)
    shared_xyz, shared_desc = np.array([[1.0, 2.0, 3.0]]), rng.randint(0, 256, (1, 32)).astype(np.uint8)
    n = 3 + rank
    r2 = np.random.RandomState(10 + rank)
    xyz = np.concatenate([shared_xyz + 0.001 * rank, r2.uniform(-5, 5, (n - 1, 3))])
    desc = np.concatenate([shared_desc, r2.randint(0, 256, (n - 1, 32)).astype(np.uint8)])
    rec = multi.pack_records(sh.rank, np.arange(n, dtype=np.int32), xyz, desc, capacity=8, device=torch.device("cpu"))   # gloo: host tensors
    assert multi.exchange_device().type == ("cuda" if torch.cuda.is_available() else "cpu")
    allrec = multi.all_gather_map(rec)
    assert allrec.shape == (world * 8, multi.RECORD_BYTES) and allrec.device == rec.device      # one collective, rank order
    st, ids, X, D = multi.unpack_records(allrec)
    from map_merge_ref import fuse_duplicates_sequential      # the rule's sequential statement (the product evaluates it on the GPU)
    keep, absorbed = fuse_duplicates_sequential(st, ids, X, D)
    q.put((rank, sh.stream_seed, rate, len(ids), int(keep.sum()), st.tolist()))
    torch.distributed.barrier()
    torch.distributed.destroy_process_group()


def test_two_rank_sharding_and_map_exchange():
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    (r0, seed0, rate0, n0, k0, st0), (r1, seed1, rate1, n1, k1, st1) = res
    assert (seed0, seed1) == (7, 8)                       # independent streams, seeds 7..
    assert abs(rate0 - 100.0) < 1e-9 and rate0 == rate1   # 2 ranks x 100 frames / max(0.5, 2.0) s
    assert n0 == n1 == 3 + 4 and st0 == st1               # every rank sees every record, same order
    assert k0 == k1 == 6                                  # the duplicated point is fused once, deterministically


def test_single_process_paths_need_no_process_group():
    from alvaar_amd import multi
    assert multi.max_over_ranks(1.5) == 1.5
    assert multi.aggregate_rate(10, 2.0) == 5.0
    rec = multi.pack_records(0, np.arange(2, dtype=np.int32), np.zeros((2, 3)), np.zeros((2, 32), np.uint8), 4)
    assert multi.all_gather_map(rec) is rec
    assert rec.device.type == multi.exchange_device().type     # records are packed where the exchange will read them: the GPU if there is one
    if not rec.is_cuda:
        import pytest
        with pytest.raises(RuntimeError):                       # the fuse has no CPU path
            multi.fuse_duplicates(rec, None)


def test_rig_cameras_partition():
    from alvaar_amd import multi
    for n in (1, 7, 8, 64, 100):
        for world in (1, 2, 3, 8):
            parts = [multi.rig_cameras(n, multi.Shard(r, world, r)) for r in range(world)]
            assert [c for p in parts for c in p] == list(range(n))          # every camera exactly once, in order
            assert max(len(p) for p in parts) - min(len(p) for p in parts) <= 1
