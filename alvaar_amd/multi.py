"""Multi-GPU plumbing: camera streams are independent, so the hot path shards one stream per GPU /
rank with NO collective on the data path.  The only collectives are
  * the max-over-ranks timing reduction of bench.py, and
  * the optional shared-map exchange named by BASELINE.json's north_star: one all_gather of fixed-size map-point
    records over RCCL/xGMI (backend "nccl" on ROCm) -- ~60 B x 3000 points = 180 KB per GPU, i.e. latency-bound; the
    reference has no multi-map behaviour to match (parity unpinned), the record layout and the fuse rule follow
    MapManager::mergeMapPoints' intent (src/slam/src/map_manager.cpp:428-513: the older point absorbs the newer).
torch.distributed is used as plumbing only; the same code runs on gloo/CPU for the world_size-2 tests."""
from __future__ import annotations

import os
from dataclasses import dataclass

import numpy as np
import torch
import torch.distributed as dist

RECORD_BYTES = 4 + 4 + 24 + 32  # stream id, point id, xyz (3 x f64), descriptor (256 bit)


@dataclass
class Shard:
    rank: int
    world: int
    local_rank: int

    @property
    def stream_seed(self) -> int:
        """seeds 7..14 for the 8-stream config"""
        return 7 + self.rank


def rig_cameras(n_cameras: int, shard: Shard) -> range:
    """cameras of a rig that this rank steps through alva_track_batch_* (block partition, first ranks one more when it does not
    divide): the cameras are independent, so a rig shards over the GPUs like streams do -- no collective on the data path"""
    base, extra = divmod(n_cameras, shard.world)
    start = shard.rank * base + min(shard.rank, extra)
    return range(start, start + base + (1 if shard.rank < extra else 0))


def shard_from_env() -> Shard:
    return Shard(int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("LOCAL_RANK", "0")))


def init_process_group(shard: Shard, backend: str | None = None) -> bool:
    if shard.world <= 1:
        return False
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    if backend is None:
        backend = "nccl" if torch.cuda.is_available() else "gloo"
    kw = {}
    if backend == "nccl":
        kw["device_id"] = torch.device("cuda", shard.local_rank)
    dist.init_process_group(backend, rank=shard.rank, world_size=shard.world, **kw)
    return True


def max_over_ranks(seconds: float, device: torch.device | str = "cpu") -> float:
    """bench.py's timing rule: the job takes as long as its slowest rank."""
    if not (dist.is_available() and dist.is_initialized()):
        return seconds
    t = torch.tensor([seconds], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def aggregate_rate(units_per_rank: int, seconds_local: float, device: torch.device | str = "cpu") -> float:
    """whole-job throughput = units processed by ALL ranks / max-over-ranks time (weak scaling)."""
    world = dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1
    return world * units_per_rank / max_over_ranks(seconds_local, device)


def pack_records(stream_id: int, ids: np.ndarray, xyz: np.ndarray, desc: np.ndarray, capacity: int) -> torch.Tensor:
    """fixed-capacity byte tensor [capacity, RECORD_BYTES]; unused rows have point id -1"""
    n = len(ids)
    assert n <= capacity and xyz.shape == (n, 3) and desc.shape == (n, 32)
    buf = np.zeros((capacity, RECORD_BYTES), np.uint8)
    rec = buf.view(np.dtype([("stream", "<i4"), ("id", "<i4"), ("xyz", "<f8", 3), ("desc", "u1", 32)]))[:, 0]
    rec["id"] = -1
    rec["stream"][:n] = stream_id
    rec["id"][:n] = ids
    rec["xyz"][:n] = xyz
    rec["desc"][:n] = desc
    return torch.from_numpy(buf)


def unpack_records(buf: torch.Tensor):
    a = buf.cpu().numpy().reshape(-1, RECORD_BYTES)
    rec = a.view(np.dtype([("stream", "<i4"), ("id", "<i4"), ("xyz", "<f8", 3), ("desc", "u1", 32)]))[:, 0]
    rec = rec[rec["id"] >= 0]
    return rec["stream"].copy(), rec["id"].copy(), rec["xyz"].copy(), rec["desc"].copy()


def all_gather_map(records: torch.Tensor) -> torch.Tensor:
    """one all_gather of every rank's fixed-size record block -> [world * capacity, RECORD_BYTES] on every rank"""
    if not (dist.is_available() and dist.is_initialized()):
        return records
    out = [torch.empty_like(records) for _ in range(dist.get_world_size())]
    dist.all_gather(out, records)
    return torch.cat(out, 0)


def _popcount_rows(x: np.ndarray) -> np.ndarray:
    return np.unpackbits(x, axis=-1).sum(-1)


def fuse_duplicates(stream, ids, xyz, desc, max_dist_m: float = 0.05, max_hamming: int = 51):
    """Deterministic fuse: a point is absorbed by an EARLIER record (lower (stream, id)) of another stream when it lies
    within max_dist_m and its descriptor is within max_hamming bits (0.2 * 256, state.hpp:60 mapMaxDescriptorDistance_).
    Returns (keep mask, absorbed_by index or -1)."""
    order = np.lexsort((ids, stream))
    keep = np.ones(len(ids), bool)
    absorbed = -np.ones(len(ids), np.int64)
    for pos, i in enumerate(order):
        prev = order[:pos]
        prev = prev[keep[prev] & (stream[prev] != stream[i])]
        if len(prev) == 0:
            continue
        d = np.linalg.norm(xyz[prev] - xyz[i], axis=1)
        cand = prev[d <= max_dist_m]
        if len(cand) == 0:
            continue
        ham = _popcount_rows(desc[cand] ^ desc[i])
        j = int(np.argmin(ham))  # first minimum = earliest record
        if ham[j] <= max_hamming:
            keep[i] = False
            absorbed[i] = cand[j]
    return keep, absorbed
