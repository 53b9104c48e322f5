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


def init_process_group(shard: Shard, backend: str | None = None, force: bool = False) -> bool:
    """force=True initialises the group for a single rank too (world_size 1): the shared-map exchange then runs through the real
    backend -- RCCL on a GPU box -- instead of the no-group short cut; MASTER_PORT must be set or free."""
    if shard.world <= 1 and not force:
        return False
    if shard.world <= 1:
        os.environ.setdefault("MASTER_PORT", "29531")
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


_REC = np.dtype([("stream", "<i4"), ("id", "<i4"), ("xyz", "<f8", 3), ("desc", "u1", 32)])


def exchange_device() -> torch.device:
    """where the map records live for the exchange: the rank's GPU whenever there is one (RCCL moves device memory; a host tensor
    is an error on the "nccl" backend), the CPU only in GPU-less (gloo) tests"""
    return torch.device("cuda", torch.cuda.current_device()) if torch.cuda.is_available() else torch.device("cpu")


def pack_records(stream_id: int, ids: np.ndarray, xyz: np.ndarray, desc: np.ndarray, capacity: int, device: torch.device | None = None) -> torch.Tensor:
    """fixed-capacity byte tensor [capacity, RECORD_BYTES] on `device` (default: exchange_device()); unused rows have point id -1"""
    n = len(ids)
    assert n <= capacity and xyz.shape == (n, 3) and desc.shape == (n, 32)
    buf = np.zeros((capacity, RECORD_BYTES), np.uint8)
    rec = buf.view(_REC)[:, 0]
    rec["id"] = -1
    rec["stream"][:n] = stream_id
    rec["id"][:n] = ids
    rec["xyz"][:n] = xyz
    rec["desc"][:n] = desc
    return torch.from_numpy(buf).to(device if device is not None else exchange_device())


def unpack_records(buf: torch.Tensor):
    a = buf.cpu().numpy().reshape(-1, RECORD_BYTES)
    rec = a.view(_REC)[:, 0]
    rec = rec[rec["id"] >= 0]
    return rec["stream"].copy(), rec["id"].copy(), rec["xyz"].copy(), rec["desc"].copy()


def all_gather_map(records: torch.Tensor) -> torch.Tensor:
    """ONE collective: every rank's fixed-size record block -> [world * capacity, RECORD_BYTES] on every rank, in rank order.  On the
    "nccl" backend (RCCL over xGMI) the block must be device memory: a host tensor is moved to the rank's GPU first."""
    if not (dist.is_available() and dist.is_initialized()):
        return records
    if dist.get_backend() == "nccl" and not records.is_cuda:
        records = records.to(exchange_device())
    records = records.contiguous()
    if dist.get_backend() == "gloo" and records.is_cuda:   # (two ranks on ONE GPU, tests: gloo moves host memory)
        host = records.cpu()
        out = torch.empty((dist.get_world_size() * host.shape[0], host.shape[1]), dtype=host.dtype)
        dist.all_gather_into_tensor(out, host)
        return out.to(records.device)
    out = torch.empty((dist.get_world_size() * records.shape[0], records.shape[1]), dtype=records.dtype, device=records.device)
    dist.all_gather_into_tensor(out, records)
    return out


def fuse_duplicates(records: torch.Tensor, ctx, max_dist_m: float = 0.05, max_hamming: int = 51):
    """Fuse the gathered records ON THE GPU (alva_fuse_map_points): a point is absorbed by the earliest surviving record of another stream
    within max_dist_m whose descriptor is within max_hamming bits (0.2 * 256, state.hpp:60 mapMaxDescriptorDistance_).
    records: [N, RECORD_BYTES] uint8 CUDA tensor (all_gather_map's output).  Returns (stream, id, keep mask, absorbed_by) as CUDA tensors
    over the valid records in (stream, id) order -- absorbed_by indexes that order."""
    import ctypes as C
    from .capi import lib, check
    if not records.is_cuda:
        raise RuntimeError("fuse_duplicates runs on the GPU: pass the device tensor all_gather_map returns (there is no CPU path)")
    rec = records.reshape(-1, RECORD_BYTES)
    ids_all = rec[:, 4:8].contiguous().view(torch.int32).reshape(-1)
    rec = rec[ids_all >= 0]
    stream = rec[:, 0:4].contiguous().view(torch.int32).reshape(-1)
    ids = rec[:, 4:8].contiguous().view(torch.int32).reshape(-1)
    order = torch.argsort(stream.to(torch.int64) * (1 << 32) + ids.to(torch.int64), stable=True)
    rec = rec[order]
    stream, ids = stream[order].contiguous(), ids[order].contiguous()
    xyz = rec[:, 8:32].contiguous().view(torch.float64).reshape(-1, 3).contiguous()
    desc = rec[:, 32:64].contiguous()
    n = int(stream.shape[0])
    keep = torch.empty(n, dtype=torch.uint8, device=rec.device)
    absorbed = torch.empty(n, dtype=torch.int32, device=rec.device)
    rounds = C.c_int(0)
    lib.alva_fuse_map_points.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_double, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
    torch.cuda.current_stream(rec.device).synchronize()
    check(lib.alva_fuse_map_points(ctx.h, n, stream.data_ptr(), xyz.data_ptr(), desc.data_ptr(), float(max_dist_m), int(max_hamming),
                                   keep.data_ptr(), absorbed.data_ptr(), C.byref(rounds)))
    return stream, ids, keep.bool(), absorbed


def system_map_records(ar, stream_id: int, capacity: int, device: torch.device | None = None):
    """The 3-D map points of one alva::System session (alva_system_debug_map_points: id, world position, descriptor medoid) as the
    fixed-capacity record block of the exchange; points without a descriptor are skipped, the newest are dropped beyond `capacity`
    (ids are handed out consecutively, so "older absorbs newer" keeps the established part of the map).  Returns (block, n_records)."""
    dev = device if device is not None else exchange_device()
    if dev.type == "cuda" and hasattr(ar, "pack_map_records") and not os.environ.get("ALVA_HOST_MAP_PACK"):
        # round 5: the block is written on the device from the resident map (records in pinned memory, descriptor tables in HBM): no
        # per-point export to the host, no numpy packing, no upload (21 ms -> well under a millisecond for ~5 000 points)
        block = torch.empty((capacity, RECORD_BYTES), dtype=torch.uint8, device=dev)
        try:
            n = ar.pack_map_records(stream_id, capacity, block)
        except Exception:   # noqa: BLE001 -- a session without a device-resident map yet: the host export below
            n = capacity + 1
        if n <= capacity:
            return block, int(n)
    ids, xyz, flags, inv, desc = ar.map_points(cap=262144)
    m = (flags[:, 0] != 0) & (flags[:, 4] > 0)   # is3d, at least one keyframe descriptor => a medoid (inspect_map_points)
    ids, xyz, desc = ids[m], xyz[m], desc[m]
    o = np.argsort(ids, kind="stable")[:capacity]
    return pack_records(stream_id, ids[o].astype(np.int32), xyz[o], desc[o], capacity, device), int(len(o))


def similarity_fit(src: np.ndarray, dst: np.ndarray):
    """Umeyama's closed-form similarity dst ~ s R src + t over matched 3-D points [n, 3] -> (s, R, t, rms residual)."""
    n = len(src)
    mu_s, mu_d = src.mean(0), dst.mean(0)
    a, b = src - mu_s, dst - mu_d
    U, D, Vt = np.linalg.svd(b.T @ a / n)
    S = np.eye(3)
    if np.linalg.det(U) * np.linalg.det(Vt) < 0:
        S[2, 2] = -1
    R = U @ S @ Vt
    var = (a ** 2).sum() / n
    s = float((D * np.diag(S)).sum() / var) if var > 0 else 1.0
    t = mu_d - s * R @ mu_s
    rms = float(np.sqrt((((s * (R @ src.T)).T + t - dst) ** 2).sum(1).mean()))
    return s, R, t, rms


def frames_are_registered(xyz_absorbed: np.ndarray, xyz_kept: np.ndarray, max_scale_dev=0.02, max_rot_deg=1.0, max_trans_m=0.05):
    """The merge's premise, checked on its own output: the fused pairs must be explained by the IDENTITY -- a similarity fitted to them may
    differ from it by less than 2 % in scale, 1 degree in rotation and 5 cm in translation.  Independent monocular maps (own gauge: own
    origin, orientation and scale) fail this -- or fuse nothing -- and are not merged; they have to be registered into one frame first."""
    if len(xyz_absorbed) < 4:
        return True, {"pairs": int(len(xyz_absorbed)), "note": "fewer than 4 pairs: nothing to fit"}
    s, R, t, rms = similarity_fit(xyz_absorbed, xyz_kept)
    ang = float(np.degrees(np.arccos(np.clip((np.trace(R) - 1) / 2, -1, 1))))
    ok = abs(s - 1) <= max_scale_dev and ang <= max_rot_deg and float(np.linalg.norm(t)) <= max_trans_m
    return ok, {"pairs": int(len(xyz_absorbed)), "scale": s, "rotation_deg": ang, "translation_m": float(np.linalg.norm(t)), "rms_m": rms}


def _split_records(records: torch.Tensor):
    """valid records of a gathered block in fuse_duplicates' (stream, id) order -> (stream, id, xyz [n,3] f64, desc [n,32] u8), CUDA tensors"""
    rec = records.reshape(-1, RECORD_BYTES)
    rec = rec[rec[:, 4:8].contiguous().view(torch.int32).reshape(-1) >= 0]
    stream = rec[:, 0:4].contiguous().view(torch.int32).reshape(-1)
    ids = rec[:, 4:8].contiguous().view(torch.int32).reshape(-1)
    order = torch.argsort(stream.to(torch.int64) * (1 << 32) + ids.to(torch.int64), stable=True)
    rec = rec[order]
    return (stream[order].contiguous(), ids[order].contiguous(), rec[:, 8:32].contiguous().view(torch.float64).reshape(-1, 3).contiguous(),
            rec[:, 32:64].contiguous())


def register_streams(records: torch.Tensor, ctx, max_hamming: int = 51, iters: int = 300, rel_tol: float = 0.03, min_inliers: int = 20):
    """Bring the maps of INDEPENDENT monocular sessions into one frame -- each has its own gauge (origin, orientation, SCALE) -- before the
    fuse rule compares positions: for every stream s > s0 (s0 = the lowest stream id in the block) a similarity x_s0 ~ s R x_s + t from
    descriptor correspondences alone.  Candidates: mutual nearest neighbours in Hamming distance between the two streams' descriptor
    medoids (alva_bf_match_hamming, both directions) within max_hamming bits; model: RANSAC over 3-point Umeyama fits (fixed seed: every
    rank derives the same transform from the same gathered block), inlier = residual below rel_tol x the spread of s0's matched points,
    refit on the inliers.  Returns {stream: dict(scale, R, t, inliers, candidates, rms)}; a stream with fewer than min_inliers stays out."""
    stream, ids, xyz, desc = _split_records(records)
    st_np = stream.cpu().numpy()
    streams = sorted(set(int(s) for s in st_np))
    out = {}
    if len(streams) < 2:
        return out
    s0 = streams[0]
    m0 = torch.from_numpy(st_np == s0).to(desc.device)
    d0, x0 = desc[m0].contiguous(), xyz[m0].cpu().numpy()
    for s in streams[1:]:
        ms = torch.from_numpy(st_np == s).to(desc.device)
        ds, xs = desc[ms].contiguous(), xyz[ms].cpu().numpy()
        if len(xs) < 3 or len(x0) < 3:
            continue
        i_s0, dist_s0 = ctx.bf_match_hamming(ds, d0)     # for every point of s: its nearest in s0
        i_0s, _ = ctx.bf_match_hamming(d0, ds)           # and back
        i_s0, dist_s0, i_0s = i_s0.cpu().numpy(), dist_s0.cpu().numpy(), i_0s.cpu().numpy()
        k = np.arange(len(xs))
        mutual = (i_0s[i_s0] == k) & (dist_s0 <= max_hamming)
        src, dst = xs[mutual], x0[i_s0[mutual]]
        info = {"candidates": int(mutual.sum()), "inliers": 0}
        if len(src) >= max(3, min_inliers):
            tol = rel_tol * float(np.sqrt(((dst - dst.mean(0)) ** 2).sum(1).mean()))
            rng = np.random.RandomState(12345 + s)
            best = None
            for _ in range(iters):
                pick = rng.choice(len(src), 3, replace=False)
                sc, R, t, _ = similarity_fit(src[pick], dst[pick])
                if not np.isfinite(sc) or sc <= 0:
                    continue
                res = np.sqrt((((sc * (R @ src.T)).T + t - dst) ** 2).sum(1))
                inl = res < tol
                if best is None or inl.sum() > best.sum():
                    best = inl
            if best is not None and best.sum() >= min_inliers:
                for _ in range(2):   # refit on the inliers, re-classify, refit
                    sc, R, t, rms = similarity_fit(src[best], dst[best])
                    best = np.sqrt((((sc * (R @ src.T)).T + t - dst) ** 2).sum(1)) < tol
                if best.sum() >= min_inliers:
                    sc, R, t, rms = similarity_fit(src[best], dst[best])
                    info.update(scale=sc, R=R, t=t, inliers=int(best.sum()), rms=rms, tol=tol)
        out[s] = info
    return out


def apply_registration(records: torch.Tensor, reg: dict) -> torch.Tensor:
    """the gathered block with every registered stream's positions moved into the reference stream's frame (x <- s R x + t)"""
    rec = records.reshape(-1, RECORD_BYTES).clone()
    valid = rec[:, 4:8].contiguous().view(torch.int32).reshape(-1) >= 0
    stream = rec[:, 0:4].contiguous().view(torch.int32).reshape(-1)
    xyz = rec[:, 8:32].contiguous().view(torch.float64).reshape(-1, 3)
    for s, r in reg.items():
        if "scale" not in r:
            continue
        msk = valid & (stream == s)
        R = torch.from_numpy(np.asarray(r["R"])).to(xyz.device)
        t = torch.from_numpy(np.asarray(r["t"])).to(xyz.device)
        xyz[msk] = r["scale"] * (xyz[msk] @ R.T) + t
    rec[:, 8:32] = xyz.contiguous().view(torch.uint8).reshape(-1, 24)
    return rec


def apply_merge(ar, my_stream: int, stream: np.ndarray, ids: np.ndarray, keep: np.ndarray, absorbed_by: np.ndarray, xyz: np.ndarray | None = None):
    """Make the fused set real for ONE session (`ar`, stream number `my_stream`): every map point of this session that the round absorbed
    into another stream's point gets that point's (stream, id) as its shared id (alva_system_set_shared_ids); two of this session's OWN
    points that ended up in the same shared point are one point -- the newer is merged into the older through the session's
    MapManager::mergeMapPoints path (alva_system_merge_map_points, map_manager.cpp:428-513).  Identical input on every rank => every
    rank derives the same shared ids.  Refuses (applies nothing) when the fused pairs contradict the one-world-frame premise."""
    stream, ids, keep, absorbed_by = (np.asarray(v) for v in (stream, ids, keep, absorbed_by))
    gone = np.flatnonzero(~keep.astype(bool))
    reg = {"pairs": int(len(gone))}
    if xyz is not None and len(gone):
        ok, reg = frames_are_registered(np.asarray(xyz)[gone], np.asarray(xyz)[absorbed_by[gone]])
        if not ok:
            return {"applied": 0, "local_merges": 0, "registered": False, "registration": reg}
    mine = gone[stream[gone] == my_stream]
    kept_idx = absorbed_by[mine]
    n_set = ar.set_shared_ids(ids[mine], stream[kept_idx], ids[kept_idx]) if len(mine) else 0
    # this session's points that share one kept point: merge the newer (higher id) into the oldest of them
    local_merges = 0
    groups: dict[int, list[int]] = {}
    for i, k in zip(mine, kept_idx):
        groups.setdefault(int(k), []).append(int(ids[i]))
    for k, members in groups.items():
        members.sort()
        if stream[k] == my_stream:          # (cannot happen with the cross-stream rule; kept for a rule that fuses inside a stream)
            members = [int(ids[k])] + members
        for newer in members[1:]:
            local_merges += int(ar.merge_map_points(newer, members[0]))
    return {"applied": int(n_set), "local_merges": local_merges, "registered": True, "registration": reg}


def map_merge_round(ar, ctx, shard: Shard, capacity: int = 16384, apply: bool = True, register: bool = False):
    """One shared-map merge (north_star: "RCCL over xGMI only for the optional shared-map merge"): pack this rank's map, ONE
    all_gather_into_tensor over the process group (RCCL when the backend is nccl), fuse the duplicates on the GPU, and APPLY the result to
    this rank's session (apply_merge: shared ids for its absorbed points, its own duplicates merged through MapManager::mergeMapPoints'
    path) -- after checking that the fused pairs agree with the one-world-frame premise (frames_are_registered).  register=True first
    moves every stream into the lowest stream's frame by a similarity estimated from descriptor correspondences (register_streams): what
    maps of independently initialised monocular sessions need.  Returns a dict of sizes and wall times; the fused set (and the
    similarities) are identical on every rank (same input, deterministic rules)."""
    import time
    t0 = time.perf_counter()
    block, n = system_map_records(ar, shard.rank, capacity)
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    allrec = all_gather_map(block)
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    reg = None
    if register:   # independent monocular maps: one frame first (similarities from descriptor correspondences), then the position rule
        reg = register_streams(allrec, ctx)
        allrec = apply_registration(allrec, reg)
    stream, ids, keep, absorbed = fuse_duplicates(allrec, ctx)
    kept = int(keep.sum().item())
    t3 = time.perf_counter()
    out = {"records_this_rank": n, "records_gathered": int(stream.shape[0]), "kept": kept, "fused": int(stream.shape[0]) - kept,
           "bytes_gathered": int(allrec.numel()), "backend": dist.get_backend() if dist.is_available() and dist.is_initialized() else None,
           "pack_us": (t1 - t0) * 1e6, "all_gather_us": (t2 - t1) * 1e6, "fuse_us": (t3 - t2) * 1e6}
    if apply:
        # the positions of the valid records in fuse_duplicates' (stream, id) order, for the registration check
        rec = allrec.reshape(-1, RECORD_BYTES)
        rec = rec[rec[:, 4:8].contiguous().view(torch.int32).reshape(-1) >= 0]
        st_all = rec[:, 0:4].contiguous().view(torch.int32).reshape(-1).to(torch.int64)
        id_all = rec[:, 4:8].contiguous().view(torch.int32).reshape(-1).to(torch.int64)
        order = torch.argsort(st_all * (1 << 32) + id_all, stable=True)
        xyz = rec[order][:, 8:32].contiguous().view(torch.float64).reshape(-1, 3).cpu().numpy()
        res = apply_merge(ar, shard.rank, stream.cpu().numpy(), ids.cpu().numpy(), keep.cpu().numpy(), absorbed.cpu().numpy().astype(np.int64), xyz)
        out.update(applied=res["applied"], local_merges=res["local_merges"], registered=res["registered"], registration=res["registration"],
                   apply_us=(time.perf_counter() - t3) * 1e6)
    if reg is not None:
        out["stream_frames"] = {int(s): {k_: (np.asarray(v).tolist() if isinstance(v, np.ndarray) else v) for k_, v in r.items()} for s, r in reg.items()}
    return out
