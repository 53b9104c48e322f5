"""Shared by test_gpu_medoid.py (device tables, medoid.hip) and test_medoid_table.py (the same record on the host): the reference's MapPoint
driven operation by operation (oracle/_ref: ref_mappoint_desc_ops) and the operation sequences."""
import ctypes as C

import numpy as np

import oracles

CAP = 64


def ref_ops(first_kf, first_desc, ops):
    """ops: [(op, kf, desc32 | None)] -> per op (medoid, has, buckets, [(key, dist)...])"""
    L = oracles.ref_lib()
    n = len(ops)
    op = np.array([o[0] for o in ops], np.int32)
    kf = np.array([o[1] for o in ops], np.int32)
    desc = np.stack([o[2] if o[2] is not None else np.zeros(32, np.uint8) for o in ops]).astype(np.uint8)
    med, has = np.zeros((n, 32), np.uint8), np.zeros(n, np.uint8)
    bk, cnt = np.zeros(n, np.int32), np.zeros(n, np.int32)
    keys, dist = np.zeros((n, CAP), np.int32), np.zeros((n, CAP), np.float32)
    L.ref_mappoint_desc_ops.argtypes = [C.c_int, C.c_void_p, C.c_int] + [C.c_void_p] * 3 + [C.c_int] + [C.c_void_p] * 6
    fd = np.ascontiguousarray(first_desc, np.uint8) if first_desc is not None else None
    rc = L.ref_mappoint_desc_ops(first_kf, fd.ctypes.data if fd is not None else None, n, op.ctypes.data, kf.ctypes.data, desc.ctypes.data, CAP,
                                 med.ctypes.data, has.ctypes.data, bk.ctypes.data, cnt.ctypes.data, keys.ctypes.data, dist.ctypes.data)
    assert rc == 0
    return [(med[i], bool(has[i]), int(bk[i]), [(int(keys[i, j]), float(dist[i, j])) for j in range(cnt[i])]) for i in range(n)]


def sequences():
    rng = np.random.RandomState(3)
    pool = rng.randint(0, 256, (6, 32)).astype(np.uint8)          # few distinct descriptors => many exact ties

    def near(d, bits):
        e = d.copy()
        for b in rng.choice(256, bits, replace=False):
            e[b // 8] ^= 1 << (b % 8)
        return e
    seqs = []
    # 1. growth through 13 -> 29 -> 59 buckets with ties, then removals in an order that hits bucket heads, absent keys, keyframe 0
    ops = [(0, k, pool[k % 3] if k % 4 else near(pool[0], 3)) for k in range(1, 41)]
    ops += [(1, k, None) for k in (40, 1, 17, 99, 5, 29, 13, 26, 39)] + [(0, 0, pool[1])] + [(1, k, None) for k in range(2, 39)] + [(1, 0, None)]
    seqs.append(("growth+ties", 0, None, ops))
    # 2. a map point born with a descriptor (the constructor of map_manager.cpp:254-327), random adds / removes, identical descriptors
    ops = []
    alive = {7}
    for _ in range(300):
        if rng.rand() < 0.6 or len(alive) < 2:
            k = int(rng.randint(0, 45))
            ops.append((0, k, pool[rng.randint(6)] if rng.rand() < 0.7 else near(pool[rng.randint(6)], int(rng.randint(1, 40)))))
            alive.add(k)
        else:
            k = int(rng.choice(sorted(alive))) if rng.rand() < 0.85 else int(rng.randint(0, 60))
            if len(alive) > 1 or k not in alive:
                ops.append((1, k, None))
                alive.discard(k)
    seqs.append(("random", 7, pool[2], ops))
    # 3. down to the last observation (desc_ released, maps cleared, bucket count kept), then life goes on in the same object
    ops = [(0, k, near(pool[3], k)) for k in range(1, 20)] + [(1, k, None) for k in range(0, 20)] + [(0, k, pool[4]) for k in (3, 50, 16)] + [(1, 50, None)]
    seqs.append(("release+reuse", 0, pool[3], ops))
    return seqs




def map_layer_log(first_kf, first_desc, ops, want):
    """What slam/map.cpp logs for these MapPoint calls (MapPt::add_desc / remove_obs): per operation a list of (op, kf, desc, rehash_to) --
    op 0 add, 1 remove, 2 clear.  The observation set decides what a removal logs; the bucket count after the reference's own insert tells
    which inserts rehash (the product reads it from its FlatHash key set, which tests/cpp/flat_hash_vs_std.cpp pins to libstdc++)."""
    buckets, present, observed = 1, set(), {first_kf}
    head = []
    if first_desc is not None:  # MapPoint(id, kf, descriptor): emplace into a 1-bucket table -> 13 buckets
        head.append((0, first_kf, first_desc, 13))
        buckets, present = 13, {first_kf}
    per_op = []
    for i, (op, kf, d) in enumerate(ops):
        bk = want[i][2]
        log = []
        if op == 0:
            observed.add(kf)
            if kf not in present:
                log.append((0, kf, d, bk if bk != buckets else 0))
                present.add(kf)
        elif kf in observed:
            observed.discard(kf)
            if not observed:
                log.append((2, -1, None, 0))
                present.clear()
            elif kf in present:
                log.append((1, kf, None, 0))
                present.discard(kf)
        buckets = bk
        per_op.append(log)
    return head, per_op
