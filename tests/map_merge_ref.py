"""TEST INFRASTRUCTURE: the sequential statement of the shared-map fuse rule (alva_fuse_map_points evaluates it as a fixed point on the
GPU).  There is no reference behaviour behind it (the reference has one map: parity unpinned); this is the definition the kernel is
tested against."""
import numpy as np


def fuse_duplicates_sequential(stream, ids, xyz, desc, max_dist_m=0.05, max_hamming=51):
    """records in (stream, id) order: a point is absorbed by an EARLIER surviving record of another stream within max_dist_m whose
    descriptor is within max_hamming bits; smallest Hamming distance, earliest record on ties.  Returns (keep, absorbed_by)."""
    order = np.lexsort((ids, stream))
    assert np.array_equal(order, np.arange(len(ids))), "records must be sorted by (stream, id)"
    keep = np.ones(len(ids), bool)
    absorbed = -np.ones(len(ids), np.int64)
    for i in range(len(ids)):
        prev = np.flatnonzero(keep[:i] & (stream[:i] != stream[i]))
        if len(prev) == 0:
            continue
        d2 = ((xyz[prev] - xyz[i]) ** 2).sum(1)
        cand = prev[d2 <= max_dist_m * max_dist_m]
        if len(cand) == 0:
            continue
        ham = np.unpackbits(desc[cand] ^ desc[i], axis=-1).sum(-1)
        j = int(np.argmin(ham))
        if ham[j] <= max_hamming:
            keep[i] = False
            absorbed[i] = cand[j]
    return keep, absorbed
