"""GPU: the optional shared-map merge (north_star extension, parity unpinned): records packed into DEVICE memory (what the
"nccl" = RCCL backend needs), fused by alva_fuse_map_points; the kernel's fixed-point evaluation equals the sequential statement of the
rule (tests/map_merge_ref.py) exactly, including absorption chains across streams and Hamming ties."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _records(n_streams, n_per, seed, dup_frac=0.5, chain=True):
    rng = np.random.RandomState(seed)
    base_xyz = rng.uniform(-5, 5, (n_per, 3))
    base_desc = rng.randint(0, 256, (n_per, 32)).astype(np.uint8)
    st, ids, xyz, desc = [], [], [], []
    for s in range(n_streams):
        for i in range(n_per):
            if rng.rand() < dup_frac:                      # the same physical point seen by this stream too
                p = base_xyz[i] + rng.normal(0, 0.012, 3)
                d = base_desc[i].copy()
                for b in rng.randint(0, 256, rng.randint(0, 40)):
                    d[b >> 3] ^= np.uint8(1 << (b & 7))
            else:
                p = rng.uniform(-5, 5, 3)
                d = rng.randint(0, 256, 32).astype(np.uint8)
            st.append(s); ids.append(i); xyz.append(p); desc.append(d)
    if chain:                                                # exact duplicates: Hamming ties -> the earliest record must win
        for s in range(1, n_streams):
            st.append(s); ids.append(n_per); xyz.append(base_xyz[0] + 0.001 * s); desc.append(base_desc[0].copy())
    st, ids = np.array(st, np.int32), np.array(ids, np.int32)
    o = np.lexsort((ids, st))
    return st[o], ids[o], np.array(xyz)[o], np.array(desc, np.uint8)[o]


@pytest.mark.parametrize("n_streams,n_per,seed", [(2, 50, 0), (8, 300, 1), (8, 3000, 2), (3, 1, 3)])
def test_fuse_kernel_equals_sequential_rule(ctx, n_streams, n_per, seed):
    import torch
    from alvaar_amd import multi
    from map_merge_ref import fuse_duplicates_sequential
    st, ids, xyz, desc = _records(n_streams, n_per, seed)
    blocks = []
    cap = int(np.bincount(st).max()) + 3
    for s in range(n_streams):
        m = st == s
        blocks.append(multi.pack_records(s, ids[m], xyz[m], desc[m], cap))
    assert all(b.is_cuda for b in blocks)                    # packed into device memory: what RCCL's all_gather moves
    allrec = torch.cat(blocks, 0)                            # = all_gather_map's layout (rank order, fixed capacity, id -1 padding)
    s2, i2, keep, absorbed = multi.fuse_duplicates(allrec, ctx)
    assert np.array_equal(s2.cpu().numpy(), st) and np.array_equal(i2.cpu().numpy(), ids)
    rk, ra = fuse_duplicates_sequential(st, ids, xyz, desc)
    assert np.array_equal(keep.cpu().numpy(), rk)
    assert np.array_equal(absorbed.cpu().numpy().astype(np.int64), ra)
    if n_per >= 50:
        assert 0 < (~rk).sum() < len(rk)
