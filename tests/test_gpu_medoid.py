"""f1: MapPoint's descriptor tables / medoids on the device (medoid.hip) against the reference's OWN MapPoint class (oracle/_ref:
ref_mappoint_desc_ops drives addObservedKeyframeId + addDesc / removeObservedKeyframeId, map_point.cpp:66-181, and reports desc_, the
distance table in the container's iteration order and its bucket count after every operation).  The device side replays the same
operations one at a time through alva_medoid_replay and dumps the table after each: medoid bytes, !desc_.empty(), the (key, distance sum)
list IN ITERATION ORDER and the bucket count must be identical after every single operation.  Sequences are built to hit what decides
the medoid: ties (repeated descriptors: a stream that revisits a view), keyframe 0 (never chosen by the removal, map_point.cpp:123),
removals of absent keys, the 13 -> 29 -> 59 bucket growth, the release when the last observation goes, re-use after it.
The operation log fed to the device is written by medoid_cases.map_layer_log, a Python MODEL of what slam/map.cpp logs (which edit
becomes which operation, the bucket count passed along with an insert that rehashes) -- not a capture of the map layer's own log.  The
real log is exercised by every System differential (tests/test_gpu_system.py compares the medoid of every map point with the reference's
on every frame, device path) and, without a GPU, by tests/test_system_host_logic.py through the host build of the same tables."""
import numpy as np
import pytest

import oracles
import medoid_cases as mc

pytestmark = [pytest.mark.gpu, pytest.mark.skipif(not oracles.ref_available(), reason="compiled reference (oracle/_ref) not present")]


@pytest.mark.parametrize("name,first_kf,first_desc,ops", mc.sequences(), ids=lambda v: v if isinstance(v, str) else None)
def test_device_tables_equal_the_reference_mappoint_after_every_operation(name, first_kf, first_desc, ops):
    import alvaar_amd
    from alvaar_amd import capi
    want = mc.ref_ops(first_kf, first_desc, ops)
    head, per_op = mc.map_layer_log(first_kf, first_desc, ops, want)
    ctx = alvaar_amd.Context(0)
    store = capi.MedoidStore(ctx)
    slot, other = 5, 9          # a second map point gets unrelated operations in the same logs: tables must not interfere
    store.replay([(slot, 3, -1, None, 0), (other, 3, -1, None, 0)] + [(slot,) + o for o in head], 16)
    for i, (op, kf, d) in enumerate(ops):
        med, has, bk, entries = want[i]
        log = [(other, 0, 1000 + i, np.full(32, i % 251, np.uint8), {0: 13, 13: 29, 29: 59}.get(i, 0))] if i < 40 else []
        log += [(slot,) + o for o in per_op[i]]
        if log:
            store.replay(log, 16)
        g_med, g_has, g_entries, g_bk, g_over = store.dump(slot)
        assert not g_over
        assert g_has == has and g_bk == bk, (name, i, op, kf, g_has, has, g_bk, bk)
        assert g_entries == entries, (name, i, op, kf, g_entries, entries)
        if has:
            assert np.array_equal(g_med, med), (name, i, op, kf)
    # the light export agrees with the dump; the bystander holds its 40 descriptors
    desc, valid, info = store.export([slot, other])
    g_med, g_has, g_entries, g_bk, _ = store.dump(slot)
    assert bool(valid[0]) == g_has and info[0, 0] == len(g_entries) and (not g_has or np.array_equal(desc[0], g_med))
    assert info[1, 0] == min(len(ops), 40) and info[1, 2] == 0
    store.close()
