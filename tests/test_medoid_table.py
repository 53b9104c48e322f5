"""The descriptor-table record of f1 (alvaar_amd/csrc/slam/medoid_table.hpp) in its HOST build -- the form the GPU-less harness of
oracle/sys_cpu.cpp runs, and the statement medoid.hip spreads over a wavefront -- against the reference's own MapPoint class, operation by
operation: desc_, !desc_.empty(), the (key, distance sum) list in the container's ITERATION order and the bucket count (sequences and
what they aim at: tests/medoid_cases.py, test_gpu_medoid.py)."""
import ctypes as C

import numpy as np
import pytest

import oracles
import medoid_cases as mc

pytestmark = pytest.mark.ref
OP = np.dtype([("op", "<i4"), ("kf", "<i4"), ("rehash_to", "<i4"), ("next", "<i4"), ("desc", "u1", 32), ("pad", "<i4", 4)])
SLOT = np.dtype([("key", "<i4"), ("next", "<i4"), ("dist", "<f4"), ("pad", "<i4"), ("desc", "u1", 32)])


def apply(table, fresh, log):
    L = oracles.ref_lib()
    L.syscpu_medoid_apply.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p]
    rec = np.zeros(len(log), OP)
    for i, (op, kf, d, rh) in enumerate(log):
        rec[i]["op"], rec[i]["kf"], rec[i]["rehash_to"], rec[i]["next"] = op, kf, rh, -1
        if d is not None:
            rec[i]["desc"] = d
    n = L.syscpu_medoid_apply(table.ctypes.data, int(fresh), len(log), rec.ctypes.data)
    assert n == table.size


def view(table):
    hdr = table[:32].view(np.int32)
    slots = table[64:64 + 48 * SLOT.itemsize].view(SLOT)
    out, s = [], int(hdr[0])
    while s != -1 and len(out) <= 48:
        out.append((int(slots[s]["key"]), float(slots[s]["dist"])))
        s = int(slots[s]["next"])
    return table[32:64], bool(hdr[5]), out, int(hdr[3]), bool(hdr[6])


@pytest.mark.parametrize("name,first_kf,first_desc,ops", mc.sequences(), ids=lambda v: v if isinstance(v, str) else None)
def test_host_tables_equal_the_reference_mappoint_after_every_operation(name, first_kf, first_desc, ops):
    want = mc.ref_ops(first_kf, first_desc, ops)
    head, per_op = mc.map_layer_log(first_kf, first_desc, ops, want)
    table = np.zeros(64 + 48 * SLOT.itemsize + 60 * 4, np.uint8)
    apply(table, True, head)
    for i, (op, kf, d) in enumerate(ops):
        med, has, bk, entries = want[i]
        apply(table, False, per_op[i])
        g_med, g_has, g_entries, g_bk, g_over = view(table)
        assert not g_over
        assert g_has == has and g_bk == bk, (name, i, op, kf, g_has, has, g_bk, bk)
        assert g_entries == entries, (name, i, op, kf, g_entries, entries)
        if has:
            assert np.array_equal(g_med, med), (name, i, op, kf)
