"""GPU: the two buffers the HOST writes straight into device memory over the PCIe BAR -- the tracker's slot table
(stages_hip.hip track_reserve) and the caller's frame buffer (alva_system_alloc_frame_buffer, AlvaAR.mem_img) -- under the shipped
allocation flag (fine-grained), through many session start-ups.

What is guarded: with hipDeviceMallocUncached for both buffers a session's second tracking frame sometimes tracked from its FIRST frame's
slot table (8 of 13 runs of test_group_sessions_equal_their_solo_runs[one_lane], round 5); fine-grained memory never did.  The failure
needs fresh sessions (recycled pages) and shows within a session's first frames, so this test starts 32 sessions one after another --
device memory churned in between, so that the allocator hands back pages other kernels have just written -- feeds each the same 34
frames through the HOST surface (memImg.write into the BAR frame buffer, slot table written by the map layer) and requires every
session's statuses, poses, keypoint ids and pixels to equal the first session's bit for bit; then the same for six sessions advanced
together in one lane."""
import numpy as np
import pytest

from alvaar_amd import synth

pytestmark = [pytest.mark.gpu]


def _record(ar, st):
    ids, px, i3 = ar.keypoints()
    return int(st), ar.pose7()[0].view(np.uint64).copy(), ids.copy(), px.view(np.uint32).copy(), list(ar.state())


def _same(a, b, what):
    assert a[0] == b[0] and a[4] == b[4], what
    assert np.array_equal(a[1], b[1]) and np.array_equal(a[2], b[2]) and np.array_equal(a[3], b[3]), what


def test_host_written_device_buffers_through_32_session_starts():
    import torch
    from alvaar_amd.system import AlvaAR
    w, h, n = 640, 480, 34
    canvas = synth.texture_canvas(w, h, 7)
    frames = [synth.gray_to_rgba(synth.frame_gray(canvas, k, w, h, noise_seed=11)) for k in range(n)]
    first = None
    for rep in range(32):
        ar = AlvaAR(w, h, cell_size=12, random_sampling=False)
        if rep == 0:
            assert ar._bar_frame, "the BAR frame buffer is the shipped path on an MI355X box (large BAR)"
        rec = []
        for k in range(n):
            pose, st = ar.findCameraPose(frames[k], 33.0 * k)     # memImg.write: host stores into device memory
            rec.append(_record(ar, st))
        ar.close()
        assert sum(r[0] == 1 for r in rec) >= 10, "the stream was meant to track"
        if first is None:
            first = rec
        else:
            for k in range(n):
                _same(rec[k], first[k], f"session start {rep}, frame {k}: differs from the first session")
        # churn: other kernels write the pages the next session's buffers may be carved from
        junk = [torch.full((1 << 20,), rep, dtype=torch.int32, device="cuda") for _ in range(8)]
        torch.cuda.synchronize()
        del junk
        torch.cuda.empty_cache()


def test_one_lane_group_start_up_repeated():
    """the round-5 failing configuration's shape: sessions of one lane, repeated start-ups (the slot tables of several sessions written
    back to back, trackers of all of them in one launch)"""
    import torch
    from alvaar_amd.system import AlvaAR, SystemGroup
    specs = [(640, 480, 40, 7), (640, 480, 12, 5), (640, 480, 24, 9), (640, 480, 12, 7)]
    n = 30
    dev = []
    for w, h, cell, seed in specs:
        canvas = synth.texture_canvas(w, h, seed)
        dev.append(torch.from_numpy(np.stack([synth.gray_to_rgba(synth.frame_gray(canvas, k, w, h)) for k in range(n)])).cuda())
    first = None
    for rep in range(30):
        group = SystemGroup([], 2)
        group.set_lockstep(True)
        group.set_lanes(1)
        sessions = [AlvaAR(w, h, cell_size=cell, random_sampling=False) for w, h, cell, seed in specs]
        group.set_sessions(sessions)
        rec = [[] for _ in specs]
        for k in range(n):
            st = group.step_device([int(fr[k].data_ptr()) for fr in dev], 33.0 * k)
            for i, ar in enumerate(sessions):
                rec[i].append(_record(ar, st[i]))
        for ar in sessions:
            ar.close()
        group.close()
        if first is None:
            first = rec
        else:
            for i in range(len(specs)):
                for k in range(n):
                    _same(rec[i][k], first[i][k], f"group start {rep}, session {i}, frame {k}: differs from the first start")
