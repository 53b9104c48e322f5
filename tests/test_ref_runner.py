"""CPU: oracle/_ref/ref_run (tests/ref_runner.py), the reproducible reference leg of the long-stream GPU differentials.

Pinned here: (1) two runs give the same records BIT FOR BIT, whatever the caller's environment and working directory; (2) the run
equals the in-process reference (sysdiff.RefSystem) in every discrete item on a stream short enough for that one to be stable;
(3) on a long stream it walks the discrete path of the committed recording (tests/golden/system_long_660_cell40.npz), the one the
HIP path equals (test_gpu_system.py::test_long_stream_equals_the_recorded_reference_run)."""
import os
import sys

import numpy as np
import pytest

from alvaar_amd import synth
import ref_runner
import sysdiff

pytestmark = [pytest.mark.ref]
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))


def _same(a, b, poses_bitwise=True):
    for k, (r, q) in enumerate(zip(a, b)):
        assert r["status"] == q["status"] and np.array_equal(r["state"], q["state"]), k
        for x, y in zip(r["kps"], q["kps"]):
            assert np.array_equal(x, y), k
        assert np.array_equal(r["kfs"], q["kfs"]), k
        for i, (x, y) in enumerate(zip(r["mps"], q["mps"])):
            if i in (1, 3) and not poses_bitwise:
                assert np.abs(x - y).max(initial=0.0) <= 1e-9, k
            else:
                assert np.array_equal(x, y), (k, i)
        if poses_bitwise:
            assert np.array_equal(r["pose7"], q["pose7"]) and np.array_equal(r["pose16"], q["pose16"]), k
        else:
            assert sysdiff.pose_diff(r["pose7"], q["pose7"]) <= 1e-9, k


def test_ref_run_is_reproducible_and_equals_the_in_process_reference(monkeypatch, tmp_path):
    w, h, n = 640, 480, 70
    canvas = synth.texture_canvas(w, h, 7)
    base = np.stack([synth.frame_gray(canvas, k, w, h) for k in range(n)])
    index = list(range(n))
    a, init_a, fin_a = ref_runner.run_reference(base, index, w, h, 40)
    monkeypatch.setenv("ALVA_SOME_LONG_VARIABLE", "x" * 4000)   # another environment, another working directory: the same bits
    monkeypatch.chdir(tmp_path)
    b, init_b, fin_b = ref_runner.run_reference(base, index, w, h, 40)
    _same(a, b)
    assert np.array_equal(init_a, init_b) and list(fin_a.keyframe_ids()) == list(fin_b.keyframe_ids())
    assert sum(r["status"] == 1 for r in a) >= 30 and len(fin_a.keyframe_ids()) >= 3
    ref = sysdiff.RefSystem(w, h, 40)
    live = []
    for k in range(n):
        st, p7, p16 = ref.step(synth.gray_to_rgba(base[k]), 33.0 * k)
        live.append(dict(status=st, pose7=p7.copy(), pose16=p16.copy(), state=ref.state().copy(), kps=tuple(x.copy() for x in ref.frame_keypoints()),
                         kfs=ref.keyframe_ids().copy(), mps=tuple(x.copy() for x in ref.map_points())))
    _same(a, live, poses_bitwise=False)   # (the in-process run's last bits follow ITS heap: DESIGN.md section 5)
    assert sysdiff.compare_keyframes(fin_a, ref, 1e-9) <= 1e-9
    ref.close()


def test_ref_run_walks_the_recorded_path_of_the_660_frame_stream():
    import make_system_long_golden as gold
    name = "660_cell40"
    w, h, n, steps, cell, cseed, nseed = gold.STREAMS[name]
    canvas = synth.texture_canvas(w, h, cseed)
    base = np.stack([synth.frame_gray(canvas, k, w, h, noise_seed=nseed) for k in range(n)])
    period = 2 * (n - 1)
    index = [(k % period) if (k % period) < n else period - (k % period) for k in range(steps)]
    rec, _, final = ref_runner.run_reference(base, index, w, h, cell)
    R = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", gold.FILES[name]))

    class View:   # make_system_long_golden.frame_record's view of one record
        def __init__(self, r):
            self.r = r

        def frame_keypoints(self):
            return self.r["kps"]

        def map_points(self):
            return self.r["mps"]

        def keyframe_ids(self):
            return self.r["kfs"]

        def state(self):
            return self.r["state"]
    for k, r in enumerate(rec):
        st, state, dig, _ = gold.frame_record(View(r), r["status"], r["pose7"])
        assert st == int(R["status"][k]) and np.array_equal(state, R["state"][k]) and np.array_equal(dig, R["digests"][k]), f"frame {k} leaves the recorded path"
        assert np.abs(r["pose7"] - R["pose7"][k]).max() <= 1e-9, k
    assert len(final.keyframe_ids()) >= 20
