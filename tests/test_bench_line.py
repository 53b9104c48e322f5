"""bench.py's ONE stdout line (bench_common.compact_line) and its --gpus launcher rules.  Round 3's line had grown to 26 KB and the
driver recorded it as unparsed: the emitter now guarantees < 4 KB, and this test pins that on last round's full record, on a record
with absurdly long free-text fields, and on a minimal one."""
import json
import os
import subprocess
import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
import bench_common  # noqa: E402

CONTRACT = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config")


def _full_record():
    return json.loads((ROOT / "profiles" / "r3n_bench_n1.json").read_text())


def test_compact_line_of_a_full_record_is_small_and_round_trips():
    full = _full_record()
    assert len(json.dumps(full)) > 20000          # the canned input IS the record that broke the driver's parse
    line = bench_common.compact_line(full)
    assert "\n" not in line and len(line) < 4096
    got = json.loads(line)
    for k in CONTRACT:
        assert k in got, k
    assert got["value"] == pytest.approx(full["value"], rel=1e-5)
    assert set(got["roofline"]) >= {"kernel", "bound", "achieved", "peak", "unit", "frac", "traffic"}
    assert got["roofline"]["frac"] == pytest.approx(full["roofline"]["frac"], rel=1e-5)
    assert set(got["cpu_baseline"]) >= {"value", "unit", "cores", "kind", "sample"}
    assert got["local_ba"]["ms_per_solve"] == pytest.approx(full["local_ba"]["ms_per_solve"], rel=1e-5)
    assert "workload" in got["config"] and "model" not in got["config"]


def test_compact_line_survives_oversized_text_and_keeps_the_contract_keys():
    full = _full_record()
    full["config"]["workload"] = "w" * 9000
    full["cpu_baseline"]["sample"] = "s" * 9000
    full["config"]["parallelism"] = "p" * 3000
    line = bench_common.compact_line(full)
    assert len(line) < 4096
    got = json.loads(line)
    for k in CONTRACT + ("roofline", "cpu_baseline"):
        assert k in got, k


def test_compact_line_carries_the_secondary_figures_as_flat_keys():
    """32 sessions, the 1280x720 System stream and configs[2]'s frame are measured before the line is printed and
    travel in it as flat keys (bench.py, run_secondary(part="line")); the line of a round-4 record with them stays < 4 KB"""
    full = json.loads((ROOT / "profiles" / "r4i_bench_detail.json").read_text())
    full["system_group32"] = next(g_ for g_ in full["system_group"] if g_["sessions"] == 32)
    line = bench_common.compact_line(full)
    assert len(line) < 4096
    got = json.loads(line)
    assert got["group32_frames_per_s"] == pytest.approx(full["system_group32"]["frames_per_s"], rel=1e-4)
    assert got["system_720p_frames_per_s"] == pytest.approx(full["system_720p"]["frames_per_s"], rel=1e-4)
    assert got["orb720_us_per_frame"] == pytest.approx(1e3 * full["config_1280x720"]["ms_per_frame"], rel=1e-4)
    assert got["orb720_k_fast_nms_GBps"] > 0 and got["tracking_chain"]["launches"] >= 1
    assert "ms_per_keyframe" in got and "value_window" in got
    for k in CONTRACT + ("roofline", "cpu_baseline"):
        assert k in got, k


def test_compact_line_of_a_minimal_record():
    line = bench_common.compact_line({"metric": "m", "value": 1.0, "unit": "frames/s", "n_gpus": 1, "steps": 2, "warmup": 0, "ms_per_step": 1000.0,
                                      "config": {"workload": "x"}})
    assert json.loads(line)["value"] == 1.0


def _run(args, env_extra):
    env = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        env.pop(k, None)
    env.update(env_extra)
    return subprocess.run([sys.executable, str(ROOT / "bench.py"), *args], capture_output=True, text=True, timeout=300, env=env)


def test_more_gpus_than_visible_is_refused_not_downgraded():
    import torch
    n = torch.cuda.device_count() + 1
    r = _run(["--gpus", str(max(n, 2))], {})
    assert r.returncode != 0 and r.stdout.strip() == ""
    assert "refusing" in r.stderr


def test_world_size_must_agree_with_gpus():
    r = _run(["--gpus", "1"], {"WORLD_SIZE": "2", "RANK": "0", "LOCAL_RANK": "0"})
    assert r.returncode != 0 and r.stdout.strip() == ""
    assert "must agree" in r.stderr


@pytest.mark.gpu
def test_bench_through_the_launcher_prints_one_parseable_line():
    """the N > 1 launcher path (torch.distributed.run, one rank per GPU) with N = 1: one stdout line, < 4 KB, n_gpus == 1, RCCL group up"""
    r = _run(["--gpus", "1", "--launcher", "--quick", "--no-cpu-baseline", "--steps", "20", "--warmup", "5"], {})
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(lines) == 1 and len(lines[0]) < 4096
    got = json.loads(lines[0])
    assert got["n_gpus"] == 1 and got["value"] > 0 and got["roofline"]["frac"] > 0
    assert got["ms_per_step"] == pytest.approx(1e3 / got["value"], rel=1e-3)
    assert (ROOT / "bench_detail.json").exists()
