"""CPU: FlatHash (alvaar_amd/csrc/slam/flat_hash.hpp) -- the map layer's keypoint tables, local maps and descriptor tables -- iterates in
exactly libstdc++'s std::unordered_map / std::unordered_set order.  The reference's numerics depend on that order (P3P sample indices,
matchToMap ties, the local BA's gauge, the descriptor medoid: SURVEY.md 8c "canonicalise ... because unordered_map order leaks"), so the
emulation is driven side by side with the REAL containers through random operation sequences -- inserts across many rehashes, erases by
key and by iterator, clears (bucket count kept), copies, swaps, range inserts -- and compared after every step (tests/cpp/flat_hash_vs_std.cpp)."""
import subprocess
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent


def test_flat_hash_iterates_like_libstdcxx(tmp_path):
    exe = tmp_path / "flat_hash_vs_std"
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-Wall", "-o", str(exe), str(ROOT / "tests" / "cpp" / "flat_hash_vs_std.cpp")])
    out = subprocess.check_output([str(exe), "10"], text=True)
    assert out.startswith("ok ") and int(out.split()[1]) > 300000, out


def test_cell_index_floor_is_std_floor(tmp_path):
    """Frame::getKeypointCellIdx (frame.cpp:313-318) floors a float quotient; the map layer does it by truncation + correction
    (FrameRec::floor_to_int, slam.hpp): equal to (int) std::floor for every cell boundary +- ulps, dense grids and random bit patterns"""
    exe = tmp_path / "cell_index_floor"
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-Wall", "-o", str(exe), str(ROOT / "tests" / "cpp" / "cell_index_floor.cpp")])
    out = subprocess.check_output([str(exe)], text=True)
    assert out.strip().endswith(" 0 mismatches") and int(out.split()[0]) > 5000000, out


def test_map_point_record_behaves_like_the_containers_it_stands_for(tmp_path):
    """slam/mp_rec.hpp (round 5): one map point's observing keyframes / per-keyframe keypoint facts / descriptor keys as sorted entries of
    a fixed record + a parallel side arena of descriptor bytes, driven side by side with std::set / std::map through random operation
    sequences and compared after every step (tests/cpp/mp_rec_vs_std.cpp)"""
    exe = tmp_path / "mp_rec_vs_std"
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-Wall", "-o", str(exe), str(ROOT / "tests" / "cpp" / "mp_rec_vs_std.cpp")])
    out = subprocess.check_output([str(exe), "8"], text=True)
    assert out.startswith("ok ") and int(out.split()[1]) > 1000000, out
