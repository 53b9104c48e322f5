"""CPU: include/alvaar_system.h and include/alvaar_hip.h are valid C (gcc) and C++ (g++) and `alva::System` exposes the
reference's method names with both the native (pointer) and the wasm32 (int offset) signatures."""
import subprocess
import tempfile
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent


def _compile(cmd, src, suffix):
    with tempfile.TemporaryDirectory() as d:
        f = Path(d) / ("t" + suffix)
        f.write_text(src)
        subprocess.check_call(cmd + ["-I", str(ROOT / "include"), "-fsyntax-only", str(f)])


def test_headers_are_plain_c():
    _compile(["gcc", "-std=c99", "-Wall", "-Werror"], '#include "alvaar_hip.h"\n#include "alvaar_system.h"\n#include "alvaar_system_testing.h"\nint main(void){return 0;}\n', ".c")


def test_system_class_signatures():
    src = r'''
#include "alvaar_system.h"
#include <cstdint>
int use(alva::System &s, const uint8_t *img, float *pose, int *pts, const double *imu) {
    s.configure(640, 480, 579.4, 579.4, 320.0, 240.0, 0, 0, 0, 0);
    s.reset();
    int a = s.findCameraPose(img, pose);
    int b = s.findCameraPoseWithIMU(img, imu, pose);
    int c = s.findPlane(pose, 250);
    int d = s.getFramePoints(pts);
    int (alva::System::*wasm_pose)(int, int) = &alva::System::findCameraPose;            // reference: system.hpp:32
    int (alva::System::*wasm_imu)(int, int, int) = &alva::System::findCameraPoseWithIMU;  // system.hpp:30
    int (alva::System::*wasm_plane)(int, int) = &alva::System::findPlane;                // system.hpp:34
    int (alva::System::*wasm_pts)(int) = &alva::System::getFramePoints;                  // system.hpp:36
    (void) wasm_pose; (void) wasm_imu; (void) wasm_plane; (void) wasm_pts;
    return a + b + c + d;
}
'''
    _compile(["g++", "-std=c++17", "-Wall", "-Werror"], src, ".cpp")
