"""The rBRIEF rotation floats (exact_sincos.hpp: a double-double evaluation rounded to float) against the C library's (float) cos / sin of
the promoted angle, on the host: same source the kernel compiles, 3 million angles over every binade up to 2 pi -- 0 mismatches, and 0
angles in the band the routine cannot prove (see the header)."""
import subprocess
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent


def test_exact_sincos_matches_libm(tmp_path):
    hdr = (ROOT / "alvaar_amd" / "csrc" / "exact_sincos.hpp").read_text().replace("#include <hip/hip_runtime.h>", "")
    (tmp_path / "exact_sincos_nohip.hpp").write_text(hdr)
    exe = tmp_path / "t"
    subprocess.check_call(["g++", "-O2", "-ffp-contract=off", f"-I{tmp_path}", "-o", str(exe), str(ROOT / "tests" / "cpp" / "exact_sincos_host.cpp"), "-lm"])
    n, mism, amb = map(int, subprocess.check_output([str(exe), "3000000"]).split())
    assert n == 3000000 and mism == 0 and amb == 0
