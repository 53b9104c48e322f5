"""Builds libalvaar_hip.so (HIP kernels + C ABI) in-tree with hipcc for gfx950.

`python -m alvaar_amd.build` or `__graft_entry__.build()`.  hipcc cross-compiles
without a GPU.  Objects are cached next to the sources and rebuilt when a
source or header is newer.
"""
from __future__ import annotations

import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor
from pathlib import Path

PKG = Path(__file__).resolve().parent
CSRC = PKG / "csrc"
ROOT = PKG.parent
LIB = PKG / "libalvaar_hip.so"

HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
# -ffp-contract=off: the float32 stages (Gaussian 7x7, min-eigenvalue, LK) must round exactly like
# the reference's non-FMA build ("RN, no FMA contraction").
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-fhip-fp32-correctly-rounded-divide-sqrt", "-Wall",
         "-Wno-unused-function", f"-I{ROOT / 'include'}"]
# one-off instrumented builds (tools/): ALVA_EXTRA_HIPFLAGS="-DALVA_KLT_COUNT" python -m alvaar_amd.build  (touch the source first)
FLAGS += os.environ.get("ALVA_EXTRA_HIPFLAGS", "").split()


def _newer(src: Path, dst: Path, deps: list[Path]) -> bool:
    if not dst.exists():
        return True
    t = dst.stat().st_mtime
    return any(p.stat().st_mtime > t for p in [src, *deps])


def build(verbose: bool = False) -> Path:
    # *.hip: kernels + C ABI; slam/*.cpp: the host-side map layer (plain C++, no HIP) behind alva_system_*
    srcs = sorted(CSRC.glob("*.hip")) + sorted((CSRC / "slam").glob("*.cpp"))
    hdrs = sorted(CSRC.glob("*.hpp")) + sorted((CSRC / "slam").glob("*.hpp")) + sorted((ROOT / "include").glob("*.h"))
    objs = []
    jobs = []
    for s in srcs:
        o = s.with_suffix(".o")
        objs.append(o)
        if _newer(s, o, hdrs):
            if s.suffix == ".cpp":
                # -mpopcnt: the descriptor medoids are popcount loops (every x86-64 CPU since 2008 has the instruction; without the flag
                # __builtin_popcountll is a bit-twiddling sequence)
                # -gline-tables-only: line numbers for tools/host_profile_gpu.py's sampled stacks (same code)
                jobs.append([HIPCC, "-x", "c++", "-O2", "-gline-tables-only", "-std=c++17", "-fPIC", "-ffp-contract=off", "-mpopcnt", "-Wall", "-c", str(s), "-o", str(o)])
            else:
                jobs.append([HIPCC, *FLAGS, "-c", str(s), "-o", str(o)])

    def run(cmd):
        if verbose:
            print(" ".join(cmd), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"hipcc failed: {' '.join(cmd)}\n{r.stdout}\n{r.stderr}")
        if verbose and r.stderr.strip():
            print(r.stderr, file=sys.stderr)

    with ThreadPoolExecutor(max_workers=min(8, max(1, len(jobs)))) as ex:
        list(ex.map(run, jobs))
    if jobs or not LIB.exists():
        run([HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", str(LIB), *map(str, objs)])
    return LIB


if __name__ == "__main__":
    print(build(verbose=True))
