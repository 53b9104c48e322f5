"""alvaar_amd -- MI355X-native hot path of AlvaAR's visual-SLAM front-end + local BA.

The compute lives in `libalvaar_hip.so` (hand-written HIP for gfx950 behind the C ABI of
`include/alvaar_hip.h`).  This package is the thin Python host mirror used by tests and
bench.py; PyTorch is used only for device memory and streams.  There is no CPU fallback:
touching any compute symbol fails loudly if the HIP library is missing.
"""
_LAZY = ("lib", "AlvaError", "Context", "Pyramid", "Orb", "Frontend", "TrackBatch", "check")


def __getattr__(name):  # lazy so that `python -m alvaar_amd.build` works before the .so exists
    if name in _LAZY:
        from . import capi
        return getattr(capi, name)
    raise AttributeError(name)
