"""CPU: the C-ABI library loads and exports every symbol include/*.h declares (alvaar_hip.h: the stages; alvaar_system.h: the drop-in
surface; alvaar_system_testing.h: the test-only hook), and the public surface header does not declare the test hook."""
import ctypes
import re
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent


def declared_symbols(header="alvaar_hip.h"):
    txt = (ROOT / "include" / header).read_text()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    txt = txt.split("#ifdef __cplusplus\n}")[0] if header == "alvaar_system.h" else txt   # (the C++ class behind the C block calls, not declares)
    return sorted(set(re.findall(r"\b(alva_[a-z0-9_]+)\s*\(", txt)))


def test_library_exports_every_declared_symbol():
    lib = ctypes.CDLL(str(ROOT / "alvaar_amd" / "libalvaar_hip.so"))
    syms = declared_symbols()
    assert len(syms) >= 10
    missing = [s for s in syms if not hasattr(lib, s)]
    assert not missing, f"declared in include/alvaar_hip.h but not exported: {missing}"
    for header in ("alvaar_system.h", "alvaar_system_testing.h"):
        syms = declared_symbols(header)
        assert syms, header
        missing = [s for s in syms if not hasattr(lib, s)]
        assert not missing, f"declared in include/{header} but not exported: {missing}"
    assert "alva_system_debug_set_init_pose" not in declared_symbols("alvaar_system.h")       # test-only: alvaar_system_testing.h
    assert declared_symbols("alvaar_system_testing.h") == ["alva_system_debug_set_init_pose"]


def test_no_cpu_fallback_without_device():
    import torch
    import alvaar_amd
    if torch.cuda.is_available():
        return
    try:
        alvaar_amd.Context(0)
    except alvaar_amd.AlvaError:
        return
    raise AssertionError("Context() must fail loudly without a HIP device")


def test_batched_driver_has_no_cpu_fallback_either():
    import torch
    import alvaar_amd
    if torch.cuda.is_available():
        return
    for make in (lambda: alvaar_amd.TrackBatch(0, 640, 480, 4, 100, 100), lambda: alvaar_amd.Frontend(0, 640, 480, 100, 500)):
        try:
            make()
        except alvaar_amd.AlvaError:
            continue
        raise AssertionError("the drivers must fail loudly without a HIP device")
