import os
import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests"))
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "ref: needs oracle/_ref/libalva_ref.so (the compiled reference)")


def _has_gpu():
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False


def pytest_collection_modifyitems(config, items):
    gpu = _has_gpu()
    import oracles
    have_ref = oracles.ref_available()
    for it in items:
        if "gpu" in it.keywords and not gpu:
            it.add_marker(pytest.mark.skip(reason="no GPU in this container"))
        if "ref" in it.keywords and not have_ref:
            it.add_marker(pytest.mark.skip(reason="oracle/_ref/libalva_ref.so not built"))


@pytest.fixture(scope="session")
def ctx():
    import alvaar_amd
    c = alvaar_amd.Context(0)
    yield c
    c.close()
