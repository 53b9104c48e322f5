"""Host round-trip probe: tiny kernel + stream sync, per call wall time."""
import sys, time, os
sys.path.insert(0, ".")
import torch
import alvaar_amd
from alvaar_amd import capi

for own in (False, True):
    ctx = alvaar_amd.Context(0, own_stream=own)
    rgba = torch.zeros((16, 64, 4), dtype=torch.uint8, device="cuda")
    gray = torch.zeros((16, 64), dtype=torch.uint8, device="cuda")
    for _ in range(20):
        ctx.rgba2gray(rgba, gray); capi.lib.alva_ctx_sync(ctx.h)
    t0 = time.perf_counter()
    N = 500
    for _ in range(N):
        ctx.rgba2gray(rgba, gray)
        capi.lib.alva_ctx_sync(ctx.h)
    dt = (time.perf_counter() - t0) / N * 1e6
    t0 = time.perf_counter()
    for _ in range(N):
        ctx.rgba2gray(rgba, gray)
    capi.lib.alva_ctx_sync(ctx.h)
    dl = (time.perf_counter() - t0) / N * 1e6
    print(f"own_stream={own}: launch+sync {dt:.1f} us/call, launch only {dl:.1f} us/call  (ROC_ACTIVE_WAIT_TIMEOUT={os.environ.get('ROC_ACTIVE_WAIT_TIMEOUT')})")
