"""alva_match_to_map: kernel times on the GPU next to the CPU restatement (same flattened problem)."""
import sys, time
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import numpy as np, torch
import alvaar_amd
from alvaar_amd import synth, capi
from oracles import orc_match_to_map, flatten_match_to_map, py_match_to_map_aux

ctx = alvaar_amd.Context(0)
for n in (1000, 4000):
    pb = synth.make_match_to_map_problem(n, 7)
    aux = py_match_to_map_aux(pb)
    t0 = time.perf_counter(); exp = orc_match_to_map(pb, aux); t_cpu = time.perf_counter() - t0
    cell_mp, local = flatten_match_to_map(pb, aux)
    d = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
    args = (pb["calib"], pb["cell_size"], aux["num_cells_w"], aux["grid_cells"], d(aux["cell_ptr"]), d(cell_mp), d(aux["kf_q"]), d(aux["kf_t"]),
            d(pb["mp_wpt"]), d(pb["mp_is3d"]), d(pb["obs_ptr"]), d(pb["obs_kf"]), d(pb["obs_px"]), d(pb["obs_desc"]), len(pb["kf_id"]) - 1,
            pb["num_kp3d"], d(local))
    fn = lambda: ctx.match_to_map(*args)
    fn(); torch.cuda.synchronize()
    kt = capi.kernel_times(fn, 20)
    t0 = time.perf_counter()
    for _ in range(50): fn()
    torch.cuda.synchronize(); t_gpu = (time.perf_counter() - t0) / 50
    print(f"n={n}: map points {len(pb['mp_id'])}, local {len(local)}, frame keypoints {len(pb['frame_kp_order'])}, matches {len(exp)} | "
          f"CPU restatement {t_cpu * 1e3:.2f} ms | GPU call {t_gpu * 1e6:.0f} us | kernels " + ", ".join(f"{k} {v[1]:.1f} us" for k, v in kt.items()))
