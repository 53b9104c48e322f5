"""One-screen summary of a consolidated run: python tools/show_final.py <tag>   (reads gpurun_out/<tag>_*)"""
import json
import sys
t = sys.argv[1]
d = json.loads(open(f"gpurun_out/{t}_bench_line.json").read().strip().splitlines()[-1])
print("value", d["value"], "window", d["value_window"]["frames_per_s"], "ms/keyframe", d.get("ms_per_keyframe"), "host-fed", d["system_surface"]["frames_per_s"])
print("roofline", {k: d["roofline"][k] for k in ("kernel", "avg_us", "frac", "traffic")}, "| BA ms", d["local_ba"]["ms_per_solve"], "| cpu", d["cpu_baseline"]["value"])
D = json.load(open(f"gpurun_out/{t}_bench_detail.json"))
g = D["system_group"]
g = list(g.values()) if isinstance(g, dict) else g
print("group:", [(x["sessions"], x["host_threads"], x["lanes"], x["lockstep"], round(x["frames_per_s"]), x["ms_tracking_step_median"], x["ms_keyframe_step_mean"]) for x in g])
c = D["config_1280x720"]
print("720p ORB+match: us/frame", round(c["ms_per_frame"] * 1e3, 1), {k: v["avg_us"] for k, v in c["kernels"].items()})
print("system_720p", round(D["system_720p"].get("frames_per_s", 0)) if isinstance(D.get("system_720p"), dict) else D.get("system_720p"))
