"""usage: python tools/show_bench.py <tag>  -- the figures of gpurun_out/<tag>_bench_n1.json + <tag>_bench_kernel_stats.csv + the PMC file"""
import json, csv, sys
tag = sys.argv[1]
d=json.load(open(f'gpurun_out/{tag}_bench_n1.json'))
print("value", round(d["value"]), "sustained", round(d["sustained"]["frames_per_s"]), "steps", d["sustained"]["steps"], "kf", d["sustained"]["keyframes"], "window kf", d["value_window"]["keyframes"], "lookahead", round(d["system_lookahead"]["frames_per_s"]), "host-fed", round(d["system_surface"]["frames_per_s"]), d["system_surface"]["caller_copy_us"])
print("720p", round(d["system_720p"]["frames_per_s"]), d["system_720p"]["ms_per_keyframe"], d["system_720p"]["tracking_frame_us"])
print("ba", d["local_ba"]["ms_per_solve"], d["local_ba"]["residual_block_iters_per_s"], "kernel_us", d["roofline_ba"]["kernel_us_per_solve"], d["roofline_ba"]["largest"])
print("streams", [(s["sessions"], round(s["frames_per_s"])) for s in d["system_streams"]], "group", [(s["sessions"], s["host_threads"], round(s["frames_per_s"])) for s in d["system_group"]])
print("frame sections", d["frame_sections_us"]["per_frame"], d["frame_sections_us"]["ms_per_keyframe"])
print("kf detail", d["frame_sections_us"]["per_keyframe_detail"])
print("klt", {k: d["klt"][k] for k in ("keypoint_levels_per_s","keypoint_levels_per_frame","kernel_us","l2_hit_rate")}, d["klt"]["pmc"]["stale"], d["klt"]["pmc"]["captured_at_commit"])
print("roofline", {k: d["roofline"][k] for k in ("bound","achieved","frac","traffic","avg_us","alg_bytes_per_launch")})
print("cpu", d["cpu_baseline"]["value"], d["cpu_baseline"]["eight_threads"]["value"], d["cpu_baseline"].get("system_cell40_shipped",{}).get("value"))
print("bounds", {k: round(v) for k,v in d["bounds"].items() if isinstance(v,(int,float))}, d["bounds"]["inputs"])
print("batch", d["local_ba_batch"]["ms_per_batch"], d["local_ba_batch"]["residual_block_iters_per_s"], d["two_view_init"]["ms_per_call"], "merge", {k: d["map_merge"].get(k) for k in ("records_this_rank","all_gather_us","fuse_us")})
print("kernels", {k: v["avg_us"] for k, v in d["kernels"].items()})
rows={r['Name'][:50]:(int(r['Calls']), float(r['AverageNs'])/1e3) for r in csv.DictReader(open(f'gpurun_out/{tag}_bench_kernel_stats.csv'))}
for k,(c,u) in sorted(rows.items(), key=lambda kv:-kv[1][0]*kv[1][1])[:14]: print("  ", k, c, round(u,1))
p=json.load(open('gpurun_out/r3_pmc_track_klt.json'))
print("pmc commit", p["commit"])
for k,v in p["kernels"].items(): print("  ", k[:30], round(v["hbm_bytes_per_launch"]/1e6,2), "MB L2", round(v["l2_hit_rate"],3))
