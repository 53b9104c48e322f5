"""GPU timeline of the tracking frame from a rocprofv3 kernel trace of tools/system_sustained.py: every kernel's start / end relative to
the frame's k_track_klt START, averaged over the tracking frames (no keyframe kernels between two trackers) of the last part of the run.
usage (GPU box): rocprofv3 --kernel-trace -d DIR -o tr --output-format csv -- python tools/system_sustained.py
                 python tools/frame_timeline.py DIR/*/tr_kernel_trace.csv"""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
short = lambda n: n.replace("(anonymous namespace)::", "").replace("alva_slam::", "").replace("void ", "").split("(")[0].split("<")[0]
ks = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), short(r["Kernel_Name"])) for r in rows)
klt = [i for i, k in enumerate(ks) if k[2] == "k_track_klt"]
chain = {"k_level0", "k_pyr_rest", "k_pyr_all", "k_track_klt", "k_pose_all", "k_track_compact", "k_p3p_pnp_s"}
acc, n = {}, 0
for f in range(max(1, len(klt) - 1500), len(klt) - 1):
    a, b, c = klt[f - 1], klt[f], klt[f + 1]
    if any(k[2] not in chain for k in ks[a:c]):
        continue   # a keyframe's kernels before or after this tracker
    t0 = ks[b][0]
    for s, e, name in ks[b:c]:   # this frame's tracker and everything up to (not including) the next frame's tracker
        v = acc.setdefault(name, [0.0, 0.0, 0])
        v[0] += (s - t0) / 1e3
        v[1] += (e - t0) / 1e3
        v[2] += 1
    v = acc.setdefault("next k_track_klt", [0.0, 0.0, 0])
    v[0] += (ks[c][0] - t0) / 1e3
    v[1] += (ks[c][1] - t0) / 1e3
    v[2] += 1
    n += 1
print(f"{n} tracking frames; us relative to the start of the frame's k_track_klt (start, end, duration):")
for k, v in sorted(acc.items(), key=lambda kv: kv[1][0] / kv[1][2]):
    print(f"  {k:36s} {v[0] / v[2]:8.1f} {v[1] / v[2]:8.1f} {(v[1] - v[0]) / v[2]:7.1f}   x{v[2]}")
