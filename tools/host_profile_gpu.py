"""Where the host thread of ONE session spends its time ON THE GPU BOX: the product (alva_system_find_camera_pose_device, frames resident in
HBM) on the bench stream under a SIGPROF sampler (tools/sampler/sampler.c; process CPU time: the session's thread spins while it waits for
the GPU, so waits show up as samples inside HipStages).  KEYFRAME=1 (default) counts only samples under create_keyframe /
process_new_keyframe.  env: FRAMES (1400), WINDOW (800), HZ (5000)"""
import os, sys, time, subprocess, collections
sys.path.insert(0, ".")
sys.path.insert(0, "tests")
import numpy as np
import ctypes as C
import bench_detail as bench

n, win, hz = int(os.environ.get("FRAMES", "1400")), int(os.environ.get("WINDOW", "800")), int(os.environ.get("HZ", "5000"))
only_kf = os.environ.get("KEYFRAME", "1") == "1"
here = os.path.dirname(os.path.abspath(__file__))
so = os.path.join(here, "sampler", "libsampler.so")
if not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(os.path.join(here, "sampler", "sampler.c")):
    subprocess.check_call(["gcc", "-O2", "-fPIC", "-shared", "-o", so, os.path.join(here, "sampler", "sampler.c"), "-ldl"])
S = C.CDLL(so)
job = bench.SystemJob(0, 7, host_copy=False)
for k in range(n - win):
    job.step()
kf0 = int(job.ar.state()[11])
S.prof_start(hz)
cpu0 = time.perf_counter()
for k in range(win):
    job.step()
cpu_s = time.perf_counter() - cpu0
S.prof_stop()
nkf = int(job.ar.state()[11]) - kf0
cnt, depth = S.prof_count(), S.prof_depth()
buf = (C.c_void_p * (cnt * depth))()
S.prof_get(buf)
a = np.frombuffer(buf, dtype=np.uint64).reshape(cnt, depth)
S.prof_sym.argtypes = [C.c_void_p, C.c_char_p, C.c_int, C.c_char_p, C.c_int, C.POINTER(C.c_ulong)]
cache, modinfo = {}, {}
def sym(pc):
    if pc in cache:
        return cache[pc]
    nb, mb, off = C.create_string_buffer(512), C.create_string_buffer(512), C.c_ulong(0)
    if not S.prof_sym(C.c_void_p(int(pc)), nb, 512, mb, 512, C.byref(off)):
        r = ("?", "?")
    else:
        r = (nb.value.decode(), os.path.basename(mb.value.decode()))
        modinfo[pc] = (mb.value.decode(), off.value)
    cache[pc] = r
    return r
names = set()
for row in a:
    for pc in row:
        if pc == 0:
            break
        names.add(sym(pc)[0])
dem = {}
if names:
    out = subprocess.run(["c++filt"], input="\n".join(sorted(names)), capture_output=True, text=True).stdout.splitlines()
    dem = dict(zip(sorted(names), out))
incl, leaf, total_slam = collections.Counter(), collections.Counter(), 0
own_pcs = collections.Counter()
per = cpu_s / max(cnt, 1)   # seconds per sample (wall clock: the harness is one busy thread)
for row in a:
    st = []
    for pc in row:
        if pc == 0:
            break
        nm, mod = sym(pc)
        st.append((dem.get(nm, nm), mod))
    own = next((i for i, (f, m) in enumerate(st) if "alva_slam::" in f), None)
    if own is None:
        continue
    if only_kf and not any("create_keyframe" in f or "process_new_keyframe" in f for f, m in st):
        continue
    total_slam += 1
    leaf[st[own][0][:120]] += 1
    own_pcs[int(row[own])] += 1
    for f in dict.fromkeys(f for f, m in st[own:] if "alva_slam::" in f):
        incl[f[:120]] += 1
print(f"{cnt} samples, {per * 1e3:.3f} ms of wall clock each, over {win} frames ({win / cpu_s:.0f} frames/s under the sampler), {nkf} keyframes; {total_slam} counted "
      f"= {total_slam * per * 1e6 / max(nkf if only_kf else win, 1):.0f} us per {'keyframe' if only_kf else 'frame'}")
win = max(nkf, 1) if only_kf else win
print("-- innermost map-layer frame (self + libc / libstdc++ callees):")
for f, c in leaf.most_common(30):
    print(f"  {100.0 * c / total_slam:5.1f} %  {c * per * 1e6 / win:7.1f} us  {f}")
print("-- inclusive:")
for f, c in incl.most_common(24):
    print(f"  {100.0 * c / total_slam:5.1f} %  {f}")
# source lines of the innermost map-layer frames (the harness builds the map layer with -g; a return address points behind the call: - 1)
bymod = collections.defaultdict(list)
for pc, c in own_pcs.items():
    if pc in modinfo:
        bymod[modinfo[pc][0]].append((pc, modinfo[pc][1]))
lines, callers = collections.Counter(), collections.Counter()
for mod, lst in bymod.items():
    # -i: the whole inline chain of every address; a chain is reported innermost first, one line per level, chains separated by the
    # next address's echo (-a)
    out = subprocess.run(["addr2line", "-e", mod, "-C", "-i", "-a"] + [hex(max(o - 1, 0)) for _, o in lst], capture_output=True, text=True).stdout.splitlines()
    chains, cur_chain = [], None
    for ln in out:
        if ln.startswith("0x"):
            cur_chain = []
            chains.append(cur_chain)
        elif cur_chain is not None:
            cur_chain.append(ln.split("/")[-1].split(" ")[0])
    for (pc, _), ch in zip(lst, chains):
        if not ch:
            continue
        lines[ch[0]] += own_pcs[pc]
        # the first level of the chain that is the map layer's own source (not libstdc++ / fortify headers)
        own_src = next((x for x in ch if x.split(":")[0] in ("slam.hpp", "map.cpp", "mapper.cpp", "frontend.cpp", "flat_hash.hpp", "se3.hpp", "medoid_table.hpp", "inspect.hpp", "track_default.cpp")), ch[-1])
        callers[own_src + ("  <- " + ch[0] if ch[0] != own_src else "")] += own_pcs[pc]
print("-- source lines (innermost map-layer frame; inlined callees count at their own lines):")
for ln, c in lines.most_common(40):
    print(f"  {100.0 * c / total_slam:5.1f} %  {ln}")
print("-- the same, attributed to the map layer's own source line (with the library line it was in):")
for ln, c in callers.most_common(45):
    print(f"  {100.0 * c / total_slam:5.1f} %  {ln}")
