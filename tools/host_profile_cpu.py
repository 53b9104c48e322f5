"""Where the host-side map layer spends its time, WITHOUT a GPU and without perf: the product's slam/*.cpp over the reference's CPU stages
(oracle/_ref: syscpu_*, test infrastructure) on the bench stream under a SIGPROF sampler (tools/sampler/sampler.c).  Only samples whose
stack passes through alva_slam:: are counted; the stage calls (CPU OpenCV / Ceres under syscpu stages) are excluded by name.
env: FRAMES (700), WINDOW (400), HZ (2000); REPLAY=1: the sampled pass runs over a tape of stage results recorded by a first pass (see
tools/host_replay_cpu.py) -- the map layer with its own working set in the caches, as over a device; KF_ONLY=1 samples keyframe frames only"""
import os, sys, time, subprocess, collections
sys.path.insert(0, ".")
sys.path.insert(0, "tests")
import numpy as np
import ctypes as C
import sysdiff
from alvaar_amd import synth

n, win, hz = int(os.environ.get("FRAMES", "700")), int(os.environ.get("WINDOW", "400")), int(os.environ.get("HZ", "5000"))
w, h, cell = 640, 480, 12
NF = 200
canvas = synth.texture_canvas(w, h, 7)
frames = [synth.gray_to_rgba(synth.frame_gray(canvas, k, w, h, noise_seed=11)) for k in range(NF)]
period = 2 * (NF - 1)
idx = lambda k: (k % period) if (k % period) < NF else period - (k % period)
here = os.path.dirname(os.path.abspath(__file__))
so = os.path.join(here, "sampler", "libsampler.so")
if not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(os.path.join(here, "sampler", "sampler.c")):
    subprocess.check_call(["gcc", "-O2", "-fPIC", "-shared", "-o", so, os.path.join(here, "sampler", "sampler.c"), "-ldl"])
S = C.CDLL(so)
s = sysdiff.CpuSystem(w, h, cell)
kf_frames = None
if os.environ.get("REPLAY"):
    s.L.syscpu_tape_new.restype = C.c_void_p
    s.L.syscpu_attach_tape.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
    tape = C.c_void_p(s.L.syscpu_tape_new())
    s.L.syscpu_attach_tape(s.h, tape, 0)
    kf_frames, made = set(), 0
    for k in range(n):
        s.step(frames[idx(k)], 33.0 * k)
        now = int(s.state()[11])
        if now != made:
            kf_frames.add(k)
        made = now
    s = sysdiff.CpuSystem(w, h, cell)
    s.L.syscpu_attach_tape(s.h, tape, 1)
for k in range(n - win):
    s.step(frames[idx(k)], 33.0 * k)
kf_only = bool(os.environ.get("KF_ONLY")) and kf_frames is not None   # sample the frames that made a keyframe in the recorded pass
S.prof_start(hz)
cpu_s, n_sampled = 0.0, 0
for k in range(n - win, n):
    on = not kf_only or k in kf_frames
    S.prof_pause(0 if on else 1)
    c0 = time.perf_counter()
    s.step(frames[idx(k)], 33.0 * k)
    if on:
        cpu_s += time.perf_counter() - c0
        n_sampled += 1
S.prof_stop()
win = max(n_sampled, 1)
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
    own = next((i for i, (f, m) in enumerate(st) if "alva_slam::" in f and "alva_slam::Stages::" not in f), None)
    if own is None:
        continue
    # callee side of the innermost map-layer frame: libc / libstdc++ helpers belong to it; anything else inside the reference library
    # (its CPU stages: OpenCV, Ceres, OpenGV, the harness's stage class, the default Stages) is stage time, not map-layer time
    if any(m.startswith("libalva_ref") for f, m in st[:own]):
        continue
    total_slam += 1
    leaf[st[own][0][:120]] += 1
    own_pcs[int(row[own])] += 1
    for f in dict.fromkeys(f for f, m in st[own:] if "alva_slam::" in f):
        incl[f[:120]] += 1
print(f"{cnt} samples, {per * 1e3:.2f} ms of CPU each, over {win} frames; {total_slam} in the map layer (host-only) = {total_slam * per * 1e6 / win:.0f} us per frame")
print("-- innermost map-layer frame (self + libc / libstdc++ callees):")
for f, c in leaf.most_common(30):
    print(f"  {100.0 * c / total_slam:5.1f} %  {c * per * 1e6 / win:7.1f} us/frame  {f}")
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
