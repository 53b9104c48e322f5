"""Compare two stage-call logs (alvaar_amd/csrc/slam/stage_trace.hpp) call by call: prints, per call, the largest difference of every
array, and stops at the first call whose discrete outputs differ or whose call sequence diverges.
usage: python tools/stage_trace_diff.py a.trace b.trace [max_calls]"""
import struct
import sys

import numpy as np

DT = {"f": np.float32, "d": np.float64, "i": np.int32, "b": np.uint8}


def read(path):
    calls = []
    with open(path, "rb") as f:
        data = f.read()
    o = 0
    while o + 20 <= len(data):
        name = data[o:o + 16].split(b"\0")[0].decode()
        cnt = struct.unpack_from("<i", data, o + 16)[0]
        o += 20
        arrs = {}
        for _ in range(cnt):
            tag = data[o:o + 8].split(b"\0")[0].decode()
            dt = chr(data[o + 8])
            nb = struct.unpack_from("<q", data, o + 9)[0]
            o += 17
            arrs[tag] = np.frombuffer(data[o:o + nb], DT[dt]).copy()
            o += nb
        calls.append((name, arrs))
    return calls


def main():
    a, b = read(sys.argv[1]), read(sys.argv[2])
    limit = int(sys.argv[3]) if len(sys.argv) > 3 else 10 ** 9
    frame = -1
    for i, ((na, xa), (nb_, xb)) in enumerate(zip(a, b)):
        if na == "new_frame":
            frame += 1
        if na != nb_:
            print(f"call {i} (frame {frame}): sequence diverges: {na} vs {nb_}")
            return
        parts, bad = [], False
        for tag in xa:
            u, v = xa[tag], xb.get(tag)
            if v is None or u.shape != v.shape:
                parts.append(f"{tag}: shape {u.shape} vs {None if v is None else v.shape}")
                bad = True
                continue
            if u.dtype in (np.int32, np.uint8):
                nd = int((u != v).sum())
                if nd:
                    parts.append(f"{tag}: {nd} differ")
                    bad = True
            elif u.size:
                d = float(np.abs(u.astype(np.float64) - v.astype(np.float64)).max())
                if d > 0:
                    parts.append(f"{tag}: {d:.2e}")
        if parts and (i < limit):
            print(f"call {i:5d} frame {frame:3d} {na:13s} " + "  ".join(parts))
        if bad:
            print("  ^ first discrete difference")
            return
    print(f"{min(len(a), len(b))} calls compared ({len(a)} vs {len(b)})")


if __name__ == "__main__":
    main()
