"""Inputs of the localisation harness as the reference's test/loc.cpp reads them -- a params.json (JSON with comments,
config/params.hpp:30) and PCD files (MapManager.cpp:68) -- written and read from Python, so that the tests can hand the SAME files to
the C++ harness (simpleslam_amd/host/loc_harness.cpp), to the Python mirror and to the CPU oracle."""
import json
import re
import struct

import numpy as np


def write_params(path, pcd_file, pcr="loam", cores=1, grid=0.5):
    """A params.json shaped like the reference's (config/params.json): comments included, the four keys the path reads."""
    text = f"""{{
    // mode has lio or lo
    "mode": "lio",

    "cores": {cores},

    /* use ndt maybe no need to downsample */
    "downSampleVoxelGridSize": {grid},

    "pcd_file": "{pcd_file}",

    "tf":{{
        "lidar_height": 2.0
    }},

    "vis": {{
        "enable" : false   // no ROS here
    }},

    "frontend" : {{
        "pcr" : "{pcr}",     // loam, ndt or vgicp
        "local_size": 100,
        "global_size": 10
    }}
}}
"""
    with open(path, "w") as f:
        f.write(text)


def read_params(path):
    """JSON with // and /* */ comments (outside strings) -> dict"""
    text = open(path).read()
    out, i, n = [], 0, len(text)
    while i < n:
        c = text[i]
        if c == '"':
            j = i + 1
            while j < n and text[j] != '"':
                j += 2 if text[j] == "\\" else 1
            out.append(text[i:j + 1]); i = j + 1
        elif text.startswith("//", i):
            while i < n and text[i] != "\n":
                i += 1
        elif text.startswith("/*", i):
            i = text.index("*/", i + 2) + 2
        else:
            out.append(c); i += 1
    return json.loads("".join(out))


def _header(fields, sizes, types, counts, n, kind):
    return ("# .PCD v0.7 - Point Cloud Data file format\nVERSION 0.7\n"
            f"FIELDS {' '.join(fields)}\nSIZE {' '.join(map(str, sizes))}\nTYPE {' '.join(types)}\nCOUNT {' '.join(map(str, counts))}\n"
            f"WIDTH {n}\nHEIGHT 1\nVIEWPOINT 0 0 0 1 0 0 0\nPOINTS {n}\nDATA {kind}\n").encode()


def write_pcd(path, pts, kind="binary"):
    """pts: (n, >=4) float32 x y z intensity.  kind: ascii | binary | binary_pcl (the padded layout PCL itself writes for
    pcl::PointXYZI: x y z _ intensity _, 32-byte records) | binary_compressed (LZF, fields one after the other)."""
    pts = np.ascontiguousarray(pts, np.float32)
    n = pts.shape[0]
    xyzi = np.ascontiguousarray(pts[:, :4])
    with open(path, "wb") as f:
        if kind == "ascii":
            f.write(_header(["x", "y", "z", "intensity"], [4] * 4, ["F"] * 4, [1] * 4, n, "ascii"))
            for p in xyzi:
                f.write(("%.9g %.9g %.9g %.9g\n" % tuple(p)).encode())
        elif kind == "binary":
            f.write(_header(["x", "y", "z", "intensity"], [4] * 4, ["F"] * 4, [1] * 4, n, "binary"))
            f.write(xyzi.tobytes())
        elif kind == "binary_pcl":
            f.write(_header(["x", "y", "z", "_", "intensity", "_"], [4, 4, 4, 1, 4, 1], ["F", "F", "F", "U", "F", "U"], [1, 1, 1, 4, 1, 12], n, "binary"))
            rec = np.zeros((n, 8), np.float32)
            rec[:, :3] = xyzi[:, :3]; rec[:, 3] = 1.0; rec[:, 4] = xyzi[:, 3]
            f.write(rec.tobytes())
        elif kind == "binary_compressed":
            f.write(_header(["x", "y", "z", "intensity"], [4] * 4, ["F"] * 4, [1] * 4, n, "binary_compressed"))
            raw = np.ascontiguousarray(xyzi.T).tobytes()          # all x, all y, all z, all intensity
            comp = lzf_compress(raw)
            f.write(struct.pack("<II", len(comp), len(raw)))
            f.write(comp)
        else:
            raise ValueError(kind)


def lzf_compress(data):
    """A valid LZF stream (the format PCL's lzfDecompress reads): greedy back references found through a 3-byte hash, literal runs
    of at most 32 bytes otherwise.  Not tuned -- it only has to exercise both token kinds of the reader."""
    out = bytearray()
    lit = bytearray()
    table = {}
    i, n = 0, len(data)

    def flush():
        nonlocal lit
        while lit:
            chunk = lit[:32]
            out.append(len(chunk) - 1); out.extend(chunk)
            lit = lit[32:]

    while i < n:
        key = data[i:i + 3]
        ref = table.get(key) if len(key) == 3 else None
        if len(key) == 3:
            table[key] = i
        if ref is not None and 0 < i - ref <= 8191:
            ln = 3
            while i + ln < n and ln < 264 and data[ref + ln] == data[i + ln]:
                ln += 1
            flush()
            off = i - ref - 1
            l2 = ln - 2
            if l2 < 7:
                out.append((l2 << 5) | (off >> 8))
            else:
                out.append((7 << 5) | (off >> 8)); out.append(l2 - 7)
            out.append(off & 0xff)
            i += ln
        else:
            lit.append(data[i]); i += 1
    flush()
    return bytes(out)


def read_pcd(path):
    """x y z intensity of a PCD written by write_pcd (ascii / binary / binary_pcl) -> (n, 4) float32"""
    raw = open(path, "rb").read()
    head_end = raw.index(b"\nDATA ")
    line_end = raw.index(b"\n", head_end + 1)
    head = raw[:line_end].decode().splitlines()
    kv = {ln.split()[0]: ln.split()[1:] for ln in head if ln and not ln.startswith("#")}
    fields, sizes, counts = kv["FIELDS"], list(map(int, kv["SIZE"])), list(map(int, kv["COUNT"]))
    n = int(kv["POINTS"][0])
    body = raw[line_end + 1:]
    if kv["DATA"][0] == "ascii":
        a = np.array([[float(v) for v in ln.split()] for ln in body.decode().splitlines()[:n]], np.float32).reshape(n, -1)
        return np.ascontiguousarray(a[:, [fields.index(k) for k in ("x", "y", "z", "intensity")]])
    assert kv["DATA"][0] == "binary"
    stride = sum(s * c for s, c in zip(sizes, counts))
    rec = np.frombuffer(body[:n * stride], np.uint8).reshape(n, stride)
    out = np.zeros((n, 4), np.float32)
    off = np.cumsum([0] + [s * c for s, c in zip(sizes, counts)])
    for j, k in enumerate(("x", "y", "z", "intensity")):
        o = off[fields.index(k)]
        out[:, j] = np.ascontiguousarray(rec[:, o:o + 4]).view(np.float32)[:, 0]
    return out
