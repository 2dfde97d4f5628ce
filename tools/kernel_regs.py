#!/usr/bin/env python3
"""Register / spill / scratch table of every gfx950 kernel in a built artefact.

    tools/kernel_regs.py [qups_amd/libqdas.so | file.o | file.hsaco | dir of .hsaco]   (default: qups_amd/libqdas.so)

Reads the AMDGPU code-object metadata (llvm-readelf --notes) of every device code object embedded in the file: hipcc stores
them as clang offload bundles ("__CLANG_OFFLOAD_BUNDLE__") in the .hip_fatbin section, one bundle per translation unit.
tests/test_build_regs.py asserts on the same table: no tiled-kernel instantiation may spill or use scratch memory.
"""
from __future__ import annotations

import os
import re
import struct
import subprocess
import sys
import tempfile

READELF = "/opt/rocm/lib/llvm/bin/llvm-readelf"
CXXFILT = "c++filt"      # binutils (gcc is part of the image); names stay mangled without it
MAGIC = b"__CLANG_OFFLOAD_BUNDLE__"


def code_objects(path: str, arch: str = "gfx950"):
    """yield the bytes of every device code object for `arch` found in `path` (a bundle container or a bare code object)"""
    blob = open(path, "rb").read()
    pos, found = 0, False
    while True:
        at = blob.find(MAGIC, pos)
        if at < 0:
            break
        found = True
        n, = struct.unpack_from("<Q", blob, at + len(MAGIC))
        q = at + len(MAGIC) + 8
        end = at + len(MAGIC)
        for _ in range(n):
            off, size, idlen = struct.unpack_from("<QQQ", blob, q)
            ident = blob[q + 24:q + 24 + idlen].decode(errors="replace")
            q += 24 + idlen
            if arch in ident and size:
                yield blob[at + off:at + off + size]
            end = max(end, at + off + size)
        pos = end
    if not found and blob[:4] == b"\x7fELF":
        yield blob


def kernel_table(path: str):
    """list of dicts {name, vgpr, agpr, sgpr, vgpr_spill, sgpr_spill, scratch, lds} for every kernel in `path`"""
    rows = []
    for co in code_objects(path):
        with tempfile.NamedTemporaryFile(suffix=".co") as f:
            f.write(co)
            f.flush()
            txt = subprocess.run([READELF, "--notes", f.name], capture_output=True, text=True).stdout
        for blk in txt.split("- .agpr_count:")[1:]:
            g = lambda k: int((re.search(r"\." + k + r":\s+(\d+)", blk) or [None, "-1"])[1])
            name = re.search(r"\.name:\s+(\S+)", blk)
            if not name:
                continue
            rows.append({"name": name.group(1), "agpr": int(re.match(r"\s*(\d+)", blk).group(1)), "vgpr": g("vgpr_count"), "sgpr": g("sgpr_count"),
                         "vgpr_spill": g("vgpr_spill_count"), "sgpr_spill": g("sgpr_spill_count"),
                         "scratch": g("private_segment_fixed_size"), "lds": g("group_segment_fixed_size")})
    return rows


def demangle(names):
    try:
        out = subprocess.run([CXXFILT], input="\n".join(names), capture_output=True, text=True).stdout.splitlines()
        return out if len(out) == len(names) else names
    except OSError:
        return names


def short(dn: str) -> str:
    """das_tile_kernel<3, HIP_vector_type<float, 2u>, false, ...> -> compact label with the template parameters named"""
    m = re.match(r"void qdas::das_tile_kernel<(.*)>\(qdas::TileParams\)", dn)
    if not m:
        return re.sub(r"^void ", "", dn)[:110]
    if "MIRQ" in dn:
        pass
    a = [x.strip() for x in re.sub(r"HIP_vector_type<double, 2u>", "f64", re.sub(r"HIP_vector_type<float, 2u>", "f32", m.group(1))).replace("unsigned int", "f16").split(",")]
    keys = ["interp", "data", "fmod", "wtab", "sym", "fb2", "fb4", "waves", "mb", "W", "nbuf", "psz", "bpc", "probe", "big", "lut", "bf", "mirq", "fold"]
    d = dict(zip(keys, a))
    flags = [k for k in ("fmod", "wtab", "sym", "fold", "mirq", "fb2", "fb4", "probe", "big", "lut", "bf") if d.get(k) == "true"]
    return f"das_tile interp={d['interp']} {d['data']} mb={d['mb']} W={d['W']} " + (" ".join(flags) if flags else "general")


def main():
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    target = sys.argv[1] if len(sys.argv) > 1 else os.path.join(root, "qups_amd", "libqdas.so")
    isdir = os.path.isdir(target)
    paths = [os.path.join(target, f) for f in sorted(os.listdir(target)) if not f.endswith(".lock")] if isdir else [target]
    rows = []
    for p in paths:
        for r in kernel_table(p):
            if isdir:                                   # a hiprtc cache: the file name is the key that qdas_plan_kernel_name() prints as "[jit <key>]"
                r["name"] = f"{r['name']} [jit {os.path.basename(p).split('.')[0]}]"      # (the symbol names the template arguments since round 5)
            rows.append(r)
    names = demangle([r["name"] for r in rows])
    print(f"# {target}: {len(rows)} kernels  (vgpr / agpr / sgpr, spilled vgpr / sgpr, scratch bytes per lane, static LDS bytes)")
    bad = 0
    for r, dn in sorted(zip(rows, names), key=lambda t: short(t[1])):
        flag = "" if (r["vgpr_spill"] == 0 and r["scratch"] == 0) else "   <-- SPILLS"
        bad += bool(flag)
        print(f"{short(dn):72s} vgpr {r['vgpr']:3d} agpr {r['agpr']:3d} sgpr {r['sgpr']:3d}  spill v {r['vgpr_spill']:3d} s {r['sgpr_spill']:3d}  "
              f"scratch {r['scratch']:4d}  lds {r['lds']:6d}{flag}")
    print(f"# kernels with spilled VGPRs or scratch: {bad}")


if __name__ == "__main__":
    main()
