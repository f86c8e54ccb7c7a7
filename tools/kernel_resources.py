#!/usr/bin/env python3
"""Per-kernel register / scratch / LDS usage read from the gfx950 code objects embedded in csrc/libdd_hotpath.so
(the AMDGPU metadata note every kernel carries): no recompilation, no GPU.

    python tools/kernel_resources.py [--spills-only]
"""
import os
import struct
import sys
import zlib

import msgpack

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(ROOT, "driving-dirty_amd", "csrc", "libdd_hotpath.so")


def _sections(blob):
    """(name, offset, size) of the sections of a 64-bit little-endian ELF image."""
    shoff, = struct.unpack_from("<Q", blob, 0x28)
    shentsize, shnum, shstrndx = struct.unpack_from("<HHH", blob, 0x3A)
    heads = [struct.unpack_from("<IIQQQQIIQQ", blob, shoff + i * shentsize) for i in range(shnum)]
    strtab = heads[shstrndx]
    out = []
    for h in heads:
        name = blob[strtab[4] + h[0]:blob.index(b"\0", strtab[4] + h[0])].decode()
        out.append((name, h[1], h[4], h[5]))
    return out


def _code_objects(fatbin):
    """The device ELF images inside a .hip_fatbin section (clang offload bundles, one per translation unit; plain or
    'CCOB'-compressed)."""
    pos = 0
    while True:
        i = fatbin.find(b"\x7fELF", pos)
        if i < 0:
            return
        shoff, = struct.unpack_from("<Q", fatbin, i + 0x28)
        shentsize, shnum = struct.unpack_from("<HH", fatbin, i + 0x3A)
        size = shoff + shentsize * shnum
        machine, = struct.unpack_from("<H", fatbin, i + 0x12)
        if machine == 224:                       # EM_AMDGPU
            yield fatbin[i:i + size]
        pos = i + 4


def kernels(lib=LIB):
    blob = open(lib, "rb").read()
    fat = next((blob[off:off + size] for name, _t, off, size in _sections(blob) if name == ".hip_fatbin"), None)
    if fat is None:
        raise SystemExit(f"{lib}: no .hip_fatbin section")
    if b"CCOB" in fat[:4096]:
        raise SystemExit("compressed offload bundles are not handled: build with --no-offload-compress")
    out = []
    for co in _code_objects(fat):
        for name, typ, off, size in _sections(co):
            if typ != 7:                         # SHT_NOTE
                continue
            p = off
            while p < off + size:
                namesz, descsz, ntype = struct.unpack_from("<III", co, p)
                p += 12
                p_name = p
                p += (namesz + 3) & ~3
                desc = co[p:p + descsz]
                p += (descsz + 3) & ~3
                if ntype == 32 and co[p_name:p_name + 6] == b"AMDGPU":
                    meta = msgpack.unpackb(desc, raw=False, strict_map_key=False)
                    out += meta.get("amdhsa.kernels", [])
    return out


def main():
    ks = kernels()
    spills_only = "--spills-only" in sys.argv
    print(f"{len(ks)} kernels in {os.path.relpath(LIB, ROOT)}")
    print(f"{'vgpr':>5} {'agpr':>5} {'sgpr':>5} {'scratch':>8} {'vspill':>7} {'lds':>7}  name")
    for k in sorted(ks, key=lambda k: k[".name"]):
        if spills_only and not (k[".private_segment_fixed_size"] or k.get(".vgpr_spill_count", 0)):
            continue
        print(f"{k['.vgpr_count']:>5} {k.get('.agpr_count', 0):>5} {k['.sgpr_count']:>5} {k['.private_segment_fixed_size']:>8} "
              f"{k.get('.vgpr_spill_count', 0):>7} {k['.group_segment_fixed_size']:>7}  {k['.name']}")


if __name__ == "__main__":
    main()
