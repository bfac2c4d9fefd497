"""What the built library's gfx950 code object says about its kernels, without a compiler or a GPU:

    python tools/code_object.py [matchering_amd/libmgx.so] [substring ...]

prints, per kernel, the bytes of scratch per lane (private_segment_fixed_size of its kernel descriptor: what the
compiler spilled), the LDS it declares statically and the size of its code.  `kernels(path)` returns the same as a
dict {demangled-ish symbol: {"scratch": .., "lds": .., "code": ..}} (tests/test_abi.py pins the hot kernels' scratch).

Layout walked here: ELF .hip_fatbin section -> "__CLANG_OFFLOAD_BUNDLE__" table -> the gfx950 entry (an ELF again) ->
its symbol table: STT_FUNC symbols are the kernels' code, "<name>.kd" objects their 64-byte descriptors (AMDHSA code
object v3+: group_segment_fixed_size at byte 0, private_segment_fixed_size at byte 4).
"""
import os
import struct
import sys


def _sections(blob, base):
    (shoff,) = struct.unpack_from("<Q", blob, base + 0x28)
    shentsize, shnum, shstrndx = struct.unpack_from("<HHH", blob, base + 0x3A)
    secs = []
    for i in range(shnum):
        name, typ, _flags, addr, off, size, link, _info, _align, entsize = struct.unpack_from(
            "<IIQQQQIIQQ", blob, base + shoff + i * shentsize)
        secs.append(dict(name=name, type=typ, addr=addr, off=off, size=size, link=link, entsize=entsize))
    strtab = secs[shstrndx]
    for s in secs:
        end = blob.index(b"\0", base + strtab["off"] + s["name"])
        s["name"] = blob[base + strtab["off"] + s["name"]:end].decode()
    return secs


def device_elf(path, arch="gfx950"):
    blob = open(path, "rb").read()
    fat = next(s for s in _sections(blob, 0) if s["name"] == ".hip_fatbin")
    at = fat["off"]
    assert blob[at:at + 24] == b"__CLANG_OFFLOAD_BUNDLE__", "not an offload bundle"
    (entries,) = struct.unpack_from("<Q", blob, at + 24)
    pos = at + 32
    for _ in range(entries):
        off, size, tsize = struct.unpack_from("<QQQ", blob, pos)
        triple = blob[pos + 24:pos + 24 + tsize].decode()
        if arch in triple:
            return blob, at + off
        pos += 24 + tsize
    raise LookupError(f"no {arch} code object in {path}")


def kernels(path=None):
    path = path or os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "matchering_amd", "libmgx.so")
    blob, base = device_elf(path)
    secs = _sections(blob, base)
    symtab = next(s for s in secs if s["type"] == 2)
    strs = secs[symtab["link"]]
    funcs, descriptors = {}, {}
    for i in range(symtab["size"] // 24):
        name, info, _other, shndx, value, size = struct.unpack_from("<IBBHQQ", blob, base + symtab["off"] + i * 24)
        end = blob.index(b"\0", base + strs["off"] + name)
        sym = blob[base + strs["off"] + name:end].decode()
        if info & 0xF == 2 and size:                       # STT_FUNC
            funcs[sym] = size
        elif sym.endswith(".kd") and shndx < len(secs):
            sec = secs[shndx]
            at = base + sec["off"] + (value - sec["addr"])
            lds, scratch = struct.unpack_from("<II", blob, at)
            descriptors[sym[:-3]] = (lds, scratch)
    return {k: {"scratch": descriptors[k][1], "lds": descriptors[k][0], "code": funcs.get(k, 0)} for k in descriptors}


if __name__ == "__main__":
    args = sys.argv[1:]
    lib = args.pop(0) if args and args[0].endswith(".so") else None
    for name, k in sorted(kernels(lib).items()):
        if not args or any(a in name for a in args):
            print(f"{name[:90]:90s} scratch {k['scratch']:5d} B/lane  static LDS {k['lds']:6d}  code {k['code']:7d} B")
