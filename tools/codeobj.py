"""gfx950 code objects out of a HIP fat binary (.so / .o): the clang offload bundles of its .hip_fatbin section, uncompressed
(`__CLANG_OFFLOAD_BUNDLE__`, entries of (offset, size, triple)), and their disassembly per kernel (llvm-objdump -d).
usage: python tools/codeobj.py <lib.so> [kernel-name-substring]   -> per kernel: instructions, s_barrier count"""
import os
import re
import struct
import subprocess
import sys
import tempfile

MAGIC = b"__CLANG_OFFLOAD_BUNDLE__"
OBJDUMP = "/opt/rocm/lib/llvm/bin/llvm-objdump"


def code_objects(path, arch="gfx950"):
    blob = open(path, "rb").read()
    out = []
    for m in re.finditer(re.escape(MAGIC), blob):
        p = m.start()
        n, = struct.unpack_from("<Q", blob, p + 24)
        o = p + 32
        for _ in range(n):
            off, size, tl = struct.unpack_from("<QQQ", blob, o)
            o += 24
            triple = blob[o:o + tl].decode()
            o += tl
            if triple.endswith(arch) and size:
                out.append(blob[p + off:p + off + size])
    return out


def kernels(path, arch="gfx950", match=None):
    """{mangled kernel symbol: [instruction text, ...]} over every code object of the file"""
    res = {}
    for elf in code_objects(path, arch):
        if match is not None and match.encode() not in elf:
            continue
        with tempfile.NamedTemporaryFile(suffix=".co") as f:
            f.write(elf)
            f.flush()
            txt = subprocess.run([OBJDUMP, "-d", "--no-show-raw-insn", f.name], capture_output=True, text=True, check=True).stdout
        cur = None
        for line in txt.splitlines():
            m = re.match(r"^[0-9a-f]+ <([^>]+)>:$", line)
            if m:
                cur = res.setdefault(m.group(1), [])
                continue
            if cur is None or not line.startswith("\t"):
                continue
            ins = line.split("//")[0].strip()
            if ins:
                cur.append(ins)
    return res


if __name__ == "__main__":
    ks = kernels(sys.argv[1], match=sys.argv[2] if len(sys.argv) > 2 else None)
    for name, ins in ks.items():
        if len(sys.argv) > 2 and sys.argv[2] not in name:
            continue
        print(len(ins), sum(i.startswith("s_barrier") for i in ins), name[:150])
