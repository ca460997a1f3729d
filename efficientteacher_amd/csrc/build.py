"""Build libet_hip.so (gfx950 code objects + C ABI) in-tree with hipcc.

``python -m efficientteacher_amd.csrc.build`` or ``__graft_entry__.build()``.
hipcc cross-compiles without a GPU; objects are cached by source mtime.
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
LIB = os.path.join(os.path.dirname(HERE), "libet_hip.so")
OBJ = os.path.join(HERE, "_obj")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
ARCH = "gfx950"

# exact-rounding files: integer/index decisions depend on fp32 results, so no FMA contraction
EXACT = {"nms.hip", "loss.hip", "pseudo_label.hip", "detect.hip", "optim.hip", "tal.hip", "augment.hip"}
COMMON = ["--offload-arch=" + ARCH, "-O3", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-function",
          "-I", os.path.join(ROOT, "include")]


def sources():
    return sorted(f for f in os.listdir(HERE) if f.endswith(".hip"))


def _deps_mtime():
    hs = [os.path.join(HERE, f) for f in os.listdir(HERE) if f.endswith(".h")]
    hs.append(os.path.join(ROOT, "include", "et_hip.h"))
    return max(os.path.getmtime(h) for h in hs)


def build(verbose=False, force=False):
    os.makedirs(OBJ, exist_ok=True)
    hdr = _deps_mtime()
    objs, rebuilt = [], False
    procs = []
    for src in sources():
        sp = os.path.join(HERE, src)
        op = os.path.join(OBJ, src[:-4] + ".o")
        objs.append(op)
        if force or not os.path.exists(op) or os.path.getmtime(op) < max(os.path.getmtime(sp), hdr):
            cmd = [HIPCC] + COMMON + (["-ffp-contract=off"] if src in EXACT else []) + ["-c", sp, "-o", op]
            if verbose:
                print(" ".join(cmd))
            procs.append((src, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)))
            rebuilt = True
    for src, p in procs:
        out, _ = p.communicate()
        if p.returncode != 0:
            sys.stderr.write(out.decode())
            raise RuntimeError(f"hipcc failed on {src}")
        if verbose and out:
            print(out.decode())
    # relink also when an object is newer than the library (an object compiled by hand, e.g. with -save-temps for an ISA check,
    # used to leave a STALE library behind: two GPU runs of r04 measured the previous build)
    if rebuilt or not os.path.exists(LIB) or any(os.path.getmtime(o) > os.path.getmtime(LIB) for o in objs):
        cmd = [HIPCC, "--offload-arch=" + ARCH, "-shared", "-fPIC", "-o", LIB] + objs
        if verbose:
            print(" ".join(cmd))
        subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    print(build(verbose=True, force="--force" in sys.argv))
