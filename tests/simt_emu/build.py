"""Build libet_emu.so: the SAME kernel sources as libet_hip.so, compiled for the host against the
SIMT emulator headers.  TEST INFRASTRUCTURE ONLY -- see include/hip/hip_runtime.h."""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
CSRC = os.path.join(ROOT, "efficientteacher_amd", "csrc")
OUT = os.path.join(HERE, "_build")
LIB = os.path.join(OUT, "libet_emu.so")
CXX = os.environ.get("ET_EMU_CXX", "/opt/rocm/lib/llvm/bin/clang++")
EXACT = {"nms.hip", "loss.hip", "pseudo_label.hip", "detect.hip", "optim.hip", "tal.hip", "augment.hip"}
FLAGS = ["-x", "c++", "-std=c++17", "-O2", "-fPIC", "-g0", "-Wno-unused-function", "-Wno-unknown-attributes",
         "-Wno-unused-value", "-fno-strict-aliasing",
         "-I", os.path.join(HERE, "include"), "-I", os.path.join(ROOT, "include")]


def build(verbose=False, force=False):
    os.makedirs(OUT, exist_ok=True)
    srcs = sorted(f for f in os.listdir(CSRC) if f.endswith(".hip"))
    hdrs = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")]
    hdrs += [os.path.join(HERE, "include", "hip", "hip_runtime.h"), os.path.join(ROOT, "include", "et_hip.h")]
    hm = max(os.path.getmtime(h) for h in hdrs)
    objs, procs, rebuilt = [], [], False
    jobs = [(os.path.join(CSRC, s), os.path.join(OUT, s[:-4] + ".o"), s in EXACT) for s in srcs]
    jobs.append((os.path.join(HERE, "emu_runtime.cpp"), os.path.join(OUT, "emu_runtime.o"), False))
    for sp, op, exact in jobs:
        objs.append(op)
        if force or not os.path.exists(op) or os.path.getmtime(op) < max(os.path.getmtime(sp), hm):
            cmd = [CXX] + FLAGS + (["-ffp-contract=off"] if exact else []) + ["-c", sp, "-o", op]
            if verbose:
                print(" ".join(cmd))
            procs.append((sp, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)))
            rebuilt = True
    for sp, p in procs:
        out, _ = p.communicate()
        if p.returncode != 0:
            sys.stderr.write(out.decode())
            raise RuntimeError(f"emulator build failed on {sp}")
    if rebuilt or not os.path.exists(LIB):
        subprocess.check_call([CXX, "-shared", "-fPIC", "-o", LIB] + objs)
    return LIB


if __name__ == "__main__":
    print(build(verbose=True, force="--force" in sys.argv))
