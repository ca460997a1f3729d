"""TEST INFRASTRUCTURE -- recipe for oracle/_ref/: the imported reference, made able to travel to the GPU box.

BASELINE.json's north_star wants "the reference's CPU path timed on the host cores of the same box in the same run".  The
reference is Python and lives at /root/reference in the build container only.  This recipe (run by __graft_entry__.build()
whenever /root/reference is present) writes into the GIT-IGNORED directory oracle/_ref/

  * the BYTE-COMPILED form (sourceless ``pkg/mod.pyc``, python 3.10 -- the GPU box runs this same image) of exactly the
    reference modules that one SSODTrainer.train_instance + update_optimizer imports: the list is DISCOVERED by running one
    reduced-size step of the live reference (oracle/time_reference_step.py --tiny) and reading sys.modules -- no hand-kept
    file list, and no reference SOURCE text is copied anywhere (compiled outputs only, like a C reference's .so);
  * utils/Arial.ttf (utils/plots.py:66 looks for the font at class-definition time and would otherwise download it);
  * MANIFEST.json (module -> sha256 of the source it was compiled from, python / torch versions).

oracle/_ref/ ships with the push like libet_hip.so does (listed in .gitignore, not in .gpurunignore).  On the GPU box
``ET_REFERENCE=oracle/_ref python -m oracle.time_reference_step`` imports it through the same oracle/ref_loader.py shims;
bench.py's cpu_baseline leg does exactly that (`cpu_baseline.kind == "reference"`).  Nothing in the product path reads it.

    python -m oracle.make_ref            # (re)build oracle/_ref/ from /root/reference
    python -m oracle.make_ref --check    # import the shipped image in a fresh process and run one tiny step
"""
import hashlib
import json
import os
import py_compile
import shutil
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, "oracle", "_ref")
LIVE = "/root/reference"

_DISCOVER = r"""
import json, os, sys
sys.path.insert(0, {root!r})
os.environ["ET_REFERENCE"] = {live!r}
from oracle import ref_loader, time_reference_step as trs
import torch
torch.set_num_threads(4)
step, _ = trs.reference_step_fn(1, 64, True)
step(0); step(1)
ref = os.path.realpath({live!r}) + os.sep
mods = {{}}
for name, m in list(sys.modules.items()):
    f = getattr(m, "__file__", None)
    if f and os.path.realpath(f).startswith(ref) and f.endswith(".py"):
        mods[name] = os.path.realpath(f)
print("MODULES " + json.dumps(mods))
"""


def discover():
    """module name -> source path, for everything the (tiny) reference step imported from the live tree"""
    p = subprocess.run([sys.executable, "-c", _DISCOVER.format(root=ROOT, live=LIVE)], capture_output=True, text=True, timeout=1800)
    for line in p.stdout.splitlines():
        if line.startswith("MODULES "):
            return json.loads(line[len("MODULES "):])
    raise RuntimeError("reference module discovery failed:\n" + p.stdout[-2000:] + p.stderr[-4000:])


def build(verbose=True):
    if not os.path.isdir(os.path.join(LIVE, "models")):
        raise RuntimeError(f"{LIVE} not present: oracle/_ref can only be built in the build container")
    mods = discover()
    tmp = OUT + ".tmp"
    shutil.rmtree(tmp, ignore_errors=True)
    os.makedirs(tmp)
    live = os.path.realpath(LIVE)
    manifest = {}
    for name, src in sorted(mods.items()):
        rel = os.path.relpath(src, live)
        dst = os.path.join(tmp, rel[:-3] + ".pyc")                 # sourceless layout: pkg/mod.pyc beside pkg/__init__.pyc
        os.makedirs(os.path.dirname(dst), exist_ok=True)
        # dfile: the path recorded in tracebacks -- the live tree's, so that a failure on the GPU box still names reference file:line
        py_compile.compile(src, cfile=dst, dfile=os.path.join(LIVE, rel), doraise=True,
                           invalidation_mode=py_compile.PycInvalidationMode.UNCHECKED_HASH)
        manifest[name] = dict(file=rel, sha256=hashlib.sha256(open(src, "rb").read()).hexdigest())
    # every directory on the way must be a package the import system accepts: the reference relies on namespace packages in places
    font = os.path.join(live, "utils", "Arial.ttf")
    if os.path.exists(font):
        os.makedirs(os.path.join(tmp, "utils"), exist_ok=True)
        shutil.copy(font, os.path.join(tmp, "utils", "Arial.ttf"))
    import torch
    json.dump(dict(what="byte-compiled image of the reference modules one SSOD train_instance imports (oracle/make_ref.py)",
                   python=sys.version.split()[0], torch=torch.__version__, modules=manifest),
              open(os.path.join(tmp, "MANIFEST.json"), "w"), indent=1, sort_keys=True)
    shutil.rmtree(OUT, ignore_errors=True)
    os.rename(tmp, OUT)
    if verbose:
        print(f"[make_ref] {OUT}: {len(manifest)} modules")
    return OUT


def check():
    """a fresh process imports ONLY the shipped image (the live tree is not on its path) and runs one tiny step"""
    env = dict(os.environ, ET_REFERENCE=OUT)
    p = subprocess.run([sys.executable, "-m", "oracle.time_reference_step", "1", "--tiny", "--seconds", "1", "--cores", "4"],
                       cwd=ROOT, env=env, capture_output=True, text=True, timeout=1800)
    if p.returncode != 0:
        raise RuntimeError("oracle/_ref does not import:\n" + p.stdout[-2000:] + p.stderr[-4000:])
    r = json.loads(p.stdout.strip().splitlines()[-1])
    assert os.path.realpath(r["where"]) == os.path.realpath(OUT), r
    return r


def up_to_date():
    """True when oracle/_ref exists and every module in its manifest still hashes to the live source (or the live tree is absent)"""
    mf = os.path.join(OUT, "MANIFEST.json")
    if not os.path.exists(mf):
        return False
    if not os.path.isdir(os.path.join(LIVE, "models")):
        return True
    try:
        man = json.load(open(mf))
        if man.get("python") != sys.version.split()[0]:
            return False
        for m in man["modules"].values():
            if hashlib.sha256(open(os.path.join(LIVE, m["file"]), "rb").read()).hexdigest() != m["sha256"]:
                return False
            if not os.path.exists(os.path.join(OUT, m["file"][:-3] + ".pyc")):
                return False
        return True
    except (OSError, ValueError, KeyError):
        return False


if __name__ == "__main__":
    if "--check" in sys.argv:
        print(json.dumps(check()))
    else:
        build()
        print(json.dumps(check()))
