"""oracle/_ref (oracle/make_ref.py): the byte-compiled image of the reference modules one SSOD step imports -- what lets
bench.py time the IMPORTED reference on the GPU box's host cores (cpu_baseline.kind == "reference").  Test infrastructure only."""
import json
import os
import subprocess
import sys

import pytest

from oracle import make_ref

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
have_live = os.path.isdir(os.path.join(make_ref.LIVE, "models"))
have_image = os.path.exists(os.path.join(make_ref.OUT, "MANIFEST.json"))


@pytest.mark.skipif(not (have_live or have_image), reason="neither the reference tree nor a built oracle/_ref is present")
def test_reference_image_imports_without_the_live_tree_and_runs_a_step():
    if have_live and not make_ref.up_to_date():
        make_ref.build(verbose=False)
    man = json.load(open(os.path.join(make_ref.OUT, "MANIFEST.json")))
    mods = man["modules"]
    # the step's own modules are in the image ...
    for need in ("trainer.ssod_trainer", "models.detector.yolo_ssod", "models.loss.ssod.ssod_loss", "utils.general",
                 "utils.self_supervised_utils", "utils.torch_utils", "configs.defaults"):
        assert need in mods, need
    # ... as compiled outputs only: no reference SOURCE text travels
    for dirpath, _, files in os.walk(make_ref.OUT):
        for f in files:
            assert f.endswith((".pyc", ".ttf", ".json")), os.path.join(dirpath, f)
    r = make_ref.check()                       # fresh process, ET_REFERENCE = the image; asserts no module came from elsewhere
    assert r["steps"] >= 2 and r["images_per_s"] > 0


def test_reference_image_is_not_tracked_and_not_in_the_product():
    """git-ignored (history stays source-only), and nothing under efficientteacher_amd/ refers to it"""
    assert "oracle/_ref/" in open(os.path.join(ROOT, ".gitignore")).read().split()
    tracked = subprocess.run(["git", "ls-files", "oracle/_ref"], cwd=ROOT, capture_output=True, text=True).stdout.strip()
    assert tracked == ""
    for dirpath, _, files in os.walk(os.path.join(ROOT, "efficientteacher_amd")):
        for f in files:
            if f.endswith(".py"):
                assert "oracle/_ref" not in open(os.path.join(dirpath, f)).read() and "make_ref" not in open(os.path.join(dirpath, f)).read()
