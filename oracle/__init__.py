"""oracle/ -- TEST INFRASTRUCTURE ONLY.

CPU restatements (numpy / plain-torch fp32) of the Efficient-Teacher SSOD hot
path, used as the checker by ``tests/``, ``__graft_entry__.smoke()`` and the
``cpu_baseline`` leg of ``bench.py``.  Nothing in ``efficientteacher_amd/`` (the
product) may import from here; the product path calls the HIP kernels behind
``include/et_hip.h`` and raises if ``libet_hip.so`` is missing.

Every function cites the reference file:line it restates.  The restatements are
pinned against the *live* reference (imported read-only through
``oracle/ref_loader.py`` in the build container, where ``/root/reference``
exists) by ``oracle/make_golden.py``; the resulting vectors are committed under
``tests/golden/`` and re-checked by ``tests/test_oracle_golden.py`` on every run.

Third-party arithmetic that is not in the reference tree:
``torchvision.ops.nms`` (requirements.txt:12 ``torchvision>=0.8.1``, call site
utils/general.py:976).  torchvision is not installed here and the reference has
no test that pins its output, so ``oracle/nms.py::nms`` restates the published
kernel semantics (stable descending sort, suppress iff IoU > thr, IoU =
inter/(a+b-inter) in fp32, no eps).  PARITY UNPINNED at that one boundary; all
other functions are pinned by the golden vectors.
"""
