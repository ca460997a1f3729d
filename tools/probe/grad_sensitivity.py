"""How sensitive are the YOLOv5l SSOD step's weight gradients to a 2^-9 relative perturbation, at the default (random) init and at
better-conditioned points?  ORACLE ONLY (fp32 CPU restatement of the reference step, oracle/step.py) -- no kernel of the product runs.

  python tools/probe/grad_sensitivity.py [Bl+Bu per side, default 1] [gamma ...]

For each BatchNorm-weight value `gamma` (1.0 = the default init): baseline gradients vs the gradients with every conv weight rounded
to bf16 (all arithmetic stays fp32), and vs the same step under torch's CPU bf16 autocast.  Prints the min / p10 / median cosine
over the conv-weight tensors.  This is the evidence behind the init tests/test_step_benchbatch.py uses for its bf16 gradient bound:
at the default init a deep train-mode-BatchNorm network is chaotic (a 2^-9 perturbation of the WEIGHTS alone decorrelates the
gradients), so a cosine against fp32 says nothing about the arithmetic there."""
import copy
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


def main():
    from efficientteacher_amd.configs import get_cfg
    from oracle import model as o_model, step as o_step
    from tests.test_step_fullsize import YAML, _inputs
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 1
    gammas = [float(a) for a in sys.argv[2:]] or [1.0, 0.5, 0.25]
    torch.set_num_threads(min(32, os.cpu_count() or 1))
    cfg = get_cfg()
    cfg.merge_from_file(YAML)
    cfg.merge_from_list(["Dataset.batch_size", 2 * B, "SSOD.fixed_accumulate", True])
    cfg.freeze()
    imgs, targets, u_str, u_ori, M_s, synth = _inputs(B, B, 640)
    for gamma in gammas:
        torch.manual_seed(0)
        student = o_model.Model.from_cfg(cfg)
        with torch.no_grad():
            for m in student.modules():
                if isinstance(m, torch.nn.BatchNorm2d):
                    m.weight.fill_(gamma)
        teacher = copy.deepcopy(student).eval()
        student.train()
        ref = o_step.ssod_step(student, teacher, imgs, targets, u_str, u_ori, M_s, cfg, synth_scores=synth)
        g0 = {n: p.grad.clone() for n, p in student.named_parameters() if p.grad is not None and p.dim() == 4 and float(p.grad.norm()) > 0}
        rows = {}
        # (a) weights rounded to bf16, arithmetic fp32
        s2 = copy.deepcopy(student)
        s2.zero_grad()
        with torch.no_grad():
            for p in s2.parameters():
                if p.dim() == 4:
                    p.copy_(p.to(torch.bfloat16).float())
        o_step.ssod_step(s2, teacher, imgs, targets, u_str, u_ori, M_s, cfg, teacher_pred=ref["teacher_pred"])
        rows["weights rounded to bf16"] = dict(s2.named_parameters())
        # (b) CPU autocast bf16
        s3 = copy.deepcopy(student)
        s3.zero_grad()
        with torch.autocast("cpu", dtype=torch.bfloat16):
            o_step.ssod_step(s3, teacher, imgs, targets, u_str, u_ori, M_s, cfg, teacher_pred=ref["teacher_pred"])
        rows["cpu autocast bf16"] = dict(s3.named_parameters())
        for what, params in rows.items():
            cos = sorted(torch.nn.functional.cosine_similarity(params[n].grad.float().flatten().double(), g.flatten().double(), 0).item()
                         for n, g in g0.items())
            print(f"gamma {gamma:5.2f}  {B}+{B} images  {what:26s}: conv-weight gradient cosine min {cos[0]:.4f}  p10 {cos[len(cos) // 10]:.4f}  "
                  f"median {cos[len(cos) // 2]:.4f}  ({len(cos)} tensors)", flush=True)


if __name__ == "__main__":
    main()
