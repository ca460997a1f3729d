"""How much do bf16 activations move the gradients of ONE YOLOv5l step at batch 1+1?  Calibration for
tests/test_step_fullsize.py: the plain-torch oracle under torch.autocast(bfloat16) (= the reference's own AMP
recipe, trainer.py:348, with bf16 instead of fp16) against the same oracle in fp32, next to the HIP bf16 path
against the fp32 oracle.  usage: python tools/probe/bf16_noise.py [width depth S]"""
import copy, json, os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench
from efficientteacher_amd.configs import get_cfg
from oracle import model as o_model, step as o_step

width, depth, S = (float(sys.argv[1]), float(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (1.0, 1.0, 640)
cfg = get_cfg(); cfg.merge_from_file(bench.YAML)
cfg.merge_from_list(["Model.width_multiple", width, "Model.depth_multiple", depth]); cfg.freeze()
torch.manual_seed(0)
torch.set_num_threads(min(32, os.cpu_count() or 1))
student = o_model.Model.from_cfg(cfg)
teacher = copy.deepcopy(student).eval()
rng = np.random.default_rng(0)
imgs, targets, u_str, u_ori, M_s = bench.make_batch(rng, 1, 1, S, "cpu")
A = 3 * ((S // 8) ** 2 + (S // 16) ** 2 + (S // 32) ** 2)
g = torch.Generator().manual_seed(99)
synth = torch.rand(1, A, 81, generator=g) ** torch.cat((torch.full((1,), 16.0), torch.full((80,), 4.0)))
with torch.no_grad():
    (tp, _), _ = teacher(u_ori)
    tp[..., 4:] = synth


def run(autocast):
    st = copy.deepcopy(student).train()
    with torch.autocast("cpu", dtype=torch.bfloat16, enabled=autocast):
        r = o_step.ssod_step(st, teacher, imgs, targets, u_str, u_ori, M_s, cfg, teacher_pred=tp.clone())
    return r, dict(st.named_parameters())


r32, p32 = run(False)
r16, p16 = run(True)
names = ("backbone.stage1.conv.weight", "backbone.stage3_2.m.4.cv2.conv.weight", "neck.C3.m.0.cv2.conv.weight",
         "head.m.1.weight", "backbone.stage2_2.cv1.bn.weight")
out = {}
for n in names:
    if n not in p32:
        continue
    a, b = p16[n].grad.float().flatten(), p32[n].grad.flatten()
    out[n] = dict(maxrel=((a - b).abs().max() / b.abs().max()).item(), l2rel=((a - b).norm() / b.norm()).item(),
                  cos=torch.nn.functional.cosine_similarity(a, b, 0).item())
print("NOISE", json.dumps(dict(cfg=[width, depth, S], sup32=r32["sup_items"], sup16=r16["sup_items"], un32=r32["un_items"],
                               un16=r16["un_items"], grads=out)))
