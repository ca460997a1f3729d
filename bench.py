"""bench.py -- SSOD images/sec of one Efficient-Teacher training step on MI355X.

    python bench.py --gpus N --steps K --warmup W
    (N > 1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...)

Metric (BASELINE.json): SSOD images/sec (teacher + student step), YOLOv5l, 640 px.
One step = SSODTrainer.train_instance on one batch of synthetic input already resident in HBM:
EMA-teacher inference on the unlabeled weak view -> NMS + pseudo-label transform -> student forward on
cat(labeled, unlabeled) -> ComputeLoss + ComputeStudentMatchLoss -> backward (+ RCCL gradient
all-reduce for N > 1) -> SGD step + ModelEMA + semi-EMA update (optimizer every step:
SSOD.fixed_accumulate).  Per-GPU work is fixed: 32 labeled + 32 unlabeled images per rank (config 3 of
BASELINE.json at N = 1; weak scaling).  value = N * 64 * K / max-over-ranks time.

Extra objects on the JSON line:
  roofline     : the dominant kernel (bf16 MFMA implicit-GEMM conv), algorithmic FLOPs per launch
                 divided by its HIP-event-measured average launch duration inside the timed region.
  cpu_baseline : the plain-torch CPU port of the same step (oracle/model.py + oracle losses/NMS) timed
                 on the host cores, rank 0 at N = 1 only, on a bounded sample.
"""
import argparse
import json
import math
import os
import sys
import time

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

YAML = os.path.join(ROOT, "efficientteacher_amd", "configs", "ssod", "coco-standard",
                    "yolov5l_coco_ssod_10_percent.yaml")
F_IMG = 111.52e9          # conv FLOPs / image forward, YOLOv5l SSOD model
PEAK_BF16 = 2.5e15        # dense bf16 MFMA peak, MI355X_MICROARCH.md


def synth_targets(rng, B):
    rows = []
    for b in range(B):
        n = int(rng.integers(1, 17))
        xy = rng.uniform(0.1, 0.9, (n, 2))
        wh = np.exp(rng.uniform(np.log(0.02), np.log(0.6), (n, 2)))
        wh = np.minimum(wh, 2 * np.minimum(xy, 1 - xy))
        rows.append(np.concatenate((np.full((n, 1), b), rng.integers(0, 80, (n, 1)), xy, wh), 1))
    return torch.from_numpy(np.concatenate(rows, 0).astype(np.float32))


def make_batch(rng, Bl, Bu, S, device):
    g = torch.Generator().manual_seed(int(rng.integers(1 << 30)))
    imgs = torch.randint(0, 256, (Bl, 3, S, S), generator=g, dtype=torch.uint8)
    u_ori = torch.randint(0, 256, (Bu, 3, S, S), generator=g, dtype=torch.uint8)
    M_s = torch.zeros(Bu, 13, dtype=torch.float64)
    for i in range(Bu):          # fixed affine: scale 0.8, translate 0.1*S, lr flip on odd images
        s = 0.8
        M_s[i] = torch.tensor([i, s, 0, 0.1 * S, 0, s, 0.1 * S, 0, 0, 1, s, 0, i % 2], dtype=torch.float64)
    f = lambda t: (t.to(device).float() / 255.0)
    return f(imgs), synth_targets(rng, Bl).to(device), f(u_ori), f(u_ori), M_s.to(device)


def build_trainer(device, rank, world, local_rank, per_rank):
    from efficientteacher_amd.configs import get_cfg
    from efficientteacher_amd.trainer import SSODTrainer
    cfg = get_cfg()
    cfg.merge_from_file(YAML)
    cfg.merge_from_list(["Dataset.batch_size", per_rank * world, "SSOD.fixed_accumulate", True])
    cfg.freeze()
    torch.manual_seed(0)
    return cfg, SSODTrainer(cfg, device, None, local_rank if world > 1 else -1, rank if world > 1 else -1, world, nb=1000)


def cpu_baseline(cfg, device, seconds=25.0):
    """The oracle (oracle/step.py: plain-torch CPU restatement of trainer/ssod_trainer.py:587-680, `kind: "port"`) timed on
    the host cores for a bounded number of 1 + 1 image steps -- and, on the way, the PARITY CHECK of the benchmarked
    configuration: the very first oracle step and one bf16 step of the HIP trainer start from the same weights and see
    the same two images / targets / M_s / injected teacher scores, and their loss terms, pseudo-label sets and NMS
    decisions are compared (what tests/test_step_fullsize.py asserts, reported here next to the throughput)."""
    import copy
    from efficientteacher_amd.trainer import SSODTrainer
    from efficientteacher_amd.utils.general import nms_ssod_padded
    from oracle import model as o_model, nms as o_nms, step as o_step
    # 32 threads: beyond that the many small layers of a 2-image batch only add oversubscription
    # (measured on the 256-core MI355X host: 256 threads are ~100x slower than 8)
    cores = min(os.cpu_count() or 1, 32)
    torch.set_num_threads(cores)
    rng = np.random.default_rng(0)
    S, Bl, Bu = cfg.Dataset.img_size, 1, 1
    c2 = cfg.clone(); c2.defrost(); c2.merge_from_list(["Dataset.batch_size", Bl + Bu]); c2.freeze()
    torch.manual_seed(0)
    tr = SSODTrainer(c2, device, None, -1, -1, 1, nb=1000)
    student = o_model.Model.from_cfg(cfg)
    student.load_state_dict({k: v.detach().cpu() for k, v in tr.model.state_dict().items()}, strict=True)
    student.train()
    teacher = copy.deepcopy(student).eval()
    imgs, targets, u_str, u_ori, M_s = make_batch(rng, Bl, Bu, S, "cpu")
    synth = torch.rand(Bu, 25200, 81) ** torch.cat((torch.full((1,), 16.0), torch.full((80,), 4.0)))
    opt = torch.optim.SGD(student.parameters(), lr=0.01, momentum=0.937, nesterov=True)

    def step():
        opt.zero_grad()
        r = o_step.ssod_step(student, teacher, imgs, targets, u_str, u_ori, M_s, cfg, synth_scores=synth)
        opt.step()
        with torch.no_grad():
            for v, m in zip(teacher.state_dict().values(), student.state_dict().values()):
                if v.dtype.is_floating_point:
                    v.mul_(0.9999).add_(m, alpha=1e-4)
        return r

    t0 = time.time()
    ref = step()                             # warm-up (allocator, oneDNN primitive caches) == the parity reference
    warm = time.time() - t0
    # ---- parity of the benchmarked configuration (bf16, YOLOv5l, 640 px) ------------------------------------
    cap = {}

    def hook(tp):
        tp[..., 4:] = synth.to(device)
        cap["tp"] = tp.detach().clone()
        return tp
    tr.teacher_pred_hook = hook
    items = tr.train_instance(imgs.to(device), targets.to(device), None, u_str.to(device), u_ori.to(device), None,
                              M_s.to(device), 2000)
    want = {**{k: ref["sup_items"][k] for k in ("box", "obj", "cls")}, **ref["un_items"]}
    loss_rel = {k: abs(float(items[k]) - v) / max(abs(v), 1e-12) for k, v in want.items()}
    dets, counts, keep, _ = nms_ssod_padded(cap["tp"], cfg.SSOD.nms_conf_thres, cfg.SSOD.nms_iou_thres)
    rd, rk = o_nms.non_max_suppression_ssod(cap["tp"].cpu().numpy(), cfg.SSOD.nms_conf_thres, cfg.SSOD.nms_iou_thres)
    keep_ok = all(int(counts[i]) == rk[i].shape[0] and np.array_equal(keep[i, :int(counts[i])].cpu().numpy(), rk[i])
                  for i in range(Bu))
    parity = dict(against="oracle/step.py (fp32 CPU restatement of the reference step), same weights and inputs, 1+1 images",
                  dtype="bf16", loss_rel_dev={k: round(v, 6) for k, v in loss_rel.items()}, max_loss_rel_dev=max(loss_rel.values()),
                  nms_keep_indices_bit_exact=bool(keep_ok), n_pseudo_labels=[int(counts.sum()), int(sum(k.shape[0] for k in rk))],
                  tolerance="loss terms 5e-2 (bf16 storage); NMS indices bit-exact on identical decoded inputs; fp32 mode "
                            "1e-4: tests/test_step_fullsize.py")
    del tr
    torch.cuda.empty_cache()
    t0, n = time.time(), 0
    if warm < seconds:                       # bounded: ~`seconds` of timed CPU work, at least one step
        while n < 1 or (time.time() - t0 + warm < seconds and n < 8):
            step(); n += 1
        dt = (time.time() - t0) / n
    else:
        n, dt = 1, warm
    base = dict(value=(Bl + Bu) / dt, unit="images/s", cores=cores, kind="port",
                sample=f"YOLOv5l SSOD step, {Bl} labeled + {Bu} unlabeled 640x640, {n} steps, plain-torch fp32 CPU port "
                       f"(oracle/step.py; all {ref['t9'].shape[0]} pseudo labels)")
    return base, parity


def _respawn(a):
    """`python bench.py --gpus N` without a launcher: start the N ranks ourselves (one process per GPU over RCCL), exactly
    as the documented torch.distributed.run command would."""
    import subprocess
    ndev = torch.cuda.device_count()
    if ndev < a.gpus:
        raise SystemExit(f"bench.py --gpus {a.gpus}: only {ndev} GPU(s) visible -- refusing to report a {ndev}-GPU number as {a.gpus}")
    port = 29500 + (os.getpid() % 2000)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={a.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    raise SystemExit(subprocess.call(cmd, env=env))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=40)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--per-rank", type=int, default=32, help="labeled (= unlabeled) images per rank")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-overlap", action="store_true", help="run the teacher on the main stream (A/B)")
    ap.add_argument("--graph", action="store_true", help="replay the step as one captured HIP graph (trainer/graph_step.py) instead of "
                    "issuing every launch from Python (A/B: the 32+32 step is GPU-bound, both take ~64 ms)")
    ap.add_argument("--host-inputs", action="store_true", help="PCIe-inclusive variant: every step receives its three "
                    "uint8 image batches from pinned host memory (never the reported `value`; noted in DESIGN.md)")
    ap.add_argument("--dump-launches", default=None, help="write (kernel, flops, bytes, ms) of every timed conv launch of the "
                    "instrumented step to this JSON file (tools/launch_table.py prints it)")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend (nccl = RCCL); 'gloo' lets two "
                    "ranks share ONE GPU to exercise the N>1 code path where only a single GPU is available")
    a = ap.parse_args()

    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        _respawn(a)
    world = int(os.environ.get("WORLD_SIZE", 1))
    rank = int(os.environ.get("RANK", 0))
    local_rank = int(os.environ.get("LOCAL_RANK", 0))
    assert world == a.gpus, f"--gpus {a.gpus} but WORLD_SIZE={world}"
    ndev = torch.cuda.device_count()
    if a.backend == "nccl" and ndev < world:
        raise SystemExit(f"bench.py --gpus {a.gpus}: only {ndev} GPU(s) visible (use --backend gloo to share one GPU for a functional check)")
    dev_index = local_rank % max(ndev, 1)
    torch.cuda.set_device(dev_index)
    device = torch.device("cuda", dev_index)
    if world > 1:
        if a.backend == "nccl":
            dist.init_process_group("nccl", device_id=device)
        else:
            dist.init_process_group(a.backend)

    from efficientteacher_amd import ops
    cfg, tr = build_trainer(device, rank, world, dev_index, a.per_rank)
    rng = np.random.default_rng(1234 + rank)
    S = cfg.Dataset.img_size
    Bl = Bu = a.per_rank
    imgs, targets, u_str, u_ori, M_s = make_batch(rng, Bl, Bu, S, device)
    g = torch.Generator(device="cpu").manual_seed(99 + rank)
    pw = torch.cat((torch.full((1,), 16.0), torch.full((80,), 4.0)))
    synth = (torch.rand(Bu, 25200, 81, generator=g) ** pw).to(device)

    def hook(tp):           # a random-init teacher scores nothing above 0.1: SURVEY.md 8(d) synthetic scores
        tp[..., 4:] = synth
        return tp
    tr.teacher_pred_hook = hook
    tr.overlap_teacher = not a.no_overlap
    if a.graph:
        tr.use_graph = True

    ni = 2000               # past the warm-up ramp's first iterations, inside warm-up like early training

    feed = None
    if a.host_inputs:           # what a data loader hands over: uint8 NCHW batches in host memory, every step
        from efficientteacher_amd.utils.prefetch import DevicePrefetcher
        host = [(t * 255).round().to(torch.uint8).cpu() for t in (imgs, u_str, u_ori)]

        def batches():
            while True:
                yield host
        feed = DevicePrefetcher(batches(), device)      # copy stream, one step ahead; /255 happens in the pack kernel

    def step(i):
        if feed is not None:
            im, us, uo = next(feed)
            return tr.train_instance(im, targets, None, us, uo, None, M_s, ni + i)
        return tr.train_instance(imgs, targets, None, u_str, u_ori, None, M_s, ni + i)

    for i in range(a.warmup):
        step(i)
    timer = ops.KernelTimer()

    def sync():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    # HIP-event timing of the conv launches (roofline leg) on a sample of the timed steps: two events per
    # launch are ~870 extra stream operations per step, ~3 % of the step if every step is instrumented
    timed = {a.steps // 2} if a.steps > 2 else set(range(a.steps))
    sync()
    t0 = time.perf_counter()
    ar_rows = []
    ddp = tr.model if hasattr(tr.model, "collect_timing") else None
    graph_default = tr.use_graph
    for i in range(a.steps):
        ops.TIMER = timer if i in timed else None
        tr.use_graph = graph_default and i not in timed      # HIP events cannot be recorded inside a replayed graph:
        if ddp is not None:                                     # the instrumented step(s) of the timed region run eagerly
            ddp.timing = i in timed
        items = step(a.warmup + i)
    ops.TIMER = None
    tr.use_graph = graph_default
    t_enq = time.perf_counter() - t0          # host time to enqueue the K steps (no device sync inside a step)
    sync()
    dt = time.perf_counter() - t0
    if ddp is not None:
        t_ar = ddp.collect_timing()
        if t_ar is not None:
            ar_rows.append(t_ar)
    if world > 1:
        t = torch.tensor([dt], device=device, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    loss_ok = all(math.isfinite(float(v)) for v in items.values())
    # outside the timed region: what issuing ONE step costs the host when the HIP queue is empty at its start (inside the
    # timed loop the host runs ahead until the queue is full and then advances at the GPU's pace, so t_enq ~ dt there)
    enq_empty = []
    for j in range(3):
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        step(a.warmup + a.steps + j)
        enq_empty.append((time.perf_counter() - t1) * 1e3)
    torch.cuda.synchronize()
    # also outside the timed region: ONE instrumented step with the teacher on the main stream (no kernel shares the GPU with
    # the timed one): the same launches' durations without the inflation the overlapped teacher stream causes
    solo = None
    try:
        t_solo = ops.KernelTimer()
        keep_overlap = tr.overlap_teacher
        tr.overlap_teacher, ops.TIMER, tr.use_graph = False, t_solo, False
        step(a.warmup + a.steps + 3)
        ops.TIMER, tr.overlap_teacher, tr.use_graph = None, keep_overlap, graph_default
        torch.cuda.synchronize()
        solo = t_solo.summary()
    except Exception:
        ops.TIMER = None

    if rank == 0:
        try:                 # the roofline leg must never cost the throughput line
            agg = timer.summary()
            if a.dump_launches:
                with open(a.dump_launches, "w") as f:
                    json.dump([dict(kernel=t, flops=fl, launches=n, bytes=nb, ms=ea.elapsed_time(eb), shape=sh)
                               for (t, fl, n, nb, ea, eb), sh in zip(timer.rows, timer.shapes)], f)
            dom = max((k for k in agg if k.startswith("conv_gemm") and "parity classes" not in k), key=lambda k: agg[k]["ms"])
            d = agg[dom]
            per_launch_flops = d["flops"] / d["launches"]
            per_launch_s = d["ms"] * 1e-3 / d["launches"]
            conv_ms = sum(v["ms"] for v in agg.values()) / len(timed)
            conv_fl = sum(v["flops"] for v in agg.values()) / len(timed)
            # HBM/fabric traffic of the same kernel from the committed PMC passes over this very command
            # (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate runs; tools/pmc_summarize.py, tools/pmc_to_traffic.py)
            traffic, traffic_src = None, None
            try:
                pt = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "pmc_traffic.json")))
                if dom in pt["kernels"]:
                    traffic, traffic_src = pt["kernels"][dom]["bytes_per_launch"], pt["source"]
            except (OSError, ValueError, KeyError):
                pass
            roof = dict(bound="mfma", kernel=dom, achieved=per_launch_flops / per_launch_s / 1e12, peak=PEAK_BF16 / 1e12,
                        unit="TFLOP/s", frac=per_launch_flops / per_launch_s / PEAK_BF16, traffic=traffic,
                        traffic_unit="bytes per launch (2*FETCH_SIZE + WRITE_SIZE)", traffic_source=traffic_src,
                        algorithmic_bytes_per_launch=d["bytes"] / d["launches"],
                        launches_per_step=d["launches"] / len(timed), instrumented_steps=len(timed),
                        avg_launch_us=per_launch_s * 1e6, algorithmic_gflop_per_launch=per_launch_flops / 1e9,
                        all_conv_kernels=dict(ms_per_step=conv_ms, tflops=conv_fl / (conv_ms * 1e-3) / 1e12,
                                              algorithmic_tflop_per_step=conv_fl / 1e12))
            if solo and dom in solo:
                sd = solo[dom]
                roof["same_kernel_teacher_not_overlapped"] = dict(
                    avg_launch_us=sd["ms"] * 1e3 / sd["launches"], frac=sd["flops"] / (sd["ms"] * 1e-3) / PEAK_BF16,
                    all_conv_ms=sum(v["ms"] for v in solo.values()),
                    note="one extra step outside the timed region with the teacher forward on the main stream: in the timed "
                         "region the teacher's launches share the GPU with the student's and lengthen them")
        except Exception as e:
            roof = dict(bound="mfma", kernel=None, achieved=None, peak=PEAK_BF16 / 1e12, unit="TFLOP/s", frac=None,
                        traffic=None, error=f"{type(e).__name__}: {e}")
        step_flop = F_IMG * Bu + 3 * F_IMG * (Bl + Bu)
        out = {
            "metric": "SSOD images/sec (teacher+student step) YOLOv5l 640px",
            "value": world * (Bl + Bu) * a.steps / dt, "unit": "images/s", "n_gpus": world, "steps": a.steps,
            "warmup": a.warmup, "ms_per_step": dt / a.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "bf16", "data": "synthetic (random-init YOLOv5l; teacher obj/cls scores "
            "replaced by U^16 / U^4 so that NMS and the pseudo-label loss do representative work)",
            "config": {"workload": f"YOLOv5l Efficient-Teacher SSOD, {Bl} labeled + {Bu} unlabeled 640px per GPU "
                                   "(BASELINE configs[2]" + ("" if Bl == 32 else ", reduced per-rank batch") + ")", "global_batch": world * (Bl + Bu), "img_size": S,
                       "parallelism": f"dp{world}", "optimizer_every_step": True,
                       "algorithmic_tflop_per_step_per_gpu": step_flop / 1e12,
                       "step_tflops_per_gpu": step_flop / (dt / a.steps) / 1e12,
                       "frac_of_bf16_mfma_peak": step_flop / (dt / a.steps) / PEAK_BF16, "loss_finite": loss_ok,
                       "rccl_ranks": dist.get_world_size() if world > 1 else 1, "env_knobs": ops.env_knobs(),
                       "grad_allreduce": (dict(bytes=int(tr.model.flat_state().grads.numel() * 4), span_ms=ar_rows[-1][0],
                                               exposed_ms=ar_rows[-1][1], note="span: first chunk launch (during backward) -> "
                                               "last collective complete; exposed: compute stream waiting after backward")
                                          if ar_rows else None),
                       "host_enqueue_ms_per_step": t_enq / a.steps * 1e3,
                       "host_enqueue_ms_empty_queue": sorted(enq_empty)[1],
                       "host_enqueue_note": "per_step is measured inside the timed loop, where the host blocks on the full HIP "
                                            "queue (back-pressure: it tracks the GPU step time); empty_queue is the median host time "
                                            "to issue one step after a device synchronise, i.e. the real launch cost",
                       "step_graph": dict(enabled=bool(graph_default and world == 1), replays=getattr(tr._graph, "replays", 0) if getattr(tr, "_graph", None) else 0,
                                          eager_instrumented_steps=len(timed)), "inputs": "host uint8 (PCIe inclusive)" if a.host_inputs else "resident in HBM"},
            "roofline": roof,
        }
        if world == 1 and not a.no_cpu_baseline:
            try:
                del tr, imgs, u_str, u_ori, synth
                torch.cuda.empty_cache()
                out["cpu_baseline"], out["parity_check"] = cpu_baseline(cfg, device)
            except Exception as e:   # never lose the GPU number to the baseline leg
                out["cpu_baseline"] = dict(value=None, unit="images/s", cores=os.cpu_count(), kind="port",
                                           sample=f"failed: {type(e).__name__}: {e}")
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
