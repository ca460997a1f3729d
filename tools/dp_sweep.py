"""First-lease sweep of the data-parallel knobs on an N-GPU node.

    python tools/dp_sweep.py --gpus 8 [--per-rank 16] [--steps 20] [--out gpurun_out/dp_sweep.json]

Runs BASELINE configs[3] (YOLOv5l SSOD, 16 + 16 images per rank; --per-rank 32 = the weak-scaling point) through bench.py for every
cell of   {captured step graph, eager}  x  ET_ALLREDUCE_CHUNK_MB {24, 48, 96}  x  ET_RCCL_CHANNELS {library default, 8, 16}
x  ET_ALLREDUCE_DTYPE {fp32, bf16: the r05 wire format, half the bytes per link}   and writes ONE JSON: per cell the images/s, ms per step, the gradient all-reduce's span and EXPOSED time (what the compute
stream waits for after backward, bench.py `grad_allreduce`), and whether the capture of the collectives was accepted.  Each cell is
its own `torch.distributed.run` launch (RCCL reads its channel count when the communicator is created).
On a single-GPU box `--gpus 1` runs the same matrix over a ONE-rank RCCL group (bench.py --force-dp): the collectives execute, the
numbers say nothing about xGMI.  Nothing here reads /root/reference.
"""
import argparse
import itertools
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run_cell(gpus, per_rank, steps, warmup, graph, chunk_mb, channels, port, timeout, wire="fp32"):
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"),
               ET_ALLREDUCE_CHUNK_MB=str(chunk_mb), ET_ALLREDUCE_DTYPE=wire)
    env.pop("ET_RCCL_CHANNELS", None)
    if channels:
        env["ET_RCCL_CHANNELS"] = str(channels)
    args = ["--gpus", str(gpus), "--per-rank", str(per_rank), "--steps", str(steps), "--warmup", str(warmup), "--no-cpu-baseline",
            "--no-weak-point", "--graph" if graph else "--no-graph"]
    if gpus == 1:
        cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--force-dp"] + args
    else:
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={gpus}", "--master-addr", "127.0.0.1",
               "--master-port", str(port), os.path.join(ROOT, "bench.py")] + args
    t0 = time.time()
    cell = dict(graph=bool(graph), chunk_mb=chunk_mb, rccl_channels=channels or "default", wire=wire)
    try:
        p = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=timeout)
        line = next((ln for ln in reversed(p.stdout.strip().splitlines()) if ln.startswith("{")), None)
        if p.returncode != 0 or line is None:
            cell.update(error=f"rc {p.returncode}", stderr_tail=p.stderr[-600:])
        else:
            d = json.loads(line)
            ar = d["config"].get("grad_allreduce") or {}
            sg = d["config"].get("step_graph") or {}
            cell.update(images_per_s=d["value"], ms_per_step=d["ms_per_step"], allreduce_span_ms=ar.get("span_ms"),
                        allreduce_exposed_ms=ar.get("exposed_ms"), allreduce_bytes=ar.get("bytes"),
                        graph_enabled=sg.get("enabled"), graph_error=sg.get("error"), rccl_env=d["config"].get("rccl_env"),
                        host_enqueue_ms_empty_queue=d["config"].get("host_enqueue_ms_empty_queue"))
    except subprocess.TimeoutExpired:
        cell.update(error=f"timeout after {timeout} s")
    cell["wall_s"] = round(time.time() - t0, 1)
    return cell


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=8)
    ap.add_argument("--per-rank", type=int, default=16)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--chunks", default="24,48,96")
    ap.add_argument("--channels", default="0,8,16", help="0 = the library's default")
    ap.add_argument("--wire", default="fp32,bf16", help="gradient all-reduce wire formats (ET_ALLREDUCE_DTYPE)")
    ap.add_argument("--timeout", type=int, default=600)
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "dp_sweep.json"))
    a = ap.parse_args()
    cells = []
    port = 29500 + (os.getpid() % 1500)
    for i, (graph, chunk, ch, wire) in enumerate(itertools.product((True, False), [int(c) for c in a.chunks.split(",")],
                                                                   [int(c) for c in a.channels.split(",")], a.wire.split(","))):
        cell = run_cell(a.gpus, a.per_rank, a.steps, a.warmup, graph, chunk, ch, port + i, a.timeout, wire)
        cells.append(cell)
        print(json.dumps(cell), flush=True)
    ok = [c for c in cells if "images_per_s" in c]
    best = max(ok, key=lambda c: c["images_per_s"]) if ok else None
    out = dict(what=f"YOLOv5l SSOD, {a.per_rank}+{a.per_rank} images per rank, {a.gpus} rank(s): step graph x all-reduce chunk x RCCL channels x wire format",
               gpus=a.gpus, per_rank=a.per_rank, steps=a.steps, cells=cells, best=best)
    os.makedirs(os.path.dirname(a.out), exist_ok=True)
    with open(a.out, "w") as f:
        json.dump(out, f, indent=1)
    print("best:", json.dumps(best))


if __name__ == "__main__":
    main()
