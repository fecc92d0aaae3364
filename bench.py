#!/usr/bin/env python3
"""bench.py — windows/sec of the detect + CAM-localize hot path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload NAME]

One "step" = one pass of the hot path (ResNet forward(s) -> CAM head -> fused ensemble
post-processing to on/off states and clipped power) over one batch of synthetic windows that
is already resident in HBM.  Every rank owns one GPU and processes its own batch (windows
shard with no collective) => weak scaling; the only distributed calls are the
barriers and the MAX of the per-rank times.

Workloads (BASELINE.json `configs`):
  resnet_fp32   (default) configs[1]: single ResNet k=7, 4096 x 1024, fp32 MFMA
  ensemble_bf16 configs[2]: 5 members k=5,7,9,15,25, bf16 MFMA, REFIT-shaped day windows (L=1440)
  ensemble_fp32 same ensemble in fp32
  long_bf16     configs[4]: ensemble, L=10080 week windows, + power estimate
  train_fp32    configs[3]: ensemble training step (forward + backward + AdamW for each of the 5 members)
                on IDEAL-shaped windows (128 x 510 per GPU), gradients all-reduced over RCCL when N > 1
Prints ONE JSON line (rank 0).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402

WORKLOADS = {
    "resnet_fp32": dict(kernels=[7], B=4096, L=1024, precision="fp32", shape="ukdale"),
    "ensemble_bf16": dict(kernels=[5, 7, 9, 15, 25], B=4096, L=1440, precision="bf16", shape="refit"),
    "ensemble_fp32": dict(kernels=[5, 7, 9, 15, 25], B=1024, L=1024, precision="fp32", shape="ukdale"),
    "long_bf16": dict(kernels=[5, 7, 9, 15, 25], B=512, L=10080, precision="bf16", shape="refit"),
    "train_fp32": dict(kernels=[5, 7, 9, 15, 25], B=128, L=510, precision="fp32", shape="ideal"),
}
HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md "HBM": 8 TB/s spec (about 6.3 TB/s achievable)
# dense MFMA peaks, /opt/skills/guides/MI355X_MICROARCH.md "Chip-level parameters"
PEAK_TFLOPS = {"fp32": 157.3, "bf16": 2500.0}


def conv_macs_per_timestep(k):
    """MACs per timestep executed by the MFMA conv kernel for one member (SURVEY §8:
    M(k) = 24640 k + 376896 minus the Cin=1 stem (64 k) and block-1 residual (64))."""
    return 24576 * k + 376832


def make_windows(shape, n, length, seed):
    from camal_amd import synth
    if shape == "refit":
        return synth.refit_windows(n, length, seed=seed)
    return synth.ukdale_windows(n, length, seed=seed)


def measured_traffic(workload, kernel_class):
    """HBM bytes per launch from the committed rocprofv3 PMC passes (profiles/traffic.json, made by
    tools/traffic.py from FETCH_SIZE x2 (gfx950 correction) + WRITE_SIZE); None if not collected."""
    try:
        with open(os.path.join(ROOT, "profiles", "traffic.json")) as f:
            return json.load(f)[workload][kernel_class]["bytes_per_launch"]
    except (OSError, KeyError, ValueError):
        return None


def train_main(args, wl, rank, local_rank, world, dev, dist):
    """configs[3]: one step = forward + backward + AdamW of every ensemble member on this rank's batch;
    with N > 1 each member's flat gradient bucket is all-reduced (mean) over RCCL before its update."""
    from camal_amd import synth
    from camal_amd.resnet import CamALResNet
    from camal_amd.trainer import TrainEngine
    B, L, ks = wl["B"], wl["L"], wl["kernels"]
    x_host, y_host = synth.ideal_windows(B, L, seed=2000 + rank)
    x = torch.from_numpy(x_host).to(dev)
    y = torch.from_numpy(y_host).to(dev)
    engines = []
    for k in ks:
        torch.manual_seed(k)  # same initial weights on every rank (DDP starts from one model)
        m = CamALResNet(kernel_size=k)
        engines.append(TrainEngine(m, dev, B, L))

    def step():
        # a member's gradient all-reduce (N > 1) runs under the next member's forward and backward
        pending = []
        for e in engines:
            e.forward_backward(x, y)
            pending.append(e.start_grad_allreduce())
        for e, p in zip(engines, pending):
            e.optimizer_step(1e-3, 1e-2, pending=p)

    def sync():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    sync()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    sync()
    dt = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    if rank == 0:
        # forward conv MACs x3 (forward, data gradient, weight gradient); the stem has no data gradient
        flops_step = 3 * 2.0 * L * B * sum(conv_macs_per_timestep(k) for k in ks)
        achieved = flops_step * args.steps / dt / 1e12
        res = {
            "metric": "windows/sec (train: forward+backward+AdamW, all members)",
            "value": world * B * args.steps / dt, "unit": "windows/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"train_fp32: E={len(ks)} k={ks} B={B}/GPU L={L} fwd+bwd+AdamW per member",
                       "windows_per_gpu_per_step": B, "window_len": L, "kernel_sizes": ks,
                       "inputs": "resident in HBM",
                       "parallelism": f"data-parallel x{world}, one all-reduce of each member's gradient bucket"},
            "roofline": {"bound": "mfma", "achieved": achieved, "peak": PEAK_TFLOPS["fp32"], "unit": "TFLOP/s",
                         "frac": achieved / PEAK_TFLOPS["fp32"], "traffic": None,
                         "kernel": "whole step (wall clock): conv forward + dgrad + wgrad MFMA kernels, BN/loss/AdamW",
                         "alg_flops_per_step": flops_step},
            "sanity": {"loss": [float(e.loss.item()) for e in engines]},
        }
        print(json.dumps(res), flush=True)


def cpu_baseline(sds, x, budget_s=15.0):
    """The oracle's torch-CPU restatement of the reference loop, timed on this host's cores
    on a bounded sample of the same windows."""
    from oracle import torch_path
    threads = min(os.cpu_count() or 1, int(os.environ.get("OMP_NUM_THREADS", "16")))
    torch.set_num_threads(threads)
    torch_path.predict_loop(sds, x[:1])  # warm-up
    t0 = time.perf_counter()
    torch_path.predict_loop(sds, x[:2])
    per_window = (time.perf_counter() - t0) / 2
    n = int(max(4, min(len(x), budget_s / max(per_window, 1e-6))))
    t0 = time.perf_counter()
    torch_path.predict_loop(sds, x[:n])
    dt = time.perf_counter() - t0
    return {"value": n / dt, "unit": "windows/s", "cores": threads, "kind": "port",
            "sample": f"{n} of the step's windows, E={len(sds)}, L={x.shape[1]}, oracle/torch_path.predict_loop "
                      f"(batch-1 forward per member and window, like core.py:290-317), {dt:.1f} s"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--workload", default="resnet_fp32", choices=sorted(WORKLOADS))
    ap.add_argument("--batch", type=int, default=0, help="windows per GPU per step (default: workload's)")
    ap.add_argument("--chunk", type=int, default=0, help="windows per internal pass")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    wl = dict(WORKLOADS[args.workload])
    if args.batch:
        wl["B"] = args.batch
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run")
    torch.cuda.set_device(local_rank)
    dev = f"cuda:{local_rank}"
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=torch.device(dev))

    if args.workload == "train_fp32":
        train_main(args, wl, rank, local_rank, world, dev, dist)
        if dist is not None:
            dist.barrier()
            dist.destroy_process_group()
        return

    from camal_amd.calibrate import calibrated_state_dicts
    from camal_amd.engine import Ensemble

    B, L, ks = wl["B"], wl["L"], wl["kernels"]
    x_host = make_windows(wl["shape"], B, L, seed=1000 + rank)
    sds = calibrated_state_dicts(ks, seed=0, sample_x=x_host[:64], device=dev, precision="fp32")
    ens = Ensemble(sds, device=dev, precision=wl["precision"], chunk_windows=args.chunk or None)
    x = torch.from_numpy(x_host).to(dev)
    out = {"prob": torch.empty(B, dtype=torch.float64, device=dev),
           "state": torch.empty(B, L, dtype=torch.float32, device=dev),
           "power": torch.empty(B, L, dtype=torch.float32, device=dev)}

    def step():
        return ens.localize(x, p_on=2000.0, scale_mul=1000.0, out=out)

    def sync():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    sync()
    ens.set_profiling(True)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    sync()
    dt = time.perf_counter() - t0
    prof = ens.get_profile()
    ens.set_profiling(False)
    if dist is not None:
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    if rank == 0:
        on_frac = float((out["state"] != 0).float().mean().item())
        vote_frac = float((out["prob"] > 0.5).double().mean().item())
        conv_n = prof["conv"][0] + prof["conv_cam_head"][0]
        conv_ms = prof["conv"][1] + prof["conv_cam_head"][1]
        flops_step = 2.0 * L * B * sum(conv_macs_per_timestep(k) for k in ks)
        achieved = flops_step * args.steps / (conv_ms * 1e-3) / 1e12 if conv_ms > 0 else 0.0
        peak = PEAK_TFLOPS[wl["precision"]]
        post_bytes_step = 4.0 * B * L * (len(ks) * vote_frac + 3)
        post_gbs = post_bytes_step * args.steps / (prof["post"][1] * 1e-3) / 1e9 if prof["post"][1] > 0 else 0.0
        res = {
            "metric": "windows/sec (detect+CAM localize)",
            "value": world * B * args.steps / dt,
            "unit": "windows/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32" if wl["precision"] == "fp32" else "bf16 (f32 accumulate)",
            "data": "synthetic",
            "config": {"workload": f"{args.workload}: E={len(ks)} k={ks} B={B}/GPU L={L} forward+CAM+post+power",
                       "windows_per_gpu_per_step": B, "window_len": L, "kernel_sizes": ks,
                       "inputs": "resident in HBM", "parallelism": f"window-sharded x{world}, no collective"},
            "roofline": {
                "bound": "mfma", "achieved": achieved, "peak": peak, "unit": "TFLOP/s",
                "frac": achieved / peak, "traffic": measured_traffic(args.workload, "conv"),
                "kernel": "camal::conv_kernel<*> + camal::block_kernel<*> (implicit-GEMM Conv1d+BN+ReLU[+skip][+CAM head], "
                          "every MFMA launch of the step)",
                "launches": conv_n, "avg_launch_ms": conv_ms / max(conv_n, 1),
                "alg_flops_per_step": flops_step, "alg_flops_per_launch": flops_step * args.steps / max(conv_n, 1),
            },
            # the fused CAM post-processing kernel against the HBM roofline: bytes that must move given the
            # votes = 4 B x (E CAM rows of the windows voted on + aggregate + state + power) per timestamp
            "roofline_post": {
                "bound": "hbm", "achieved": post_gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": post_gbs / HBM_PEAK_GBS, "traffic": measured_traffic(args.workload, "post"),
                "kernel": "camal::post_kernel (ensemble average + vote gate + moving average + states + power)",
                "launches": prof["post"][0], "avg_launch_ms": prof["post"][1] / max(prof["post"][0], 1),
                "alg_bytes_per_launch": post_bytes_step * args.steps / max(prof["post"][0], 1),
            },
            "kernel_ms_per_step": {k: v[1] / args.steps for k, v in prof.items()},
            "sanity": {"state_on_fraction": on_frac, "windows_voted_on": vote_frac},
        }
        if not args.no_cpu_baseline and world == 1:  # rank 0 at N = 1 only: the other ranks would idle at the barrier
            res["cpu_baseline"] = cpu_baseline(sds, x_host)
        print(json.dumps(res), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
