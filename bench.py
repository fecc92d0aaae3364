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
}
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
                "frac": achieved / peak, "traffic": None,
                "kernel": "camal::conv_kernel<*> (implicit-GEMM Conv1d+BN+ReLU, all MFMA layers)",
                "launches": conv_n, "avg_launch_ms": conv_ms / max(conv_n, 1),
                "alg_flops_per_step": flops_step,
            },
            "kernel_ms_per_step": {k: v[1] / args.steps for k, v in prof.items()},
            "sanity": {"state_on_fraction": on_frac, "windows_voted_on": vote_frac},
        }
        if not args.no_cpu_baseline:
            res["cpu_baseline"] = cpu_baseline(sds, x_host)
        print(json.dumps(res), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
