#!/usr/bin/env python3
"""bench.py — windows/sec of the detect + CAM-localize hot path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload NAME] [--also auto|all|none]

One "step" = one pass of the hot path (ResNet forward(s) -> CAM head -> fused ensemble
post-processing to on/off states and clipped power) over one batch of synthetic windows that
is already resident in HBM.  Every rank owns one GPU and processes its own batch (windows
shard with no collective) => weak scaling; the only distributed calls on the
inference path are the barriers and the MAX of the per-rank times.

Workloads (BASELINE.json `configs`):
  resnet_fp32   (headline) configs[1]: single ResNet k=7, 4096 x 1024, fp32 MFMA
  ensemble_bf16 configs[2]: 5 members k=5,7,9,15,25, bf16 MFMA, REFIT-shaped day windows (L=1440)
  ensemble_fp32 same ensemble in fp32 (1024 x 1024)
  long_bf16     configs[4]: ensemble, L=10080 week windows, + power estimate
  train_fp32    configs[3]: ensemble training step (forward + backward + AdamW for each of the 5 members)
                on IDEAL-shaped windows (128 x 510 per GPU), gradients all-reduced over RCCL when N > 1
  cfg0 (CPU)    configs[0]: E=5, 64 x 1024 on the host cores = `cpu_baseline` (oracle/torch_path)

Prints ONE JSON line (rank 0): the headline workload's contract fields + `roofline`, `roofline_post`,
`cpu_baseline`, and under `also` the same figures for every other workload, timed in the same run.

`--gpus N` with N > 1 and no WORLD_SIZE in the environment: this process only spawns the N rank processes
(children, before anything touches a GPU) and relays rank 0's line; under torch.distributed.run the ranks are
already there and it runs as one of them.
"""
import argparse
import json
import math
import os
import socket
import subprocess
import sys
import threading
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

WORKLOADS = {
    "resnet_fp32": dict(kernels=[7], B=4096, L=1024, precision="fp32", shape="ukdale", config="configs[1]"),
    "ensemble_bf16": dict(kernels=[5, 7, 9, 15, 25], B=4096, L=1440, precision="bf16", shape="refit", config="configs[2]"),
    "ensemble_fp32": dict(kernels=[5, 7, 9, 15, 25], B=1024, L=1024, precision="fp32", shape="ukdale",
                          config="configs[0]'s ensemble in fp32 on the GPU"),
    "long_bf16": dict(kernels=[5, 7, 9, 15, 25], B=512, L=10080, precision="bf16", shape="refit", config="configs[4]"),
    "train_fp32": dict(kernels=[5, 7, 9, 15, 25], B=128, L=510, precision="fp32", shape="ideal", config="configs[3]"),
}
ORDER = ["resnet_fp32", "ensemble_bf16", "ensemble_fp32", "long_bf16", "train_fp32"]
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


def rocprof_avg_ms(workload, kernel_class):
    """Average duration (ms) of that kernel class in the committed `rocprofv3 --kernel-trace --stats` run of the same
    workload (profiles/kernel_durations.json, made by tools/durations.py from the kernel_stats csv); None if absent.
    A HIP-event pair around ONE short launch also times the dispatch and completion hand-offs (about 3 us on a
    13-25 us post-processing kernel), which rocprofv3's begin / end stamps do not."""
    try:
        with open(os.path.join(ROOT, "profiles", "kernel_durations.json")) as f:
            return json.load(f)[workload][kernel_class]["avg_ms"]
    except (OSError, KeyError, ValueError):
        return None


class Ranks:
    """What the timing protocol needs from torch.distributed (or nothing at N = 1)."""

    def __init__(self, rank, world, dev, dist):
        self.rank, self.world, self.dev, self.dist = rank, world, dev, dist

    def sync(self):
        import torch
        torch.cuda.synchronize()
        if self.dist is not None:
            self.dist.barrier()
        torch.cuda.synchronize()

    def max(self, v):
        if self.dist is None:
            return v
        import torch
        t = torch.tensor([v], dtype=torch.float64, device="cpu" if self.dist.get_backend() == "gloo" else self.dev)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
        return float(t.item())


def warm_up(step, steps, warmup, R, min_seconds=0.0):
    """The W untimed warm-up steps.  -> (steps to time, seconds of the last warm step): `steps` as given, or with
    min_seconds enough steps for a timed region of that length (the `also` workloads; same count on every rank)."""
    for _ in range(max(warmup - 1, 0)):
        step()
    R.sync()
    t0 = time.perf_counter()
    if warmup > 0:
        step()
    R.sync()
    est = R.max(time.perf_counter() - t0)  # one warm step, for sizing only
    if min_seconds > 0 and est > 0:
        steps = max(steps, int(math.ceil(min_seconds / est)))
    return steps, est


def run_inference(name, wl, steps, warmup, R, args, min_seconds=0.0, with_cpu_inputs=False):
    """One inference workload -> its result dict (the contract's fields for that workload)."""
    import torch
    from camal_amd.calibrate import calibrated_state_dicts
    from camal_amd.engine import Ensemble

    B, L, ks = wl["B"], wl["L"], wl["kernels"]
    dev = R.dev
    x_host = make_windows(wl["shape"], B, L, seed=1000 + R.rank)
    sds = calibrated_state_dicts(ks, seed=0, sample_x=x_host[:64], device=dev, precision="fp32")
    ens = Ensemble(sds, device=dev, precision=wl["precision"], chunk_windows=args.chunk or None)
    x = torch.from_numpy(x_host).to(dev)
    out = {"prob": torch.empty(B, dtype=torch.float64, device=dev),
           "state": torch.empty(B, L, dtype=torch.float32, device=dev),
           "power": torch.empty(B, L, dtype=torch.float32, device=dev)}

    def step():
        return ens.localize(x, p_on=2000.0, scale_mul=1000.0, out=out)

    steps, _ = warm_up(step, steps, warmup, R, min_seconds)
    R.sync()
    ens.set_profiling(True)
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    R.sync()
    dt = R.max(time.perf_counter() - t0)
    prof = ens.get_profile()
    ens.set_profiling(False)

    res = None
    if R.rank == 0:
        on_frac = float((out["state"] != 0).float().mean().item())
        vote_frac = float((out["prob"] > 0.5).double().mean().item())
        conv_n = prof["conv"][0] + prof["conv_cam_head"][0]
        conv_ms = prof["conv"][1] + prof["conv_cam_head"][1]
        flops_step = 2.0 * L * B * sum(conv_macs_per_timestep(k) for k in ks)
        achieved = flops_step * steps / (conv_ms * 1e-3) / 1e12 if conv_ms > 0 else 0.0
        peak = PEAK_TFLOPS[wl["precision"]]
        post_bytes_step = 4.0 * B * L * (len(ks) * vote_frac + 3)
        post_gbs = post_bytes_step * steps / (prof["post"][1] * 1e-3) / 1e9 if prof["post"][1] > 0 else 0.0
        res = {
            "value": R.world * B * steps / dt, "unit": "windows/s", "steps": steps, "warmup": warmup,
            "ms_per_step": dt / steps * 1e3,
            "dtype": "f32" if wl["precision"] == "fp32" else "bf16 (f32 accumulate)",
            "config": {"workload": f"{name} = {wl['config']}: E={len(ks)} k={ks} B={B}/GPU L={L} forward+CAM+post+power",
                       "windows_per_gpu_per_step": B, "window_len": L, "kernel_sizes": ks,
                       "inputs": "resident in HBM", "parallelism": f"window-sharded x{R.world}, no collective"},
            "roofline": {
                "bound": "mfma", "achieved": achieved, "peak": peak, "unit": "TFLOP/s",
                "frac": achieved / peak, "traffic": measured_traffic(name, "conv"),
                "kernel": "camal::conv_kernel<*> + camal::block_kernel<*> (implicit-GEMM Conv1d+BN+ReLU[+skip][+CAM head], "
                          "every MFMA launch of the step)",
                "launches": conv_n, "avg_launch_ms": conv_ms / max(conv_n, 1),
                "alg_flops_per_step": flops_step, "alg_flops_per_launch": flops_step * steps / max(conv_n, 1),
                "frac_wall_clock": flops_step * steps / dt / 1e12 / peak,
            },
            # the fused CAM post-processing kernel against the HBM roofline: bytes that must move given the
            # votes = 4 B x (E CAM rows of the windows voted on + aggregate + state + power) per timestamp
            "roofline_post": {
                "bound": "hbm", "achieved": post_gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": post_gbs / HBM_PEAK_GBS, "traffic": measured_traffic(name, "post"),
                "kernel": "camal::post_stream_kernel (ensemble average + vote gate + moving average + states + power; post_kernel for the cases it does not take)",
                "launches": prof["post"][0], "avg_launch_ms": prof["post"][1] / max(prof["post"][0], 1),
                "alg_bytes_per_launch": post_bytes_step * steps / max(prof["post"][0], 1),
            },
            "kernel_ms_per_step": {k: v[1] / steps for k, v in prof.items()},
            "sanity": {"state_on_fraction": on_frac, "windows_voted_on": vote_frac},
        }
        rp_ms = rocprof_avg_ms(name, "post")
        if rp_ms:
            rp = res["roofline_post"]
            rp["rocprof_avg_launch_ms"] = rp_ms
            rp["achieved_rocprof"] = rp["alg_bytes_per_launch"] / (rp_ms * 1e-3) / 1e9
            rp["frac_rocprof"] = rp["achieved_rocprof"] / HBM_PEAK_GBS
            rp["note"] = ("`achieved` = algorithmic bytes / HIP-event duration of the launch (contract); an event pair around one "
                          "13-25 us kernel also times the dispatch and completion hand-offs; `achieved_rocprof` uses the kernel's own "
                          "begin / end stamps from the committed rocprofv3 run of this workload (profiles/)")
        if wl["precision"] == "bf16" and not args.no_extras:
            res["sanity"]["bf16_vs_fp32"] = bf16_agreement(sds, x, out, dev)
        # the same box's plain streaming kernels (tools/hbm_probe.hip) as denominators beside the 8 TB/s spec:
        # a 1 GiB float4 copy (nothing cached), and the post kernel's own read/write mix at its own size in
        # back-to-back launches (the Infinity Cache holds those buffers: an upper bound the pipeline cannot see,
        # its CAM rows were written tens of milliseconds of HBM traffic earlier)
        try:
            if args.no_extras:
                raise RuntimeError("skipped (--no-extras)")
            sys.path.insert(0, os.path.join(ROOT, "tools"))
            import hbm_probe
            nr = max(1, int(round(len(ks) * vote_frac)) + 1)
            copy = hbm_probe.probe(1 << 28, 1, 1, device=dev, blocks=(1024, 4096, 16384), reps=5)
            mix = hbm_probe.probe(B * L, min(nr, 6), 2, device=dev, blocks=(4096,), reps=10, unrolls=(1, 4))
            rp = res["roofline_post"]
            rp["copy_gbs"], rp["same_mix_cached_gbs"] = copy, mix
            rp["frac_of_copy"] = post_gbs / copy if copy else None
        except Exception as e:  # noqa: BLE001 — a measurement extra, never fatal
            res["roofline_post"]["copy_gbs"] = f"probe failed: {e}"
        # the same box's register-only MFMA loops (tools/mfma_probe.hip) beside the spec peak: on constant operands the
        # pipes reach the spec figure; on random bf16 operands the part lowers its clock (data toggling against the power
        # limit) and the loop that does nothing else stops near 0.76-0.79 of it
        try:
            if args.no_extras:
                raise RuntimeError("skipped (--no-extras)")
            import mfma_probe
            rf = res["roofline"]
            kinds = ("bf16", "bf16_random") if wl["precision"] == "bf16" else ("fp32",)
            rf["mfma_probe_tflops"] = {k: mfma_probe.probe(k, device=dev, waves=(8,), blocks=(1024,), launch_ms=(20.0,)) for k in kinds}
            ceil = rf["mfma_probe_tflops"].get("bf16_random") or rf["mfma_probe_tflops"].get("fp32")
            rf["frac_of_probe"] = achieved / ceil if ceil else None
        except Exception as e:  # noqa: BLE001
            res["roofline"]["mfma_probe_tflops"] = f"probe failed: {e}"
    if R.rank == 0 and name == "ensemble_bf16" and R.world == 1 and not args.no_extras:
        res["host_to_host"] = host_to_host(ens, wl, B * steps / dt)
    ens.close()
    del ens, out, x
    torch.cuda.empty_cache()
    return res


def host_to_host(ens, wl, resident_rate, n_default=1 << 17, n_scale=1 << 20):
    """End to end as CamAL.predict's callers see it: NumPy float32 windows in host memory in, states + probabilities +
    power in host memory out, through Ensemble.localize_host (chunks through pinned buffers, H2D / D2H on their own
    streams under the kernels, one byte per state and float32 power over PCIe, worker threads widening into the result
    arrays).  Not `value` (that is HBM-resident).  Two passes after a warm-up that pins the buffers: the API's default
    dtypes (float64 states and power, the reference's) on n_default windows, and configs[2]'s scale — n_scale >= 1 M
    windows — with the compact result dtypes (int8 states, float32 power: 5 bytes per timestamp of host memory)."""
    import numpy as np
    L = wl["L"]
    chunk = ens.stream_chunk_windows(L)
    base = make_windows(wl["shape"], chunk, L, seed=5)
    ens.localize_host(base[:chunk // 2 + 1], p_on=2000.0, scale_mul=1000.0)  # pins the buffers, starts the workers
    res = {"unit": "windows/s", "chunk_windows": chunk, "pcie_bytes_per_window": 4 * L + 5 * L + 8, "resident_windows_per_s": resident_rate}

    def timed(N, **kw):
        x = np.tile(base, (N // chunk + 1, 1))[:N]
        t0 = time.perf_counter()
        states, prob, power = ens.localize_host(x, p_on=2000.0, scale_mul=1000.0, **kw)
        dt = time.perf_counter() - t0
        r = {"value": N / dt, "windows": N, "seconds": dt, "frac_of_resident": N / dt / resident_rate if resident_rate else None,
             "states_dtype": str(states.dtype), "power_dtype": str(power.dtype),
             "host_result_bytes": int(states.nbytes + power.nbytes + prob.nbytes), "state_on_fraction": float((states[::97] != 0).mean())}
        del x, states, prob, power
        return r

    res["default_dtypes"] = dict(timed(n_default), what="NumPy float32 in -> float64 states, float64 power, probabilities out")
    res["at_scale_compact"] = dict(timed(n_scale, states_dtype=np.int8, power_dtype=np.float32),
                                   what="configs[2] scale: >= 1 M windows, int8 states + float32 power out (bounded host memory)")
    res["value"] = res["at_scale_compact"]["value"]
    return res


def bf16_agreement(sds, x, out16, dev, n=512):
    """On/off states of the bf16 path against the fp32 path of this library on the first n windows of the step:
    agreement over ALL timestamps of every window whose fp32 vote is at least 2e-3 away from 0.5 (the fp32
    criterion of the parity tests), not only the firmly voting ones."""
    import torch
    from camal_amd.engine import Ensemble
    n = min(n, x.shape[0])
    ens32 = Ensemble(sds, device=dev, precision="fp32")
    r32 = ens32.localize(x[:n], p_on=2000.0, scale_mul=1000.0)
    torch.cuda.synchronize()
    p32, p16 = r32["prob"], out16["prob"][:n]
    counted = (p32 - 0.5).abs() >= 2e-3
    same = (r32["state"] == out16["state"][:n])
    agree = float(same[counted].float().mean().item()) if bool(counted.any()) else 1.0
    flips = int(((p32 > 0.5) != (p16 > 0.5))[counted].sum().item())
    res = {"windows": n, "windows_counted": int(counted.sum().item()), "state_agreement_all_timestamps": agree,
           "vote_flips": flips, "max_abs_prob_diff": float((p32 - p16).abs().max().item())}
    ens32.close()
    return res


def run_train(name, wl, steps, warmup, R, args, min_seconds=0.0):
    """configs[3]: one step = forward + backward + AdamW of every ensemble member on this rank's batch;
    with N > 1 each member's flat gradient bucket is all-reduced (mean) over RCCL before its update."""
    import torch
    from camal_amd import synth
    from camal_amd.resnet import CamALResNet
    from camal_amd.trainer import TrainEngine
    B, L, ks = wl["B"], wl["L"], wl["kernels"]
    dev = R.dev
    x_host, y_host = synth.ideal_windows(B, L, seed=2000 + R.rank)
    x = torch.from_numpy(x_host).to(dev)
    y = torch.from_numpy(y_host).to(dev)
    engines = []
    for k in ks:
        torch.manual_seed(k)  # same initial weights on every rank (and TrainEngine broadcasts rank 0's anyway)
        engines.append(TrainEngine(CamALResNet(kernel_size=k), dev, B, L))

    from camal_amd.trainer import MemberStreams
    members = MemberStreams(engines)  # one HIP stream per member: independent members overlap on the device

    def step():
        members.step(x, y, 1e-3, 1e-2)

    steps, _ = warm_up(step, steps, warmup, R, min_seconds)
    R.sync()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    R.sync()
    dt = R.max(time.perf_counter() - t0)
    res = None
    if R.rank == 0:
        # forward conv MACs x3 (forward, data gradient, weight gradient); the stem has no data gradient
        flops_step = 3 * 2.0 * L * B * sum(conv_macs_per_timestep(k) for k in ks)
        achieved = flops_step * steps / dt / 1e12
        res = {
            "metric": "windows/sec (train: forward+backward+AdamW, all members)",
            "value": R.world * B * steps / dt, "unit": "windows/s", "steps": steps, "warmup": warmup,
            "ms_per_step": dt / steps * 1e3, "dtype": "f32",
            "config": {"workload": f"{name} = {wl['config']}: E={len(ks)} k={ks} B={B}/GPU L={L} fwd+bwd+AdamW per member",
                       "windows_per_gpu_per_step": B, "window_len": L, "kernel_sizes": ks,
                       "inputs": "resident in HBM",
                       "parallelism": f"data-parallel x{R.world}, one all-reduce of each member's gradient bucket"
                                      + (" over RCCL" if R.world > 1 else " (no-op at N = 1)")},
            "roofline": {"bound": "mfma", "achieved": achieved, "peak": PEAK_TFLOPS["fp32"], "unit": "TFLOP/s",
                         "frac": achieved / PEAK_TFLOPS["fp32"], "traffic": None,
                         "kernel": "whole step (wall clock): conv forward + dgrad + wgrad MFMA kernels, BN/loss/AdamW",
                         "alg_flops_per_step": flops_step},
            "sanity": {"loss": [float(e.loss.item()) for e in engines]},
        }
    for e in engines:
        e.close()
    del engines
    torch.cuda.empty_cache()
    if res is not None and R.world == 1 and not args.no_extras:
        res["host_to_host"] = train_host_to_host(wl, dev)
    return res


def train_host_to_host(wl, dev, k=7, n_batches=48):
    """End to end as CamAL.train's callers see it, for ONE member (the reference trains them one after the other,
    core.py:104-158): NumPy windows and labels in host memory -> BasedClassifTrainer.train(1), i.e. an epoch of
    shuffled batches with forward + backward + AdamW, the epoch's loss / accuracy read back, the best-state snapshot.
    Beside it the same member's step on HBM-resident tensors.  Not `value`."""
    import numpy as np
    import torch
    from camal_amd import synth
    from camal_amd.resnet import CamALResNet
    from camal_amd.trainer import BasedClassifTrainer, TrainEngine, TSLoader
    B, L = wl["B"], wl["L"]
    x, y = synth.ideal_windows(B * n_batches, L, seed=77)
    torch.manual_seed(k)
    tr = BasedClassifTrainer(CamALResNet(kernel_size=k), TSLoader(x, y, batch_size=B, shuffle=True), device=dev,
                             learning_rate=1e-3, weight_decay=0)
    tr.train(1)  # warm-up epoch: the training set moves into HBM here
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    tr.train(2)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 2
    eng = tr.engine
    xd, yd = torch.from_numpy(x[:B]).to(dev), torch.from_numpy(y[:B]).to(dev)
    for _ in range(5):
        eng.forward_backward(xd, yd)
        eng.optimizer_step(1e-3, 0.0)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n_batches):
        eng.forward_backward(xd, yd)
        eng.optimizer_step(1e-3, 0.0)
    torch.cuda.synchronize()
    dr = time.perf_counter() - t0
    eng.close()
    return {"value": B * n_batches / dt, "unit": "windows/s", "member_kernel_size": k, "windows_per_epoch": B * n_batches,
            "seconds_per_epoch": dt, "resident_step_windows_per_s": B * n_batches / dr, "frac_of_resident": dr / dt,
            "what": "BasedClassifTrainer.train on NumPy arrays (one member, no validation loader): epoch wall clock incl. "
                    "shuffle, per-epoch read-back and state snapshot, against the same steps on resident tensors"}


def cpu_baseline(gpu_sds=None, gpu_x=None):
    """the oracle's torch-CPU restatement of the reference's predict loop (core.py:286-332, one
    batch-1 forward per member and window) on configs[0] exactly — E=5 (k=5,7,9,15,25), 64 x 1024 UK-DALE-shaped
    windows, seed 0 — on this host's cores: one warm-up, then the median of 3 runs.  Second figure: the same
    algorithm with a batched forward per member."""
    import numpy as np
    import torch
    from camal_amd import synth, weights
    from oracle import torch_path
    # the GPU box gives this job a share of the machine's cores (OMP_NUM_THREADS there); os.cpu_count() is the
    # whole machine
    avail = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    threads = min(os.cpu_count() or 1, avail, int(os.environ.get("OMP_NUM_THREADS", "0")) or avail)
    torch.set_num_threads(threads)
    ks = [5, 7, 9, 15, 25]
    x = synth.ukdale_windows(64, 1024, seed=0)
    sds = [weights.make_state_dict(k, seed=0, head_gain=1.0) for k in ks]

    def med3(fn):
        ts = []
        for _ in range(3):
            t0 = time.perf_counter()
            fn()
            ts.append(time.perf_counter() - t0)
        return float(np.median(ts)), ts

    torch_path.predict_loop(sds, x[:4])  # warm-up
    t_loop, all_loop = med3(lambda: torch_path.predict_loop(sds, x))
    torch_path.predict_batched(sds, x[:8])
    t_bat, all_bat = med3(lambda: torch_path.predict_batched(sds, x))
    model = ""
    try:
        with open("/proc/cpuinfo") as f:
            model = next((ln.split(":", 1)[1].strip() for ln in f if ln.startswith("model name")), "")
    except OSError:
        pass
    return {"value": 64 / t_loop, "unit": "windows/s", "cores": threads, "kind": "port",
            "sample": "configs[0] exactly: E=5 k=[5,7,9,15,25], 64 x 1024 UK-DALE-shaped windows seed 0, "
                      "oracle/torch_path.predict_loop (batch-1 forward per member and window, core.py:290-317), "
                      f"1 warm-up + median of 3 runs ({', '.join(f'{t:.2f}' for t in all_loop)} s)",
            "batched": {"value": 64 / t_bat, "unit": "windows/s",
                        "sample": "same windows and members, oracle/torch_path.predict_batched (one forward per member over "
                                  f"the 64 windows), median of 3 ({', '.join(f'{t:.2f}' for t in all_bat)} s)"},
            "cpu_model": model, "host_cpus": os.cpu_count()}


# ---------------------------------------------------------------------------------------------------------------
def spawn_ranks(args):
    """--gpus N without a launcher: start the N ranks as children (nothing here has touched a GPU), pass rank 0's
    stdout through, exit with the worst return code."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    for r in range(args.gpus):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(args.gpus), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=None if r == 0 else sys.stderr))
    rc = 0
    alive = list(procs)
    while alive:
        time.sleep(0.2)
        for p in list(alive):
            code = p.poll()
            if code is None:
                continue
            alive.remove(p)
            if code != 0 and rc == 0:
                rc = code
                for q in alive:  # a rank died: the others would wait at a barrier for ever
                    q.terminate()
    sys.exit(rc)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--workload", default=None, choices=sorted(WORKLOADS),
                    help="headline workload (default resnet_fp32 = configs[1]); naming one turns `--also auto` off")
    ap.add_argument("--also", default="auto", choices=["auto", "all", "none"],
                    help="time the other workloads too (auto: only when --workload is the default headline)")
    ap.add_argument("--also-seconds", type=float, default=2.0, help="minimum timed region of each `also` workload")
    ap.add_argument("--batch", type=int, default=0, help="windows per GPU per step (default: workload's)")
    ap.add_argument("--chunk", type=int, default=0, help="windows per internal pass")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true",
                    help="only the timed steps (profiling runs): no bf16-vs-fp32 check, host-to-host pass or HBM probe")
    ap.add_argument("--deadline", type=float, default=900.0, help="seconds after which rank 0 prints what it has")
    ap.add_argument("--dry-run", action="store_true",
                    help="CPU rehearsal of the launch path: ranks, gloo process group, barrier, MAX; no GPU work")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        spawn_ranks(args)
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")

    # stdout carries ONE JSON line and nothing else: libraries that print there (RCCL's version banner at communicator
    # creation) are sent to stderr for the whole run; the line itself goes to the real stdout
    sys.stdout.flush()
    real_stdout = os.fdopen(os.dup(1), "w")
    os.dup2(2, 1)

    import torch
    dist = None
    # CAMAL_BENCH_REHEARSAL=1 (tests, a one-GPU box): every rank runs on cuda:0 and the process group is "gloo" (RCCL
    # refuses two ranks on one device), so that the WHOLE N > 1 flow of this file — ranks other than 0, barriers, MAX,
    # the data-parallel training step with its all-reduce — runs on real kernels before the first multi-GPU node sees it
    rehearsal = os.environ.get("CAMAL_BENCH_REHEARSAL", "0") == "1"
    if rehearsal:
        local_rank = 0
    dev = f"cuda:{local_rank}"
    if not args.dry_run:
        torch.cuda.set_device(local_rank)
    # a launcher's environment (torch.distributed.run sets WORLD_SIZE and MASTER_*) gets its process group even for one
    # rank: barrier and MAX then run through RCCL exactly as they do for N > 1
    if world > 1 or ("WORLD_SIZE" in os.environ and "MASTER_PORT" in os.environ):
        import datetime
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.dry_run:
            dist.init_process_group("gloo", timeout=datetime.timedelta(seconds=120))
        elif rehearsal:
            dist.init_process_group("gloo", timeout=datetime.timedelta(seconds=300))
        else:
            dist.init_process_group("nccl", device_id=torch.device(dev), timeout=datetime.timedelta(seconds=300))
    if args.dry_run:
        t = torch.tensor([float(rank)], dtype=torch.float64)
        if dist is not None:
            dist.barrier()
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dist.destroy_process_group()
        if rank == 0:
            print(json.dumps({"dry_run": True, "n_gpus": world, "max_rank_seen": int(t.item())}), file=real_stdout, flush=True)
        return

    R = Ranks(rank, world, dev, dist)
    explicit = args.workload is not None
    args.workload = args.workload or "resnet_fp32"
    wl = dict(WORKLOADS[args.workload])
    if args.batch:
        wl["B"] = args.batch
    also = [] if args.also == "none" or (args.also == "auto" and explicit) else \
        [n for n in ORDER if n != args.workload]

    line = {}
    printed = threading.Lock()

    def emit():
        if rank == 0 and line and printed.acquire(blocking=False):
            print(json.dumps(line), file=real_stdout, flush=True)

    def on_deadline():  # a hung collective or kernel must not cost the driver the line rank 0 already has
        for n in also:
            line.setdefault("also", {}).setdefault(n, {"error": f"not finished within --deadline {args.deadline:.0f} s"})
        emit()
        os._exit(0 if rank != 0 or line else 3)

    dog = threading.Timer(args.deadline, on_deadline)
    dog.daemon = True
    dog.start()

    runner = run_train if args.workload == "train_fp32" else run_inference
    head = runner(args.workload, wl, args.steps, args.warmup, R, args)
    if rank == 0:
        line.update({"metric": head.pop("metric", "windows/sec (detect+CAM localize)"), "value": head.pop("value"),
                     "unit": head.pop("unit"), "n_gpus": world, "steps": head.pop("steps"), "warmup": head.pop("warmup"),
                     "ms_per_step": head.pop("ms_per_step"), "higher_is_better": True, "scaling": "weak",
                     "vs_baseline": None, "dtype": head.pop("dtype"), "data": "synthetic",
                     "process_group": dist.get_backend() if dist is not None else None})
        line.update(head)
    for n in also:
        r = None
        try:
            fn = run_train if n == "train_fp32" else run_inference
            r = fn(n, dict(WORKLOADS[n]), args.steps, min(args.warmup, 3), R, args, min_seconds=args.also_seconds)
        except Exception as e:  # noqa: BLE001 — one workload failing must not cost the others their figures
            if world > 1:
                # the other ranks are inside collectives: the job fails, but rank 0 first prints the line it has (the
                # headline is complete at this point; the watchdog is a daemon thread and would die with the exception)
                line.setdefault("also", {})[n] = {"error": f"{type(e).__name__}: {e}"}
                dog.cancel()
                emit()
                raise
            r = {"error": f"{type(e).__name__}: {e}"}
        if rank == 0:
            line.setdefault("also", {})[n] = r
    if rank == 0 and not args.no_cpu_baseline and world == 1:  # N = 1 only: the other ranks would idle at the barrier
        line["cpu_baseline"] = cpu_baseline()
    dog.cancel()
    emit()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
