"""post-processing kernel alone, on the shapes bench.py times: GB/s of algorithmic bytes (HIP events of the `post`
class).  python tools/post_bench.py  (env CAMAL_POST_STREAM=0: the tiled kernel; CAMAL_POST_STREAM_BLOCKS=n)"""
import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from camal_amd import synth, weights
from camal_amd.calibrate import calibrated_state_dicts
from camal_amd.engine import Ensemble

def run(name, ks, B, L, mode, precision):
    xh = synth.refit_windows(B, L, seed=1) if L != 1024 else synth.ukdale_windows(B, L, seed=1)
    x = torch.from_numpy(xh).to("cuda:0")
    if mode == "mixed":
        sds = calibrated_state_dicts(ks, seed=0, sample_x=xh[:64], device="cuda:0", precision="fp32")
    else:
        hb = (10.0, -10.0) if mode == "all_off" else (-10.0, 10.0)
        sds = [weights.make_state_dict(k, seed=0, head_gain=(0.0 if mode == "all_off" else 1.0), head_bias=hb) for k in ks]
    ens = Ensemble(sds, device="cuda:0", precision=precision)
    out = {"prob": torch.empty(B, dtype=torch.float64, device="cuda:0"), "state": torch.empty(B, L, device="cuda:0"), "power": torch.empty(B, L, device="cuda:0")}
    for _ in range(2): ens.localize(x, p_on=2000.0, scale_mul=1000.0, out=out)
    torch.cuda.synchronize(); ens.set_profiling(True)
    for _ in range(6): ens.localize(x, p_on=2000.0, scale_mul=1000.0, out=out)
    torch.cuda.synchronize(); p = ens.get_profile()
    n, ms = p["post"]
    frac_on = float((out["prob"] > 0.5).double().mean())
    byt = 4.0 * B * L * (len(ks) * frac_on + 3)
    print(f"{name:28s} votes on {frac_on:.2f}  post {1e3 * ms / n:6.1f} us/launch  {byt / (ms / n * 1e-3) / 1e9:7.0f} GB/s", flush=True)
    chk = (float(out["state"].double().sum()), float(out["power"].double().sum()))
    ens.close()
    return chk

if __name__ == "__main__":
    r = []
    r.append(run("E1 4096x1024 fp32 mixed", [7], 4096, 1024, "mixed", "fp32"))
    r.append(run("E5 4096x1440 bf16 mixed", [5, 7, 9, 15, 25], 4096, 1440, "mixed", "bf16"))
    r.append(run("E5 4096x1440 bf16 all_on", [5, 7, 9, 15, 25], 4096, 1440, "all_on", "bf16"))
    r.append(run("E5 4096x1440 bf16 all_off", [5, 7, 9, 15, 25], 4096, 1440, "all_off", "bf16"))
    r.append(run("E5 512x10080 bf16 mixed", [5, 7, 9, 15, 25], 512, 10080, "mixed", "bf16"))
    print("checksums", r)
