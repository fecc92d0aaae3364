import sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch
from camal_amd import synth, weights
from camal_amd.engine import Ensemble
B, L = 4096, 1440
x = torch.from_numpy(synth.refit_windows(B, L, seed=1)).to("cuda:0")
for name, hb in (("all_off", (10.0, -10.0)), ("all_on", (-10.0, 10.0))):
    sds = [weights.make_state_dict(k, seed=0, head_gain=(0.0 if name == "all_off" else 1.0), head_bias=hb) for k in (5, 7, 9, 15, 25)]
    ens = Ensemble(sds, device="cuda:0", precision="bf16")
    out = {"prob": torch.empty(B, dtype=torch.float64, device="cuda:0"), "state": torch.empty(B, L, device="cuda:0"), "power": torch.empty(B, L, device="cuda:0")}
    for _ in range(2): ens.localize(x, p_on=2000.0, scale_mul=1000.0, out=out)
    torch.cuda.synchronize(); ens.set_profiling(True)
    for _ in range(5): ens.localize(x, p_on=2000.0, scale_mul=1000.0, out=out)
    torch.cuda.synchronize(); p = ens.get_profile()
    n, ms = p["post"]
    frac_on = float((out["prob"] > 0.5).double().mean())
    byt = 4.0 * B * L * (5 * frac_on + 3)
    print(name, "votes on", frac_on, "post us/launch", 1e3 * ms / n, "GB/s", byt / (ms / n * 1e-3) / 1e9, flush=True)
    ens.close()
