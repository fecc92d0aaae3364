"""Register-only MFMA loop rate of this box (tools/mfma_probe.hip): the measured matrix-pipe ceiling beside the guide's
peak figures.  python tools/mfma_probe.py -> one JSON line; bench.py calls probe() for `roofline.mfma_probe_tflops`."""
import ctypes as C
import json
import os

HERE = os.path.dirname(os.path.abspath(__file__))
LIB = os.path.join(HERE, "libmfma_probe.so")


def probe(kind, device="cuda:0", waves=(4, 8), blocks=(256, 512, 1024), launch_ms=(2.0, 20.0), detail=None):
    """Best TFLOP/s of the MFMA-only loop over a few launch shapes; kind "bf16" (16x16x32), "fp32" (32x32x2) or "bf16_random" (bf16 on
    random operands, a different pair per accumulator tile: the toggle rate of real data).
    launch_ms: target launch lengths (a short one shows the rate before the clock settles, a long one the sustained
    rate); None if the probe library is not built.  detail: a dict that receives every shape's figure."""
    if not os.path.exists(LIB):
        return None
    import torch
    lib = C.CDLL(LIB)
    lib.mfma_probe.restype = C.c_double
    lib.mfma_probe.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.POINTER(C.c_double), C.c_void_p]
    k = {"bf16": 0, "fp32": 1, "bf16_random": 2, "bf16_const_own_regs": 3}[kind]
    scratch = torch.zeros(16, dtype=torch.float32, device=device)
    stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    cyc = 8 * 16 if k != 1 else 4 * 64   # issue cycles of one loop iteration at full rate
    best = 0.0
    for ms in launch_ms:
        for w in waves:
            for g in blocks:
                per_simd = g * w / 1024.0                       # waves per SIMD over the launch (256 CUs x 4 SIMDs)
                iters = max(16, int(ms * 1e-3 * 2.4e9 / (cyc * max(per_simd, 1.0))))
                sec = C.c_double(0.0)
                t = lib.mfma_probe(k, g, w, iters, 5, scratch.data_ptr(), C.byref(sec), stream)
                if t > 0:
                    best = max(best, t)
                    if detail is not None:
                        detail[f"{ms:g}ms_w{w}_g{g}"] = {"tflops": round(t, 1), "launch_ms": round(sec.value * 1e3, 3)}
    return best or None


if __name__ == "__main__":
    out = {}
    for kind in ("bf16", "bf16_const_own_regs", "bf16_random", "fp32"):
        d = {}
        out[kind] = probe(kind, detail=d)
        out[kind + " detail"] = d
    print(json.dumps(out))
