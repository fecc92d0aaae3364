"""Streaming-kernel bandwidth of this box at the post-processing kernel's sizes and read/write mixes
(tools/hbm_probe.hip).  python tools/hbm_probe.py -> one JSON line; bench.py calls probe() for its denominators."""
import ctypes as C
import json
import os

HERE = os.path.dirname(os.path.abspath(__file__))
LIB = os.path.join(HERE, "libhbm_probe.so")


def probe(n_floats, nr, nw, device="cuda:0", blocks=(512, 1024, 2048, 4096, 8192, 16384), reps=20, unrolls=(1, 2, 4, 8), detail=None):
    """Best GB/s over a few grid sizes and loads-in-flight per lane (unrolls) for nr read + nw write streams of n_floats
    floats each; None if the probe library is not built.  detail: a dict that receives {(unroll, blocks): GB/s}."""
    if not os.path.exists(LIB):
        return None
    import torch
    lib = C.CDLL(LIB)
    lib.hbm_probe.restype = C.c_double
    lib.hbm_probe_u.restype = C.c_double
    lib.hbm_probe_u.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]
    n = int(n_floats) // 4 * 4
    src = torch.rand(nr * n, dtype=torch.float32, device=device)
    dst = torch.empty(nw * n, dtype=torch.float32, device=device)
    best = 0.0
    for u in unrolls:
        for g in blocks:
            t = lib.hbm_probe_u(src.data_ptr(), dst.data_ptr(), n, nr, nw, g, reps, u, C.c_void_p(torch.cuda.current_stream().cuda_stream))
            if t > 0:
                gbs = 4.0 * n * (nr + nw) / t / 1e9
                best = max(best, gbs)
                if detail is not None:
                    detail[f"u{u}_g{g}"] = round(gbs, 1)
    return best or None


if __name__ == "__main__":
    out = {}
    for name, n, nr, nw in [("copy_1GiB", 1 << 28, 1, 1), ("post_E1_4096x1024 (2r:2w)", 4096 * 1024, 2, 2),
                            ("post_E5_4096x1440 (4r:2w ~ 55% voted on)", 4096 * 1440, 4, 2),
                            ("post_E5_all_on (6r:2w)", 4096 * 1440, 6, 2), ("post_all_off (1r:2w)", 4096 * 1440, 1, 2),
                            ("read_heavy (6r:1w)", 4096 * 1440, 6, 1)]:
        d = {}
        out[name] = probe(n, nr, nw, detail=d)
        out[name + " detail"] = d
    print(json.dumps(out))
