"""Experiment (round 3): do CU-masked streams let the HBM-bound kernels of the inference step run beside the
matrix kernels?  Measures, on 8192 patches:
  a. the LocalStage eval forward on an ordinary stream and on streams masked to 248 / 240 / 224 CUs;
  b. an HBM-bound kernel (max-pool over conv1's 0.9 GB map) on an ordinary stream and on streams masked to 8 / 16 / 32 CUs;
  c. both at once: the forward on the big partition, the pool loop on the small one.
usage (GPU box): python tools/exp_cu_mask.py
Result (profiles/r03_cu_mask_experiment.md): not a route - a compute unit moves ~27 GB/s with these kernels, so 32 CUs reach
0.9 TB/s where the whole chip reaches 4.9: the HBM-bound kernels need every CU just as the matrix kernels do.
"""
import ctypes as C
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "blurry-edges_amd"))
sys.path.insert(0, ROOT)
import models  # noqa: E402
from be_hip import native, synth  # noqa: E402

_hip = C.CDLL("libamdhip64.so")         # the runtime torch already loaded (same soname)

DEV = torch.device("cuda:0")
NCU = torch.cuda.get_device_properties(0).multi_processor_count


def masked_stream(lo, hi):
    """stream on CUs [lo, hi) in the driver's bit order"""
    words = (NCU + 31) // 32
    m = (C.c_uint32 * words)()
    for i in range(lo, hi):
        m[i // 32] |= 1 << (i % 32)
    out = C.c_void_p()
    rc = _hip.hipExtStreamCreateWithCUMask(C.byref(out), C.c_uint32(words), m)
    assert rc == 0, f"hipExtStreamCreateWithCUMask: {rc}"
    return torch.cuda.ExternalStream(out.value, device=DEV)


def timed(stream, fn, iters=20, warm=3):
    with torch.cuda.stream(stream):
        for _ in range(warm):
            fn()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(stream)
        for _ in range(iters):
            fn()
        b.record(stream)
    b.synchronize()
    return a.elapsed_time(b) / iters


def main():
    print("CUs", NCU)
    model = models.LocalStage().to(DEV)
    model.load_state_dict({k: torch.from_numpy(v) for k, v in synth.local_stage_state_dict().items()})
    model.eval()
    model.streams = 1
    x = torch.from_numpy(synth.uniform_patches(8192, name="cu_mask")).to(DEV)
    big = torch.randn(8192, 21, 21, 64, device=DEV)
    pool_bytes = big.numel() * 4 + 8192 * 11 * 11 * 64 * 4
    fwd = lambda: model(x)
    pool = lambda: native.maxpool_nhwc(big, 3, 2, 1)
    with torch.no_grad():
        ref = model(x).clone()
        cur = torch.cuda.current_stream()
        print("a. forward, ordinary stream: %.3f ms" % timed(cur, fwd))
        for n in (248, 240, 232, 224):
            s = masked_stream(0, n)
            t = timed(s, fwd)
            with torch.cuda.stream(s):
                same = torch.equal(model(x), ref)
            s.synchronize()
            print("a. forward on %d CUs: %.3f ms  bit-identical %s" % (n, t, same))
        t = timed(cur, pool, iters=50)
        print("b. pool, ordinary stream: %.3f ms  %.2f TB/s" % (t, pool_bytes / t / 1e9))
        for n in (8, 16, 24, 32):
            s = masked_stream(NCU - n, NCU)
            t = timed(s, pool, iters=50)
            print("b. pool on the last %d CUs: %.3f ms  %.2f TB/s" % (n, t, pool_bytes / t / 1e9))
        for n in (8, 16, 24, 32):
            sg, st = masked_stream(0, NCU - n), masked_stream(NCU - n, NCU)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            ev = []
            for s_, f, it in ((sg, fwd, 20), (st, pool, 200)):
                with torch.cuda.stream(s_):
                    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    a.record(s_)
                    for _ in range(it):
                        f()
                    b.record(s_)
                    ev.append((a, b, it))
            torch.cuda.synchronize()
            tf = ev[0][0].elapsed_time(ev[0][1]) / ev[0][2]
            tp = ev[1][0].elapsed_time(ev[1][1]) / ev[1][2]
            print("c. together, %d + %d CUs: forward %.3f ms, pool %.3f ms (%.2f TB/s) [pool loop ran %.0f ms, forward loop %.0f ms]"
                  % (NCU - n, n, tf, tp, pool_bytes / tp / 1e9, tp * ev[1][2], tf * ev[0][2]))


if __name__ == "__main__":
    main()
