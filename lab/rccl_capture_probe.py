#!/usr/bin/env python3
"""Can this stack (PyTorch 2.10 + ROCm 7 + RCCL 2.26) capture a collective into a hipGraph?  ('capture the
collectives if this RCCL / torch allows - state the version-specific reason if not'.)  One rank, backend nccl: an all_reduce between
two kernels inside torch.cuda.graph, replayed; then the same with the collective issued on a side stream behind an event, the way
be_hip.dp.GradSync issues its buckets.  Prints what happened; exits 0 either way."""
import os
import socket
import sys
import time
import traceback

os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
import torch
import torch.distributed as dist


def main():
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1, device_id=dev)
    print("torch", torch.__version__, "rccl", torch.cuda.nccl.version(), flush=True)
    x = torch.ones(1 << 20, device=dev)
    dist.all_reduce(x)                                             # communicator up before any capture
    torch.cuda.synchronize()
    for mode in ("same_stream", "side_stream_event"):
        try:
            a = torch.full((1 << 20,), 2.0, device=dev)
            g = torch.cuda.CUDAGraph()
            side = torch.cuda.Stream()
            s = torch.cuda.Stream()
            s.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(s):
                with torch.cuda.graph(g, stream=s, capture_error_mode="thread_local"):
                    a.mul_(3.0)
                    if mode == "same_stream":
                        dist.all_reduce(a)
                    else:
                        ev = torch.cuda.Event()
                        ev.record()
                        side.wait_event(ev)
                        with torch.cuda.stream(side):
                            h = dist.all_reduce(a, async_op=True)
                        h.wait()                                   # the capturing stream waits for the collective
                        torch.cuda.current_stream().wait_stream(side)
                    a.add_(1.0)
            torch.cuda.synchronize()
            a.fill_(2.0)
            t0 = time.perf_counter()
            for _ in range(100):
                g.replay()
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) / 100 * 1e6
            # one replay from 2.0: (2 * 3) summed over one rank + 1 = 7; hundred replays: x -> 3 x + 1 each time
            a.fill_(2.0)
            g.replay()
            torch.cuda.synchronize()
            print(f"{mode}: captured and replayed, value {float(a[0])} (expected 7.0), {dt:.1f} us per replay", flush=True)
        except Exception as e:
            print(f"{mode}: FAILED {type(e).__name__}: {e}", flush=True)
            traceback.print_exc()
            try:
                torch.cuda.synchronize()
            except Exception:
                pass
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
    sys.exit(0)
