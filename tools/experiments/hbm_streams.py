#!/usr/bin/env python3
"""What plain streaming kernels reach on this GPU for the traffic mix of a Bottleneck tail (two fp16 reads + one write) and
for a pure write, as a yardstick for the HBM-bound conv launches."""
import torch

dev = "cuda:0"
n = 4096000 * 512          # elements of the 128->512 16x16 tail at 16000 images
a = torch.randn(n // 8, device=dev).half().repeat(8)
b = torch.randn(n // 8, device=dev).half().repeat(8)
out = torch.empty_like(a)


def timeit(fn, nbytes, name, iters=10):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / iters
    print(f"{name:28s}: {ms * 1e3:8.1f} us  {nbytes / ms / 1e6:6.0f} GB/s", flush=True)


timeit(lambda: torch.add(a, b, out=out), 3 * n * 2, "out = a + b (2R + 1W)")
timeit(lambda: out.copy_(a), 2 * n * 2, "copy (1R + 1W)")
timeit(lambda: out.fill_(1.0), n * 2, "fill (1W)")
timeit(lambda: torch.relu_(out), 2 * n * 2, "relu_ in place (1R + 1W)")
timeit(lambda: a.sum(), n * 2, "sum (1R)")
