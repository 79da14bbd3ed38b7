#!/usr/bin/env python3
"""sr_bgemm at the GEMM shapes of one HAT x4 training step (batch 4, 64x64 LR: 16,384 tokens): forward, data gradient and weight
gradient of the Linear layers and of the im2col'd 3x3 convs, bf16 operands (autocast) and exact fp32.
Prints us, TFLOP/s and the HBM-side GB/s of the operands + result (each counted once)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from studiosr_amd import autograd as AG  # noqa: E402


def timeit(fn, iters=20, warm=3):
    for _ in range(warm):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


def main():
    dev = torch.device("cuda")
    T = 16384
    cases = [("qkv", T, 540, 180), ("proj", T, 180, 180), ("fc1", T, 360, 180), ("fc2", T, 180, 360), ("cab1 (im2col)", T, 60, 1620), ("cab2 (im2col)", T, 180, 540),
             ("conv 180 (im2col)", T, 180, 1620)]
    for bf in (True, False):
        print("bf16 operands" if bf else "fp32")
        for name, M, N, K in cases:
            X, W, Y = torch.randn(M, K, device=dev), torch.randn(N, K, device=dev), torch.randn(M, N, device=dev)
            dX, dW = torch.empty_like(X), torch.zeros_like(W)
            ks = int(os.environ["KS"]) if os.environ.get("KS") else AG._ksplit(N, K, M)
            with AG.autocast_state(bf):
                f = {"fwd": lambda: AG.bgemm(X, W, Y, M, N, K, (K, 1), (1, K), (N, 1)),
                     "dgrad": lambda: AG.bgemm(Y, W, dX, M, K, N, (N, 1), (K, 1), (K, 1)),
                     "wgrad": lambda: (dW.zero_() if ks > 1 else None, AG.bgemm(Y, X, dW, N, K, M, (1, N), (K, 1), (K, 1), ksplit=ks))}
                for kind, fn in f.items():
                    if os.environ.get("KS") and kind != "wgrad":
                        continue
                    us = timeit(fn)
                    nbytes = 4 * (M * K + N * K + M * N)
                    print(f"  {name:18s} {kind:5s} M={M} N={N} K={K}: {us:7.1f} us  {2 * M * N * K / us / 1e6:7.1f} TF/s  {nbytes / us / 1e3:7.0f} GB/s", flush=True)


if __name__ == "__main__":
    main()
