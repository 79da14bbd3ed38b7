#!/usr/bin/env python3
"""Per-K-step latency of ONE workgroup of the bf16 training GEMM (M = 128, N = 64, K = 16384 tokens, no split-K): weight-gradient
layout (both operands row-contiguous) against forward layout (both k-contiguous)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from studiosr_amd import autograd as AG  # noqa: E402
from bgemm_bench import timeit  # noqa: E402

dev = torch.device("cuda")
K = 16384
for (M, N) in [(128, 64), (512, 192), (540, 180)]:
    Y, X = torch.randn(K, M, device=dev), torch.randn(K, N, device=dev)
    dW = torch.zeros(M, N, device=dev)
    A2, B2 = torch.randn(M, K, device=dev), torch.randn(N, K, device=dev)
    with AG.autocast_state(True):
        us = timeit(lambda: AG.bgemm(Y, X, dW, M, N, K, (1, M), (N, 1), (N, 1)))
        print(f"wgrad layout  M={M} N={N} K={K} ksplit=1: {us:8.1f} us = {us / (K / 32) * 1e3:6.0f} ns / K step")
        us = timeit(lambda: AG.bgemm(A2, B2, dW, M, N, K, (K, 1), (1, K), (N, 1)))
        print(f"fwd layout    M={M} N={N} K={K} ksplit=1: {us:8.1f} us = {us / (K / 32) * 1e3:6.0f} ns / K step")
