#!/usr/bin/env python3
"""Forward throughput of every model family at the BASELINE.json single-GPU shapes (default architectures, bf16 operands,
synthetic weights / inputs, HIP-graph replay): EDSR x4 b16 (config 2), SwinIR x4 b8 (config 3), RCAN x4 b16, HAT x4 b4.
Prints one JSON line per model: ms per forward, HR-Mpix/s, achieved TFLOP/s against the 2.5 PFLOP/s bf16 MFMA peak.
`KIND:B` overrides the batch, `KIND:B:N` also keeps N independent batches in flight (own graph / stream / workspace each, as
bench.py does): what a serving loop gets out of the launch-latency-bound models."""
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import studiosr_amd as S  # noqa: E402
from studiosr_amd.runtime import GraphedForward  # noqa: E402

GFLOP_PER_TILE = {"EDSR": 411.67, "SwinIR": 135.56, "RCAN": 130.40, "HAT": 207.76,  # BASELINE.md section 2 (64x64 LR tile, x4)
                  # the reference's other shipped SwinIR (from_pretrained(light=True), swinir.py:418-427: embed 60, 4 x 6 blocks, pixelshuffledirect), 72 x 72
                  # padded tokens: per token 24 x (qkv 21,600 + attention 15,360 + proj 7,200 + mlp 28,800) + convs 379,080 = 2.13 MFLOP
                  "SwinIR-light": 11.04}
CASES = {"EDSR": 16, "SwinIR": 8, "RCAN": 16, "HAT": 4, "SwinIR-light": 8}


def build(kind: str):
    if kind == "SwinIR-light":
        return S.SwinIR(scale=4, embed_dim=60, depths=[6, 6, 6, 6], num_heads=[6, 6, 6, 6], upsampler="pixelshuffledirect")
    return getattr(S, kind)(scale=4)


def main():
    which = sys.argv[1:] or list(CASES)
    dev = torch.device("cuda")
    for kind in which:
        parts = kind.split(":")  # "HAT:16" overrides the batch size, "HAT:4:2" also runs 2 batches in flight
        kind = parts[0]
        B = int(parts[1]) if len(parts) > 1 and parts[1] else CASES[kind]
        inflight = int(parts[2]) if len(parts) > 2 else 1
        torch.manual_seed(0)
        m = build(kind).eval().to(dev).set_precision(os.environ.get("MB_PREC", "bf16"))  # MB_PREC=fp32x3: what inference() runs
        x = torch.rand(B, 3, 64, 64, device=dev)
        with torch.no_grad():
            pipes = []
            for _ in range(inflight):
                ws_i = S.runtime.Workspace(dev)

                def fwd(inp, ws_i=ws_i):
                    m._ws = ws_i
                    return m(inp)

                st = torch.cuda.Stream()
                with torch.cuda.stream(st):
                    pipes.append((GraphedForward(fwd, x), st, ws_i))
            torch.cuda.synchronize()

            def run(n):
                for i in range(n):
                    g, st, _ = pipes[i % len(pipes)]
                    with torch.cuda.stream(st):
                        g.replay()

            run(3 * inflight)
            torch.cuda.synchronize()
            n = 20 * inflight
            t0 = time.perf_counter()
            run(n)
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) / n
        tf = B * GFLOP_PER_TILE[kind] / dt / 1e3
        print(json.dumps({"model": f"{kind} x4", "precision": os.environ.get("MB_PREC", "bf16"), "batch": B, "in_flight": inflight, "ms": round(dt * 1e3, 3), "hr_mpix_per_s": round(B * 256 * 256 / 1e6 / dt, 1),
                          "tflops": round(tf, 1), "frac_bf16_mfma_peak": round(tf / 2500.0, 4)}), flush=True)
        del pipes, m
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
