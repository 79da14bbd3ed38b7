#!/usr/bin/env python3
"""BASELINE config 4: SwinIR x4 on ONE large LR image, row-strip sharded with per-layer halo exchange.

  single process, all strips on one GPU (checks the sharded result against the unsharded forward):
      python tools/strip_bench.py --size 2048 --strips 8 --check
  one process per GPU (RCCL point-to-point halos):
      python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 tools/strip_bench.py --size 2048

Prints one JSON line (rank 0): HR megapixels / second of the whole image.
"""
import argparse
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--size", type=int, default=2048)
    ap.add_argument("--strips", type=int, default=8, help="strips in this process when not launched by torch.distributed.run")
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--precision", default="bf16")
    ap.add_argument("--check", action="store_true", help="compare with the unsharded forward (single process only)")
    args = ap.parse_args()

    import torch.distributed as dist

    import studiosr_amd as S
    from studiosr_amd.strips import DistStripComm, LocalStripComm

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=dev)
        comm = DistStripComm()
    else:
        comm = LocalStripComm(args.strips)

    torch.manual_seed(0)
    model = S.SwinIR(scale=4).eval().to(dev).set_precision(args.precision)
    x = torch.rand(1, 3, args.size, args.size, generator=torch.Generator().manual_seed(0)).to(dev)

    with torch.no_grad():
        for _ in range(args.warmup):
            out = model.forward_strips(x, comm)
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            out = model.forward_strips(x, comm)
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / args.steps
        res = {"workload": f"SwinIR x4, one {args.size}x{args.size} LR image, {comm.world} row strips", "n_gpus": world,
               "strips": comm.world, "precision": args.precision, "s_per_image": round(dt, 4),
               "hr_mpix_per_s": round((args.size * 4) ** 2 / 1e6 / dt, 2), "includes": "halo exchange + final gather of the HR strips"}
        if args.check and world == 1:
            ref = model(x)
            res["equals_unsharded"] = bool(torch.equal(out, ref))
            res["max_abs_diff"] = float((out - ref).abs().max())
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            model(x)
            torch.cuda.synchronize()
            res["unsharded_s_per_image"] = round(time.perf_counter() - t0, 4)
        res["peak_mem_gib"] = round(torch.cuda.max_memory_allocated() / 2**30, 2)
    if rank == 0:
        print(json.dumps(res))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
