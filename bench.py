#!/usr/bin/env python3
"""Headline benchmark: HR megapixels / second of SwinIR x4 on 64x64 LR tiles (BASELINE.json).

    python bench.py [--gpus N] [--steps K] [--warmup W]

One "step" = one eval-mode forward of the default SwinIR x4 (embed 180, 6x6 blocks, window 8) over one
batch of 8 synthetic 64x64 LR tiles (BASELINE.json configs[2]), bf16 operands / fp32 accumulate, inputs
resident in HBM, whole forward replayed from one HIP graph.  With N > 1 (launched by torch.distributed.run,
one rank per GPU) every rank processes its own batch of 8 tiles -- tiles are independent, so there is no
data-path collective ("weak" scaling); the barrier + max-over-ranks timing uses RCCL.

The JSON line also carries
  roofline      dominant kernel (the one-launch Swin block kernel `sr_swin_block3_kernel`, C ABI sr_swin_block) vs the bf16 MFMA
                peak: median of 200 launches timed live with HIP events on the launch stream; `forward` = whole-forward fraction.
  cpu_baseline  the CPU oracle (oracle/, a PyTorch-fp32 restatement pinned to the reference) timed on this host: all the
                cores this process may use (count and CPU model stated) on one full step, plus a 1-thread leg on one tile.
  parity        PSNR of the HIP output against the oracle output on the sample, and the metric's
                "PSNR delta" against a fixed synthetic target.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

TILE = 64
BATCH = int(os.environ.get("SR_BENCH_BATCH", "8"))  # 8 = the metric's workload; the override is for experiments only
SCALE = 4
FLOP_PER_LR_PIXEL = 26_150_616  # 2*MAC over conv/Linear/QK^T/AV, default SwinIR x4
PADDED = 72  # eval-mode pad 64 -> 72 (swinir.py:249-255)
MFMA_BF16_PEAK_TFLOPS = 2500.0  # MI355X_MICROARCH.md: ~2.5 PF dense bf16
CPU_SAMPLE_TILES = 8  # one full step of the workload
CPU_SAMPLE_FORWARDS = 5


def build_model(device):
    import studiosr_amd as S

    torch.manual_seed(0)
    model = S.SwinIR(scale=SCALE).eval()  # reference defaults: C 180, depths 6x6, heads 6, ws 8, mlp 2.0
    return model.to(device).set_precision("bf16")


def time_dominant_kernel(model, x, groups: int = 40, per_group: int = 5):
    """Duration of ONE launch of the dominant kernel, `sr_swin_block3_kernel` (C ABI sr_swin_block) -- one whole Swin block
    (LayerNorm1 + QKV + shifted-window attention + proj + residual, LayerNorm2 + fc1 + GELU + fc2 + residual) at the
    bench shape: 648 windows = 41,472 tokens, 36 launches per forward -- timed with HIP events on the launch stream
    (torch's current stream is the stream the C-ABI call enqueues on): `groups` event pairs around `per_group` back-to-back
    launches each (200 launches), MEDIAN of the group averages.  Its ALGORITHMIC FLOPs per token:
    2*MAC of qkv (180->540) 194,400 + QK^T and AV (6 heads x 64 keys x 30) 46,080 + proj (180->180) 64,800
    + fc1 (180->360) 129,600 + fc2 (360->180) 129,600 = 564,480 FLOP.
    Returns (median ms, min ms, algorithmic FLOPs per launch, launches timed)."""
    from studiosr_amd.models import swinir as SW

    cdt = torch.bfloat16
    P = model._get_packed(cdt)
    ws_ = model._workspace(x.device)
    lp = P["layers"][0]
    geo, bp = lp["geo"], lp["blocks"][1]  # a shifted block (mask path included)
    B = x.shape[0]
    t = ws_.get("ta", (B, PADDED, PADDED, geo.Cp), torch.float32)
    o = ws_.get("tb", (B, PADDED, PADDED, geo.Cp), torch.float32)

    def launch():
        SW.run_swin_block(bp, geo, t, o, ws_, cdt, bp["shift"])

    for _ in range(10):
        launch()
    torch.cuda.synchronize()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(groups)]
    for e0, e1 in ev:
        e0.record()
        for _ in range(per_group):
            launch()
        e1.record()
    torch.cuda.synchronize()
    ms = sorted(e0.elapsed_time(e1) / per_group for e0, e1 in ev)
    flops = float(B * PADDED * PADDED) * 564_480.0
    return ms[len(ms) // 2], ms[0], flops, groups * per_group


# MFMAs the block kernel executes per 64-token window (padded shapes: 180 -> 192 channels, 30 -> 32 per head, 360 -> 384 hidden):
# QKV 864 + QK^T 96 + PV 96 + proj 288 + fc1 576 + fc2 576, each v_mfma_f32_16x16x32_bf16 = 16,384 FLOP
EXECUTED_FLOP_PER_WINDOW = 2496 * 16384.0


def profiled_clock_ghz():
    """Shader clock the chip holds under the dominant kernel, from the committed diagnostic run (profiles/*_block_kernel_clock.txt,
    tools/wgtrace_blk3.py on a -DSR_WGTRACE build: s_memtime cycles / s_memrealtime time of every workgroup, median).  The profile carries the hash of
    the kernel sources it was taken on (`# kernel_src_sha16 = ...`); a profile of other sources (or without the line) is stale and not quoted: None."""
    import glob
    import re

    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "*_block_kernel_clock.txt")))
    if not files:
        return None
    text = open(files[-1]).read()
    m0 = re.search(r"#\s*kernel_src_sha16\s*=\s*([0-9a-f]+)", text)
    if not m0 or m0.group(1) != dominant_kernel_src_sha16():
        return None
    m = re.search(r"clock_ghz_median\s*=\s*([0-9.]+)", text)
    return float(m.group(1)) if m else None


DOMINANT_KERNEL_SOURCES = ("studiosr_amd/csrc/sr_swin_block3.hip", "studiosr_amd/csrc/sr_swin_stream.h", "studiosr_amd/csrc/sr_common.h")


def dominant_kernel_src_sha16() -> str:
    """sha256[:16] over the sources of the dominant kernel (sr_swin_block3_kernel): a committed PMC profile is only quoted while it still
    describes this code (tools/profile_bench.sh writes the hash into the profile's first line)."""
    import hashlib

    h = hashlib.sha256()
    for f in DOMINANT_KERNEL_SOURCES:
        h.update(open(os.path.join(ROOT, f), "rb").read())
    return h.hexdigest()[:16]


def profiled_hbm_traffic():
    """HBM bytes per launch of the dominant kernel from the committed rocprofv3 PMC passes (profiles/*_bench_hbm_counters.txt,
    collected by tools/profile_bench.sh: FETCH_SIZE and WRITE_SIZE in separate runs, KiB per dispatch).  gfx950 correction
    (MI355X_MICROARCH.md, HBM): FETCH_SIZE reports half the bytes of wide coalesced reads -> doubled; WRITE_SIZE is exact.
    Returns None when no profile is present (the counters cannot be read from inside the process)."""
    import glob
    import re

    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "*_bench_hbm_counters.txt")))
    if not files:
        return None
    vals, key, prof_sha = {}, None, None
    for line in open(files[-1]):
        m0 = re.match(r"#\s*kernel_src_sha16\s*=\s*([0-9a-f]+)", line)
        if m0:
            prof_sha = m0.group(1)
        if line.startswith("("):
            key = ("swin_block3_kernel" in line or "swin_block_kernel<true>" in line) and ", 648)" in line
        elif key:
            m = re.match(r"\s+(FETCH_SIZE|WRITE_SIZE)\s+([0-9.]+)", line)
            if m:
                vals[m.group(1)] = float(m.group(2))
    if len(vals) != 2:
        return None
    cur = dominant_kernel_src_sha16()
    stale = prof_sha != cur  # (a profile without the hash line predates round 4: stale by definition)
    return dict(bytes=None if stale else int((2.0 * vals["FETCH_SIZE"] + vals["WRITE_SIZE"]) * 1024), source=os.path.relpath(files[-1], ROOT),
                fetch_kib_raw=vals["FETCH_SIZE"], write_kib=vals["WRITE_SIZE"], profile_kernel_src_sha16=prof_sha, kernel_src_sha16=cur, stale=stale)


TRAIN_KERNEL_SOURCES = ("studiosr_amd/csrc/sr_tr_block.hip", "studiosr_amd/csrc/sr_tr_wgrad.hip", "studiosr_amd/csrc/sr_tr_attn.hip", "studiosr_amd/csrc/sr_tr_attn_lds.hip")


def profiled_train_traffic():
    """HBM bytes per launch of the training step's largest kernel -- sr_tr_wgrad_wide_kernel, the launch that holds a block's four nn.Linear weight-gradient jobs (grid 256;
    before the wide tiles: sr_tr_wgrad_kernel with a HAB's six jobs, grid 1536) -- from the committed PMC passes (profiles/*_train_hbm_counters_HAT.txt, tools/pmc_train.sh;
    FETCH_SIZE doubled per the gfx950 rule).  None when absent or taken on other kernel sources (the profile's first line carries their hash)."""
    import glob
    import hashlib
    import re

    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "*_train_hbm_counters_HAT.txt")))
    if not files:
        return None
    h = hashlib.sha256()
    for f in TRAIN_KERNEL_SOURCES:
        h.update(open(os.path.join(ROOT, f), "rb").read())
    text = open(files[-1]).read()
    m0 = re.search(r"#\s*kernel_src_sha16\s*=\s*([0-9a-f]+)", text)
    stale = not m0 or m0.group(1) != h.hexdigest()[:16]
    vals = {}
    name = "sr_tr_wgrad_wide_kernel (one block's four nn.Linear weight-gradient jobs, 42 launches per step)"
    for m in re.finditer(r"\('tr_wgrad_wide_kernel[^\n]*, 256\)\n\s+(FETCH_SIZE|WRITE_SIZE)\s+([0-9.]+)", text):
        vals[m.group(1)] = float(m.group(2))
    if len(vals) != 2:
        vals, name = {}, "sr_tr_wgrad_kernel (one HAB's weight-gradient jobs, 36 launches per step)"
        for m in re.finditer(r"\('tr_wgrad_kernel[^\n]*, 1536\)\n\s+(FETCH_SIZE|WRITE_SIZE)\s+([0-9.]+)", text):
            vals[m.group(1)] = float(m.group(2))
    if len(vals) != 2:
        return None
    return dict(kernel=name, bytes=None if stale else int((2.0 * vals["FETCH_SIZE"] + vals["WRITE_SIZE"]) * 1024),
                fetch_kib_raw=vals["FETCH_SIZE"], write_kib=vals["WRITE_SIZE"], source=os.path.relpath(files[-1], ROOT), stale=stale)


def cpu_model_string() -> str:
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def physical_cores(avail: int) -> int:
    """Physical cores among the CPUs this process may run on (distinct (physical id, core id) pairs of /proc/cpuinfo); SMT siblings
    add little to an fp32 GEMM-bound forward, and a thread count above the core count is what made round 2's baseline swing 2.6x."""
    try:
        allowed = os.sched_getaffinity(0)
        cores, cpu, phys = set(), None, None
        for line in open("/proc/cpuinfo"):
            if line.startswith("processor"):
                cpu = int(line.split(":")[1])
            elif line.startswith("physical id"):
                phys = line.split(":")[1].strip()
            elif line.startswith("core id") and cpu in allowed:
                cores.add((phys, line.split(":")[1].strip()))
        if cores:
            return len(cores)
    except (OSError, ValueError, AttributeError):
        pass
    return avail


def cgroup_cpu_limit():
    """CPU quota of this container in cores (cgroup v2 cpu.max or v1 cfs quota), or None when unlimited / unreadable.  The GPU boxes
    expose every host CPU in the affinity mask but schedule the container on a fraction of them: threads beyond the quota only contend."""
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if q != "max":
            return max(1, int(int(q) / int(per)))
    except (OSError, ValueError):
        pass
    try:
        q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
        per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
        if q > 0:
            return max(1, q // per)
    except (OSError, ValueError):
        pass
    return None


def cpu_baseline_and_parity(model, device):
    """Oracle (CPU, fp32) on CPU_SAMPLE_TILES tiles of the same synthetic workload: throughput + parity."""
    from oracle import metrics as OMT
    from oracle import models as OM

    sd = {k: (v.detach().float() if v.is_floating_point() else v.detach()).cpu() for k, v in model.state_dict().items()}
    cfg = model.get_model_config()
    g = torch.Generator().manual_seed(0)
    x = torch.rand(CPU_SAMPLE_TILES, 3, TILE, TILE, generator=g)
    avail = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    quota = cgroup_cpu_limit()
    # all the physical cores this process may run on; with no readable CPU quota, at most 32 threads (the default run has to stay within minutes)
    cores = int(os.environ.get("SR_CPU_THREADS", "0")) or min(physical_cores(avail), quota if quota else 32)
    cores = max(1, min(cores, avail))
    torch.set_num_threads(cores)
    with torch.inference_mode():
        OM.swinir_forward(sd, x[:1], cfg)  # warm-up
        times = []
        for _ in range(CPU_SAMPLE_FORWARDS):  # ~10-20 s of CPU work
            t0 = time.perf_counter()
            ref = OM.swinir_forward(sd, x, cfg)
            times.append(time.perf_counter() - t0)
        dt = sorted(times)[len(times) // 2]
        torch.set_num_threads(1)  # 1-thread leg: one tile, best of 2
        t1 = []
        for _ in range(2):
            t0 = time.perf_counter()
            OM.swinir_forward(sd, x[:1], cfg)
            t1.append(time.perf_counter() - t0)
        torch.set_num_threads(cores)
    mpix = CPU_SAMPLE_TILES * (TILE * SCALE) ** 2 / 1e6
    cpu = dict(value=round(mpix / dt, 5), unit="HR-Mpix/s", cores=cores, kind="port", cpu_model=cpu_model_string(), logical_cpus_available=avail, cgroup_cpu_quota=quota,
               one_thread=dict(value=round((TILE * SCALE) ** 2 / 1e6 / min(t1), 5), unit="HR-Mpix/s", sample="one tile, best of 2 forwards"),
               sample=f"one step of the same workload ({CPU_SAMPLE_TILES} tiles, SwinIR x4 eval, 64x64 LR, fp32, torch CPU, {cores} threads): median of "
                      f"{CPU_SAMPLE_FORWARDS} forwards after 1 warm-up, {round(sum(times), 1)} s of CPU work; spread {round(min(times), 2)}-{round(max(times), 2)} s per forward")

    tgt = torch.rand(CPU_SAMPLE_TILES, 3, TILE * SCALE, TILE * SCALE, generator=g)

    def to_u8(t):
        return (t.permute(0, 2, 3, 1) * 255.0).round().clip(0, 255).to(torch.uint8).numpy()

    par = {}
    ref_u8, tgt_u8 = to_u8(ref), to_u8(tgt)
    for prec in ("fp32", "bf16"):
        model.set_precision(prec)
        with torch.no_grad():
            y = model(x.to(device)).cpu()
        mse = float(((y - ref) ** 2).mean())
        par[f"psnr_vs_oracle_{prec}_db"] = round(10 * np.log10(1.0 / max(mse, 1e-20)), 3)
        par[f"max_abs_diff_{prec}"] = float((y - ref).abs().max())
        y_u8 = to_u8(y)
        d = [abs(OMT.compute_psnr(y_u8[i], tgt_u8[i], y_only=True, crop_border=SCALE) - OMT.compute_psnr(ref_u8[i], tgt_u8[i], y_only=True, crop_border=SCALE))
             for i in range(CPU_SAMPLE_TILES)]
        par[f"psnr_delta_{prec}_db"] = float(max(d))
    model.set_precision("bf16")
    return cpu, par


def run_strips(args, model, device, rank, world) -> None:
    """BASELINE configs[3]: SwinIR x4 on ONE size x size LR image, one window-aligned row strip per GPU, per-layer halo exchange
    between strip neighbours (RCCL send/recv; studiosr_amd/strips.py).  value = HR megapixels of the whole image / time per
    image (strong scaling: the image is fixed, N strips).  With one GPU the strips run in this process (LocalStripComm) and the
    result is compared bit for bit with the unsharded forward; with N > 1 rank 0 checks its gathered image against the unsharded
    forward it computes itself."""
    import torch.distributed as dist

    from studiosr_amd.strips import DistStripComm, LocalStripComm

    comm = DistStripComm() if world > 1 else LocalStripComm(int(os.environ.get("SR_BENCH_STRIPS", "8")))
    size = args.size
    x = torch.rand(1, 3, size, size, generator=torch.Generator().manual_seed(0)).to(device)
    steps, warmup = max(1, min(args.steps, 5)), max(1, min(args.warmup, 2))

    def sync():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    with torch.no_grad():
        for _ in range(warmup):
            out = model.forward_strips(x, comm)
        sync()
        t0 = time.perf_counter()
        for _ in range(steps):
            out = model.forward_strips(x, comm)
        torch.cuda.synchronize()
        sync()
        elapsed = time.perf_counter() - t0
        if world > 1:
            t = torch.tensor([elapsed], dtype=torch.float64, device=device)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            elapsed = float(t.item())
        if rank == 0:
            ref = model(x)
            equal = bool(torch.equal(out, ref))
            maxdiff = float((out - ref).abs().max())
            dt = elapsed / steps
            hp = (size // 8 + 1) * 8
            flops = hp * hp * FLOP_PER_LR_PIXEL
            print(json.dumps({
                "metric": "HR megapixels/sec at SwinIR x4, one large LR image, row strips with halo exchange", "value": round((size * SCALE) ** 2 / 1e6 / dt, 3),
                "unit": "HR-Mpix/s", "n_gpus": world, "steps": steps, "warmup": warmup, "ms_per_step": round(dt * 1e3, 3), "higher_is_better": True,
                "scaling": "strong", "vs_baseline": None, "dtype": "bf16", "data": "synthetic",
                "config": {"workload": f"SwinIR x4 eval forward, ONE {size}x{size} LR image, {comm.world} row strips, halo exchange per layer + gather of the HR strips",
                           "strips": comm.world, "launch": "eager"},
                "equals_unsharded": equal, "max_abs_diff": maxdiff,
                "roofline": {"bound": "mfma", "achieved": round(flops / dt / 1e12, 2), "peak": MFMA_BF16_PEAK_TFLOPS * world, "unit": "TFLOP/s",
                             "frac": round(flops / dt / 1e12 / (MFMA_BF16_PEAK_TFLOPS * world), 4), "traffic": None},
                "cpu_baseline": None,
            }), flush=True)


def run_train(args, device, rank, world) -> None:
    """BASELINE configs[4]: HAT x4 (defaults: embed 180, 6 x (6 HAB + OCAB), window 16, DropPath 0.1) training step of the reference
    Trainer (studiosr/engine/trainer.py:97-109: bf16 autocast context, forward, L1 loss, backward, Adam 2e-4 / (0.9, 0.99), MultiStepLR)
    on synthetic DIV2K-shape batches: global batch 4 x N (the reference's 32 at N = 8), 64x64 LR -> 256x256 HR, one rank per GPU,
    DistributedDataParallel (RCCL all-reduce of 83 MB of fp32 gradients overlapped with backward).  value = training samples / s over
    all ranks; a step = forward + backward + optimizer step.  FLOPs per sample = 3 x 207.76 GF.  Under the
    autocast context the contractions of forward and backward round their operands to bf16 and run on the bf16 matrix cores (fp32
    accumulate; parameters, gradients, activations and everything element-wise stay fp32), so the roofline is the bf16 MFMA peak."""
    import torch.distributed as dist
    from torch.nn.parallel import DistributedDataParallel as DDP

    import studiosr_amd as S

    torch.manual_seed(0)
    kind = {"hat": "HAT", "swinir": "SwinIR"}[args.model]
    model = getattr(S, kind)(scale=4).to(device).train()  # reference defaults (hat.py:389-406 / swinir.py:259-274)
    cfg = model.get_training_config()
    from studiosr_amd.optim import Adam  # torch.optim.Adam (what studiosr_amd.Trainer builds): one flat launch on the fused path

    opt = Adam(model.parameters(), model=model, lr=cfg.get("learning_rate", 2e-4), betas=(cfg.get("beta1", 0.9), cfg.get("beta2", 0.99)))
    sched = torch.optim.lr_scheduler.MultiStepLR(opt, milestones=cfg.get("milestones", [250000]), gamma=cfg.get("gamma", 0.5))
    net = DDP(model, device_ids=[device.index], output_device=device.index) if world > 1 else model
    per_rank = 4
    torch.manual_seed(1234 + rank)  # seed + rank as data/handler.py:86-88
    x, y = torch.rand(per_rank, 3, TILE, TILE, device=device), torch.rand(per_rank, 3, TILE * SCALE, TILE * SCALE, device=device)
    steps, warmup = max(1, min(args.steps, 20)), max(1, min(args.warmup, 3))

    def step():
        with torch.autocast(device_type="cuda", dtype=torch.bfloat16):
            loss = torch.nn.functional.l1_loss(net(x), y)
        loss.backward()
        opt.step()
        opt.zero_grad(set_to_none=True)
        sched.step()
        return loss

    def sync():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(warmup):
        step()
    sync()
    t0 = time.perf_counter()
    for _ in range(steps):
        loss = step()
    torch.cuda.synchronize()
    sync()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    if rank == 0:
        dt = elapsed / steps
        gf_fwd = {"hat": 207.76e9, "swinir": 107.11e9}[args.model]  # forward GFLOP per 64 x 64 training patch (SwinIR: train mode does not pad 64 -> 72)
        tflops = 3 * gf_fwd * per_rank * world / dt / 1e12
        fused = getattr(model, "_fast_plan", None) is not None
        print(json.dumps({
            "metric": f"training samples/sec at {kind} x4, 64x64 LR patches, L1 + Adam step", "value": round(per_rank * world / dt, 3), "unit": "samples/s", "n_gpus": world,
            "steps": steps, "warmup": warmup, "ms_per_step": round(dt * 1e3, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "bf16",
            "data": "synthetic", "config": {"workload": ("HAT x4 (embed 180, 6x(6 HAB + OCAB), ws 16)" if args.model == "hat" else "SwinIR x4 (embed 180, 6x6 blocks, ws 8)") +
                                                         " training step: forward + backward + Adam, per-rank batch 4, 64x64 LR / 256x256 HR",
                                             "global_batch": per_rank * world, "parallelism": f"ddp{world}" if world > 1 else "single",
                                             "path": ("fused launch sequences replayed from C launch plans, one autograd node per RHAG (studiosr_amd/fasttrain.py, C ABI v10 sr_plan_run)" if fused else
                                                      "generic engine (one strided batched GEMM / row kernel per op, studiosr_amd/autograd.py)")},
            "final_loss": loss.item(),
            "roofline": {"bound": "mfma", "achieved": round(tflops, 2), "peak": 2500.0 * world, "unit": "TFLOP/s", "frac": round(tflops / (2500.0 * world), 4),
                         "traffic": ((profiled_train_traffic() or {}).get("bytes") if args.model == "hat" else None), "traffic_detail": profiled_train_traffic() if args.model == "hat" else None,
                         "note": "bf16 operands / fp32 accumulate for the contractions (autocast), fp32 everywhere else; against the fp32 MFMA peak (157.3 TFLOP/s) the same step is "
                                 + str(round(tflops / (157.3 * world), 3))},
            "cpu_baseline": None,
        }), flush=True)


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--no-graph", action="store_true", help="launch eagerly instead of replaying a HIP graph")
    ap.add_argument("--skip-cpu", action="store_true", help="skip the CPU baseline / parity leg")
    ap.add_argument("--inflight", type=int, default=2, help="independent batches in flight per GPU (one HIP graph + stream + workspace each)")
    ap.add_argument("--mode", choices=["tiles", "strips", "train"], default="tiles",
                    help="tiles: the headline metric (independent 64x64 tiles); strips: BASELINE config 4, ONE large LR image cut into one row strip per GPU with per-layer halo exchange; "
                         "train: BASELINE config 5, HAT x4 training step, per-rank batch 4, DistributedDataParallel over RCCL")
    ap.add_argument("--size", type=int, default=2048, help="--mode strips: LR image side")
    ap.add_argument("--model", choices=["hat", "swinir"], default="hat", help="--mode train: HAT x4 (BASELINE config 5) or SwinIR x4 (swinir.py:391-402); both on the fused path of studiosr_amd/fasttrain.py")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        # `python bench.py --gpus N` without a launcher: start N fresh rank processes (one per GPU) through
        # torch.distributed.run and relay rank 0's JSON line.  This parent has not touched the GPU (importing torch does not
        # initialise HIP) and never execs: the ranks are ordinary child processes.
        import socket
        import subprocess

        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
               "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        raise SystemExit(subprocess.run(cmd).returncode)
    if world != args.gpus:
        raise SystemExit(f"bench.py --gpus {args.gpus} but WORLD_SIZE={world}: launch one rank per GPU")
    import torch.distributed as dist

    device = torch.device("cuda", local_rank)
    torch.cuda.set_device(device)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=device)  # RCCL on ROCm

    from studiosr_amd import _lib
    from studiosr_amd.runtime import GraphedForward

    if not os.path.exists(_lib.LIB_PATH):  # a fresh checkout: the library is a (git-ignored) build artefact
        if rank == 0:
            _lib.build()
        if world > 1:
            dist.barrier()

    if args.mode == "train":
        run_train(args, device, rank, world)
        if world > 1:
            dist.barrier()
            dist.destroy_process_group()
        return
    model = build_model(device)
    if args.mode == "strips":
        run_strips(args, model, device, rank, world)
        if world > 1:
            dist.barrier()
            dist.destroy_process_group()
        return
    g = torch.Generator().manual_seed(1234 + rank)
    x = torch.rand(BATCH, 3, TILE, TILE, generator=g).to(device)

    def fwd_m(m, inp):
        with torch.no_grad():
            return m(inp)

    def fwd(inp):
        return fwd_m(model, inp)

    # `inflight` independent pipelines (own module copy = own workspace, own HIP graph, own stream): a batch of 8 tiles is
    # 648 windows on 256 CUs, so the tail of one step's kernels leaves CUs idle that the next step's kernels can use.
    n_pipe = 1 if args.no_graph else max(1, args.inflight)
    if args.no_graph:
        pipes = [(fwd, torch.cuda.current_stream())]
    else:
        from studiosr_amd.runtime import Workspace

        pipes = []
        for i in range(n_pipe):
            ws_i = Workspace(device)  # weights are shared; every pipeline owns its activations

            def fwd_i(inp, ws_i=ws_i):
                model._ws = ws_i
                return fwd(inp)

            st = torch.cuda.Stream()
            with torch.cuda.stream(st):
                gf = GraphedForward(fwd_i, x)
            pipes.append((gf, st, ws_i))  # the graph holds raw pointers into ws_i: keep it alive with the pipeline
            if os.environ.get("SR_BENCH_SERIAL"):
                torch.cuda.synchronize()
                print(f"[bench] pipeline {i} captured", file=sys.stderr, flush=True)
        torch.cuda.synchronize()

    def run_steps(n):
        for i in range(n):
            f, st = pipes[i % len(pipes)][:2]
            if args.no_graph:
                f(x)
            else:
                with torch.cuda.stream(st):
                    f.replay()
                if os.environ.get("SR_BENCH_SERIAL"):
                    torch.cuda.synchronize()
                    print(f"[bench] step {i} on pipeline {i % len(pipes)} done", file=sys.stderr, flush=True)

    run_steps(args.warmup)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    barrier()
    t0 = time.perf_counter()
    run_steps(args.steps)
    torch.cuda.synchronize()
    barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    ms_per_step = elapsed / args.steps * 1e3
    hr_mpix_per_step = world * BATCH * (TILE * SCALE) ** 2 / 1e6
    value = hr_mpix_per_step / (elapsed / args.steps)

    # the same K steps with ONE batch in flight (every step waits for nothing but its own predecessor on one stream).  A batch that is alone on the GPU runs as
    # four part batches of two tiles on the model's own streams (SwinIR.part_batches: a latency knob -- it costs throughput when a second batch is in flight,
    # so the pipelines above do not use it); its graph is captured here.
    one_parts = 1
    one = pipes[:1]
    if not args.no_graph and BATCH % 4 == 0 and not os.environ.get("SR_SWIN_PARTS"):
        from studiosr_amd.runtime import Workspace

        ws_1 = Workspace(device)

        def fwd_1(inp):
            model._ws = ws_1
            return fwd(inp)

        st1 = torch.cuda.Stream()
        model.part_batches = one_parts = int(os.environ.get("SR_BENCH_ONE_PARTS", "4"))  # (the override is for experiments only)
        try:
            with torch.cuda.stream(st1):
                gf1 = GraphedForward(fwd_1, x)
        finally:
            model.part_batches = 1
        torch.cuda.synchronize()
        one = [(gf1, st1, ws_1)]
    pipes_all, pipes[:] = list(pipes), one
    run_steps(min(args.warmup, 3))
    barrier()
    t0 = time.perf_counter()
    run_steps(args.steps)
    torch.cuda.synchronize()
    barrier()
    elapsed1 = time.perf_counter() - t0
    pipes[:] = pipes_all
    if not args.no_graph:
        model._ws = pipes_all[-1][2]  # time_dominant_kernel() below launches the block kernel on the stream tensors of a throughput pipeline's last forward (real
        # activations -- a fresh, zero-filled workspace would be timed at another power level)
    if world > 1:
        t = torch.tensor([elapsed1], dtype=torch.float64, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed1 = float(t.item())

    if rank == 0:
        k_ms, k_ms_min, k_flops, k_n = time_dominant_kernel(model, x)
        achieved = k_flops / (k_ms * 1e-3) / 1e12
        executed = BATCH * (PADDED // 8) ** 2 * EXECUTED_FLOP_PER_WINDOW
        clock = profiled_clock_ghz()
        fwd_flops = BATCH * PADDED * PADDED * FLOP_PER_LR_PIXEL  # per GPU per step, reference semantics (padded tile)
        fwd_tflops = fwd_flops / (ms_per_step * 1e-3) / 1e12
        traffic = profiled_hbm_traffic()
        # fp32 residual stream in + out (Cp = 192 channels) + the block's packed bf16 weights and biases, per launch
        algo_bytes = 2 * BATCH * PADDED * PADDED * 192 * 4 + 2 * (192 * 576 + 192 * 192 + 2 * 192 * 384) + 4 * 6 * 64 * 64
        roof = dict(
            bound="mfma", kernel="sr_swin_block3_kernel / C ABI sr_swin_block (one whole Swin block per launch: LN1 + QKV + shifted-window attention + proj + residual + LN2 + MLP + "
                                 "residual; one 64-token window per 4-wave workgroup, 648 workgroups, 36 launches per forward)",
            achieved=round(achieved, 2), peak=MFMA_BF16_PEAK_TFLOPS, unit="TFLOP/s", frac=round(achieved / MFMA_BF16_PEAK_TFLOPS, 4),
            traffic=(traffic or {}).get("bytes"), traffic_detail=traffic, algorithmic_hbm_bytes=algo_bytes, kernel_ms=round(k_ms, 5),
            kernel_ms_min=round(k_ms_min, 5), launches_timed=k_n, executed_over_algorithmic_flops=round(executed / k_flops, 4),
            clock_ghz_held=clock, frac_of_peak_at_held_clock=(round(achieved / (MFMA_BF16_PEAK_TFLOPS * clock / 2.4), 4) if clock else None),
            forward=dict(achieved=round(fwd_tflops, 2), frac=round(fwd_tflops / MFMA_BF16_PEAK_TFLOPS, 4),
                         frac_unpadded=round(fwd_tflops * (TILE * TILE) / (PADDED * PADDED) / MFMA_BF16_PEAK_TFLOPS, 4),
                         gflop_per_step=round(fwd_flops / 1e9, 2)),
        )
        out = {
            "metric": "HR megapixels/sec at SwinIR x4, 64x64 LR tiles",
            "value": round(value, 3),
            "unit": "HR-Mpix/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 4),
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "bf16",
            "data": "synthetic",
            "config": {"workload": "SwinIR x4 (embed 180, 6x6 blocks, ws 8) eval forward, 64x64 LR tiles, batch 8 per GPU",
                       "tiles_per_step": world * BATCH, "launch": "eager" if args.no_graph else "hipGraph replay", "batches_in_flight": len(pipes)},
            "one_batch_in_flight": {"ms_per_step": round(elapsed1 / args.steps * 1e3, 4), "value": round(hr_mpix_per_step / (elapsed1 / args.steps), 3), "part_batches": one_parts,
                                    "forward_frac": round(BATCH * PADDED * PADDED * FLOP_PER_LR_PIXEL / (elapsed1 / args.steps) / 1e12 / MFMA_BF16_PEAK_TFLOPS, 4)},
            "roofline": roof,
        }
        if not args.skip_cpu and world == 1:
            cpu, par = cpu_baseline_and_parity(model, device)
            out["cpu_baseline"] = cpu
            out["parity"] = par
        elif not args.skip_cpu:
            out["cpu_baseline"] = None  # reported on rank 0 at N=1 only
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
