"""Multi-GPU paths on real devices (skipped on a one-GPU box): `bench.py --gpus 2` starts its own ranks (one per GPU, RCCL),
--mode tiles = independent tile batches (no data-path collective), --mode strips = ONE image in row strips with per-layer halo
exchange over `DistStripComm` (nccl p2p), checked bit for bit against the unsharded forward on rank 0."""
import json
import os
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
needs2 = pytest.mark.skipif(torch.cuda.device_count() < 2, reason="needs 2 visible GPUs")


def _bench(*args):
    env = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *args], capture_output=True, text=True, timeout=900, env=env)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-4000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, p.stdout[-2000:]
    return json.loads(lines[0])


@needs2
def test_bench_self_launches_two_ranks_tiles():
    r = _bench("--gpus", "2", "--steps", "5", "--warmup", "2", "--skip-cpu")
    assert r["n_gpus"] == 2 and r["scaling"] == "weak" and r["config"]["tiles_per_step"] == 16 and r["value"] > 0


@needs2
def test_strips_two_ranks_nccl_equal_the_unsharded_forward():
    r = _bench("--gpus", "2", "--mode", "strips", "--size", "200", "--steps", "2", "--warmup", "1")
    assert r["n_gpus"] == 2 and r["config"]["strips"] == 2
    assert r["equals_unsharded"] is True and r["max_abs_diff"] == 0.0


@needs2
def test_tile_parallel_with_the_hip_forward_on_nccl(tmp_path):
    script = tmp_path / "tp.py"
    script.write_text(
        "import os, sys, torch, torch.distributed as dist\n"
        f"sys.path.insert(0, {ROOT!r})\n"
        "import studiosr_amd as S\n"
        "from studiosr_amd.parallel import TileParallel\n"
        "lr = int(os.environ['LOCAL_RANK']); torch.cuda.set_device(lr); dev = torch.device('cuda', lr)\n"
        "dist.init_process_group('nccl', device_id=dev)\n"
        "torch.manual_seed(0)\n"
        "m = S.SwinIR(scale=2, embed_dim=60, depths=[2], num_heads=[6]).to(dev).eval().set_precision('bf16')\n"
        "x = torch.rand(5, 3, 16, 16, generator=torch.Generator().manual_seed(1)).to(dev)\n"
        "with torch.no_grad():\n"
        "    ref = m(x)\n"
        "    out = TileParallel(lambda t: m(t))(x)\n"
        "assert torch.equal(out, ref), float((out - ref).abs().max())\n"
        "dist.barrier(); dist.destroy_process_group(); print('OK')\n"
    )
    p = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
                        "--master-port", "29611", str(script)], capture_output=True, text=True, timeout=600)
    assert p.returncode == 0 and p.stdout.count("OK") == 2, p.stdout[-2000:] + p.stderr[-4000:]


# ----------------------------------------------------------------------------- more than one PROCESS on the one GPU (gloo collectives)
def _run_ranks(tmp_path, body: str, nproc: int, port: int):
    """torch.distributed.run with `nproc` ranks that all use cuda:0 and the gloo backend (RCCL refuses two ranks on one device);
    the collectives carry device tensors, the compute is the real HIP forward."""
    script = tmp_path / "ranks.py"
    script.write_text(
        "import os, sys, torch, torch.distributed as dist\n"
        f"sys.path.insert(0, {ROOT!r})\n"
        "import studiosr_amd as S\n"
        "torch.cuda.set_device(0); dev = torch.device('cuda', 0)\n"
        "dist.init_process_group('gloo')\n"
        "rank, world = dist.get_rank(), dist.get_world_size()\n" + body +
        "\ndist.barrier(); dist.destroy_process_group(); print('RANK_OK')\n"
    )
    env = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    p = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={nproc}", "--master-addr", "127.0.0.1",
                        "--master-port", str(port), str(script)], capture_output=True, text=True, timeout=900, env=env)
    assert p.returncode == 0 and p.stdout.count("RANK_OK") == nproc, p.stdout[-2000:] + p.stderr[-4000:]


def test_row_strips_three_processes_with_the_hip_forward_equal_the_unsharded_forward(tmp_path):
    """DistStripComm (one strip per PROCESS, paired isend / irecv halos, gather of the HR strips) driving the HIP kernels: the result of
    every rank must be bit-identical to the unsharded forward (round 1 only ran the protocol with CPU stand-ins)."""
    _run_ranks(tmp_path, """
from studiosr_amd.strips import DistStripComm
torch.manual_seed(0)
m = S.SwinIR(scale=4, embed_dim=180, depths=[2, 2], num_heads=[6, 6]).to(dev).eval().set_precision('bf16')
with torch.no_grad():
    for p_ in m.parameters():
        if p_.ndim == 1:
            p_.add_(torch.randn_like(p_) * 0.1)
x = torch.rand(1, 3, 37, 52, generator=torch.Generator().manual_seed(1)).to(dev)   # eval pad -> 40 x 56: 5 window rows over 3 strips
with torch.no_grad():
    ref = m(x)
    out = m.forward_strips(x, DistStripComm())
assert out.shape == ref.shape and torch.equal(out, ref), float((out - ref).abs().max())
""", 3, 29631)


def test_row_strips_two_processes_at_config4_size_equal_the_unsharded_forward(tmp_path):
    """BASELINE config 4's image size (2048 x 2048 LR, eval pad -> 2056: 257 window rows) over TWO processes with DistStripComm: paired isend / irecv halos of
    every shifted block and every conv, gather of the 8192-wide HR strips -- bit-identical to the unsharded forward on every rank (reduced depth keeps it short:
    the exchange pattern per block does not depend on the depth)."""
    _run_ranks(tmp_path, """
from studiosr_amd.strips import DistStripComm
torch.manual_seed(0)
m = S.SwinIR(scale=4, embed_dim=180, depths=[2, 2], num_heads=[6, 6]).to(dev).eval().set_precision('bf16')
x = torch.rand(1, 3, 2048, 2048, generator=torch.Generator().manual_seed(1)).to(dev)
with torch.no_grad():
    ref = m(x)
    out = m.forward_strips(x, DistStripComm())
assert out.shape == (1, 3, 8192, 8192) and torch.equal(out, ref), float((out - ref).abs().max())
""", 2, 29633)


def test_tile_parallel_two_processes_with_the_hip_forward(tmp_path):
    _run_ranks(tmp_path, """
from studiosr_amd.parallel import TileParallel
torch.manual_seed(0)
m = S.SwinIR(scale=2, embed_dim=60, depths=[2], num_heads=[6]).to(dev).eval().set_precision('bf16')
x = torch.rand(5, 3, 16, 16, generator=torch.Generator().manual_seed(1)).to(dev)
with torch.no_grad():
    ref = m(x)
    out = TileParallel(lambda t: m(t))(x)
assert torch.equal(out, ref), float((out - ref).abs().max())
""", 2, 29632)


@needs2
def test_model_on_a_non_default_device():
    """model.to('cuda:1') with device 0 current -- launches must follow the tensors' device."""
    import studiosr_amd as S

    torch.manual_seed(0)
    m0 = S.EDSR(scale=2, n_feats=32, n_resblocks=2).eval()
    x = torch.rand(1, 3, 12, 12)
    with torch.no_grad():
        y0 = m0.to("cuda:0")(x.to("cuda:0")).cpu()
        assert torch.cuda.current_device() == 0
        y1 = m0.to("cuda:1")(x.to("cuda:1")).cpu()
    assert torch.equal(y0, y1)
