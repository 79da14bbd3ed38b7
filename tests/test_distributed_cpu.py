"""N > 1 path on CPU: two gloo ranks shard a tile batch, run a stand-in forward (the CPU oracle -- the HIP forward
needs a GPU) and all-gather the HR tiles; the result must equal the single-process result."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import golden_cfg, golden_sd, load_golden


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, n_tiles, out_path):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from oracle import models as OM
        from studiosr_amd.parallel import TileParallel

        g = load_golden("f11_edsr_x2")
        sd, cfg = golden_sd(g), golden_cfg(g)
        torch.manual_seed(0)
        tiles = torch.rand(n_tiles, 3, 8, 8)
        fn = lambda t: OM.edsr_forward(sd, t, cfg)  # noqa: E731
        with torch.no_grad():
            out = TileParallel(fn)(tiles)
            ref = fn(tiles)
        assert out.shape == ref.shape
        torch.testing.assert_close(out, ref, rtol=1e-5, atol=1e-6)
        if rank == 0:
            torch.save(out, out_path)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("n_tiles", [5, 1])
def test_tile_parallel_two_ranks_gloo(tmp_path, n_tiles):
    torch.set_num_threads(1)
    port = _free_port()
    out_path = str(tmp_path / "out.pt")
    mp.spawn(_worker, args=(2, port, n_tiles, out_path), nprocs=2, join=True)
    assert torch.load(out_path).shape == (n_tiles, 3, 16, 16)


# ----------------------------------------------------------------------------- row strips of one image (section 8e, config 4)
def _strip_case(world, seed=0):
    """Per-rank send buffers of different content (strip heights differ; halos have a fixed size)."""
    g = torch.Generator().manual_seed(seed)
    return [torch.randn(1, 4, 6, 5, generator=g) + 10 * r for r in range(world)]


def _strip_worker(rank, world, port):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from studiosr_amd.strips import DistStripComm, LocalStripComm, strip_partition

        comm, local = DistStripComm(), LocalStripComm(world)
        assert comm.world == world and comm.local_ranks == [rank]
        sends = _strip_case(world)
        for cyclic in (True, False):
            for name in ("shift_up", "shift_down"):
                want = [torch.full_like(s, -1.0) for s in sends]
                getattr(local, name)(sends, want, cyclic)
                got = torch.full_like(sends[rank], -1.0)
                getattr(comm, name)([sends[rank]], [got], cyclic)
                assert torch.equal(got, want[rank]), (name, cyclic, rank)

        # the SW-MSA protocol on a per-token stand-in: strips of roll(x, -sh) via shift_up, roll back via shift_down
        H, W, C, ws, sh = 8 * (2 * world + 1), 8, 3, 8, 4
        full = torch.arange(H * W * C, dtype=torch.float32).view(1, H, W, C)
        a, b = strip_partition(H // ws, world)[rank]
        r0, r1 = a * ws, b * ws
        buf = torch.zeros(1, (r1 - r0) + sh, W, C)
        buf[:, : r1 - r0] = full[:, r0:r1]
        comm.shift_up([buf[:, :sh]], [buf[:, r1 - r0 :]], cyclic=True)
        rolled = torch.roll(full, -sh, dims=1)
        assert torch.equal(buf[:, sh:], rolled[:, r0:r1])
        buf[:, sh:] += 1.0  # "the block"
        comm.shift_down([buf[:, r1 - r0 :]], [buf[:, :sh]], cyclic=True)
        assert torch.equal(buf[:, : r1 - r0], full[:, r0:r1] + 1.0)

        # 1-row conv halos: zeros at the image border
        ext = torch.full((1, (r1 - r0) + 2, W, C), -7.0)
        ext[:, 1:-1] = full[:, r0:r1]
        comm.shift_down([ext[:, -2:-1]], [ext[:, :1]], cyclic=False)
        comm.shift_up([ext[:, 1:2]], [ext[:, -1:]], cyclic=False)
        padded = torch.nn.functional.pad(full, (0, 0, 0, 0, 1, 1))
        assert torch.equal(ext, padded[:, r0 : r1 + 2])

        # uneven gather along the row dimension of an NCHW result
        mine = full[:, r0:r1].permute(0, 3, 1, 2).contiguous()
        assert torch.equal(comm.gather_rows([mine], dim=2), full.permute(0, 3, 1, 2))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_strip_halo_exchange_gloo(world):
    torch.set_num_threads(1)
    mp.spawn(_strip_worker, args=(world, _free_port()), nprocs=world, join=True)


def test_strip_partition_and_local_comm():
    from studiosr_amd.strips import LocalStripComm, strip_partition

    assert strip_partition(257, 8) == [(0, 33), (33, 65), (65, 97), (97, 129), (129, 161), (161, 193), (193, 225), (225, 257)]
    assert strip_partition(5, 5) == [(i, i + 1) for i in range(5)]
    with pytest.raises(ValueError):
        strip_partition(3, 4)
    c = LocalStripComm(3)
    s = [torch.full((2,), float(r)) for r in range(3)]
    d = [torch.empty(2) for _ in range(3)]
    c.shift_up(s, d, cyclic=True)
    assert [float(t[0]) for t in d] == [1.0, 2.0, 0.0]
    c.shift_up(s, d, cyclic=False)
    assert [float(t[0]) for t in d] == [1.0, 2.0, 0.0] and float(d[2].abs().sum()) == 0.0
    c.shift_down(s, d, cyclic=True)
    assert [float(t[0]) for t in d] == [2.0, 0.0, 1.0]
    c.shift_down(s, d, cyclic=False)
    assert [float(t[0]) for t in d] == [0.0, 0.0, 1.0]
