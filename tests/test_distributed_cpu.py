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
