"""world_size-2 gloo test of the N>1 path: tile-partitioned rendering + ONE gather
(raytracingpbr_amd/distributed.py) on CPU, with the oracle standing in for the GPU renderer.
Checks that rank 0 ends with exactly the single-process frame."""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, out_path):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ["OMP_NUM_THREADS"] = "2"
    import torch.distributed as dist
    from cases import case_by_name
    from oracle_backend import OracleRenderer
    from raytracingpbr_amd.distributed import render_distributed
    dist.init_process_group("gloo", rank=rank, world_size=world)
    case = case_by_name("cornell_v3_8b_wide")
    r = OracleRenderer(case.scene, case.cfg, threads=2)
    tg = render_distributed(r, 4, rank, world, tile=(16, 16))
    if rank == 0:
        np.save(out_path, r.image_buffer)
    assert tg.layout.world == world
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_tile_gather_matches_single_process(tmp_path, world):
    import torch.multiprocessing as mp
    from cases import case_by_name
    from oracle_backend import OracleRenderer
    out = str(tmp_path / "gathered.npy")
    mp.spawn(_worker, args=(world, _free_port(), out), nprocs=world, join=True)
    case = case_by_name("cornell_v3_8b_wide")
    ref = OracleRenderer(case.scene, case.cfg)
    ref.sample(4)
    got = np.load(out)
    assert np.array_equal(got.view(np.uint32), ref.image_buffer.view(np.uint32))
    assert np.all(got[..., 3] == 4.0)
