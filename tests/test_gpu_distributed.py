"""Rehearsal of the sharded path with the REAL kernel: 2 and 4 ranks share the one GPU of the box and exchange through
gloo (RCCL refuses two ranks on one device; the 8-GPU RCCL run is the driver's).  Checks tgt_off != 0 launches,
the ping-pong and the in-place exchange against the unsharded run."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import ROOT

pytestmark = pytest.mark.gpu
N, STEPS = 16384, 4


def _run(rank, world, port, path):
    sys.path.insert(0, ROOT)
    import nbody_amd  # noqa: F401
    from nbody_amd import synthetic
    from nbody_amd.distributed import ShardedSystem, shard_range
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    if world > 1:
        dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    lo, hi = shard_range(N, rank, world)
    pos, vel = synthetic.body4_f32(N, lo, hi)
    sysm = ShardedSystem(N, torch.from_numpy(pos), torch.from_numpy(vel), synthetic.EPS, 1e-2, dev)
    for _ in range(STEPS):
        sysm.step()
    torch.cuda.synchronize()
    if world > 1:
        vels = [torch.zeros_like(sysm.vel) for _ in range(world)]
        dist.all_gather(vels, sysm.vel)
        allv = torch.cat(vels)
    else:
        allv = sysm.vel
    if rank == 0:
        np.savez(path, pos=sysm.positions.cpu().numpy(), vel=allv.cpu().numpy())
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 4])
def test_ranks_on_one_gpu_match_single(nb, tmp_path, world):
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    one, two = str(tmp_path / "one.npz"), str(tmp_path / "two.npz")
    mp.spawn(_run, args=(1, 0, one), nprocs=1, join=True)
    mp.spawn(_run, args=(world, port, two), nprocs=world, join=True)
    a, b = np.load(one), np.load(two)
    # the sharded launch may pick another register blocking / source split: equal to fp32 rounding, not bitwise
    assert np.abs(a["pos"] - b["pos"]).max() < 5e-7 and np.abs(a["vel"] - b["vel"]).max() < 1e-5
    p0, _ = nb.synthetic.body4_f32(N)
    assert np.abs(b["pos"][:, :3] - p0[:, :3]).max() > 1e-6 and np.array_equal(b["pos"][:, 3], p0[:, 3])
