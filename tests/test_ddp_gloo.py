"""world_size-2 gloo test of the data-parallel path on the CPU: sharding of the pair
stream, parameter broadcast, flat-bucket gradient all-reduce == gradient of the
global batch computed by one process."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from mm_masking_amd import ddp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _model(seed):
    torch.manual_seed(seed)
    return torch.nn.Sequential(torch.nn.Conv2d(1, 4, 3, padding=1), torch.nn.ReLU(), torch.nn.Conv2d(4, 1, 1))


def _data(idx):
    g = torch.Generator().manual_seed(1000 + idx)
    return torch.rand(1, 1, 16, 16, generator=g), torch.rand(1, 1, 16, 16, generator=g)


def _worker(rank, world, port, out):
    os.environ.update({"MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(port)})
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        model = _model(seed=rank)                # different init per rank on purpose
        sync = ddp.FlatGradSync(model)
        sync.sync_params(0)
        opt = torch.optim.Adam(model.parameters(), lr=1e-2)
        idx = ddp.shard_indices(8, rank, world)
        for step in range(2):
            sync.zero_grad()
            x = torch.cat([_data(i + 8 * step)[0] for i in idx])
            y = torch.cat([_data(i + 8 * step)[1] for i in idx])
            loss = ((model(x) - y) ** 2).mean()
            loss.backward()
            sync()
            opt.step()
        out[rank] = torch.cat([p.detach().flatten() for p in model.parameters()])
        assert all(p.grad.data_ptr() == v.data_ptr() for p, v in zip(sync.params, sync.views))
    finally:
        dist.destroy_process_group()


def test_two_rank_allreduce_equals_single_process_global_batch():
    world = 2
    port = _free_port()
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_worker, args=(world, port, out), nprocs=world, join=True)
    assert torch.equal(out[0], out[1])            # ranks stay in lock step
    # single process over the same global batches
    model = _model(seed=0)
    opt = torch.optim.Adam(model.parameters(), lr=1e-2)
    for step in range(2):
        opt.zero_grad()
        x = torch.cat([_data(i + 8 * step)[0] for i in range(8)])
        y = torch.cat([_data(i + 8 * step)[1] for i in range(8)])
        ((model(x) - y) ** 2).mean().backward()
        opt.step()
    ref = torch.cat([p.detach().flatten() for p in model.parameters()])
    assert torch.allclose(out[0], ref, atol=1e-6)


def test_shard_indices_cover_global_batch():
    got = sorted(i for r in range(4) for i in ddp.shard_indices(32, r, 4, start=64))
    assert got == list(range(64, 96))
    assert ddp.shard_indices(8, 1, 2) == [1, 3, 5, 7]
