"""world_size-2 gloo tests of the data-parallel path on the CPU: sharding of the pair stream, parameter
broadcast, flat-bucket gradient all-reduce == gradient of the global batch computed by one process -- on a toy
model (the collective's arithmetic) and on the policy's REAL host logic (`LearnICPWeightPolicy`: the 46-tensor module
tree with its twice-applied decoder blocks, the batch-global min-max normalisation reduced over the ranks, the
amax-normalised mask, `FlatGradSync`, Adam), which on the CPU runs through the nn.Module mirror of the U-Net;
the dICP / radar kernels have no CPU path, so the real-kernel variant of this test is tests/test_gpu_ddp.py."""
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


# ------------------------------------------------------------------------------------------------------------------
# the policy's real host logic
def _policy(seed):
    from mm_masking_amd import train_icp_weights as trn
    from mm_masking_amd.icp_weight_policy import LearnICPWeightPolicy
    p = trn.default_params(torch.device("cpu"))
    p.update({"dropout": 0.0, "unet_backend": "torch"})
    torch.manual_seed(seed)
    return LearnICPWeightPolicy(p), p


def _scan(idx, hw=64):
    """Pair `idx`: an image whose value range depends on the pair (a per-rank min-max normalisation would show)."""
    g = torch.Generator().manual_seed(2000 + idx)
    img = torch.rand(1, hw, hw, generator=g) * (0.4 + 0.15 * (idx % 4)) + 0.05 * (idx % 3)
    tgt = (torch.rand(1, hw, hw, generator=g) > 0.7).float()
    return img, tgt


def _policy_step(model, opt, idx, sync=None):
    from mm_masking_amd import train_icp_weights as trn  # noqa: F401
    img = torch.cat([_scan(i)[0] for i in idx])
    tgt = torch.cat([_scan(i)[1] for i in idx])
    scan = {"fft_data": img, "fft_cfar": torch.zeros_like(img), "raw_pc": torch.zeros(len(idx), 4, 3)}
    if sync is not None:
        sync.zero_grad()
    else:
        opt.zero_grad()
    mask = model(scan, {"pc": torch.zeros(len(idx), 4, 6)}, None, mask_only=True)
    loss = torch.nn.BCELoss()(mask, tgt)          # the trainer's mask loss (train_icp_weights.py:223-226)
    loss.backward()
    if sync is not None:
        sync()
    opt.step()
    return float(loss)


def _policy_worker(rank, world, port, out):
    os.environ.update({"MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(port)})
    torch.set_num_threads(2)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from mm_masking_amd import train_icp_weights as trn
        model, p = _policy(seed=10 + rank)        # different init per rank on purpose: rank 0's is broadcast
        assert model.global_minmax                # the default inside a multi-rank job
        model.train()
        sync = ddp.FlatGradSync(model)
        sync.sync_params(0)
        opt = trn.make_optimizer(model, p)
        losses = []
        for step in range(2):
            losses.append(_policy_step(model, opt, ddp.shard_indices(4, rank, world, start=4 * step), sync))
        lt = torch.tensor(losses, dtype=torch.float64)
        dist.all_reduce(lt)
        out[rank] = (torch.cat([q.detach().flatten() for q in model.parameters()]), (lt / world).tolist())
    finally:
        dist.destroy_process_group()


def test_policy_host_logic_two_ranks_equal_one_process():
    """2 ranks x 2 pairs through LearnICPWeightPolicy + FlatGradSync + Adam == 1 process x 4 pairs: same parameters after two
    steps and the same mean loss.  Needs the min-max normalisation to be global over the ranks (its default in a
    multi-rank job: the shards' value ranges differ), the decoder's shared weights to accumulate both applications on
    every rank, and the flat bucket to average the 46 gradients."""
    world = 2
    port = _free_port()
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_policy_worker, args=(world, port, out), nprocs=world, join=True)
    assert torch.equal(out[0][0], out[1][0])
    from mm_masking_amd import train_icp_weights as trn
    model, p = _policy(seed=10)
    assert not model.global_minmax               # single process: nothing to reduce over
    model.train()
    opt = trn.make_optimizer(model, p)
    losses = [_policy_step(model, opt, list(range(4 * step, 4 * step + 4))) for step in range(2)]
    ref = torch.cat([q.detach().flatten() for q in model.parameters()])
    assert len(ref) == 1769905
    # BCE mean over 2 x (2 pairs) = mean over 4 pairs; Adam amplifies summation-order noise of near-zero gradients,
    # so compare the parameter UPDATE (lr 1e-4, two steps) rather than bits
    assert torch.allclose(out[0][0], ref, atol=2e-5), float((out[0][0] - ref).abs().max())
    assert abs(out[0][1][0] - losses[0]) < 1e-6 and abs(out[0][1][1] - losses[1]) < 1e-4
