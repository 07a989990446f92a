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
        res = {}
        for overlap in (False, True):             # one all-reduce of the block / the three buckets of the native backward
            model, p = _policy(seed=10 + rank)    # different init per rank on purpose: rank 0's is broadcast
            assert model.global_minmax            # the default inside a multi-rank job
            model.train()
            sync = ddp.FlatGradSync(model, overlap=overlap)
            sync.sync_params(0)
            opt = trn.make_optimizer(model, p)
            losses = []
            for step in range(2):
                losses.append(_policy_step(model, opt, ddp.shard_indices(4, rank, world, start=4 * step), sync))
            lt = torch.tensor(losses, dtype=torch.float64)
            dist.all_reduce(lt)
            res[overlap] = (torch.cat([q.detach().flatten() for q in model.parameters()]), (lt / world).tolist(), sync.calls,
                            sync.buckets_last)
        out[rank] = res
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
    both = out[0], out[1]
    out = {r: both[r][False] for r in range(2)}
    assert torch.equal(out[0][0], out[1][0])
    # the bucketed form (three all-reduces per step, in the order the native backward completes its gradients: decoder + final
    # layer, encoder blocks 3-5, encoder blocks 0-2) gives the single all-reduce's result bit for bit, on every rank
    for r in range(2):
        flat_b, losses_b, calls_b, ranges = both[r][True]
        assert torch.equal(flat_b, out[r][0]) and losses_b == out[r][1]
        assert calls_b == 6 and both[r][False][2] == 2 and both[r][False][3] == [(0, 1769905)]
        assert len(ranges) == 3 and sorted(ranges) == [ranges[2], ranges[1], ranges[0]]           # tail, middle, head
        assert ranges[2][0] == 0 and ranges[2][1] == ranges[1][0] and ranges[1][1] == ranges[0][0] and ranges[0][1] == 1769905
        assert ranges[2][1] - ranges[2][0] < 20000 < ranges[0][1] - ranges[0][0] < ranges[1][1] - ranges[1][0]
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


# ------------------------------------------------------------------------------------------------------------------
# fit(): every collective of the epoch loop pairs up across the ranks (the final validation used to run on
# the main rank alone while a forward holds the global min-max all-reduce)
class _Batches:
    """A fixed list of (scan, map, T) batches of `hw` x `hw` images, different per rank."""

    def __init__(self, rank, n_batches, first, hw=32, b=2):
        self.items = []
        for k in range(n_batches):
            idx = [first + 2 * (k * b + j) + rank for j in range(b)]
            img = torch.cat([_scan(i, hw)[0] for i in idx])
            T = torch.eye(4).repeat(b, 1, 1)
            T[:, 0, 3] = torch.tensor([0.1 * (i % 5) for i in idx])
            self.items.append({"loc_data": {"fft_data": img, "fft_cfar": (img > img.mean(dim=(1, 2), keepdim=True)).float(), "raw_pc": torch.rand(b, 4, 3),
                                            "filtered_pc": torch.rand(b, 4, 3)},
                               "map_data": {"pc": torch.rand(b, 4, 6)},
                               "transforms": {"T_ml_init": T, "T_ml_gt": torch.eye(4).repeat(b, 1, 1)}})

    def __len__(self):
        return len(self.items)

    def __iter__(self):
        return iter(self.items)


def _fit_worker(rank, world, port, out, ckpt_dir):
    os.environ.update({"MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(port)})
    torch.set_num_threads(2)
    from mm_masking_amd import icp_weight_policy as pol
    from mm_masking_amd import train_icp_weights as trn
    # the model exists BEFORE the process group does: the global min-max default must still resolve to "on"
    model, p = _policy(seed=20 + rank)
    p.update({"num_epochs": 2, "loss_cfar_mask_weight": 1.0, "loss_map_pts_mask_weight": 0.0})
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        assert model.global_minmax
        n_forward = [0]

        # the dICP / extract_weights kernels have no CPU path: host-logic stand-ins with the same shapes (the test is
        # about which rank enters which collective, not about their arithmetic)
        def stats_stub(mask, pts):
            w = mask.mean(dim=(1, 2)).unsqueeze(1).expand(-1, pts.shape[1])
            z = torch.zeros(())
            return (w, z, 0.0, 0.0, 0.0, 0.0), [z] * 6

        def icp_stub(scan_pc, map_pc, T_init, weights):
            n_forward[0] += 1
            return T_init * (1.0 + 0.01 * weights.mean())
        pol._extract_weights_stats = stats_stub
        model.icp = icp_stub
        sync = ddp.FlatGradSync(model)
        sync.sync_params(0)
        opt = trn.make_optimizer(model, p)
        hist = trn.fit(model, _Batches(rank, 2, 0), _Batches(rank, 1, 100), opt, p, ckpt_dir, grad_sync=sync,
                       is_main=(rank == 0), log=lambda *_: None)
        flat = torch.cat([q.detach().flatten() for q in model.parameters()])
        out[rank] = (flat, hist.get("final_acc"), n_forward[0], sorted(os.listdir(ckpt_dir)) if rank == 0 else None,
                     (hist["best_norm"], hist["acc"]))
    finally:
        dist.destroy_process_group()


def test_fit_two_ranks_final_validation_on_all_ranks(tmp_path):
    """fit() on 2 gloo ranks with is_main = (rank == 0): returns on both ranks (no rank is left alone in a collective),
    both run the same number of forwards incl. the final validation, both end with the best policy's parameters, and
    only the main rank wrote files."""
    world = 2
    port = _free_port()
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_fit_worker, args=(world, port, out, str(tmp_path)), nprocs=world, join=True)
    assert out[0][2] == out[1][2] > 0
    assert out[0][1] is not None and out[1][1] is not None          # final validation ran on both
    # the validation metric is the mean over the ranks' shards (the reference selects on the whole validation set,
    # train_icp_weights.py:534-537): every rank holds the same history and takes the same best-policy decisions
    assert out[0][4] == out[1][4] and out[0][1] == out[1][1]
    assert torch.equal(out[0][0], out[1][0])                        # identical (best) weights everywhere
    best = torch.load(os.path.join(str(tmp_path), "best_policy.pt"), weights_only=True)
    # best_policy.pt holds the state_dict in module order: same order as parameters()
    ref = torch.cat([v.flatten() for k, v in best.items()])
    assert torch.equal(out[0][0], ref)
    assert {"best_policy.pt", "epoch_0.pt", "epoch_1.pt", "resume.pt"} <= set(out[0][3])


# ------------------------------------------------------------------------------------------------------------------
# bench.py's launcher (python bench.py --gpus N without WORLD_SIZE): argument / exit-code logic, no GPU, no children
def _bench():
    import importlib.util
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("bench_under_test", os.path.join(root, "bench.py"))
    mod = importlib.util.module_from_spec(spec)
    sys.modules["bench_under_test"] = mod
    spec.loader.exec_module(mod)
    return mod


class _Proc:
    def __init__(self, rc, out):
        self.returncode, self.stdout = rc, out


def test_bench_launcher_command_and_exit_codes(capsys, monkeypatch):
    bench = _bench()
    cmd = bench.launcher_command(["--gpus", "4", "--steps", "7"], 4, 29517, python="py")
    assert cmd[:4] == ["py", "-m", "torch.distributed.run", "--nnodes=1"]
    assert cmd[cmd.index("--nproc-per-node") + 1] == "4"
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and cmd[cmd.index("--master-port") + 1] == "29517"
    assert cmd[-5].endswith("bench.py") and cmd[-4:] == ["--gpus", "4", "--steps", "7"]
    # rank 0's line is found behind backend banners; other JSON-looking lines without "metric" are not it
    good = '{"metric": "scan-pairs/s", "value": 1.0, "n_gpus": 2}'
    text = "Gloo banner: connected\n{\"not\": \"it\"}\n" + good + "\ntrailing noise\n"
    assert bench.pick_result_line(text) == good
    assert bench.pick_result_line("nothing here\n") is None
    monkeypatch.setenv("WORLD_SIZE", "7")       # must not leak into the children
    seen = {}

    def runner_ok(c, e):
        seen["cmd"], seen["env"] = c, e
        return _Proc(0, text)
    assert bench.launch_ranks(["--gpus", "2"], 2, runner=runner_ok) == 0
    assert capsys.readouterr().out.strip() == good                  # exactly one line on stdout
    assert "WORLD_SIZE" not in seen["env"] and seen["env"]["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"
    assert seen["cmd"][seen["cmd"].index("--nproc-per-node") + 1] == "2"
    # a failing rank: its exit code is relayed and nothing goes to stdout
    assert bench.launch_ranks(["--gpus", "2"], 2, runner=lambda c, e: _Proc(17, "partial\n")) == 17
    assert capsys.readouterr().out == ""
    # clean exit without a result line is an error too
    assert bench.launch_ranks(["--gpus", "2"], 2, runner=lambda c, e: _Proc(0, "no json\n")) == 3
