"""Benchmark of the hot path: scan-pairs/s through one full training step
(GO-CFAR -> peak extraction -> polar->Cartesian -> mask U-Net (bf16) -> weight
sampling -> 10-iteration point-to-plane Huber dICP -> loss -> backward -> Adam) on
synthetic 400x3360 radar scans against 20k-point lidar submaps.

  python bench.py --gpus N --steps K --warmup W
  (N > 1: one rank per GPU over RCCL.  Under torch.distributed.run (WORLD_SIZE set) this process is a rank; started
   plainly with --gpus N > 1 it is the launcher: before any GPU call it spawns N fresh rank processes through
   ``python -m torch.distributed.run``, relays rank 0's JSON line and exits non-zero if any rank failed.)

Prints ONE JSON line on rank 0 (contract: task prompt / DESIGN.md §6):
  value      = whole-job scan-pairs/s, inputs resident in HBM before the timed region
  roofline   = the brute-force NN kernel against the HBM roofline named by
               BASELINE.json's north_star (plus the binding fp32 VALU roofline)
  cpu_baseline = the oracle's CPU port of the same step on the host cores (rank 0, N=1)
"""
import argparse
import ctypes
import json
import os
import sys
import time

# A rank of an RCCL job (one process per GPU): eight hardware queues instead of ROCm's four, so that the step's streams -- the
# caller's, the weight-gradient side stream, the gradient all-reduce's communication stream, RCCL's own -- do not share queues
# (0.2 ms per step in the one-rank rehearsal).  Before anything initialises HIP; an explicit setting
# wins; not for gloo rehearsals, where several ranks share one GPU and more queues per process means slower time-slicing.
if "WORLD_SIZE" in os.environ and os.environ.get("MMK_BENCH_BACKEND", "nccl") == "nccl":
    os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")


def usable_cores():
    """Host cores this process may actually use: affinity mask, capped by the cgroup
    CPU quota (a GPU box exposes every core of the host but grants a share of them)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return max(1, min(n, int(os.environ.get("MMK_BENCH_MAX_CORES", "16"))))


# must be set before torch / libgomp start their thread pools (a value the user exported is kept, and handed on to the ranks)
_USER_OMP = os.environ.get("OMP_NUM_THREADS")
os.environ.setdefault("OMP_NUM_THREADS", str(usable_cores()))

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0        # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
MFMA_BF16_PEAK_TFLOPS = 2500.0   # dense bf16 MFMA peak
UNET_GFLOP_FWD_PER_SAMPLE = 28.85  # SURVEY.md §8a Group U (hooked on the reference module); fwd+bwd = 3x
VALU_PEAK_TFLOPS = 157.3     # fp32 vector peak (spec)
N_PAD, M_VALID, M_PAD, DIM, ICP_ITERS = 5120, 20000, 20480, 2, 10


_T_START = time.time()


def progress(msg):
    """Progress lines on stderr (the JSON result is the only thing on stdout)."""
    if int(os.environ.get("RANK", "0")) == 0:
        print("[bench %7.1fs] %s" % (time.time() - _T_START, msg), file=sys.stderr, flush=True)


def launcher_command(argv, n, port, python=None):
    """The command the launcher starts: torch.distributed.run with one rank per GPU on this node, rendezvous on
    127.0.0.1 (the container's hostname may not resolve), followed by this script and its own arguments."""
    return [python or sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n),
            "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + list(argv)


def pick_result_line(stdout_text):
    """Rank 0's JSON line out of everything the ranks wrote to stdout (backends print banners there): the LAST line
    that parses as a JSON object with a "metric" key; None when there is none."""
    found = None
    for line in stdout_text.splitlines():
        line = line.strip()
        if not (line.startswith("{") and line.endswith("}")):
            continue
        try:
            obj = json.loads(line)
        except ValueError:
            continue
        if isinstance(obj, dict) and "metric" in obj:
            found = line
    return found


def launch_ranks(argv, n, runner=None):
    """--gpus N > 1 without WORLD_SIZE: be the launcher.  Nothing in this process has touched the GPU (no torch.cuda
    call, no HIP call): the ranks are fresh child processes, never an exec of a process that initialised the device.
    Returns the exit code: the children's when non-zero, 3 when they succeeded without printing a result line."""
    import socket
    import subprocess
    port = int(os.environ.get("MASTER_PORT", "0"))
    if port == 0:
        with socket.socket() as s:
            s.bind(("127.0.0.1", 0))
            port = s.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")      # dmabuf IPC only on this pool (RCCL needs it)
    # each of the n ranks gets its share of the host cores (this process's own default above is the whole host)
    env["OMP_NUM_THREADS"] = _USER_OMP if _USER_OMP is not None else str(max(1, usable_cores() // n))
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE"):
        env.pop(k, None)
    cmd = launcher_command(argv, n, port)
    progress("launcher: " + " ".join(cmd))
    run = runner or (lambda c, e: subprocess.run(c, env=e, stdout=subprocess.PIPE, text=True))
    proc = run(cmd, env)
    line = pick_result_line(proc.stdout or "")
    if proc.returncode != 0:
        sys.stderr.write("[bench] a rank failed: torch.distributed.run exited with %d\n" % proc.returncode)
        if proc.stdout:
            sys.stderr.write(proc.stdout[-4000:])
        return proc.returncode
    if line is None:
        sys.stderr.write("[bench] the ranks exited cleanly but rank 0 printed no result line\n")
        return 3
    # whatever else the ranks wrote to stdout (RCCL's NCCL_DEBUG=INFO lines, for one) goes to this process's stderr: stdout stays
    # the one result line
    for other in (proc.stdout or "").splitlines():
        if other.strip() and other.strip() != line.strip():
            sys.stderr.write("[rank stdout] " + other + "\n")
    print(line)
    sys.stdout.flush()
    return 0


def nn_algorithmic_bytes(B):
    """SURVEY.md §8d / BASELINE.md §4: 4*d*N + 4*d*M + 8*N bytes per pair per launch."""
    return B * (4 * DIM * N_PAD + 4 * DIM * M_PAD + 8 * N_PAD)


def cpu_baseline(params_like, sample_pairs=16, timed_steps=3, steps_8_threads=2):
    """The cpu_baseline leg — the only part of bench.py that touches oracle/: the oracle's
    PyTorch-CPU port of the same train step (oracle/train_ref.py) timed on a bounded sample:
    `sample_pairs` pairs per step (16 = the reference's train batch, train_icp_weights.py:374),
    1 warm-up + `timed_steps` steps on every usable core, then `steps_8_threads` steps on 8 threads (the
    reference's OMP_NUM_THREADS=8, scripts/setup_container.sh:25)."""
    from mm_masking_amd import synthetic
    from oracle import radar_ref, train_ref
    torch.set_num_threads(usable_cores())
    raw = synthetic.make_batch(list(range(sample_pairs)), device="cpu", m_valid=M_VALID, m_pad=M_PAD)
    fft = raw["fft_polar"].numpy()
    # CFAR / polar->Cartesian are outside the reference's step (cached by its Dataset):
    cfar = radar_ref.cfar_mask(fft, 0.0596, diff=False)
    pcs = radar_ref.extract_pc(cfar, 0.0596, raw["azimuths"].numpy(), raw["az_times"].numpy(), diff=False)
    pc = np.zeros((sample_pairs, N_PAD, 3), np.float32)
    for b, p in enumerate(pcs):
        pc[b, :min(len(p), N_PAD)] = p[:N_PAD]
    cart = radar_ref.radar_polar_to_cartesian_diff(fft, raw["azimuths"].numpy(), 0.0596)
    batch = {"fft_data": torch.from_numpy(cart), "raw_pc": torch.from_numpy(pc), "filtered_pc": torch.from_numpy(pc),
             "map_pc": raw["map_pc"], "T_init": raw["T_init"], "T_gt": raw["T_gt"]}
    step = train_ref.TrainStepRef(icp_type="pt2pl", loss_fn={"name": "huber", "metric": 1.0}, max_iter=ICP_ITERS,
                                  dim=DIM, dropout=0.05)
    progress("cpu_baseline: warm-up step (%d threads)" % torch.get_num_threads())
    step.step(batch)
    t0 = time.time()
    for i in range(timed_steps):
        step.step(batch)
        progress("cpu_baseline: timed step %d/%d done, %.1f s" % (i + 1, timed_steps, time.time() - t0))
    dt = time.time() - t0
    res = {"value": sample_pairs * timed_steps / dt, "unit": "pairs/s", "cores": torch.get_num_threads(),
           "kind": "port",
           "sample": "%d timed steps of B=%d after 1 warm-up (fp32 U-Net + 10-iter pt2pl Huber ICP fwd+bwd + Adam; "
                     "CFAR/polar->Cartesian outside the step as in the reference's cached Dataset), %.1f s"
                     % (timed_steps, sample_pairs, dt)}
    if steps_8_threads > 0 and torch.get_num_threads() != 8:
        n_all = torch.get_num_threads()
        torch.set_num_threads(8)
        os.environ["OMP_NUM_THREADS"] = "8"           # the C restatement of the NN search reads it per parallel region
        try:
            import ctypes as _ct
            _ct.CDLL("libgomp.so.1").omp_set_num_threads(8)
        except OSError:
            pass
        t0 = time.time()
        for i in range(steps_8_threads):
            step.step(batch)
            progress("cpu_baseline: 8-thread step %d/%d done, %.1f s" % (i + 1, steps_8_threads, time.time() - t0))
        dt8 = time.time() - t0
        res["value_8_threads"] = sample_pairs * steps_8_threads / dt8
        res["sample_8_threads"] = "%d timed steps of B=%d on 8 threads (the reference's OMP_NUM_THREADS), %.1f s" % (
            steps_8_threads, sample_pairs, dt8)
        torch.set_num_threads(n_all)
    return res


def conv_stack_rate(model, raw, params, device, reps=3):
    """MFMA leg of the measurement: the mask U-Net alone (forward + backward), HIP events on
    the current stream, against the dense bf16 MFMA peak."""
    from mm_masking_amd import train_icp_weights as trn
    batch = trn.prepare_batch(raw, params, max_loc_pts=N_PAD)
    B = batch["loc_data"]["fft_data"].shape[0]
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
    f_ms, b_ms = [], []
    for _ in range(reps + 1):
        model.zero_grad(set_to_none=True)
        ev[0].record()
        mask = model(batch["loc_data"], batch["map_data"], None, mask_only=True)
        ev[1].record()
        mask.sum().backward()
        ev[2].record()
        torch.cuda.synchronize(device)
        f_ms.append(ev[0].elapsed_time(ev[1]))
        b_ms.append(ev[1].elapsed_time(ev[2]))
    f, b = min(f_ms[1:]), min(b_ms[1:])
    h, w = batch["loc_data"]["fft_data"].shape[1:3]
    gf = UNET_GFLOP_FWD_PER_SAMPLE * (h * w) / (640.0 * 640.0)   # conv FLOPs scale with the pixel count
    tf = 3 * gf * B / (f + b)                                  # GFLOP / ms = TFLOP/s
    return {"bound": "mfma", "unet_fwd_ms": f, "unet_bwd_ms": b, "achieved": tf, "peak": MFMA_BF16_PEAK_TFLOPS,
            "unit": "TFLOP/s", "frac": tf / MFMA_BF16_PEAK_TFLOPS,
            "flop": "3 x %.2f GFLOP/sample (fwd + data grad + weight grad), B=%d" % (gf, B),
            "note": "layers with <= 16 channels at 640x640 are HBM-bound (arithmetic intensity ~70 FLOP/B)"}


def grid_engine_block(model, one_step, first, B, L, steps=5):
    """Side measurement (never `value`): the same step with the exact uniform-grid NN engine (identical
    correspondences), its launches timed with the same HIP-event hook."""
    from mm_masking_amd import _lib
    from mm_masking_amd.dICP.ICP import _IcpFunction
    engines = (model.ICP_alg.nn_search, model.ICP_alg_inference.nn_search)
    model.ICP_alg.nn_search = model.ICP_alg_inference.nn_search = "grid"
    try:
        for i in range(2):
            one_step(first + i)
        torch.cuda.synchronize()
        cap = steps * ICP_ITERS + 8
        _lib.check(L.mmk_nn_profile_begin(cap))
        t0 = time.perf_counter()
        for i in range(steps):
            one_step(first + 2 + i)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        ms = (ctypes.c_float * cap)()
        n_rec = ctypes.c_int32(0)
        _lib.check(L.mmk_nn_profile_end(ms, cap, ctypes.byref(n_rec)))
        g_ms = np.array(ms[:min(n_rec.value, cap)], dtype=np.float64)
        act = float(_IcpFunction.last_active[:ICP_ITERS].float().sum(dim=1).mean().item())
        avg_s = float(g_ms.mean()) * 1e-3
        alg = nn_algorithmic_bytes(1) * act
        return {"bound": "hbm", "kernel": "grid_nn_kernel<2>", "achieved": alg / avg_s / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": alg / avg_s / 1e9 / HBM_PEAK_GBS, "traffic": None, "avg_launch_us": avg_s * 1e6,
                "launches_timed": int(len(g_ms)), "active_pairs_per_launch": act, "algorithmic_bytes_per_launch": alg,
                "ms_per_step": dt / steps * 1e3, "pairs_per_s": B * steps / dt,
                "note": "same algorithmic bytes as the brute-force launch (the engine reads a binned copy of the target)"}
    finally:
        model.ICP_alg.nn_search, model.ICP_alg_inference.nn_search = engines


def timed_steps(fn, n, device):
    """HIP-event time of n calls of fn on the current stream, after one untimed call (ms per call)."""
    fn(0)
    torch.cuda.synchronize(device)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(n):
        fn(1 + i)
    e1.record()
    torch.cuda.synchronize(device)
    return e0.elapsed_time(e1) / n


def valid_scan_points(batch):
    """Non-padding rows of the scan clouds (the reference pads with all-zero rows: icp_weight_dataset.py:379-381)."""
    return (batch["loc_data"]["filtered_pc"] != 0).any(dim=-1).sum(dim=1).float()


def side_blocks(model, opt, lw, params, device, B, steps=5):
    """Side measurements on rank 0 at N=1 (never `value`):
      sparse_scene  the same step on the scenes of rounds 1-2 (2 450-2 970 valid scan points per pair)
      dim3          the same step with the SE(3) / 6x6 solve (params["icp_dim"] = 3, 3-D map; SURVEY.md §8d config 3 variant)
      inference     validate_policy-style forward (model.eval(), no_grad, ICP_alg_inference: 50 iterations, early exit;
                    train_icp_weights.py:71-177) at B = 16 (configs[1]) and B = 32, in pairs/s"""
    from mm_masking_amd import synthetic
    from mm_masking_amd import train_icp_weights as trn
    out = {}
    raws = [synthetic.make_batch(list(range(5000 + i * B, 5000 + (i + 1) * B)), device=device, m_valid=M_VALID, m_pad=M_PAD,
                                 density="sparse") for i in range(2)]

    def step_on(rs, prm):
        def f(i):
            trn.train_step(model, trn.prepare_batch(rs[i % len(rs)], prm, max_loc_pts=N_PAD), opt, lw, device)
        return f
    ms = timed_steps(step_on(raws, params), steps, device)
    out["sparse_scene"] = {"ms_per_step": ms, "pairs_per_s": B / ms * 1e3,
                           "scan_pts_valid_mean": float(valid_scan_points(trn.prepare_batch(raws[0], params, max_loc_pts=N_PAD)).mean()),
                           "note": "labelled side line: the round-1/2 scene density (synthetic.DENSITY['sparse'])"}
    del raws
    raws3 = [synthetic.make_batch(list(range(6000 + i * B, 6000 + (i + 1) * B)), device=device, m_valid=M_VALID, m_pad=M_PAD, dim=3)
             for i in range(2)]
    dim_was = model.icp_dim
    model.icp_dim = 3
    try:
        ms3 = timed_steps(step_on(raws3, params), steps, device)
    finally:
        model.icp_dim = dim_was
    out["dim3"] = {"ms_per_step": ms3, "pairs_per_s": B / ms3 * 1e3,
                   "note": "same step with icp_dim=3: SE(3) pose, 6x6 Gauss-Newton, 3-D nearest neighbour (nn_mfma_kernel<3>: two chained "
                           "MFMAs per 32 x 32 pairs) on a map with heights and tilted normals (synthetic.make_pair(dim=3))"}
    del raws3
    model.eval()
    try:
        inf = {}
        for b in (16, 32):
            rawi = [synthetic.make_batch(list(range(7000 + i * b, 7000 + (i + 1) * b)), device=device, m_valid=M_VALID, m_pad=M_PAD,
                                         dataset_type="val") for i in range(2)]
            its = []

            def f(i):
                batch = trn.prepare_batch(rawi[i % 2], params, max_loc_pts=N_PAD)
                with torch.no_grad():
                    T, _, _ = model(batch["loc_data"], batch["map_data"], batch["transforms"]["T_ml_init"])
                    trn.eval_validation_loss(T, batch["transforms"]["T_ml_gt"])
                its.append(model.ICP_alg_inference.last_iterations)
            msi = timed_steps(f, steps, device)
            inf["B%d" % b] = {"ms_per_batch": msi, "pairs_per_s": b / msi * 1e3,
                              "icp_iterations_run": [int(v) for v in its[1:] if v is not None]}
        inf["note"] = ("validate_policy path: prepare_batch + U-Net forward + extract_weights + ICP_alg_inference (differentiable=False, "
                       "max 50 iterations, tolerance 1e-5, early exit polled every 8 iterations) + eval_validation_loss; "
                       "T_init ~ N(0, 2 m / 0.6 rad) as the reference's validation split")
        out["inference"] = inf
    finally:
        model.train()
    return out


def pose_parity(model, params, device, pairs=2):
    """Part of the cpu_baseline leg: GPU dICP vs the CPU restatement on identical inputs
    (the oracle as checker, outside the timed region)."""
    from mm_masking_amd import synthetic
    from mm_masking_amd import train_icp_weights as trn
    from mm_masking_amd.dICP.ICP import ICP
    from oracle import dicp_ref
    raw = synthetic.make_batch(list(range(1000, 1000 + pairs)), device=device, m_valid=M_VALID, m_pad=M_PAD)
    batch = trn.prepare_batch(raw, params, max_loc_pts=N_PAD)
    src = batch["loc_data"]["filtered_pc"]
    w = torch.rand(src.shape[:2], device=device)
    lf = {"name": "huber", "metric": 1.0}
    icp = ICP("pt2pl", differentiable=True, max_iterations=ICP_ITERS, tolerance=1e-5)
    wg = w.clone().requires_grad_(True)
    T = icp.icp(src, raw["map_pc"], T_init=raw["T_init"], weight=wg, trim_dist=5.0, loss_fn=lf, dim=DIM)["T"]
    idx = T.grad_fn.saved_tensors[3].cpu().numpy()
    ref = dicp_ref.ICPRef("pt2pl", differentiable=False, max_iterations=ICP_ITERS, tolerance=1e-5)
    out = ref.icp(src.cpu(), raw["map_pc"].cpu(), T_init=raw["T_init"].cpu(), weight=w.cpu(), trim_dist=5.0,
                  loss_fn=lf, dim=DIM)
    mism = 0
    for k in range(out["num_iter"]):
        act = out["hist"]["active"][k].numpy()
        mism += int((idx[k][act] != out["hist"]["idx"][k].numpy()[act]).sum())
    Tg, Tr = T.detach().cpu().numpy(), out["T"].numpy()
    return {"nn_idx_mismatches": mism, "max_trans_err_m": float(np.abs(Tg[:, :2, 3] - Tr[:, :2, 3]).max()),
            "max_rot_err_rad": float(np.abs(np.arctan2(Tg[:, 1, 0], Tg[:, 0, 0]) - np.arctan2(Tr[:, 1, 0], Tr[:, 0, 0])).max()),
            "pairs": pairs, "iters": ICP_ITERS}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=32, help="scan pairs per GPU per step (BASELINE configs[2])")
    ap.add_argument("--distinct", type=int, default=2, help="distinct synthetic batches kept resident in HBM")
    ap.add_argument("--nn", choices=["brute", "grid"], default=None,
                    help="nearest-neighbour engine of the dICP (default: the dICP config, 'brute' = north_star's kernel)")
    ap.add_argument("--network-input", choices=["cartesian", "polar"], default="cartesian",
                    help="'polar' = the reference's network_input_type='polar' option (U-Net on the 400x3360 polar image, "
                         "SURVEY 8f.3): a side measurement, not the headline configuration")
    ap.add_argument("--settle", type=int, default=40, help="untimed settling steps before the warm-up (0 = none)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-grid", action="store_true", help="skip the side measurement of the exact grid NN engine")
    ap.add_argument("--no-parity", action="store_true")
    ap.add_argument("--no-side", action="store_true", help="skip the side measurements (sparse scenes, dim 3, inference)")
    ap.add_argument("--force-dist", action="store_true",
                    help="run the N-rank code path whatever N is: a process group is initialised (RCCL for --gpus 1 too), the gradient "
                         "all-reduce(s) and the global min-max all-reduce are issued, the line carries the `ddp` block")
    ap.add_argument("--no-overlap", action="store_true",
                    help="ONE gradient all-reduce between backward and step instead of the three bucket all-reduces that overlap "
                         "the backward pass (mm_masking_amd/ddp.py)")
    ap.add_argument("--local-minmax", action="store_true",
                    help="(diagnostic) keep the min-max normalisation per rank in a --force-dist / N-rank run: no collective in the forward pass")
    args = ap.parse_args()

    if (args.gpus > 1 or args.force_dist) and "WORLD_SIZE" not in os.environ:
        raise SystemExit(launch_ranks(sys.argv[1:], args.gpus))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d: launch with torch.distributed.run --nproc-per-node %d"
                         % (args.gpus, world, args.gpus))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the hot path is HIP kernels with no CPU fallback")
    device = torch.device("cuda", local_rank % torch.cuda.device_count())
    torch.cuda.set_device(device)
    dist_on = world > 1 or args.force_dist
    if dist_on:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        # "nccl" = RCCL over xGMI; MMK_BENCH_BACKEND=gloo only for rehearsing the rank logic on a 1-GPU box
        backend = os.environ.get("MMK_BENCH_BACKEND", "nccl")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=device)
        else:
            dist.init_process_group(backend)

    from mm_masking_amd import _lib, ddp, synthetic
    from mm_masking_amd import train_icp_weights as trn
    from mm_masking_amd.dICP.ICP import ICP
    from mm_masking_amd.icp_weight_policy import LearnICPWeightPolicy
    L = _lib.lib()
    if args.nn is not None:
        ICP.NN_SEARCH_OVERRIDE = args.nn

    params = trn.default_params(device)
    params.update({"icp_type": "pt2pl", "icp_loss_fn": {"name": "huber", "metric": 1.0}, "icp_dim": DIM,
                   "max_iter": ICP_ITERS, "dropout": 0.05})
    polar = args.network_input == "polar"
    if polar:
        # a (400,3360) mask cannot be compared with the 640x640 map-point image: that loss is off, as it
        # has to be upstream (train_icp_weights.py:223-226 would raise on the shape mismatch)
        params.update({"network_input_type": "polar", "network_output_type": "polar", "loss_map_pts_mask_weight": 0.0})
    lw = trn.loss_weights_from(params)
    torch.manual_seed(1234)
    model = LearnICPWeightPolicy(params).to(device)
    model.train()
    opt = trn.make_optimizer(model, params)
    if args.local_minmax:
        params["global_minmax"] = False
        model.global_minmax = False
    elif args.force_dist:
        params["global_minmax"] = True          # the 2C-float MAX all-reduce in front of the first layer, as in an N-rank job
        model.global_minmax = True
        from mm_masking_amd import unet_hip
        unet_hip.FORCE_COLLECTIVES = True
    sync = ddp.FlatGradSync(model, overlap=not args.no_overlap, force_collective=args.force_dist) if dist_on else None
    if sync is not None:
        sync.sync_params(0)

    # synthetic input, resident in HBM before anything is timed; rank r owns pairs r::world
    B = args.batch
    progress("generating %d synthetic pairs per rank" % (B * max(1, args.distinct)))
    raws = []
    for i in range(max(1, args.distinct)):
        idx = ddp.shard_indices(B * world, rank, world, start=i * B * world)
        raws.append(synthetic.make_batch(idx, device=device, m_valid=M_VALID, m_pad=M_PAD))

    def one_step(i):
        batch = trn.prepare_batch(raws[i % len(raws)], params, max_loc_pts=N_PAD)
        return trn.train_step(model, batch, opt, lw, device, grad_sync=sync)

    def fence():
        torch.cuda.synchronize(device)
        if dist_on:
            dist.barrier()
            torch.cuda.synchronize(device)

    # Untimed settling before the warm-up proper (a guard against start-up transients: allocator growth,
    # code-object loading): run until five consecutive steps are within 10 % of the fastest step seen,
    # at most `--settle` steps.
    if args.settle > 0:
        progress("settling (untimed, at most %d steps)" % args.settle)
        best, streak = float("inf"), 0
        # (with more than one rank every step holds a collective: all ranks must run the same number of
        # steps, so the count is fixed there instead of adaptive)
        n_settle = args.settle if not dist_on else min(args.settle, 25)
        for i in range(n_settle):
            t_s = time.perf_counter()
            one_step(i)
            torch.cuda.synchronize(device)
            d_s = time.perf_counter() - t_s
            best = min(best, d_s)
            streak = streak + 1 if d_s <= 1.10 * best else 0
            if not dist_on and i >= 5 and streak >= 5:
                break
    import gc
    gc.collect()
    gc.freeze()      # keep the collector from re-scanning the long-lived objects inside the timed region
    progress("warm-up: %d steps" % args.warmup)
    for i in range(args.warmup):
        one_step(i)
    fence()          # (nothing but this fence between the warm-up and the timed steps: an idle GPU drops its clocks)
    progress("timing %d steps" % args.steps)
    cap = args.steps * ICP_ITERS + 8
    _lib.check(L.mmk_nn_profile_begin(cap))
    if sync is not None:
        sync.timing, sync.exposed, sync.calls = [], [], 0
    step_ev = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps + 1)]
    ms0 = torch.cuda.memory_stats(device)
    t0 = time.perf_counter()
    step_ev[0].record()
    for i in range(args.steps):
        loss, _ = one_step(args.warmup + i)
        step_ev[i + 1].record()
    host_dt = time.perf_counter() - t0        # host-side enqueue time (the GPU may still be running)
    fence()
    dt = time.perf_counter() - t0
    per_step = [step_ev[i].elapsed_time(step_ev[i + 1]) for i in range(args.steps)]
    progress("per-step GPU ms: " + " ".join("%.1f" % v for v in per_step))
    progress("host time of the enqueue loop %.1f ms/step (with the launch queue full: it mostly waits for the GPU)" % (host_dt / args.steps * 1e3))
    ms1 = torch.cuda.memory_stats(device)
    progress("allocator during the timed region: %d device mallocs, %d frees, %d retries; reserved %.2f GiB" % (
        ms1.get("num_device_alloc", 0) - ms0.get("num_device_alloc", 0),
        ms1.get("num_device_free", 0) - ms0.get("num_device_free", 0),
        ms1.get("num_alloc_retries", 0) - ms0.get("num_alloc_retries", 0),
        ms1.get("reserved_bytes.all.current", 0) / 2 ** 30))
    ms = (ctypes.c_float * cap)()
    n_rec = ctypes.c_int32(0)
    _lib.check(L.mmk_nn_profile_end(ms, cap, ctypes.byref(n_rec)))
    nn_ms = np.array(ms[:min(n_rec.value, cap)], dtype=np.float64)
    ddp_block = None
    if dist_on:
        # what the process group saw, from the group itself: every rank contributes its own wall time and the mean
        # duration of its gradient all-reduces (events on the stream the collective is ordered on)
        ar_ms = [a.elapsed_time(b) for a, b in (sync.timing or [])]
        ex_ms = [a.elapsed_time(b) for a, b in (sync.exposed or [])]
        sync.timing = sync.exposed = None
        per_step = sync.calls / float(args.steps)
        mine = torch.tensor([dt, float(np.sum(ar_ms)) / args.steps if ar_ms else float("nan"), float(np.max(ar_ms)) if ar_ms else float("nan"),
                             float(np.mean(ex_ms)) if ex_ms else float("nan")], dtype=torch.float64, device=device)
        everyone = [torch.zeros_like(mine) for _ in range(dist.get_world_size())]
        dist.all_gather(everyone, mine)
        rows = torch.stack(everyone).cpu().numpy()
        dt = float(rows[:, 0].max())                # MAX over ranks, as the contract says
        ddp_block = {"backend": dist.get_backend(), "world_size_seen": dist.get_world_size(), "rank_count_reporting": int(len(rows)),
                     "allreduce_bytes": sync.allreduce_bytes(), "allreduce_calls_per_step": per_step,
                     "overlap": bool(sync.overlap), "bucket_elements": [hi - lo for lo, hi in (sync.buckets_last or [])],
                     "allreduce_ms": float(np.nanmean(rows[:, 1])), "allreduce_ms_max_over_ranks_and_steps": float(np.nanmax(rows[:, 2])),
                     "allreduce_exposed_ms": None if np.isnan(rows[:, 3]).all() else float(np.nanmean(rows[:, 3])),
                     "ms_per_step_min_over_ranks": float(rows[:, 0].min()) / args.steps * 1e3,
                     "ms_per_step_max_over_ranks": float(rows[:, 0].max()) / args.steps * 1e3,
                     "devices_visible": torch.cuda.device_count(), "global_minmax": bool(getattr(model, "global_minmax", False)),
                     "note": "fp32 sum all-reduce of the U-Net backward's own gradient block (mm_masking_amd/ddp.py): with overlap in the three "
                             "buckets in which the backward completes it (decoder + final layer, encoder blocks 3-5, encoder blocks 0-2), each "
                             "on a communication stream behind its own completion event; allreduce_ms = per-step sum of the event times "
                             "around the collectives (mean over ranks; it includes waiting for the slowest rank), allreduce_exposed_ms = time "
                             "the step's own stream waits for the communication stream before the optimizer (what the overlap leaves)"}

    # host cost of enqueueing one step, measured with an EMPTY launch queue (synchronise, time the Python call, synchronise):
    # inside the timed loop the host runs ahead until the queue is full and then waits for the GPU, so the loop's host
    # time says nothing about the host's own cost
    host_idle = []
    for i in range(5):
        torch.cuda.synchronize(device)
        t_h = time.perf_counter()
        one_step(args.warmup + args.steps + i)
        host_idle.append(time.perf_counter() - t_h)
    torch.cuda.synchronize(device)
    host_ms = float(np.median(host_idle)) * 1e3
    progress("host enqueue cost %.1f ms/step (empty queue, median of 5)" % host_ms)
    progress("timed region done: %.1f ms/step" % (dt / args.steps * 1e3))
    if len(nn_ms) == args.steps * ICP_ITERS:
        by_it = nn_ms.reshape(args.steps, ICP_ITERS).mean(axis=0) * 1e3
        progress("NN launch us by ICP iteration: " + " ".join("%.0f" % v for v in by_it))
    result = None
    if rank == 0:
        nn_avg_s = float(nn_ms.mean()) * 1e-3 if len(nn_ms) else float("nan")
        # frozen pairs (||delta|| < tolerance) skip the NN kernel: price a launch by the pairs that ran
        active_pairs = float(B)
        from mm_masking_amd.dICP.ICP import _IcpFunction
        if _IcpFunction.last_active is not None:
            act = _IcpFunction.last_active[:ICP_ITERS].float().sum(dim=1)       # (K,): active pairs per launch, last step
            active_pairs = float(act.mean().item())
        alg_bytes = nn_algorithmic_bytes(1) * active_pairs
        achieved = alg_bytes / nn_avg_s / 1e9
        # source rows a launch actually scans: blocks of 512 all-zero padding rows behind a pair's first zero row take that
        # row's result instead (csrc/mmk_icp.hip: src_zero_scan_kernel) -- the evaluation count follows the rows scanned
        scanned = float(N_PAD)
        if model.ICP_alg.nn_search == "brute":
            src_pts = trn.prepare_batch(raws[0], params, max_loc_pts=N_PAD)["loc_data"]["filtered_pc"]
            zero = (src_pts == 0).all(dim=-1)                                     # (B,N)
            first_zero = torch.where(zero.any(dim=1), zero.float().argmax(dim=1), torch.full((B,), N_PAD, device=device))
            blk_zero = zero.view(B, N_PAD // 512, 512).all(dim=2)
            starts = torch.arange(0, N_PAD, 512, device=device).view(1, -1)
            skipped = blk_zero & (first_zero.view(-1, 1) < starts)
            scanned = float((N_PAD - 512 * skipped.sum(dim=1)).float().mean().item())
        evals = active_pairs * scanned * M_PAD       # distance evaluations per launch
        # HBM bytes per launch from the PMC counters: NOT measured in this run -- read from a committed file that a separate
        # rocprofv3 --pmc pass over the dICP alone wrote (scripts/pmc_nn.sh -> profiles/r05_nn_traffic.json, FETCH_SIZE and
        # WRITE_SIZE in passes of their own, FETCH_SIZE doubled per the gfx950 correction); null when no such file exists.
        # `traffic_source` says so in the line itself.
        traffic, traffic_source = None, None
        for name in ("r05_nn_traffic.json", "r04_nn_traffic.json", "r03_nn_traffic.json", "r02_nn_traffic.json"):
            tr_file = os.path.join(ROOT, "profiles", name)
            if os.path.exists(tr_file) and model.ICP_alg.nn_search == "brute":
                tj = json.load(open(tr_file))
                traffic = tj.get("hbm_bytes_per_launch")
                traffic_source = ("committed PMC pass profiles/%s (dICP alone, %s us per launch there, scene density %s); not "
                                  "collected in this run" % (name, tj.get("avg_launch_us", "?"), tj.get("density", "sparse (round 2)")))
                break
        valid = valid_scan_points(trn.prepare_batch(raws[0], params, max_loc_pts=N_PAD))
        # bytes a launch actually reads + writes: the scanned source rows (x,y), the target planes, one (idx, d2) per row
        alg_bytes_read = active_pairs * (4 * DIM * scanned + 4 * DIM * M_PAD + 8 * N_PAD)
        if model.ICP_alg.nn_search == "grid":
            nn_kernel = "grid_nn_kernel<2>"
        else:
            nn_kernel = "nn_mfma_kernel<2>"
        # every pair is one element of a v_mfma_f32_32x32x16_bf16 result tile: 1 024 pairs per instruction, one instruction
        # per 32 cycles and SIMD (MI355X_MICROARCH.md, cycle constants), 1 024 SIMDs at 2.4 GHz
        peak_pairs = 1024 * 1024 / 32.0 * 2.4e9
        binding = {"pipe": "bf16 matrix cores (pair pricing) + fp32 VALU (chunk minima, exact re-scan)",
                   "pair_evals_per_s": evals / nn_avg_s, "peak_pair_evals_per_s": peak_pairs,
                   "frac": evals / nn_avg_s / peak_pairs,
                   "equivalent_dense_bf16_tflops": evals / nn_avg_s * 32 / 1e12,
                   "note": "16 k-slots (2 x 16 flop) per pair: exact three-way bf16 splits of -2p, t and |t|^2, fp32 accumulation; "
                           "1.1 vector instructions per 64 pairs beside the MFMAs (profiles/r05_nn_pmc_counters.json; the fp32 VALU "
                           "kernel of rounds 1-2, removed in round 5, issued 3.3)"}
        result = {
            "metric": "scan-pairs/s (mask-CNN + 10-iter dICP fwd+bwd)", "value": B * world * args.steps / dt,
            "unit": "pairs/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "bf16", "data": "synthetic", "host_enqueue_ms_per_step": host_ms,
            "config": {"workload": ("BASELINE configs[2]: train_icp_weights step fwd+bwd, batch=%d per GPU, 10 dICP "
                                    "iters, point-to-plane Huber, dim=2; 400x3360 polar radar -> GO-CFAR + peaks + "
                                    "polar->Cartesian 640x640 -> U-Net (hand-written NHWC bf16 MFMA kernels, fp32 masters) -> dICP (fp32 points, "
                                    "fp64 normal equations) -> rot+trans+mask_pts loss -> Adam" % B) if not polar else
                                   ("NOT the headline configuration: network_input_type='polar' variant of configs[2] (U-Net on the "
                                    "400x3360 polar image, 3.3x the pixels; rot+trans loss), batch=%d per GPU" % B),
                       "batch_per_gpu": B, "global_batch": B * world, "scan_pts_pad": N_PAD, "map_pts": M_VALID,
                       "map_pts_pad": M_PAD, "icp_iters": ICP_ITERS, "parallelism": "dp%d" % world,
                       "scene_density": "survey (SURVEY.md 8d: 3 500-5 000 valid scan points of 5 120)",
                       "scan_pts_valid_mean": float(valid.mean()), "scan_pts_valid_min": float(valid.min()),
                       "scan_pts_valid_max": float(valid.max()),
                       "final_loss": float(loss)},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_source,
                         "kernel": nn_kernel,
                         "nn_engine": model.ICP_alg.nn_search, "launches_timed": int(len(nn_ms)),
                         "avg_launch_us": nn_avg_s * 1e6, "algorithmic_bytes_per_launch": alg_bytes,
                         "algorithmic_bytes_rows_read_per_launch": alg_bytes_read,
                         "achieved_rows_read": alg_bytes_read / nn_avg_s / 1e9,
                         "active_pairs_per_launch": active_pairs, "source_rows_scanned_per_pair": scanned,
                         "note": "north_star names the HBM roofline; brute force does %.3g distance evaluations per "
                                 "launch over those bytes (~2.2 kFLOP/B), so what binds is the arithmetic pipe that prices the "
                                 "pairs, not HBM" % evals,
                         "binding": binding},
        }
        if ddp_block is not None:
            result["ddp"] = ddp_block
        if world == 1:
            result["conv_stack"] = conv_stack_rate(model, raws[0], params, device)
            if model.ICP_alg.nn_search == "brute" and not args.no_grid:
                result["roofline_grid"] = grid_engine_block(model, one_step, args.warmup + args.steps, B, L)
            if not args.no_side and not polar:
                progress("side measurements: sparse scenes, dim 3, inference")
                result.update(side_blocks(model, opt, lw, params, device, B))
        if world == 1 and not args.no_cpu_baseline:
            result["cpu_baseline"] = cpu_baseline(params)
            if not args.no_parity:
                progress("cpu_baseline: pose parity of the GPU dICP vs the CPU restatement")
                result["cpu_baseline"]["pose_parity"] = pose_parity(model, params, device)
        print(json.dumps(result))
        sys.stdout.flush()
    if dist_on:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
