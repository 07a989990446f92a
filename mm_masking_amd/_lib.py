"""ctypes binding of libmmk_hip.so (the C ABI declared in include/mmk.h).

This is the stub a maintainer of the reference would add: the
reference has no FFI of its own, so every entry point replaces a *Python* call
site of mm_masking (cited in include/mmk.h).  PyTorch is only the owner of the
device buffers and of the HIP stream the kernels are enqueued on.

The library is mandatory: there is NO CPU or PyTorch fallback.  ``lib()`` raises
if the shared object is missing or fails to load, and every wrapper raises if a
tensor is not a contiguous tensor of the expected dtype on a HIP device.
"""
import ctypes
import os
import sys
import subprocess

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
_ROOT = os.path.dirname(_HERE)
SO_PATH = os.environ.get("MMK_LIB", os.path.join(_HERE, "libmmk_hip.so"))   # MMK_LIB: A/B another build (development)
SOURCES = ["mmk_api.hip", "mmk_icp.hip", "mmk_radar.hip", "mmk_unet.hip", "mmk_unet_driver.hip", "mmk_loader.hip", "mmk_loss.hip"]

_lib = None

c_f32p = ctypes.c_void_p
c_vp = ctypes.c_void_p


class MmkError(RuntimeError):
    pass


class ReadJob(ctypes.Structure):
    """mmk_read_job of include/mmk.h (one row-copy of the batched loader)."""
    _fields_ = [("path", ctypes.c_char_p), ("header_bytes", ctypes.c_int64), ("rows", ctypes.c_int32),
                ("row_bytes", ctypes.c_int32), ("col0", ctypes.c_int32), ("ncols", ctypes.c_int32), ("roll", ctypes.c_int32),
                ("reserved", ctypes.c_int32), ("dst", ctypes.c_void_p)]


class IcpParams(ctypes.Structure):
    """mmk_icp_params of include/mmk.h."""
    _fields_ = [("B", ctypes.c_int32), ("N", ctypes.c_int32), ("M", ctypes.c_int32),
                ("tgt_cols", ctypes.c_int32), ("dim", ctypes.c_int32), ("icp_type", ctypes.c_int32),
                ("loss", ctypes.c_int32), ("loss_k", ctypes.c_float), ("trim_dist", ctypes.c_float),
                ("tolerance", ctypes.c_float), ("max_iter", ctypes.c_int32), ("save_state", ctypes.c_int32),
                ("check_every", ctypes.c_int32), ("nn_method", ctypes.c_int32)]


class ConvDesc(ctypes.Structure):
    """mmk_conv_desc of include/mmk.h."""
    _fields_ = [("x1", ctypes.c_void_p), ("x2", ctypes.c_void_p), ("C1", ctypes.c_int32), ("C2", ctypes.c_int32),
                ("wpack", ctypes.c_void_p), ("bias", ctypes.c_void_p),
                ("y1", ctypes.c_void_p), ("relu_src1", ctypes.c_void_p), ("O1", ctypes.c_int32),
                ("accumulate1", ctypes.c_int32), ("scale1", ctypes.c_float),
                ("y2", ctypes.c_void_p), ("relu_src2", ctypes.c_void_p), ("O2", ctypes.c_int32),
                ("accumulate2", ctypes.c_int32), ("scale2", ctypes.c_float),
                ("B", ctypes.c_int32), ("H", ctypes.c_int32), ("W", ctypes.c_int32), ("relu", ctypes.c_int32),
                ("leaky_slope", ctypes.c_float), ("drop_p", ctypes.c_float), ("seed", ctypes.c_uint32), ("pool_y", ctypes.c_void_p),
                ("pool_arg", ctypes.c_void_p)]


class UNetDesc(ctypes.Structure):
    """mmk_unet_desc of include/mmk.h."""
    _fields_ = [("B", ctypes.c_int32), ("H", ctypes.c_int32), ("W", ctypes.c_int32), ("cin", ctypes.c_int32),
                ("x", ctypes.c_void_p), ("pre", ctypes.c_void_p), ("params", ctypes.POINTER(ctypes.c_void_p)),
                ("drop_p", ctypes.c_float), ("seed", ctypes.c_uint32), ("leaky_slope", ctypes.c_float), ("norm", ctypes.c_int32),
                ("workspace", ctypes.c_void_p), ("workspace_bytes", ctypes.c_size_t), ("mask", ctypes.c_void_p),
                ("keep_full_res", ctypes.c_int32)]


FINAL_BWD_WS_FLOATS = 16384        # MMK_FINAL_BWD_WS_FLOATS of include/mmk.h
ICP_TYPES = {"pt2pt": 0, "pt2pl": 1}
LOSSES = {None: 0, "none": 0, "l2": 0, "cauchy": 1, "huber": 2}
NN_METHODS = {"brute": 0, "grid": 1}


def build(verbose=False):
    """Compile the HIP sources for gfx950 into mm_masking_amd/libmmk_hip.so (hipcc cross-compiles without a GPU): one
    object per source (in parallel, only the stale ones), then one link.  -ffp-contract=off is part of the numerical
    contract."""
    from concurrent.futures import ThreadPoolExecutor
    csrc = os.path.join(_HERE, "csrc")
    srcs = [os.path.join(csrc, s) for s in SOURCES]
    common = [os.path.join(csrc, "mmk_common.h"), os.path.join(csrc, "mmk_unet_shared.h"), os.path.join(_ROOT, "include", "mmk.h")]
    incs = [os.path.join(csrc, f) for f in sorted(os.listdir(csrc)) if f.endswith(".inc")]
    extra = {"mmk_unet.hip": incs, "mmk_loader.hip": incs}
    objdir = os.path.join(csrc, "_obj")
    os.makedirs(objdir, exist_ok=True)
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    flags = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-I", os.path.join(_ROOT, "include")]
    # mmk_unet.hip: MFMA results in VGPRs instead of AGPRs.  The thin-layer kernels are bound by vector-instruction issue and the
    # compiler's default parks their accumulators in AGPRs, which vector instructions cannot read: every output value then costs
    # a v_accvgpr_read (4 184 of them in the file, 96 with the option; same arithmetic, same or better occupancy; HISTORY.md 10.17)
    per_file = {"mmk_unet.hip": ["-mllvm", "-amdgpu-mfma-vgpr-form"]}
    jobs, objs = [], []
    for src in srcs:
        obj = os.path.join(objdir, os.path.basename(src)[:-4] + ".o")
        objs.append(obj)
        deps = [src] + common + extra.get(os.path.basename(src), []) + [os.path.abspath(__file__)]
        if not (os.path.exists(obj) and all(os.path.getmtime(obj) >= os.path.getmtime(d) for d in deps)):
            jobs.append([hipcc] + flags + per_file.get(os.path.basename(src), []) + ["-c", src, "-o", obj])
    if not jobs and os.path.exists(SO_PATH) and all(os.path.getmtime(SO_PATH) >= os.path.getmtime(o) for o in objs):
        return SO_PATH

    def run(cmd):
        if verbose:
            print(" ".join(cmd), flush=True)
        try:
            subprocess.check_call(cmd)
        except subprocess.CalledProcessError:
            # a hipcc that does not know a backend option of per_file: same code without it (slower kernels, same results)
            plain = [c for c in cmd if c not in sum(per_file.values(), [])]
            if len(plain) == len(cmd):
                raise
            print("libmmk_hip: retrying without %s" % " ".join(sorted(set(cmd) - set(plain))), file=sys.stderr, flush=True)
            subprocess.check_call(plain)
    with ThreadPoolExecutor(max_workers=max(1, min(len(jobs), os.cpu_count() or 1))) as ex:
        list(ex.map(run, jobs))
    run([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", SO_PATH] + objs)
    return SO_PATH


def _declare(lib):
    i32, f32, sz = ctypes.c_int32, ctypes.c_float, ctypes.c_size_t
    P = ctypes.POINTER(IcpParams)
    sig = {
        "mmk_version": (ctypes.c_int, []),
        "mmk_last_error": (ctypes.c_char_p, []),
        "mmk_icp_workspace_bytes": (sz, [P]),
        "mmk_icp_forward": (ctypes.c_int, [P] + [c_vp] * 10 + [c_vp, sz, ctypes.POINTER(ctypes.c_int), c_vp]),
        "mmk_icp_backward": (ctypes.c_int, [P] + [c_vp] * 11 + [c_vp, sz, c_vp]),
        "mmk_icp_status": (ctypes.c_int, [P, c_vp, sz, c_vp, c_vp]),
        "mmk_pose_loss_fwd": (ctypes.c_int, [c_vp, i32, c_vp, c_vp]),
        "mmk_pose_loss_bwd": (ctypes.c_int, [c_vp, i32, c_vp, c_vp, c_vp, c_vp]),
        "mmk_bce_ws_bytes": (sz, []),
        "mmk_bce_mean_fwd": (ctypes.c_int, [c_vp, c_vp, ctypes.c_int64, c_vp, sz, c_vp, c_vp]),
        "mmk_bce_mean_bwd": (ctypes.c_int, [c_vp, c_vp, ctypes.c_int64, c_vp, c_vp, c_vp]),
        "mmk_icp_partials_count": (sz, [P]),
        "mmk_icp_accumulate": (ctypes.c_int, [P] + [c_vp] * 8 + [c_vp]),
        "mmk_icp_solve_update": (ctypes.c_int, [P] + [c_vp] * 7 + [c_vp]),
        "mmk_nn_padded_m": (i32, [i32]),
        "mmk_pack_target": (ctypes.c_int, [c_vp, i32, i32, i32, i32, c_vp, c_vp]),
        "mmk_nn_workspace_bytes": (sz, [i32, i32, i32, i32]),
        "mmk_nn_search": (ctypes.c_int, [c_vp, c_vp, c_vp, i32, i32, i32, i32, c_vp, c_vp, c_vp, sz, c_vp]),
        "mmk_nn_profile_begin": (ctypes.c_int, [i32]),
        "mmk_nn_profile_end": (ctypes.c_int, [ctypes.POINTER(ctypes.c_float), i32, ctypes.POINTER(ctypes.c_int32)]),
        "mmk_conv3x3_packed_elems": (sz, [i32, i32, i32]),
        "mmk_conv3x3_pack_weights": (ctypes.c_int, [c_vp, i32, i32, i32, c_vp, c_vp]),
        "mmk_conv3x3_pack_weights_batch": (ctypes.c_int, [i32, c_vp, c_vp, c_vp, i32, c_vp, c_vp]),
        "mmk_conv3x3": (ctypes.c_int, [ctypes.POINTER(ConvDesc), c_vp]),
        "mmk_conv3x3_wgrad_unpack_batch": (ctypes.c_int, [i32, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp]),
        "mmk_conv3x3_wgrad_slices": (i32, [i32, i32, i32, i32, i32, i32]),
        "mmk_conv3x3_wgrad_partial": (ctypes.c_int, [c_vp, c_vp, i32, i32, c_vp, i32, i32, i32, i32, c_vp, i32, c_vp]),
        "mmk_conv_bwd_fused": (ctypes.c_int, [c_vp, c_vp, c_vp, ctypes.c_float, i32, i32, i32, i32, c_vp, c_vp, i32, c_vp]),
        "mmk_conv8x16_bwd_fused": (ctypes.c_int, [c_vp, c_vp, c_vp, ctypes.c_float, i32, i32, i32, c_vp, c_vp, i32, c_vp]),
        "mmk_conv16x8_bwd_fused": (ctypes.c_int, [c_vp, c_vp, c_vp, c_vp, ctypes.c_float, i32, i32, i32, c_vp, c_vp, c_vp, i32, c_vp]),
        "mmk_conv3x3_pool_fusable": (ctypes.c_int32, [i32, i32, i32, i32, i32]),
        "mmk_channel_minmax": (ctypes.c_int, [c_vp, i32, i32, ctypes.c_int64, c_vp, c_vp, c_vp, c_vp]),
        "mmk_conv_first": (ctypes.c_int, [c_vp, i32, c_vp, c_vp, c_vp, i32, i32, i32, f32, c_vp, c_vp]),
        "mmk_conv_first_wgrad_ws_bytes": (sz, [i32]),
        "mmk_conv_first_wgrad": (ctypes.c_int, [c_vp, i32, c_vp, c_vp, i32, i32, i32, c_vp, c_vp, c_vp, sz, c_vp]),
        "mmk_maxpool2_fwd": (ctypes.c_int, [c_vp, i32, i32, i32, i32, c_vp, c_vp]),
        "mmk_maxpool2_bwd": (ctypes.c_int, [c_vp, c_vp, i32, i32, i32, i32, f32, f32, c_vp, c_vp]),
        "mmk_maxpool2_fwd_arg": (ctypes.c_int, [c_vp, i32, i32, i32, i32, c_vp, c_vp, c_vp]),
        "mmk_maxpool2_bwd_arg": (ctypes.c_int, [c_vp, c_vp, i32, i32, i32, i32, f32, c_vp, c_vp]),
        "mmk_upsample_fwd": (ctypes.c_int, [c_vp, i32, i32, i32, i32, i32, i32, c_vp, c_vp]),
        "mmk_upsample_bwd": (ctypes.c_int, [c_vp, i32, i32, i32, i32, i32, i32, c_vp, f32, f32, c_vp, c_vp]),
        "mmk_final_fwd": (ctypes.c_int, [c_vp, c_vp, c_vp, ctypes.c_int64, c_vp, c_vp]),
        "mmk_final_bwd": (ctypes.c_int, [c_vp, c_vp, c_vp, c_vp, ctypes.c_int64, f32, f32, c_vp, c_vp, c_vp, c_vp, c_vp]),
        "mmk_mask_normalize": (ctypes.c_int, [c_vp, i32, ctypes.c_int64, c_vp, c_vp, c_vp, c_vp]),
        "mmk_final_bwd_normalized": (ctypes.c_int, [c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, i32, ctypes.c_int64, f32, f32, c_vp, c_vp,
                                                    c_vp, c_vp, c_vp, c_vp, c_vp]),
        "mmk_bn_forward_stats": (ctypes.c_int, [c_vp, ctypes.c_int64, i32, c_vp, c_vp, f32, f32, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp]),
        "mmk_bn_apply": (ctypes.c_int, [c_vp, ctypes.c_int64, i32, c_vp, f32, ctypes.c_uint32, c_vp, c_vp]),
        "mmk_bn_backward": (ctypes.c_int, [c_vp, c_vp, f32, c_vp, ctypes.c_int64, i32, c_vp, c_vp, c_vp, f32, i32, c_vp, c_vp, c_vp,
                                           c_vp, c_vp, c_vp]),
        "mmk_unet_workspace_bytes": (sz, [i32, i32, i32, i32]),
        "mmk_unet_scratch_bytes": (sz, [i32, i32, i32, i32]),
        "mmk_unet_forward": (ctypes.c_int, [ctypes.POINTER(UNetDesc), c_vp]),
        "mmk_unet_backward": (ctypes.c_int, [ctypes.POINTER(UNetDesc), c_vp, ctypes.POINTER(ctypes.c_void_p), c_vp, sz, c_vp]),
        "mmk_unet_backward_buckets": (ctypes.c_int, [ctypes.POINTER(UNetDesc), c_vp, ctypes.POINTER(ctypes.c_void_p), c_vp, sz,
                                                     ctypes.POINTER(ctypes.c_void_p), c_vp]),
        "mmk_unet_grad_bucket": (i32, [i32, ctypes.POINTER(i32), ctypes.POINTER(i32)]),
        "mmk_unet_tensor": (ctypes.c_int, [i32, i32, i32, i32, i32, ctypes.POINTER(sz), ctypes.POINTER(i32), ctypes.POINTER(i32),
                                           ctypes.POINTER(i32)]),
        "mmk_cfar_mask": (ctypes.c_int, [c_vp, i32, i32, i32, i32, i32, i32, i32, f32, f32, i32, f32, c_vp, c_vp]),
        "mmk_extract_peaks_workspace_bytes": (sz, [i32, i32, i32, i32]),
        "mmk_extract_peaks": (ctypes.c_int, [c_vp, i32, i32, i32, f32, c_vp, c_vp, c_vp, i32, f32, i32, c_vp, c_vp,
                                             c_vp, sz, c_vp]),
        "mmk_polar_to_cart": (ctypes.c_int, [c_vp, c_vp, c_vp, c_vp, i32, i32, i32, i32, f32, i32, i32, c_vp, c_vp]),
        "mmk_polar_to_cart_pair": (ctypes.c_int, [c_vp, c_vp, c_vp, c_vp, c_vp, i32, i32, i32, i32, f32, i32, i32, c_vp, c_vp,
                                                  c_vp]),
        "mmk_cart_to_polar": (ctypes.c_int, [c_vp, c_vp, c_vp, c_vp, i32, i32, i32, i32, i32, ctypes.c_double, c_vp, c_vp]),
        "mmk_sample_weights_fwd": (ctypes.c_int, [c_vp, c_vp, i32, i32, i32, i32, i32, i32, f32, c_vp, c_vp]),
        "mmk_host_read_rows": (ctypes.c_int, [ctypes.c_char_p, ctypes.c_int64, i32, i32, i32, i32, i32, c_vp]),
        "mmk_host_read_rows_batch": (ctypes.c_int, [ctypes.POINTER(ReadJob), i32, i32]),
        "mmk_u8_to_float": (ctypes.c_int, [c_vp, c_vp, ctypes.c_int64, c_vp, c_vp]),
        "mmk_sample_weights_bwd_ws_bytes": (sz, [i32, i32]),
        "mmk_sample_weights_bwd": (ctypes.c_int, [c_vp, c_vp, i32, i32, i32, i32, i32, i32, f32, c_vp, c_vp, sz, c_vp]),
        "mmk_weight_stats": (ctypes.c_int, [c_vp, c_vp, i32, i32, i32, c_vp, c_vp, c_vp]),
        "mmk_bev_raster": (ctypes.c_int, [c_vp, i32, i32, i32, i32, f32, c_vp, c_vp]),
    }
    for name, (res, args) in sig.items():
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args
    return sig


EXPORTED = None


def lib():
    """The loaded C-ABI library; raises MmkError when it is absent (no fallback)."""
    global _lib, EXPORTED
    if _lib is None:
        if not os.path.exists(SO_PATH):
            raise MmkError("libmmk_hip.so is not built: run `python -c 'import __graft_entry__ as g; g.build()'` "
                           "(hipcc --offload-arch=gfx950).  The HIP path has no CPU fallback.")
        try:
            loaded = ctypes.CDLL(SO_PATH)
        except OSError as e:
            raise MmkError("cannot load %s: %s" % (SO_PATH, e)) from e
        EXPORTED = _declare(loaded)
        _lib = loaded
    return _lib


def check(rc):
    if rc != 0:
        raise MmkError("libmmk_hip: %s (code %d)" % (lib().mmk_last_error().decode(errors="replace"), rc))


def stream_ptr(device=None):
    """The current PyTorch HIP stream of ``device`` as a void*.  The C entry points launch on the calling
    thread's *current* HIP device (include/mmk.h), so a tensor that lives on another device than the
    current one would meet a foreign stream handle: refuse that instead of faulting."""
    if device is not None:
        idx = device.index if isinstance(device, torch.device) else int(device)
        if idx is not None and idx != torch.cuda.current_device():
            raise MmkError("tensors live on cuda:%d but the current HIP device is cuda:%d: call torch.cuda.set_device() "
                           "(one process per GPU) or wrap the call in `with torch.cuda.device(...)`"
                           % (idx, torch.cuda.current_device()))
    return ctypes.c_void_p(torch.cuda.current_stream(device).cuda_stream)


def ptr(t, dtype=None, name="tensor"):
    """Device pointer of a contiguous HIP tensor (None -> NULL)."""
    if t is None:
        return ctypes.c_void_p(0)
    if not t.is_cuda:
        raise MmkError("%s must live on a HIP device (got %s); the kernels have no CPU path" % (name, t.device))
    if dtype is not None and t.dtype != dtype:
        raise MmkError("%s must be %s (got %s)" % (name, dtype, t.dtype))
    if not t.is_contiguous():
        raise MmkError("%s must be contiguous" % name)
    return ctypes.c_void_p(t.data_ptr())


def dev_f32(t, device):
    """Reference call sites hand over CPU or device tensors of any float type
    (icp_weight_policy.py:130-134 moves them itself); normalise to contiguous
    fp32 on `device`."""
    return t.detach().to(device=device, dtype=torch.float32).contiguous()
