"""ORACLE — TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

CPU restatement of the differentiable-ICP stage of the hot path (SURVEY.md §8a
Group I, stages I1-I7), written as an autograd-unrolled PyTorch program so the
gradient of the pose with respect to the per-point ``weight`` comes from
autograd and independently checks the hand-derived HIP backward kernels.

PARITY UNPINNED: the reference's arithmetic lives in ``lisusdaniil/dICP``
(git submodule ``external/dICP``, no pinned commit; /root/reference/.gitmodules:4-6,
requirements.txt:11), absent from /root/reference, and the reference holds no
test or golden vector for it.  What is anchored on the reference is the call
contract (mm_masking/icp_weight_policy.py:54-55,281-288;
mm_masking/icp_weight_dataset.py:59-61,379-381,395-398):

    ICP(icp_type, config_path, differentiable, max_iterations, tolerance)
    .target_pad_val
    .icp(source (B,N,3), target (B,M,6|3), T_init=(B,4,4), weight=(B,N),
         trim_dist=5.0, loss_fn={"name": "cauchy"|"huber", "metric": k}, dim=2|3)
        -> {"T": (B,4,4)}   differentiable w.r.t. ``weight``

Normative arithmetic (DESIGN.md §3 repeats it; the HIP kernels follow it
operation for operation so that correspondences stay bit-exact over all
iterations):

  per point, fp32, every product/sum individually rounded, left to right
    I1  p  = (R00*sx + R01*sy [+ R02*sz]) + tx ...          (dim 2: x,y block)
    I2  j* = argmin_j fmaf-form squared distance (oracle/nn_search.c), ties -> lowest j
    I3  e_vec = q - p ; d2 = (ex*ex + ey*ey) [+ ez*ez] ; keep = d2 < trim*trim
        pt2pt: r2 = d2                 pt2pl: e = (nx*ex + ny*ey) [+ nz*ez], r2 = e*e
        cauchy: rho = 1/(1 + r2/(k*k)) huber: r = sqrt(r2), rho = r<=k ? 1 : k/r
        w = (weight*keep)*rho
    I4  Jacobian rows of the left perturbation T <- Exp(delta) T,
        delta = (x,y,theta) [dim 2] or (rho(3),phi(3)) [dim 3]:
        pt2pt rows G = [I | -[p]x]; pt2pl row J = n^T G, rotational part
        fl(fl(a*b) - fl(c*d));  wJ = w*J (fp32)
  per pair, fp64
        A = sum_i sum_rows wJ^T J   (exact fp64 products of fp32 values; upper
                                     triangle computed, mirrored to the lower)
        b = sum_i sum_rows wJ^T e
    I5  delta = A^-1 b  (Cholesky; not positive definite -> delta = 0)
    I6  T_{k+1} = fp32( Exp(delta) @ fp64(T_k) ); a pair freezes once
        ||delta||_2 < tolerance.
"""
import math

import numpy as np
import torch

from . import _clib

DEFAULT_TARGET_PAD_VAL = 1000.0
_SMALL_TH2 = 1e-8


def se_exp(delta, dim):
    """Closed-form exponential map, fp64.  delta (B,3) for dim 2 (x, y, theta),
    (B,6) for dim 3 (rho, phi; translation first as pylgmath's
    Transformation(xi_ab=...) used at icp_weight_dataset.py:275).  -> (B,4,4)."""
    B = delta.shape[0]
    E = torch.zeros(B, 4, 4, dtype=delta.dtype)
    E[:, 3, 3] = 1.0
    if dim == 2:
        x, y, th = delta[:, 0], delta[:, 1], delta[:, 2]
        th2 = th * th
        small = th2 < _SMALL_TH2
        ths = torch.where(small, torch.ones_like(th), th)
        a = torch.where(small, 1.0 - th2 / 6.0 + th2 * th2 / 120.0, torch.sin(ths) / ths)
        bb = torch.where(small, th * (0.5 - th2 / 24.0 + th2 * th2 / 720.0), (1.0 - torch.cos(ths)) / ths)
        c, s = torch.cos(th), torch.sin(th)
        rows = [
            torch.stack([c, -s, torch.zeros_like(c), a * x - bb * y], dim=1),
            torch.stack([s, c, torch.zeros_like(c), bb * x + a * y], dim=1),
            torch.stack([torch.zeros_like(c), torch.zeros_like(c), torch.ones_like(c), torch.zeros_like(c)], dim=1),
            torch.stack([torch.zeros_like(c), torch.zeros_like(c), torch.zeros_like(c), torch.ones_like(c)], dim=1),
        ]
        return torch.stack(rows, dim=1)
    rho, phi = delta[:, 0:3], delta[:, 3:6]
    th2 = (phi * phi).sum(dim=1)
    small = th2 < _SMALL_TH2
    th2s = torch.where(small, torch.ones_like(th2), th2)
    th = torch.sqrt(th2s)
    A_ = torch.where(small, 1.0 - th2 / 6.0 + th2 * th2 / 120.0, torch.sin(th) / th)
    B_ = torch.where(small, 0.5 - th2 / 24.0 + th2 * th2 / 720.0, (1.0 - torch.cos(th)) / th2s)
    C_ = torch.where(small, 1.0 / 6.0 - th2 / 120.0 + th2 * th2 / 5040.0, (th - torch.sin(th)) / (th2s * th))
    z = torch.zeros_like(th2)
    K = torch.stack([
        torch.stack([z, -phi[:, 2], phi[:, 1]], dim=1),
        torch.stack([phi[:, 2], z, -phi[:, 0]], dim=1),
        torch.stack([-phi[:, 1], phi[:, 0], z], dim=1)], dim=1)
    K2 = K @ K
    I3 = torch.eye(3, dtype=delta.dtype).expand(B, 3, 3)
    R = I3 + A_[:, None, None] * K + B_[:, None, None] * K2
    V = I3 + B_[:, None, None] * K + C_[:, None, None] * K2
    t = (V @ rho.unsqueeze(-1)).squeeze(-1)
    top = torch.cat([R, t.unsqueeze(-1)], dim=2)
    bot = torch.tensor([0.0, 0.0, 0.0, 1.0], dtype=delta.dtype).expand(B, 1, 4)
    return torch.cat([top, bot], dim=1)


def _gather(t, idx):
    return torch.gather(t, 1, idx)


def transform_points(src, Tk, dim):
    """Stage I1: list of (B,N) coordinate tensors of p = R s + t."""
    sx, sy, sz = src[..., 0], src[..., 1], src[..., 2]

    def r(i, j):
        return Tk[:, i, j].unsqueeze(1)

    if dim == 2:
        px = (r(0, 0) * sx + r(0, 1) * sy) + r(0, 3)
        py = (r(1, 0) * sx + r(1, 1) * sy) + r(1, 3)
        return [px, py]
    px = ((r(0, 0) * sx + r(0, 1) * sy) + r(0, 2) * sz) + r(0, 3)
    py = ((r(1, 0) * sx + r(1, 1) * sy) + r(1, 2) * sz) + r(1, 3)
    pz = ((r(2, 0) * sx + r(2, 1) * sy) + r(2, 2) * sz) + r(2, 3)
    return [px, py, pz]


def per_point_terms(src, tgt, Tk, idx, omega, icp_type, loss_name, loss_k, trim_dist, dim):
    """Stages I1, I3, I4 for given correspondences.  Returns J (B,N,nr,p),
    e (B,N,nr), w (B,N), p (B,N,dim).  dtype follows ``src`` (fp32 normative,
    fp64 for finite-difference checks)."""
    dt = src.dtype
    P = transform_points(src, Tk, dim)
    px, py = P[0], P[1]
    if dim == 3:
        pz = P[2]
    idx64 = idx.long()
    Q = [_gather(tgt[..., c], idx64) for c in range(dim)]
    Ev = [Q[c] - P[c] for c in range(dim)]
    d2 = Ev[0] * Ev[0] + Ev[1] * Ev[1]
    if dim == 3:
        d2 = d2 + Ev[2] * Ev[2]
    trim = torch.tensor(trim_dist, dtype=dt)
    keep = (d2 < trim * trim).to(dt)
    one = torch.ones_like(px)
    zero = torch.zeros_like(px)
    if icp_type == "pt2pl":
        Nn = [_gather(tgt[..., 3 + c], idx64) for c in range(dim)]
        e = Nn[0] * Ev[0] + Nn[1] * Ev[1]
        if dim == 3:
            e = e + Nn[2] * Ev[2]
        r2 = e * e
        if dim == 2:
            Jrows = [[Nn[0], Nn[1], Nn[1] * px - Nn[0] * py]]
        else:
            Jrows = [[Nn[0], Nn[1], Nn[2],
                      py * Nn[2] - pz * Nn[1], pz * Nn[0] - px * Nn[2], px * Nn[1] - py * Nn[0]]]
        erows = [e]
    else:
        r2 = d2
        if dim == 2:
            Jrows = [[one, zero, -py], [zero, one, px]]
        else:
            Jrows = [[one, zero, zero, zero, pz, -py],
                     [zero, one, zero, -pz, zero, px],
                     [zero, zero, one, py, -px, zero]]
        erows = Ev
    k = torch.tensor(loss_k, dtype=dt)
    if loss_name == "cauchy":
        rho = 1.0 / (1.0 + r2 / (k * k))
    elif loss_name == "huber":
        rr = torch.sqrt(r2.detach())
        r_safe = torch.sqrt(torch.where(r2 > 0, r2, torch.ones_like(r2)))
        rho = torch.where(rr <= k, one, k / r_safe)
    elif loss_name in ("l2", "none", None):
        rho = one
    else:
        raise ValueError("unknown loss_fn name %r" % (loss_name,))
    w = (omega * keep) * rho
    J = torch.stack([torch.stack(row, dim=-1) for row in Jrows], dim=-2)
    e = torch.stack(erows, dim=-1)
    return J, e, w, torch.stack(P, dim=-1)


def normal_equations(J, e, w):
    wJ = w[..., None, None] * J
    A = torch.einsum("bnrp,bnrq->bpq", wJ.double(), J.double())
    b = torch.einsum("bnrp,bnr->bp", wJ.double(), e.double())
    # the upper triangle is normative; mirror it so that A is exactly symmetric
    A = torch.triu(A) + torch.triu(A, 1).transpose(1, 2)
    return A, b


def solve_spd(A, b):
    """delta = A^-1 b by Cholesky in fp64; a pair whose A is not positive
    definite gets delta = 0."""
    L, info = torch.linalg.cholesky_ex(A.detach())
    ok = (info == 0) & torch.isfinite(A.detach()).all(dim=(1, 2))
    p = A.shape[-1]
    eye = torch.eye(p, dtype=A.dtype).expand_as(A)
    A_safe = torch.where(ok[:, None, None], A, eye)
    b_safe = torch.where(ok[:, None], b, torch.zeros_like(b))
    delta = torch.linalg.solve(A_safe, b_safe.unsqueeze(-1)).squeeze(-1)
    return delta, ok


class ICPRef:
    """CPU restatement of ``dICP.ICP.ICP`` as the hot path uses it."""

    def __init__(self, icp_type="pt2pt", config_path=None, differentiable=True,
                 max_iterations=100, tolerance=1e-12, target_pad_val=DEFAULT_TARGET_PAD_VAL):
        assert icp_type in ("pt2pt", "pt2pl")
        self.icp_type = icp_type
        self.differentiable = differentiable
        self.max_iterations = int(max_iterations)
        self.tolerance = float(tolerance)
        self.target_pad_val = float(target_pad_val)

    def icp(self, source, target, T_init=None, weight=None, trim_dist=5.0,
            loss_fn=None, dim=3, dtype=torch.float32, fixed_idx=None):
        src = source.to(dtype)
        tgt = target.to(dtype)
        B, N, _ = src.shape
        if self.icp_type == "pt2pl" and tgt.shape[-1] < 6:
            raise ValueError("pt2pl needs target normals: target must be (B,M,6)")
        Tk = torch.eye(4, dtype=dtype).repeat(B, 1, 1) if T_init is None else T_init.to(dtype)
        omega = torch.ones(B, N, dtype=dtype) if weight is None else weight.to(dtype)
        loss_name = None if loss_fn is None else loss_fn.get("name")
        loss_k = 1.0 if loss_fn is None else float(loss_fn.get("metric", 1.0))
        tgt_xy = np.ascontiguousarray(tgt[..., :dim].detach().float().numpy())
        active = torch.ones(B, dtype=torch.bool)
        hist = {"idx": [], "T": [Tk.detach().clone()], "delta": [], "active": []}
        with torch.set_grad_enabled(self.differentiable and torch.is_grad_enabled()):
            for k in range(self.max_iterations):
                if not bool(active.any()):
                    break
                if fixed_idx is not None:
                    idx = fixed_idx[k]
                else:
                    p = torch.stack(transform_points(src.detach(), Tk.detach(), dim), dim=-1)
                    idx_np, _ = _clib.nn_search(p.float().numpy(), tgt_xy)
                    idx = torch.from_numpy(idx_np)
                J, e, w, _ = per_point_terms(src, tgt, Tk, idx, omega, self.icp_type,
                                             loss_name, loss_k, trim_dist, dim)
                A, b = normal_equations(J, e, w)
                delta, ok = solve_spd(A, b)
                delta = torch.where(active[:, None], delta, torch.zeros_like(delta))
                E = se_exp(delta, dim)
                Tn = (E @ Tk.double()).to(dtype)
                Tk = torch.where(active[:, None, None], Tn, Tk)
                hist["idx"].append(idx.clone())
                hist["delta"].append(delta.detach().clone())
                hist["active"].append(active.clone())
                hist["T"].append(Tk.detach().clone())
                conv = delta.detach().norm(dim=1) < self.tolerance
                active = active & ~conv
        return {"T": Tk, "hist": hist, "num_iter": len(hist["delta"])}
