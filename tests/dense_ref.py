"""Independent dense PyTorch (float64, autograd) re-derivation of the rasterizer's math.

Used ONLY to pin the CPU oracle (tests/test_oracle_*.py): the reference ships no tests or golden
vectors, so the oracle's forward AND its hand-written backward are checked against
this autograd implementation, which shares no code with oracle/ and is built from the formulas of
the reference's own Python helpers:

* SH evaluation      -- utils/sh_utils.py:57-112 (eval_sh), constants :24-51
* R from quaternion  -- utils/general_utils.py:78-99 (build_rotation), Sigma = (RS)(RS)^T
                        scene/gaussian_model.py:27-32
* projection         -- utils/graphics_utils.py:22-29 (geom_transform_points), +1e-7 on w
* EWA cov2D, low-pass 0.3, conic, 3-sigma radius / tile rect, alpha thresholds
                     -- CF/cuda_rasterizer/forward.cu:77-116,216-240,330-356 (math only; dense, no tiles
                        except the rect-membership mask that defines which pixels a Gaussian may touch)

Everything is O(P * W * H): keep P <= a few hundred and images <= 64x64.
"""
from __future__ import annotations

import math

import torch

SH_C0 = 0.28209479177387814
SH_C1 = 0.4886025119029199
SH_C2 = [1.0925484305920792, -1.0925484305920792, 0.31539156525252005, -1.0925484305920792, 0.5462742152960396]
SH_C3 = [-0.5900435899266435, 2.890611442640554, -0.4570457994644658, 0.3731763325901154,
         -0.4570457994644658, 1.445305721320277, -0.5900435899266435]


def eval_sh(deg, sh, dirs):
    """sh: (P,K,3), dirs: (P,3) unit.  Returns (P,3) (before +0.5 / clamp)."""
    x, y, z = dirs[:, 0:1], dirs[:, 1:2], dirs[:, 2:3]
    res = SH_C0 * sh[:, 0]
    if deg > 0:
        res = res - SH_C1 * y * sh[:, 1] + SH_C1 * z * sh[:, 2] - SH_C1 * x * sh[:, 3]
    if deg > 1:
        xx, yy, zz, xy, yz, xz = x * x, y * y, z * z, x * y, y * z, x * z
        res = (res + SH_C2[0] * xy * sh[:, 4] + SH_C2[1] * yz * sh[:, 5] + SH_C2[2] * (2 * zz - xx - yy) * sh[:, 6]
               + SH_C2[3] * xz * sh[:, 7] + SH_C2[4] * (xx - yy) * sh[:, 8])
    if deg > 2:
        res = (res + SH_C3[0] * y * (3 * xx - yy) * sh[:, 9] + SH_C3[1] * xy * z * sh[:, 10]
               + SH_C3[2] * y * (4 * zz - xx - yy) * sh[:, 11] + SH_C3[3] * z * (2 * zz - 3 * xx - 3 * yy) * sh[:, 12]
               + SH_C3[4] * x * (4 * zz - xx - yy) * sh[:, 13] + SH_C3[5] * z * (xx - yy) * sh[:, 14]
               + SH_C3[6] * x * (xx - 3 * yy) * sh[:, 15])
    return res


def build_rotation(q):
    """Rotation from (r,x,y,z) WITHOUT normalising (the kernels do not; CF forward.cu:130)."""
    r, x, y, z = q[:, 0], q[:, 1], q[:, 2], q[:, 3]
    R = torch.stack([
        1 - 2 * (y * y + z * z), 2 * (x * y - r * z), 2 * (x * z + r * y),
        2 * (x * y + r * z), 1 - 2 * (x * x + z * z), 2 * (y * z - r * x),
        2 * (x * z - r * y), 2 * (y * z + r * x), 1 - 2 * (x * x + y * y)], dim=1).reshape(-1, 3, 3)
    return R  # row-major standard rotation matrix


def render_dense(means3D, opacities, viewmatrix, projmatrix, campos, bg, W, H, tanfovx, tanfovy, *,
                 scales=None, rotations=None, cov3D_precomp=None, colors_precomp=None, shs=None, sh_degree=0,
                 scale_modifier=1.0, means2D_offset=None, mask=None, helpers=None):
    """`helpers`: optional tests/reference_helpers.ReferenceHelpers -- SH evaluation, scaling-rotation matrix and the
    projective transform are then the REFERENCE's own Python functions (imported from /root/reference/utils), not the
    restatements above.  All tensor inputs float64.  viewmatrix/projmatrix are the reference's transposed matrices
    (row-vector convention: p_hom = [x,y,z,1] @ M).  Returns dict(color (C,H,W), radii, mask, depth,
    n_contrib, final_T)."""
    dt = torch.float64
    P = means3D.shape[0]
    ones = torch.ones(P, 1, dtype=dt)
    hom = torch.cat([means3D, ones], 1)
    p_view = hom @ viewmatrix           # (P,4)
    p_hom = hom @ projmatrix
    p_w = 1.0 / (p_hom[:, 3] + 0.0000001)
    p_proj = p_hom[:, :3] * p_w[:, None]
    if helpers is not None:
        p_proj = helpers.geom_transform_points(means3D, projmatrix)      # utils/graphics_utils.py:22-29
    visible = p_view[:, 2] > 0.2

    if cov3D_precomp is None and helpers is not None:
        L = helpers.build_scaling_rotation(scale_modifier * scales, rotations)   # utils/general_utils.py:101-110
        Sigma = L @ L.transpose(1, 2)                                            # scene/gaussian_model.py:27-32
    elif cov3D_precomp is None:
        R = build_rotation(rotations)
        S = torch.diag_embed(scale_modifier * scales)
        L = R @ S
        Sigma = L @ L.transpose(1, 2)
    else:
        c = cov3D_precomp
        Sigma = torch.stack([c[:, 0], c[:, 1], c[:, 2], c[:, 1], c[:, 3], c[:, 4], c[:, 2], c[:, 4], c[:, 5]],
                            1).reshape(-1, 3, 3)

    focal_x = W / (2.0 * tanfovx)
    focal_y = H / (2.0 * tanfovy)
    tx, ty, tz = p_view[:, 0], p_view[:, 1], p_view[:, 2]
    limx, limy = 1.3 * tanfovx, 1.3 * tanfovy
    txc = torch.clamp(tx / tz, -limx, limx) * tz
    tyc = torch.clamp(ty / tz, -limy, limy) * tz
    zero = torch.zeros_like(tz)
    J = torch.stack([focal_x / tz, zero, -(focal_x * txc) / (tz * tz),
                     zero, focal_y / tz, -(focal_y * tyc) / (tz * tz),
                     zero, zero, zero], 1).reshape(-1, 3, 3)
    Wr = viewmatrix[:3, :3].T            # world->view rotation (standard, row-major)
    Tm = J @ Wr                          # (P,3,3)
    cov = Tm @ Sigma @ Tm.transpose(1, 2)
    a = cov[:, 0, 0] + 0.3
    b = cov[:, 0, 1]
    c_ = cov[:, 1, 1] + 0.3
    det = a * c_ - b * b
    conic_a, conic_b, conic_c = c_ / det, -b / det, a / det
    mid = 0.5 * (a + c_)
    lam = mid + torch.sqrt(torch.clamp(mid * mid - det, min=0.1))
    radius = torch.ceil(3.0 * torch.sqrt(lam)).detach()

    ndc = p_proj[:, :2]
    if means2D_offset is not None:
        ndc = ndc + means2D_offset[:, :2]
    px = ((ndc[:, 0] + 1.0) * W - 1.0) * 0.5
    py = ((ndc[:, 1] + 1.0) * H - 1.0) * 0.5

    gx, gy = (W + 15) // 16, (H + 15) // 16
    pxd, pyd = px.detach(), py.detach()
    rminx = torch.clamp(torch.trunc((pxd - radius) / 16), 0, gx)
    rminy = torch.clamp(torch.trunc((pyd - radius) / 16), 0, gy)
    rmaxx = torch.clamp(torch.trunc((pxd + radius + 15) / 16), 0, gx)
    rmaxy = torch.clamp(torch.trunc((pyd + radius + 15) / 16), 0, gy)
    visible = visible & (det != 0) & ((rmaxx - rminx) * (rmaxy - rminy) > 0)
    radii = torch.where(visible, radius, torch.zeros_like(radius)).to(torch.int32)

    if colors_precomp is not None:
        colors = colors_precomp
    else:
        d = means3D - campos[None]
        d = d / d.norm(dim=1, keepdim=True)
        sh_val = eval_sh(sh_degree, shs, d) if helpers is None else helpers.eval_sh(sh_degree, shs, d)
        colors = torch.clamp_min(sh_val + 0.5, 0.0)

    # order: (depth, index) ascending == the (tile|depth) radix sort restricted to any one tile
    depth32 = p_view[:, 2].detach().to(torch.float32)   # keys use fp32 depth bits
    order = torch.tensor(sorted(range(P), key=lambda i: (float(depth32[i]), i)), dtype=torch.long)

    ys, xs = torch.meshgrid(torch.arange(H, dtype=dt), torch.arange(W, dtype=dt), indexing="ij")
    pixx, pixy = xs.reshape(-1), ys.reshape(-1)
    tilex, tiley = torch.floor(pixx / 16), torch.floor(pixy / 16)

    o = order
    dx = px[o, None] - pixx[None]
    dy = py[o, None] - pixy[None]
    power = -0.5 * (conic_a[o, None] * dx * dx + conic_c[o, None] * dy * dy) - conic_b[o, None] * dx * dy
    alpha = torch.clamp_max(opacities.reshape(-1)[o, None] * torch.exp(power), 0.99)
    member = (visible[o, None] & (tilex[None] >= rminx[o, None]) & (tilex[None] < rmaxx[o, None])
              & (tiley[None] >= rminy[o, None]) & (tiley[None] < rmaxy[o, None]))
    ok = member & (power <= 0) & (alpha >= 1.0 / 255.0)
    A = torch.where(ok, alpha, torch.zeros_like(alpha))
    one_minus = 1.0 - A
    T_incl = torch.cumprod(one_minus, 0)
    T_excl = torch.cat([torch.ones(1, T_incl.shape[1], dtype=dt), T_incl[:-1]], 0)
    stop = ok & ((T_excl * one_minus).detach() < 0.0001)
    alive = (torch.cumsum(stop.to(torch.int64), 0) == 0)     # inclusive: the stopping Gaussian is not blended
    Aeff = torch.where(alive, A, torch.zeros_like(A))
    T_incl = torch.cumprod(1.0 - Aeff, 0)
    T_excl = torch.cat([torch.ones(1, T_incl.shape[1], dtype=dt), T_incl[:-1]], 0)
    wgt = Aeff * T_excl                                      # (P,N)
    T_final = T_incl[-1] if P > 0 else torch.ones(W * H, dtype=dt)
    color = (wgt.T @ colors[o]).T + T_final[None] * bg[:, None]
    out = {"color": color.reshape(-1, H, W), "radii": radii, "final_T": T_final.reshape(H, W)}
    if mask is not None:
        out["mask"] = (wgt.T @ mask.reshape(-1)[o]).reshape(1, H, W)
        out["depth"] = (wgt.T @ p_view[:, 2].detach()[o]).reshape(1, H, W)
    # n_contrib: 1-based position within the TILE list is tile-dependent; expose the contributing matrix instead
    out["contrib"] = (Aeff > 0)
    out["order"] = order
    return out
