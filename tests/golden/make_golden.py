#!/usr/bin/env python
"""Generates the golden fixtures tests/golden/*.npz.

The reference (CUDA-only) cannot be imported or run here and ships no fixtures, so these
vectors come from the INDEPENDENT dense float64 autograd re-derivation in tests/dense_ref.py -- not from the
oracle and not from the HIP kernels, both of which are tested AGAINST them.  Each file holds the fp32 inputs
of one rasterizer call and the expected image / radii / gradients (computed in fp64 from those fp32 inputs).

    python tests/golden/make_golden.py      # rewrites the .npz files (deterministic)
"""
import math
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from seganygaussians_amd import scenes  # noqa: E402
from tests.dense_ref import render_dense  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))

CASES = {
    "rgb_precomp_bg":   dict(P=140, W=48, H=40, C=3, seed=0, bg="random"),
    "features32":       dict(P=160, W=40, H=40, C=32, seed=3),
    "sh3_mask_rotated": dict(P=110, W=44, H=36, C=3, seed=13, with_shs=True, sh_degree=3, use_mask=True, rotated=True, bg="random"),
}


def make_case(name, P, W, H, C, seed, bg=None, with_shs=False, sh_degree=0, use_mask=False, rotated=False):
    focal = 0.9 * W
    sc = scenes.make_scene(P, W, H, focal, C, math.log(0.12), 0.5, seed=seed, with_shs=with_shs, z_range=(1.0, 6.0))
    if rotated:
        cam = scenes.orbit_camera(W, H, focal, 0.25, 0.1)
        sc.means3D[:, 2] += 1.0
    else:
        cam = scenes.look_at_camera(W, H, focal)
    rng = np.random.default_rng(seed + 100)
    bgv = rng.uniform(0, 1, C).astype(np.float32) if bg == "random" else np.zeros(C, np.float32)
    mask = rng.uniform(0, 1, P).astype(np.float32) if use_mask else None
    dL = (scenes.make_grad_image(C, H, W, seed=seed + 1) * (W * H)).astype(np.float32)
    dLm = rng.normal(0, 1, (H, W)).astype(np.float32) if use_mask else None

    t64 = lambda a, g=True: None if a is None else torch.tensor(np.asarray(a, np.float64), requires_grad=g)
    means3D, opac = t64(sc.means3D), t64(sc.opacities)
    scales_t, rots_t = t64(sc.scales), t64(sc.rotations)
    cols_t = None if with_shs else t64(sc.features)
    shs_t = t64(sc.shs) if with_shs else None
    m2d = torch.zeros(P, 3, dtype=torch.float64, requires_grad=True)
    mask_t = t64(mask) if use_mask else None
    ref = render_dense(means3D, opac, t64(cam.viewmatrix, False), t64(cam.projmatrix, False), t64(cam.campos, False),
                       t64(bgv, False), W, H, cam.tanfovx, cam.tanfovy, scales=scales_t, rotations=rots_t,
                       colors_precomp=cols_t, shs=shs_t, sh_degree=sh_degree, means2D_offset=m2d, mask=mask_t)
    loss = (ref["color"] * torch.tensor(dL, dtype=torch.float64)).sum()
    mask_grad = None
    if use_mask:
        # dL/dout_mask reaches only dL_dmask in the reference (DEPTH/cuda_rasterizer/backward.cu:516)
        mloss = (ref["mask"][0] * torch.tensor(dLm, dtype=torch.float64)).sum()
        mask_grad = torch.autograd.grad(mloss, mask_t, retain_graph=True)[0]
    loss.backward()
    out = dict(
        means3D=sc.means3D, scales=sc.scales, rotations=sc.rotations, opacities=sc.opacities,
        viewmatrix=cam.viewmatrix, projmatrix=cam.projmatrix, campos=cam.campos, bg=bgv,
        image_width=np.int32(W), image_height=np.int32(H), tanfovx=np.float64(cam.tanfovx),
        tanfovy=np.float64(cam.tanfovy), channels=np.int32(C), sh_degree=np.int32(sh_degree), dL_dout_color=dL,
        exp_color=ref["color"].detach().numpy(), exp_radii=ref["radii"].numpy().astype(np.int32),
        exp_dL_dmeans3D=means3D.grad.numpy(), exp_dL_dmeans2D=m2d.grad.numpy(), exp_dL_dopacity=opac.grad.numpy(),
        exp_dL_dscales=scales_t.grad.numpy(), exp_dL_drotations=rots_t.grad.numpy())
    if with_shs:
        out.update(shs=sc.shs, exp_dL_dsh=shs_t.grad.numpy())
    else:
        out.update(colors_precomp=sc.features, exp_dL_dcolors=cols_t.grad.numpy())
    if use_mask:
        out.update(mask=mask, dL_dout_mask=dLm, exp_mask=ref["mask"].detach().numpy(), exp_depth=ref["depth"].detach().numpy(),
                   exp_dL_dmask=mask_grad.numpy())
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **out)
    print(name, "visible", int((ref["radii"] > 0).sum()), "of", P)


if __name__ == "__main__":
    for n, kw in CASES.items():
        make_case(n, **kw)
