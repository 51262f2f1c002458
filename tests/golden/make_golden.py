#!/usr/bin/env python
"""Generates the golden fixtures tests/golden/*.npz.

The reference ships no fixtures.  These vectors come from the dense float64 autograd re-derivation in
tests/dense_ref.py with the REFERENCE'S OWN Python helpers plugged in (tests/reference_helpers.py imports
/root/reference/utils/{sh_utils,general_utils,graphics_utils}.py: eval_sh, build_scaling_rotation, geom_transform_points,
getWorld2View2, getProjectionMatrix) -- not from the oracle and not from the HIP kernels, both of which are tested AGAINST
them.  The same run cross-checks dense_ref's restated helpers (what the GPU box, which has no /root/reference, uses)
against the imported ones: every expected tensor must agree to 2e-6 (|q| = 1 only to fp32 rounding), the cameras of seganygaussians_amd/scenes.py must
equal the reference's matrices (bit for bit for the front camera), and the rotation gradient (the rasterizer differentiates R(q) WITHOUT
normalising q, forward.cu:130, while the Python helper normalises) must agree on the component tangential to q.
Each file holds the fp32 inputs of one rasterizer call and the expected image / radii / gradients (fp64 from those inputs).
The real pin of the rasterizer itself is oracle/_ref (tests/test_zz_reference_pin.py); this one covers the Python-side
conventions (matrix layout, SH basis, quaternion order).

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
from tests import reference_helpers as rh  # noqa: E402

HELPERS = rh.ReferenceHelpers()

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

    # the cameras of scenes.py against the reference's own matrix builders (scene/cameras.py:56-65)
    gu = HELPERS.graphics_utils
    fovx, fovy = 2 * math.atan(W / (2 * focal)), 2 * math.atan(H / (2 * focal))
    proj_ref = gu.getProjectionMatrix(znear=0.01, zfar=100.0, fovX=fovx, fovY=fovy).transpose(0, 1)
    view_ref = torch.tensor(gu.getWorld2View2(cam.R, cam.T)).transpose(0, 1)
    # (getWorld2View2 inverts Rt twice, so a rotated camera agrees to rounding, the front camera bit for bit)
    assert np.allclose(view_ref.numpy(), cam.viewmatrix, rtol=0, atol=1e-6), "world_view_transform"
    full_ref = (view_ref.unsqueeze(0).bmm(proj_ref.unsqueeze(0))).squeeze(0)
    assert np.allclose(full_ref.numpy(), cam.projmatrix, rtol=0, atol=1e-5), "full_proj_transform"
    if not rotated:
        assert np.array_equal(view_ref.numpy(), cam.viewmatrix) and np.array_equal(full_ref.numpy(), cam.projmatrix)
    assert np.allclose(view_ref.inverse()[3, :3].numpy(), cam.campos, atol=1e-6), "camera_center"

    t64 = lambda a, g=True: None if a is None else torch.tensor(np.asarray(a, np.float64), requires_grad=g)

    def run(helpers):
        means3D, opac = t64(sc.means3D), t64(sc.opacities)
        scales_t, rots_t = t64(sc.scales), t64(sc.rotations)
        cols_t = None if with_shs else t64(sc.features)
        shs_t = t64(sc.shs) if with_shs else None
        m2d = torch.zeros(P, 3, dtype=torch.float64, requires_grad=True)
        mask_t = t64(mask) if use_mask else None
        ref = render_dense(means3D, opac, t64(cam.viewmatrix, False), t64(cam.projmatrix, False), t64(cam.campos, False),
                           t64(bgv, False), W, H, cam.tanfovx, cam.tanfovy, scales=scales_t, rotations=rots_t,
                           colors_precomp=cols_t, shs=shs_t, sh_degree=sh_degree, means2D_offset=m2d, mask=mask_t,
                           helpers=helpers)
        loss = (ref["color"] * torch.tensor(dL, dtype=torch.float64)).sum()
        mask_grad = None
        if use_mask:
            # dL/dout_mask reaches only dL_dmask in the reference (DEPTH/cuda_rasterizer/backward.cu:516)
            mloss = (ref["mask"][0] * torch.tensor(dLm, dtype=torch.float64)).sum()
            mask_grad = torch.autograd.grad(mloss, mask_t, retain_graph=True)[0]
        loss.backward()
        return ref, means3D, opac, scales_t, rots_t, cols_t, shs_t, m2d, mask_grad

    ref, means3D, opac, scales_t, rots_t, cols_t, shs_t, m2d, mask_grad = run(HELPERS)
    own = run(None)
    # dense_ref's restated helpers == the imported ones, on this fixture (to ~1e-7: the reference's build_rotation divides
    # by |q|, and the fixture's fp32 quaternions are unit only to fp32 rounding)
    for a, b, what in ((ref["color"], own[0]["color"], "color"), (means3D.grad, own[1].grad, "dmeans3D"),
                       (opac.grad, own[2].grad, "dopacity"), (scales_t.grad, own[3].grad, "dscales"),
                       (m2d.grad, own[7].grad, "dmeans2D")):
        assert torch.allclose(a.detach(), b.detach(), rtol=2e-6, atol=2e-7 * float(b.detach().abs().max())), what
    assert torch.equal(ref["radii"], own[0]["radii"])
    q = torch.tensor(np.asarray(sc.rotations, np.float64))
    g_free = own[4].grad                                    # d/dq of R(q) as the kernels define it (no normalisation)
    g_tan = g_free - (g_free * q).sum(1, keepdim=True) * q / (q * q).sum(1, keepdim=True)
    assert torch.allclose(rots_t.grad, g_tan / q.norm(dim=1, keepdim=True), rtol=1e-5, atol=2e-6 * float(g_free.abs().max())), \
        "rotation gradient: tangential component vs the reference's normalising build_rotation"
    rots_t = own[4]
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
