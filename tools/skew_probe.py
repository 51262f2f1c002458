"""Does the backward blend's tile scheduling (common.h: contiguous runs per XCD + queues with stealing) cope with a scene that
fills only part of the image?  cfg3 with all Gaussians, with a random half of them, and with the half that projects into the
lower / right / upper-left part of the image: per-stage times and the backward's time per walked list entry.
   python tools/skew_probe.py"""
import os, sys
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from seganygaussians_amd import install_dropin, scenes, _lib
install_dropin()
from seganygaussians_amd import rasterizer as R
from diff_gaussian_rasterization_contrastive_f import GaussianRasterizationSettings as GRS
cfg = scenes.CONFIGS["cfg3"]; C, W, H, P = cfg["C"], cfg["W"], cfg["H"], cfg["P"]
dev = torch.device("cuda", 0)
_, _, GR = R.make_rasterizer(C)
scene = scenes.make_scene(P, W, H, cfg["focal"], C, cfg["ls_mean"], cfg["ls_std"], seed=0)
cam = scenes.look_at_camera(W, H, cfg["focal"])
t = lambda a: torch.as_tensor(np.ascontiguousarray(a, np.float32)).to(dev)
m = np.asarray(scene.means3D).reshape(-1, 3)
rng = np.random.default_rng(3)
subsets = {"all": np.ones(P, bool), "random half": rng.random(P) < 0.5, "lower half of the image": m[:, 1] > 0,
           "right half": m[:, 0] > 0, "upper left quarter x2": (m[:, 0] < 0) & (m[:, 1] < 0)}
dL = t(scenes.make_grad_image(C, H, W, seed=1))
s = GRS(image_height=H, image_width=W, tanfovx=cam.tanfovx, tanfovy=cam.tanfovy, bg=torch.zeros(C, device=dev), scale_modifier=1.0,
        viewmatrix=t(cam.viewmatrix), projmatrix=t(cam.projmatrix), sh_degree=0, campos=t(cam.campos), prefiltered=False, debug=False)
rast = GR(s)
for name, keep in subsets.items():
    sel = lambda a, k=keep: t(np.asarray(a).reshape(P, -1)[k])
    means3D, feats, opac, scales, rots = [sel(a).requires_grad_(True) for a in (scene.means3D, scene.features, scene.opacities, scene.scales, scene.rotations)]
    def step():
        for l in (means3D, feats, opac, scales, rots): l.grad = None
        m2 = torch.zeros_like(means3D, requires_grad=True)
        color, radii = rast(means3D=means3D, means2D=m2, shs=None, colors_precomp=feats, opacities=opac, scales=scales, rotations=rots, cov3D_precomp=None)
        torch.autograd.backward(color, grad_tensors=dL)
    for _ in range(20): step()
    torch.cuda.synchronize()
    _lib.profile_enable(True)
    acc = {k: 0.0 for k in _lib.MI_STAGES}
    for _ in range(10):
        step(); torch.cuda.synchronize()
        ms = _lib.profile_read()
        for k in acc: acc[k] += ms[k] / 10
    _lib.profile_enable(False)
    print(f"{name:28s} P {int(keep.sum()):8d}  blend_fwd {acc['blend_fwd']:.3f}  blend_bwd {acc['blend_bwd']:.3f}  tile_sort {acc['tile_sort']:.3f}  emit {acc['emit']:.3f} ms")
