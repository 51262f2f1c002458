import os, sys, time
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from seganygaussians_amd import install_dropin, scenes
install_dropin()
from seganygaussians_amd import rasterizer as R
from diff_gaussian_rasterization_contrastive_f import GaussianRasterizationSettings
t_start = time.perf_counter()
cfg = scenes.CONFIGS["cfg3"]; C, W, H, P = cfg["C"], cfg["W"], cfg["H"], cfg["P"]
dev = torch.device("cuda", 0)
_, _, GR = R.make_rasterizer(C)
scene = scenes.make_scene(P, W, H, cfg["focal"], C, cfg["ls_mean"], cfg["ls_std"], seed=0)
cam = scenes.look_at_camera(W, H, cfg["focal"])
t = lambda a: torch.as_tensor(np.ascontiguousarray(a, np.float32)).to(dev)
means3D = t(scene.means3D).requires_grad_(True); feats = t(scene.features).requires_grad_(True)
opac = t(scene.opacities).requires_grad_(True); scales = t(scene.scales).requires_grad_(True); rots = t(scene.rotations).requires_grad_(True)
s = GaussianRasterizationSettings(image_height=H, image_width=W, tanfovx=cam.tanfovx, tanfovy=cam.tanfovy, bg=torch.zeros(C, device=dev),
    scale_modifier=1.0, viewmatrix=t(cam.viewmatrix), projmatrix=t(cam.projmatrix), sh_degree=0, campos=t(cam.campos), prefiltered=False, debug=False)
rast = GR(s); dL = t(scenes.make_grad_image(C, H, W, seed=1))
print("setup s", time.perf_counter() - t_start)
times = []
for i in range(60):
    torch.cuda.synchronize(); a = time.perf_counter()
    for l in (means3D, feats, opac, scales, rots): l.grad = None
    m2 = torch.zeros_like(means3D, requires_grad=True)
    color, radii = rast(means3D=means3D, means2D=m2, shs=None, colors_precomp=feats, opacities=opac, scales=scales, rotations=rots, cov3D_precomp=None)
    torch.autograd.backward(color, grad_tensors=dL)
    torch.cuda.synchronize(); times.append((time.perf_counter() - a) * 1e3)
print(" ".join(f"{x:.2f}" for x in times))
