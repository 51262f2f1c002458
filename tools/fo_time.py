"""Stage times of the features-only backward (blend_bwd, HIP events) for the library MI_RAST_LIB selects: cfg3 / cfg5, median of n steps.
   MI_RAST_LIB=seganygaussians_amd/libmi_rast_<variant>.so python tools/fo_time.py [cfg3 cfg5] [--full]   (--full: the default backward)"""
import os, sys, json
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import seganygaussians_amd
from seganygaussians_amd import _lib, scenes, rasterizer as R
seganygaussians_amd.install_dropin()
from diff_gaussian_rasterization_contrastive_f import GaussianRasterizationSettings
cfgs = [a for a in sys.argv[1:] if not a.startswith("--")] or ["cfg3"]
full = "--full" in sys.argv
dev = torch.device("cuda:0")
for name in cfgs:
    cfg = scenes.CONFIGS[name]
    C, W, H, P = cfg["C"], cfg["W"], cfg["H"], cfg["P"]
    scene = scenes.scene_of_config(name, seed=0, P=P, with_shs=False)
    cam = scenes.look_at_camera(W, H, cfg["focal"])
    t = lambda a: torch.as_tensor(np.ascontiguousarray(a, np.float32)).to(dev)
    means3D, opac, scales, rots = (t(x).requires_grad_(True) for x in (scene.means3D, scene.opacities, scene.scales, scene.rotations))
    feats = t(scene.features).requires_grad_(True)
    _, _, GR = R.make_rasterizer(C)
    rast = GR(GaussianRasterizationSettings(image_height=H, image_width=W, tanfovx=cam.tanfovx, tanfovy=cam.tanfovy, bg=torch.zeros(C, device=dev),
              scale_modifier=1.0, viewmatrix=t(cam.viewmatrix), projmatrix=t(cam.projmatrix), sh_degree=0, campos=t(cam.campos), prefiltered=False, debug=False))
    dL = t(scenes.make_grad_image(C, H, W, seed=1))
    R.enable_features_only_backward(not full)
    def step():
        for l in (means3D, opac, scales, rots, feats):
            l.grad = None
        m2 = torch.zeros_like(means3D, requires_grad=True)
        col, _ = rast(means3D=means3D, means2D=m2, shs=None, colors_precomp=feats, opacities=opac, scales=scales, rotations=rots, cov3D_precomp=None)
        torch.autograd.backward(col, grad_tensors=dL)
    for _ in range(30):
        step()
    torch.cuda.synchronize()
    _lib.profile_enable(True)
    v = []
    for _ in range(25):
        step(); torch.cuda.synchronize()
        v.append(_lib.profile_read()["blend_bwd"])
    _lib.profile_enable(False)
    import time
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(100):
        step()
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 100
    v = np.sort(np.asarray(v))
    print(f"{os.path.basename(os.environ.get('MI_RAST_LIB', 'libmi_rast.so')):24s} {name} {'full' if full else 'features-only'}: blend_bwd median {np.median(v):.4f} ms (p10 {v[2]:.4f} p90 {v[-3]:.4f}); step {1e3 * dt:.4f} ms", flush=True)
