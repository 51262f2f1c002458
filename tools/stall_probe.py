"""Where do the occasional 30-80 ms stalls of a step come from?  Per-step wall times of `bench.py`'s step (cfg5 by default),
with the caching allocator's device malloc / free counters and Python's garbage collections logged next to every slow step.
   python tools/stall_probe.py [cfg] [steps]"""
import gc, os, sys, time
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from seganygaussians_amd import install_dropin, scenes
install_dropin()
from seganygaussians_amd import rasterizer as R
cfgname = sys.argv[1] if len(sys.argv) > 1 else "cfg5"
nsteps = int(sys.argv[2]) if len(sys.argv) > 2 else 300
cfg = scenes.CONFIGS[cfgname]; C, W, H, P = cfg["C"], cfg["W"], cfg["H"], cfg["P"]
dev = torch.device("cuda", 0)
from diff_gaussian_rasterization_contrastive_f import GaussianRasterizationSettings as GRS
_, _, GR = R.make_rasterizer(C)
scene = scenes.make_scene(P, W, H, cfg["focal"], C, cfg["ls_mean"], cfg["ls_std"], seed=0)
cam = scenes.look_at_camera(W, H, cfg["focal"])
t = lambda a: torch.as_tensor(np.ascontiguousarray(a, np.float32)).to(dev)
means3D = t(scene.means3D).requires_grad_(True); feats = t(scene.features).requires_grad_(True)
opac = t(scene.opacities).requires_grad_(True); scales = t(scene.scales).requires_grad_(True); rots = t(scene.rotations).requires_grad_(True)
s = GRS(image_height=H, image_width=W, tanfovx=cam.tanfovx, tanfovy=cam.tanfovy, bg=torch.zeros(C, device=dev),
        scale_modifier=1.0, viewmatrix=t(cam.viewmatrix), projmatrix=t(cam.projmatrix), sh_degree=0, campos=t(cam.campos), prefiltered=False, debug=False)
rast = GR(s); dL = t(scenes.make_grad_image(C, H, W, seed=1))
events = []
gc.callbacks.append(lambda phase, info: events.append((time.perf_counter(), "gc-" + phase, info.get("generation"), info.get("collected"))))
def stats():
    m = torch.cuda.memory_stats(dev)
    return (m.get("num_device_alloc", 0), m.get("num_device_free", 0), m.get("num_alloc_retries", 0), m.get("reserved_bytes.all.current", 0) >> 20)
times, marks = [], []
for i in range(nsteps):
    torch.cuda.synchronize(); a = time.perf_counter(); s0 = stats()
    for l in (means3D, feats, opac, scales, rots): l.grad = None
    m2 = torch.zeros_like(means3D, requires_grad=True)
    ta = time.perf_counter()
    color, radii = rast(means3D=means3D, means2D=m2, shs=None, colors_precomp=feats, opacities=opac, scales=scales, rotations=rots, cov3D_precomp=None)
    tb = time.perf_counter()
    torch.autograd.backward(color, grad_tensors=dL)
    tc = time.perf_counter()
    torch.cuda.synchronize(); b = time.perf_counter()
    times.append((b - a) * 1e3)
    marks.append((i, (b - a) * 1e3, (tb - ta) * 1e3, (tc - tb) * 1e3, (b - tc) * 1e3, s0, stats(), [e[1:] for e in events if a <= e[0] <= b]))
med = float(np.median(times))
print(f"{cfgname}: {nsteps} steps, median {med:.3f} ms, mean {np.mean(times):.3f}, max {max(times):.1f}")
for m in marks:
    if m[1] > 2.5 * med:
        print(f"step {m[0]}: {m[1]:.1f} ms = forward call {m[2]:.1f} + backward call {m[3]:.1f} + final sync {m[4]:.1f}; "
              f"allocator (device mallocs, frees, retries, reserved MiB) {m[5]} -> {m[6]}; gc {m[7]}")
