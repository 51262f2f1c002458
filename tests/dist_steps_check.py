"""Helper of tests/test_gpu_api.py::test_three_steps_on_single_rank_rccl_group (run as a subprocess: it initialises a process group).
Three consecutive training-like steps (render a view, backward, all-reduce of the feature gradients, feature update) in two ways:
  plain : everything on the current stream, no torch.distributed;
  dist  : a single-rank RCCL ("nccl") group -- dist.allreduce_grads_async starts the collective without stalling the compute
          stream, the feature update waits for it on a side stream, and the NEXT forward is handed the update's event through
          rasterizer.set_features_ready_event (its geometry stages run ahead, only its blend stage waits).
Prints one JSON line: the relative difference of the step-k gradients and of the final features between the two ways."""
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
from seganygaussians_amd import install_dropin, scenes  # noqa: E402

install_dropin()
from seganygaussians_amd import rasterizer as R  # noqa: E402
from seganygaussians_amd.dist import ViewShardedStep, allreduce_grads_async  # noqa: E402
from diff_gaussian_rasterization_contrastive_f import GaussianRasterizationSettings, GaussianRasterizer  # noqa: E402

dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
P, W, H, C, LR, STEPS = 60_000, 480, 272, 32, 50.0, 3
sc = scenes.make_scene(P, W, H, 400.0, C, np.log(0.03), 0.7, seed=4)
t = lambda a: torch.as_tensor(np.ascontiguousarray(a, np.float32)).to(dev)
means3D, opac, scales, rots, feats0 = t(sc.means3D), t(sc.opacities).requires_grad_(True), t(sc.scales), t(sc.rotations), t(sc.features)
dL = t(scenes.make_grad_image(C, H, W, seed=2))


def render_backward(feats, view):
    cam = scenes.orbit_camera(W, H, 400.0, 0.05 * view, 0.02 * view)
    st = GaussianRasterizationSettings(image_height=H, image_width=W, tanfovx=cam.tanfovx, tanfovy=cam.tanfovy, bg=torch.zeros(C, device=dev),
                                       scale_modifier=1.0, viewmatrix=t(cam.viewmatrix), projmatrix=t(cam.projmatrix), sh_degree=0,
                                       campos=t(cam.campos), prefiltered=False, debug=False)
    color, _ = GaussianRasterizer(st)(means3D=means3D, means2D=torch.zeros_like(means3D), shs=None, colors_precomp=feats, opacities=opac,
                                      scales=scales, rotations=rots, cov3D_precomp=None)
    opac.grad = None
    torch.autograd.backward(color, grad_tensors=dL)
    # what depends on the feature VALUES of this step (dL/dfeatures does not): the image and the opacity gradient
    return color.detach().clone(), opac.grad.clone()


def run(use_dist):
    feats = feats0.clone().requires_grad_(True)
    grads, images = [], []
    side = torch.cuda.Stream(device=dev)
    step = ViewShardedStep([feats])
    for k in range(STEPS):
        if use_dist:
            feats.grad = None
            images.append(render_backward(feats, k))        # (one view per rank and step: rank 0 of 1 renders view k)
            g = feats.grad
            ev, keep = allreduce_grads_async([g])
            with torch.cuda.stream(side):                    # the optimizer step: after the collective, off the compute stream
                side.wait_event(ev)
                feats.data.add_(g, alpha=-LR)
                upd = torch.cuda.Event()
                upd.record(side)
            g.record_stream(side)
            R.set_features_ready_event(upd)                  # the next forward's blend stage waits for the update; its geometry does not
            grads.append((g, upd))
        else:
            assert step.world_size == 1
            step(1, lambda v: images.append(render_backward(feats, k)))     # zeroes the grads, renders view k, (no-op) all-reduce
            feats.data.add_(feats.grad, alpha=-LR)
            grads.append((feats.grad, None))
    torch.cuda.synchronize(dev)
    return [g.clone() for g, _ in grads], feats.detach().clone(), images


plain_g, plain_f, plain_i = run(False)
import torch.distributed as dist  # noqa: E402
os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29547")
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
dist_g, dist_f, dist_i = run(True)
dist.destroy_process_group()
rel = lambda a, b: float((a - b).norm() / b.norm())
# how much the images would differ had a step read the features of the step before (same view, stale features)
stale = []
for k in range(1, STEPS):
    f_prev = feats0.clone()
    for j in range(k - 1):
        f_prev.add_(plain_g[j], alpha=-LR)
    stale.append(rel(render_backward(f_prev.requires_grad_(True), k)[0], plain_i[k][0]))
out = {"grad_rel": [rel(a, b) for a, b in zip(dist_g, plain_g)], "feat_rel": rel(dist_f, plain_f),
       "image_rel": [rel(a[0], b[0]) for a, b in zip(dist_i, plain_i)], "dopacity_rel": [rel(a[1], b[1]) for a, b in zip(dist_i, plain_i)],
       "moved": rel(plain_f, feats0), "stale_image_rel": stale}
import ctypes  # noqa: E402
ctypes.CDLL(None).fflush(None)
print(json.dumps(out), flush=True)
