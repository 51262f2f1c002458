#!/usr/bin/env python
"""bench.py -- BASELINE.json's headline metric on MI355X.

metric : train views/sec (fwd+bwd), 1080p, 1M Gaussians, 32-D features  (BASELINE config 3;
         config 4 when --gpus N > 1: one view per GPU per step + RCCL all-reduce of the (P,32)
         feature gradient).
step   : ONE forward + backward of the rasterizer hot path over one synthetic view, through the
         drop-in Python API (GaussianRasterizer -> autograd backward), inputs resident in HBM.
         Every allocation, zero-fill and the num_rendered host sync are inside the timed region.

Usage:  python bench.py [--gpus N --steps K --warmup W]
        N > 1:  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1
                --master-port P bench.py --gpus N --steps K --warmup W

Prints ONE JSON line (rank 0).  Extra objects:
  roofline     dominant kernel: algorithmic bytes per launch (SURVEY.md 8(d) formula) / its
               average duration measured live with HIP events on the launch stream; peak 8000 GB/s.
  cpu_baseline the CPU oracle (kind "port") on this box's host cores, one full view (rank 0, N == 1).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

HBM_PEAK_GBPS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec peak


def algorithmic_bytes(c, C):
    """SURVEY.md 8(d) per-stage algorithmic bytes for one view (fp32, C channels)."""
    P, V, R, E, L, N, Tn = c["P"], c["V"], c["R"], c["E"], c["L"], c["N"], c["tiles"]
    p = (c["sort_bits"] + 7) // 8
    # binning rows keep the REFERENCE algorithm's byte counts (scan / duplicate / 45-bit sort / ranges), mapped
    # onto the stages of our pipeline that replace them (DESIGN.md "Binning")
    st = {
        "preprocess": 44 * P + 8 * P + 52 * V,
        "tile_scan": 8 * P,
        "emit": 8 * P + 12 * V + 12 * R,
        "tile_sort": (24 * p + 8) * R + 8 * R + 8 * Tn,
        "depth_sort": 0,
        "blend_fwd": (28 + 4 * C) * E + (4 * C + 8) * N,
        "blend_bwd": (28 + 4 * C) * L + (4 * C + 8) * N + 2 * 4 * (C + 6) * L,
        "grad_zero_init": 4 * (24 + C) * P,
        "geom_bwd": 4 * P * 2 + (88 + 128) * V,
    }
    st["total"] = sum(st.values())
    return st


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--config", default="cfg3", help="cfg3 (default, headline) | cfg5 | cfg1")
    ap.add_argument("--points", type=int, default=None, help="override Gaussian count (debug only; invalidates the metric)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-threads", type=int, default=0)
    ap.add_argument("--dist-single", action="store_true",
                    help="testing only: initialise torch.distributed (RCCL) with a single rank and run the N > 1 step")
    args = ap.parse_args()

    import numpy as np
    import torch

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus > 1 and world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} needs torch.distributed.run with {args.gpus} ranks (WORLD_SIZE={world})")
    dev = torch.device("cuda", local_rank)
    torch.cuda.set_device(dev)
    dist = None
    if world > 1 or args.dist_single:
        import torch.distributed as dist
        if args.dist_single and "MASTER_ADDR" not in os.environ:
            os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT="29533", RANK="0", WORLD_SIZE="1")
        dist.init_process_group("nccl", device_id=dev)

    from seganygaussians_amd import _lib, install_dropin, scenes
    install_dropin()
    from diff_gaussian_rasterization_contrastive_f import GaussianRasterizationSettings
    from seganygaussians_amd.dist import allreduce_grads_async
    from seganygaussians_amd.rasterizer import make_rasterizer, set_features_ready_event

    cfg = scenes.CONFIGS[args.config]
    C, W, H = cfg["C"], cfg["W"], cfg["H"]
    P = cfg["P"] if args.points is None else args.points
    _, _, GaussianRasterizer = make_rasterizer(C)
    scene = scenes.make_scene(P, W, H, cfg["focal"], C, cfg["ls_mean"], cfg["ls_std"], seed=0)
    # one camera per rank: rank 0 is the canonical front view of config 3; other ranks orbit (config 4)
    cam = scenes.look_at_camera(W, H, cfg["focal"]) if rank == 0 else \
        scenes.orbit_camera(W, H, cfg["focal"], 0.05 * rank, 0.02 * rank)
    t = lambda a: torch.as_tensor(np.ascontiguousarray(a, np.float32)).to(dev)
    means3D = t(scene.means3D).requires_grad_(True)
    feats = t(scene.features).requires_grad_(True)
    opac = t(scene.opacities).requires_grad_(True)
    scales = t(scene.scales).requires_grad_(True)
    rots = t(scene.rotations).requires_grad_(True)
    settings = GaussianRasterizationSettings(
        image_height=H, image_width=W, tanfovx=cam.tanfovx, tanfovy=cam.tanfovy, bg=torch.zeros(C, device=dev),
        scale_modifier=1.0, viewmatrix=t(cam.viewmatrix), projmatrix=t(cam.projmatrix), sh_degree=0,
        campos=t(cam.campos), prefiltered=False, debug=False)
    rasterizer = GaussianRasterizer(settings)
    dL = t(scenes.make_grad_image(C, H, W, seed=1))
    leaves = [means3D, feats, opac, scales, rots]

    state = {}

    def step():
        for l in leaves:
            l.grad = None
        # config 4: the features of this step are "ready" when the previous step's gradient all-reduce is (in training the
        # optimizer step sits between the two).  Only the blend stage of the forward reads them; the geometry stages of this
        # view -- SAGA trains the feature rows alone, scene/gaussian_model_ff.py:154-162 -- run while the gradients travel.
        ev, _keep = state.pop("pending", (None, None))
        if ev is not None:
            set_features_ready_event(ev)
        means2D = torch.zeros_like(means3D, requires_grad=True)
        color, radii = rasterizer(means3D=means3D, means2D=means2D, shs=None, colors_precomp=feats, opacities=opac,
                                  scales=scales, rotations=rots, cov3D_precomp=None)
        torch.autograd.backward(color, grad_tensors=dL)
        if dist is not None:
            # sum the per-Gaussian feature gradients of the N views over RCCL/xGMI: one flat 128-MB bucket, asynchronous
            state["pending"] = allreduce_grads_async([feats.grad])
        state["radii"] = radii

    def barrier():
        ev, _keep = state.pop("pending", (None, None))
        if ev is not None:
            ev.synchronize()  # the last step's all-reduce belongs to the timed region
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize(dev)

    for _ in range(args.warmup):
        step()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    barrier()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        tt = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
    ms_per_step = 1e3 * elapsed / args.steps
    value = world * args.steps / elapsed

    # ---- counters + live per-stage HIP-event timing (separate, un-timed steps) -------------------
    roofline = None
    stages_ms = {}
    counters = {}
    if rank == 0:
        _lib.profile_enable(True)
        acc = {k: 0.0 for k in _lib.MI_STAGES}
        nprof = max(3, min(10, args.steps))
        for _ in range(nprof):
            step()
            torch.cuda.synchronize(dev)
            ms = _lib.profile_read()
            for k in acc:
                acc[k] += ms[k]
        _lib.profile_enable(False)
        stages_ms = {k: v / nprof for k, v in acc.items()}
        # counters: one extra (un-timed) native forward whose opaque buffers we can inspect
        from seganygaussians_amd.rasterizer import rasterize_gaussians_native
        e = torch.empty(0)
        # E and L are counters of the REFERENCE algorithm (positions in its full tile lists): full-list mode for this call
        prev_mode = _lib.load().mi_rast_set_full_lists(1)
        with torch.no_grad():
            num_rendered, _c, _r, _g, _b, imgbuf = rasterize_gaussians_native(
                C, False, settings.bg, means3D, feats, opac, None, scales, rots, 1.0, e, settings.viewmatrix,
                settings.projmatrix, cam.tanfovx, cam.tanfovy, H, W, e, 0, settings.campos, False, False)
        _lib.load().mi_rast_set_full_lists(prev_mode)
        _, ioff = _lib.image_layout(W, H)
        tiles_x, tiles_y = (W + 15) // 16, (H + 15) // 16
        nc = imgbuf[ioff["n_contrib"]:ioff["n_contrib"] + 4 * W * H].view(torch.int32).reshape(H, W)
        pad = torch.zeros((tiles_y * 16, tiles_x * 16), dtype=torch.int32, device=dev)
        pad[:H, :W] = nc
        L = int(pad.reshape(tiles_y, 16, tiles_x, 16).amax(dim=(1, 3)).sum().item())
        cons = imgbuf[ioff["tile_consumed"]:ioff["tile_consumed"] + 4 * tiles_x * tiles_y].view(torch.int32)
        E = int(cons.sum().item())
        V = int((state["radii"] > 0).sum().item())
        counters = dict(P=P, V=V, R=int(num_rendered), E=E, L=L, N=W * H, tiles=tiles_x * tiles_y,
                        sort_bits=32 + int(_lib.load().mi_rast_get_higher_msb(tiles_x * tiles_y)))
        ab = algorithmic_bytes(counters, C)
        dom = max((k for k in stages_ms), key=lambda k: stages_ms[k])
        achieved = ab[dom] / (stages_ms[dom] * 1e-3) / 1e9 if stages_ms[dom] > 0 else 0.0
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "traffic.json")
        if os.path.exists(tpath):
            try:
                traffic = json.load(open(tpath)).get(dom, {}).get("bytes_per_launch")
            except Exception:
                traffic = None
        roofline = {"bound": "hbm", "kernel": dom, "achieved": round(achieved, 1), "peak": HBM_PEAK_GBPS,
                    "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBPS, 4), "traffic": traffic,
                    "algorithmic_bytes": ab[dom], "kernel_ms": round(stages_ms[dom], 4),
                    "whole_view": {"algorithmic_bytes": ab["total"],
                                   "achieved": round(ab["total"] / (ms_per_step * 1e-3) / 1e9, 1),
                                   "frac": round(ab["total"] / (ms_per_step * 1e-3) / 1e9 / HBM_PEAK_GBPS, 4),
                                   # also against what a float4 copy reaches on this part (6.3 TB/s)
                                   "frac_of_copy_rate": round(ab["total"] / (ms_per_step * 1e-3) / 1e9 / 6300.0, 4)}}

    # ---- CPU baseline: the oracle on this box's host cores (rank 0, N == 1 only) ------------------
    cpu_baseline = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        from oracle import saga_oracle as so
        if args.cpu_threads > 0:
            so.set_num_threads(args.cpu_threads)
        inp = so.Inputs(means3D=scene.means3D, opacities=scene.opacities, viewmatrix=cam.viewmatrix,
                        projmatrix=cam.projmatrix, campos=cam.campos, bg=np.zeros(C, np.float32), image_width=W,
                        image_height=H, tanfovx=cam.tanfovx, tanfovy=cam.tanfovy, channels=C,
                        colors_precomp=scene.features, scales=scene.scales, rotations=scene.rotations)
        dLn = dL.cpu().numpy()
        c0 = time.perf_counter()
        fo = so.forward(inp)
        so.backward(inp, fo, dLn)
        cpu_s = time.perf_counter() - c0
        cpu_baseline = {"value": round(1.0 / cpu_s, 4), "unit": "views/s", "cores": so.num_threads(), "kind": "port",
                        "sample": f"1 full view fwd+bwd of {args.config} (P={P}, {W}x{H}, C={C}) in {cpu_s:.2f} s, "
                                  f"OpenMP over Gaussians/tiles, nproc={os.cpu_count()}"}

    if rank == 0:
        out = {
            "metric": "train views/sec (fwd+bwd), 1080p, 1M Gaussians, 32-D features" if args.config == "cfg3"
            else f"train views/sec (fwd+bwd), {args.config}",
            "value": round(value, 3), "unit": "views/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 4), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"BASELINE {args.config}: {P} Gaussians, {W}x{H}, {C}-D features, fwd+bwd, "
                                   f"1 view/GPU/step" + (", RCCL all-reduce of (P,C) feature grads" if world > 1 else ""),
                       "parallelism": f"view-sharded x{world}", "counters": counters,
                       "lists": "lean (product default: only overlaps that pass the exact-conservative cull are listed; "
                                "counters E/L from one full-list call)",
                       "arithmetic": "f32 throughout; C=32 forward accumulation = exact 3-way bf16 split of f32 operands, "
                                     "six partial products on the bf16 matrix pipe, f32 accumulate (f32 rounding level)",
                       "stages_ms": {k: round(v, 4) for k, v in stages_ms.items()}},
            "roofline": roofline, "cpu_baseline": cpu_baseline,
        }
    if dist is not None:
        dist.destroy_process_group()
    if rank == 0:
        # RCCL prints a version banner through C stdio, which reaches a redirected stdout only when it is flushed: do that
        # first, so that the JSON line is the LAST line of stdout
        import ctypes
        try:
            ctypes.CDLL(None).fflush(None)
        except Exception:  # noqa: BLE001
            pass
        sys.stdout.flush()
        print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
