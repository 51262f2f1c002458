#!/usr/bin/env python
"""bench.py -- BASELINE.json's headline metric on MI355X.

metric : train views/sec (fwd+bwd), 1080p, 1M Gaussians, 32-D features  (BASELINE config 3;
         config 4 when --gpus N > 1: one view per GPU per step + RCCL all-reduce of the (P,32)
         feature gradient).
step   : ONE forward + backward of the rasterizer hot path over one synthetic view, through the
         drop-in Python API (GaussianRasterizer -> autograd backward), inputs resident in HBM.
         Every allocation, zero-fill and the num_rendered host read-back are inside the timed region.

Usage:  python bench.py [--gpus N --steps K --warmup W] [--config cfg3|cfg5|cfg2|cfg1]
        N > 1:  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1
                --master-port P bench.py --gpus N --steps K --warmup W
        cfg2 = BASELINE config 2: 1M Gaussians, 1080p, SH degree 3 RGB + mask + depth, FORWARD only, through the
        diff_gaussian_rasterization_depth drop-in; cfg5 = 5M Gaussians, 1600x1063, 64-D, fwd+bwd.
        Defaults: N = 1, K = 100 timed steps, W = 10 warm-up steps behind untimed settling blocks (--settle seconds, see
        main()); the whole default run, CPU baseline and parity check included, takes about half a minute.
        --fast-exp / --ref-on-gpu / --dist-single: reporting and testing aids (see their help texts).

Prints ONE JSON line (rank 0).  Extra objects:
  roofline     dominant kernel: algorithmic bytes per launch (SURVEY.md 8(d) formula) / its
               average duration measured live with HIP events on the launch stream; peak 8000 GB/s.
               `stages` lists EVERY stage the same way; a stage whose frac exceeds 1 is flagged
               "algorithm replaced" (the reference's byte count for a pass this pipeline does not perform:
               never read it as bandwidth).  `whole_view.frac` leaves the replaced stages' bytes out (the number to read),
               `whole_view.frac_survey_bytes` is SURVEY.md 8(d)'s literal byte table, `whole_view.frac_traffic` the HBM bytes the
               counters saw.
               `traffic` / `alu` (matrix / vector pipe busy fractions of the dominant kernel) come from the committed PMC passes
               (profiles/traffic_<cfg>.json, alu_<cfg>.json) and are null unless those were taken with the library being timed
               (stamp = mi_rast_version(): a hash of sources, headers and flags).
  cpu_baseline the CPU oracle (kind "port", gcc -O3 -march=native on this box) on this box's host cores, --cpu-views (3) full
               views (rank 0, N == 1).
  timing       median / p10 / p90 of ms per step over untimed blocks of ten steps after the timed region.
  parity       that same oracle run compared with the GPU outputs of the benchmarked configuration (product default
               lists): image and gradients, max-norm criterion of the tests + norm-wise error.
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


def algorithmic_bytes(c, C, forward_only=False, sh_coeffs=0, extra=0):
    """SURVEY.md 8(d) per-stage algorithmic bytes for one view (fp32, C channels; `extra` = mask + depth planes of the
    DEPTH variant, `sh_coeffs` > 0 adds the SH evaluation of the preprocess pass)."""
    P, V, R, E, L, N, Tn = c["P"], c["V"], c["R"], c["E"], c["L"], c["N"], c["tiles"]
    p = (c["sort_bits"] + 7) // 8
    Cx = C + extra
    # binning rows keep the REFERENCE algorithm's byte counts (scan / duplicate / 45-bit sort / ranges), mapped
    # onto the stages of our pipeline that replace them (DESIGN.md "Binning")
    st = {
        "preprocess": 44 * P + 8 * P + 52 * V + ((12 * sh_coeffs + 15) * V if sh_coeffs else 0),
        "tile_scan": 8 * P,
        "emit": 8 * P + 12 * V + 12 * R,
        "tile_sort": (24 * p + 8) * R + 8 * R + 8 * Tn,
        "blend_fwd": (28 + 4 * Cx) * E + (4 * Cx + 8) * N,
    }
    if not forward_only:
        st.update({
            "blend_bwd": (28 + 4 * C) * L + (4 * C + 8) * N + 2 * 4 * (C + 6) * L,
            "grad_zero_init": 4 * (24 + C) * P,
            "geom_bwd": 4 * P * 2 + (88 + 128) * V,
        })
    st["total"] = sum(st.values())
    return st


def _close(got, want, rtol=1e-4):
    """(fraction outside |d| <= rtol*|want| + rtol*max|want|, norm-wise error) -- tests/helpers.py's criterion."""
    import numpy as np
    got = np.asarray(got, np.float64).ravel()
    want = np.asarray(want, np.float64).ravel()
    if want.size == 0:
        return 0.0, 0.0
    scale = float(np.abs(want).max())
    err = np.abs(got - want)
    frac = float((~(err <= rtol * np.abs(want) + rtol * max(scale, 1e-30))).mean())
    nrm = float(np.linalg.norm(want))
    return frac, (float(np.linalg.norm(got - want)) / nrm if nrm > 0 else 0.0)


def _rows_outside(got, want, rtol=1e-4):
    """tests/helpers.py: row_error_report -- share of the non-zero rows r with ||got_r - want_r||_inf > rtol * (||want_r||_inf +
    median over non-zero rows of ||want_r'||_inf)."""
    import numpy as np
    want = np.asarray(want, np.float64)
    want = want.reshape(want.shape[0], -1)
    got = np.asarray(got, np.float64).reshape(want.shape)
    mag = np.abs(want).max(axis=1)
    nz = mag > 0
    if not nz.any():
        return 0.0
    err = np.abs(got - want).max(axis=1)
    return float(((err > rtol * (mag + float(np.median(mag[nz])))) & nz).sum() / nz.sum())


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--config", default="cfg3", help="cfg3 (default, headline) | cfg5 | cfg2 (forward only) | cfg1 | cfg3s / cfg5s (the same "
                                                     "sizes under the second synthetic law, scenes.make_surface_scene)")
    ap.add_argument("--ply", default=None, help="a trained 3DGS point_cloud.ply (scene/gaussian_model.py:271-322) instead of the synthetic "
                                                "Gaussians: with --cameras, runs --config's workload (its channel count and passes) on the real scene")
    ap.add_argument("--cameras", default=None, help="a COLMAP scene directory (sparse/0/{cameras,images}.bin|txt): rank r renders its "
                                                    "camera --camera-index + r (sorted by image name like the reference's training list)")
    ap.add_argument("--camera-index", type=int, default=0)
    ap.add_argument("--images", default=None, help="with --cameras: the image folder the reference run was trained on (its -i option, e.g. "
                                                   "<scene>/images_4): the render takes the size of the camera's image FILE like the reference's "
                                                   "loadCam; default: the COLMAP camera model's size (= the full-resolution `images` folder)")
    ap.add_argument("--resolution", type=float, default=-1, help="with --cameras: the reference's --resolution (utils/camera_utils.py:20-40): "
                                                                 "-1 = camera size, width capped at 1600; 1/2/4/8 = divisor; else target width")
    ap.add_argument("--feature-ply", default=None, help="with --ply: feature rows from a FeatureGaussianModel.save_ply file instead of seeded noise")
    ap.add_argument("--dump-grads", default=None,
                    help="testing aid: after everything else, one more step from zeroed gradients; rank 0 saves the (all-reduced) "
                         "feature gradient and every rank its camera to this directory (.npy)")
    ap.add_argument("--sustained-seconds", type=float, default=2.0,
                    help="after the K timed steps: views/s over at least this many seconds of back-to-back steps with ONE synchronisation at "
                         "the end (`sustained`; what a 10 000-iteration training run sees; 0: skip)")
    ap.add_argument("--points", type=int, default=None, help="override Gaussian count (debug only; invalidates the metric)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-threads", type=int, default=0)
    ap.add_argument("--rs-ag", action="store_true",
                    help="N > 1: exchange the feature gradients as reduce-scatter -> rank-local feature update (plain SGD, lr "
                         "--rs-ag-lr, on this rank's 1/N of the rows) -> all-gather of the updated rows (dist.sharded_update_async) "
                         "instead of one all-reduce; the next view's blend stage waits for the all-gather only")
    ap.add_argument("--rs-ag-lr", type=float, default=1e-4)
    ap.add_argument("--dist-single", action="store_true",
                    help="testing only: initialise torch.distributed (RCCL) with a single rank and run the N > 1 step")
    ap.add_argument("--settle", type=float, default=3.0,
                    help="minimum seconds of untimed settling blocks before the W warm-up steps (0 under a profiler: the "
                         "trace would hold thousands of views)")
    ap.add_argument("--dist-blocks", type=int, default=20,
                    help="untimed blocks of ten steps after the timed region for the median / p10 / p90 of ms per step (0: skip)")
    ap.add_argument("--cpu-views", type=int, default=3, help="views the CPU baseline is timed over (SURVEY.md 8(d): >= 3)")
    ap.add_argument("--fast-exp", action="store_true",
                    help="run the product in its MI_RAST_FAST_EXP mode (v_exp_f32 instead of expf; noted in "
                         "config.arithmetic -- the headline number is the default mode)")
    ap.add_argument("--exact-exp", action="store_true",
                    help="forward blend with the device library's expf for every pair (MI_RAST_EXACT_EXP) instead of the product "
                         "default, the hybrid form (expf only where a pixel comes within 4e-6 of the alpha >= 1/255 cut): A/B aid")
    ap.add_argument("--tile-fwd", action="store_true",
                    help="A/B aid: 32/64-channel forward on the tile-batched kernel (MI_RAST_TILE_FWD) instead of the wave-per-quadrant one; "
                         "a comparison kernel of the PROFILING build: run with MI_RAST_LIB=seganygaussians_amd/libmi_rast_prof.so "
                         "(python -m seganygaussians_amd.build --profiling)")
    ap.add_argument("--equal-runs", action="store_true",
                    help="A/B aid: the blend kernels' XCD runs at equal tile counts (MI_RAST_EQUAL_RUNS) instead of equal modelled work")
    ap.add_argument("--features-only-grad", action="store_true",
                    help="adds a separately labelled result `features_only_backward` (never the headline): the same fwd+bwd steps with the "
                         "opt-in features-only backward of the drop-in on (rasterizer.enable_features_only_backward / "
                         "MI_RAST_FEATURES_ONLY_BACKWARD=1: SAGA's feature training optimises the feature rows only, the reference computes "
                         "the seven geometry gradients all the same and nobody reads them); with --frozen-geometry also the two opt-ins together")
    ap.add_argument("--frozen-geometry", action="store_true",
                    help="adds a SECOND, separately labelled result `frozen_geometry` (never the headline): the same fwd+bwd steps with the "
                         "opt-in per-camera geometry cache of the drop-in on (rasterizer.GeometryCache: SAGA's feature training optimises the "
                         "feature rows only and revisits its cameras), cycling over --views cameras; first visits are inside its timed region")
    ap.add_argument("--views", type=int, default=16, help="with --frozen-geometry: number of cameras cycled through")
    ap.add_argument("--frozen-passes", type=int, default=8, help="with --frozen-geometry: visits per camera in the timed region (the first one fills the cache)")
    ap.add_argument("--ref-on-gpu", action="store_true",
                    help="reporting only, after the timed region: also time oracle/_ref (the reference's own kernels, translated "
                         "test-only by oracle/build_ref.py) on the same workload and GPU")
    args = ap.parse_args()

    import numpy as np
    import torch

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus > 1 and world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} needs torch.distributed.run with {args.gpus} ranks (WORLD_SIZE={world})")
    # MI_BENCH_SHARE_GPU=1 + MI_BENCH_DIST_BACKEND=gloo (testing only, tests/test_gpu_api.py): every rank on cuda:0 with a CPU-staged
    # collective -- RCCL refuses two ranks on one device --, so that the N > 1 code path of THIS file (settle-flag all-reduce, per-block
    # MAX, asynchronous gradient all-reduce + features-ready event, every rank's own camera) runs with a real second rank on a
    # one-GPU box.  Its timings mean nothing.
    share_gpu = os.environ.get("MI_BENCH_SHARE_GPU") == "1"
    backend = os.environ.get("MI_BENCH_DIST_BACKEND", "nccl")
    dev = torch.device("cuda", 0 if share_gpu else local_rank)
    torch.cuda.set_device(dev)
    dist = None
    if world > 1 or args.dist_single:
        import torch.distributed as dist
        if args.dist_single and "MASTER_ADDR" not in os.environ:
            os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT="29533", RANK="0", WORLD_SIZE="1")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)

    # Who takes part in the exchange (config.comm of the JSON line): every rank's device as the driver sees it, gathered over the SAME
    # process group the gradients travel on -- a line whose n_gpus is N then shows N distinct devices (or shows that it does not).
    comm_ranks = None
    if dist is not None:
        pr_ = torch.cuda.get_device_properties(dev)
        ident = {"rank": rank, "local_rank": local_rank, "device_index": dev.index, "name": pr_.name,
                 "uuid": str(getattr(pr_, "uuid", "")) or None,
                 "pci": ("%04x:%02x:%02x" % (getattr(pr_, "pci_domain_id", 0), getattr(pr_, "pci_bus_id", 0), getattr(pr_, "pci_device_id", 0)))
                 if hasattr(pr_, "pci_bus_id") else None,
                 "pid": os.getpid(), "visible_devices": os.environ.get("HIP_VISIBLE_DEVICES") or os.environ.get("ROCR_VISIBLE_DEVICES")}
        comm_ranks = [None] * dist.get_world_size()
        dist.all_gather_object(comm_ranks, ident)

    from seganygaussians_amd import _lib, install_dropin, scenes
    if args.tile_fwd:
        from seganygaussians_amd import build as _build
        if os.path.realpath(os.environ.get("MI_RAST_LIB", "")) != os.path.realpath(_build.PROF_LIB_PATH):
            sys.exit("--tile-fwd selects a comparison kernel that only the profiling build holds: "
                     f"MI_RAST_LIB={_build.PROF_LIB_PATH} python bench.py --tile-fwd ...  (python -m seganygaussians_amd.build --profiling builds it)")
    install_dropin()
    from seganygaussians_amd import rasterizer as R
    from seganygaussians_amd.dist import allreduce_grads_async, sharded_update_async

    cfg = scenes.CONFIGS[args.config]
    C, W, H = cfg["C"], cfg["W"], cfg["H"]
    fwd_only = args.config == "cfg2"      # BASELINE config 2: RGB (SH degree 3) + mask + depth, forward
    data = "synthetic"
    if args.ply:
        # the named scenes of BASELINE configs 2-5 (garden / bicycle) where the data exists: Gaussians from the trained 3DGS ply,
        # cameras from the scene's COLMAP model, both as the reference loads them (seganygaussians_amd/colmap_io.py)
        from seganygaussians_amd import colmap_io
        if not args.cameras:
            raise SystemExit("--ply needs --cameras <COLMAP scene directory>")
        scene = colmap_io.load_3dgs_scene(args.ply, C, seed=0, feature_ply=args.feature_ply)
        P = scene.means3D.shape[0]
        cams = colmap_io.read_colmap_cameras(args.cameras)
        cc = cams[(args.camera_index + rank) % len(cams)]
        # the reference sizes the render from the image FILE (utils/camera_utils.py:21): --images <dir> (e.g. the scene's images_4)
        # reads the size of this camera's file there; without it the COLMAP camera model's size stands in (= the `images` folder)
        img_size = None
        if args.images:
            img_path = os.path.join(args.images, cc.name)
            if not os.path.isfile(img_path):
                raise SystemExit(f"--images {args.images}: no file {cc.name} for camera {args.camera_index + rank} of {args.cameras} "
                                 "(the folder must be the one the reference run was trained on, e.g. <scene>/images_4)")
            img_size = colmap_io.image_size_of(img_path)
        cam = colmap_io.to_camera(cc, args.resolution, image_size=img_size)
        W, H = cam.image_width, cam.image_height
        data = f"file: {os.path.basename(os.path.dirname(os.path.abspath(args.ply))) or args.ply} ({P} Gaussians), camera {cc.name}"
    else:
        P = cfg["P"] if args.points is None else args.points
        scene = scenes.scene_of_config(args.config, seed=0, P=P, with_shs=fwd_only)
        # one camera per rank: rank 0 is the canonical front view of config 3; other ranks orbit (config 4)
        cam = scenes.look_at_camera(W, H, cfg["focal"]) if rank == 0 else \
            scenes.orbit_camera(W, H, cfg["focal"], 0.05 * rank, 0.02 * rank)
    t = lambda a: torch.as_tensor(np.ascontiguousarray(a, np.float32)).to(dev)
    means3D = t(scene.means3D).requires_grad_(not fwd_only)
    opac = t(scene.opacities).requires_grad_(not fwd_only)
    scales = t(scene.scales).requires_grad_(not fwd_only)
    rots = t(scene.rotations).requires_grad_(not fwd_only)
    if fwd_only:
        from diff_gaussian_rasterization_depth import GaussianRasterizationSettings, GaussianRasterizer
        shs = t(scene.shs)
        mask = torch.ones(P, device=dev)
        feats = None
        leaves = []
    else:
        from diff_gaussian_rasterization_contrastive_f import GaussianRasterizationSettings
        _, _, GaussianRasterizer = R.make_rasterizer(C)
        feats = t(scene.features).requires_grad_(True)
        leaves = [means3D, feats, opac, scales, rots]
    settings = GaussianRasterizationSettings(
        image_height=H, image_width=W, tanfovx=cam.tanfovx, tanfovy=cam.tanfovy, bg=torch.zeros(C, device=dev),
        scale_modifier=1.0, viewmatrix=t(cam.viewmatrix), projmatrix=t(cam.projmatrix), sh_degree=3 if fwd_only else 0,
        campos=t(cam.campos), prefiltered=False, debug=False)
    rasterizer = GaussianRasterizer(settings)
    dL = t(scenes.make_grad_image(C, H, W, seed=1))

    state = {}
    host_trace = [] if os.environ.get("MI_BENCH_HOST_TRACE") else None
    t_start = time.perf_counter()

    def phase(msg):   # MI_BENCH_TRACE=1: where a run is (stderr, every rank), for hangs that only show with several ranks
        if os.environ.get("MI_BENCH_TRACE"):
            print(f"[bench rank {rank}] {time.perf_counter() - t_start:8.3f} s  {msg}", file=sys.stderr, flush=True)

    def step():
        if args.fast_exp or args.tile_fwd or args.exact_exp or args.equal_runs:
            with R.forward_flags(fast_exp=True if args.fast_exp else None, tile_fwd=True if args.tile_fwd else None,
                                 exact_exp=True if args.exact_exp else None, equal_runs=True if args.equal_runs else None):
                return step_()
        return step_()

    def step_():
        for l in leaves:
            l.grad = None
        if fwd_only:
            with torch.no_grad():
                means2D = torch.zeros_like(means3D)
                color, omask, odepth, radii = rasterizer(means3D=means3D, means2D=means2D, opacities=opac, mask=mask,
                                                         shs=shs, colors_precomp=None, scales=scales, rotations=rots,
                                                         cov3D_precomp=None)
            state.update(radii=radii, color=color, mask=omask, depth=odepth)
            return
        # config 4: the features of this step are "ready" when the previous step's gradient all-reduce is (in training the
        # optimizer step sits between the two).  Only the blend stage of the forward reads them; the geometry stages of this
        # view -- SAGA trains the feature rows alone, scene/gaussian_model_ff.py:154-162 -- run while the gradients travel.
        ev, _keep = state.pop("pending", (None, None))
        if ev is not None:
            R.set_features_ready_event(ev)
        t_a = time.perf_counter()
        means2D = torch.zeros_like(means3D, requires_grad=True)
        color, radii = rasterizer(means3D=means3D, means2D=means2D, shs=None, colors_precomp=feats, opacities=opac,
                                  scales=scales, rotations=rots, cov3D_precomp=None)
        t_b = time.perf_counter()
        torch.autograd.backward(color, grad_tensors=dL)
        if host_trace is not None:   # MI_BENCH_HOST_TRACE: host time of the two calls of every step (no extra syncs)
            host_trace.append((t_a, t_b, time.perf_counter()))
        if dist is not None and not state.get("solo"):
            # sum the per-Gaussian feature gradients of the N views over RCCL/xGMI: one flat 128-MB bucket, asynchronous
            # (not in rank 0's reporting-only steps behind the timed regions: a collective only one rank enters never completes)
            if args.rs_ag:
                # reduce-scatter -> this rank updates its 1/N of the feature rows -> all-gather of the updated rows: the same bytes
                # on the wire, but only the all-gather (and an optimizer pass over 1/N of the rows) in front of the next blend
                lr = args.rs_ag_lr
                state["pending"] = sharded_update_async(feats, feats.grad, lambda prow, grow, lo, hi: prow.add_(grow, alpha=-lr))
            else:
                state["pending"] = allreduce_grads_async([feats.grad])
        state.update(radii=radii, color=color.detach(), means2D=means2D)

    def barrier():
        ev, _keep = state.pop("pending", (None, None))
        if ev is not None:
            ev.synchronize()  # the last step's all-reduce belongs to the timed region
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize(dev)

    # Settling, untimed and before the W warm-up steps: blocks of max(5, W) steps until the blocks of the last second (at least five) all lie
    # within 5 % of the fastest block seen, for at least 3 s and at most 20 s of wall time.  The FIRST process on a freshly
    # provisioned box runs erratically for its first seconds: measured on cfg5, blocks of ten steps took
    # 59, 72, 19, 3.5, 57, 34, 3.6, 3.5 ms/step (and a timed region right after that still caught a burst: 6.2 ms/step),
    # while the same command started again on the same box gives 19 (first step: code-object load), 3.55, 3.55 and then
    # stays there.  The kernels' own durations are normal throughout -- the host thread is what stalls (each view has one
    # host read-back, the reference API's `num_rendered`), presumably while the box's image is still paging in.  The W
    # warm-up steps alone (15 ms of work) do not cover that; waiting it out is not part of the measurement.
    block_ms, block_end = [], []
    nb = max(5, args.warmup)
    t_settle = time.perf_counter()
    phase("settling")
    while True:
        barrier()
        phase(f"settling block {len(block_ms)}")
        tb = time.perf_counter()
        for _ in range(nb):
            step()
        barrier()
        now = time.perf_counter()
        block_ms.append(round(1e3 * (now - tb) / nb, 3))
        block_end.append(now)
        # the trailing window: the last five blocks, and every block that ended within the last second (five blocks of a
        # 3-ms step are 0.15 s -- too short to call a box settled whose stalls come seconds apart)
        last = [m for m, te in zip(block_ms, block_end) if now - te <= 1.0]
        last = block_ms[-5:] if len(last) < 5 else last
        settled = ((len(block_ms) >= 5 and max(last) <= 1.05 * min(block_ms) and now - t_settle >= args.settle) or now - t_settle >= 20.0
                   or args.settle <= 0)
        if dist is not None:   # every rank must run the same number of steps (each step holds a collective)
            flag = torch.tensor([0 if settled else 1], device=dev, dtype=torch.int32)
            dist.all_reduce(flag, op=dist.ReduceOp.MAX)
            settled = int(flag.item()) == 0
        if settled:
            break
    phase("warm-up")
    for _ in range(args.warmup):
        step()
    barrier()
    phase("timed region")
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
    # Sustained throughput, reporting only: >= --sustained-seconds of steps back to back, bracketed like the timed region.  A
    # 10 000-iteration training run (README.md:82 of the reference) never pauses; the K-step figure above is a burst behind settling
    # blocks with a synchronisation each, and the chip settles a few per cent lower under an uninterrupted load.
    sustained = None
    phase("sustained region")
    if args.sustained_seconds > 0:
        n_sus = max(args.steps, int(args.sustained_seconds / max(elapsed / args.steps, 1e-5)) + 1)
        if dist is not None:   # every rank the same count
            tn = torch.tensor([n_sus], device=dev, dtype=torch.int64)
            dist.all_reduce(tn, op=dist.ReduceOp.MAX)
            n_sus = int(tn.item())
        barrier()
        ts = time.perf_counter()
        for _ in range(n_sus):
            step()
        barrier()
        sus = time.perf_counter() - ts
        if dist is not None:
            tt = torch.tensor([sus], device=dev, dtype=torch.float64)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            sus = float(tt.item())
        sustained = {"value": round(world * n_sus / sus, 3), "unit": "views/s", "ms_per_step": round(1e3 * sus / n_sus, 4),
                     "steps": n_sus, "seconds": round(sus, 3),
                     "note": "back-to-back steps, one synchronisation at the end, after the K timed steps"}
    # distribution (median + p10 / p90), reporting only and AFTER the timed region: blocks of ten steps, one
    # synchronisation per block, max over ranks per block
    dist_blocks = []
    phase("distribution blocks")
    for _ in range(max(0, args.dist_blocks)):
        barrier()
        tb = time.perf_counter()
        for _ in range(10):
            step()
        barrier()
        dtb = time.perf_counter() - tb
        if dist is not None:
            tt = torch.tensor([dtb], device=dev, dtype=torch.float64)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            dtb = float(tt.item())
        dist_blocks.append(1e2 * dtb)
    timing = None
    if dist_blocks:
        q = np.percentile(np.asarray(dist_blocks), [10, 50, 90])
        timing = {"blocks": len(dist_blocks), "steps_per_block": 10, "ms_per_step_p10": round(float(q[0]), 4),
                  "ms_per_step_median": round(float(q[1]), 4), "ms_per_step_p90": round(float(q[2]), 4),
                  "note": "untimed blocks after the K timed steps; `value` and `ms_per_step` are the mean over exactly K steps"}
    # N > 1: what the exchange costs on the critical path -- the same ranks' step time with the collective skipped (state["solo"],
    # every rank) against the step time with it, ten steps each, max over the ranks
    comm = None
    if dist is not None:
        def _block(n=10):
            barrier()
            tb_ = time.perf_counter()
            for _ in range(n):
                step()
            barrier()
            tt_ = torch.tensor([time.perf_counter() - tb_], device=dev, dtype=torch.float64)
            dist.all_reduce(tt_, op=dist.ReduceOp.MAX)
            return 1e3 * float(tt_.item()) / n
        phase("exposed-communication blocks")
        with_ms = _block()
        state["solo"] = True
        solo_ms = _block()
        state["solo"] = False
        try:
            rccl = ".".join(str(v) for v in torch.cuda.nccl.version()) if dist.get_backend() and "nccl" in str(dist.get_backend()) else None
        except Exception:  # noqa: BLE001
            rccl = None
        comm = {"backend": str(dist.get_backend()), "rccl_version": rccl, "nranks": dist.get_world_size(),
                "devices": comm_ranks, "distinct_devices": len({(d or {}).get("uuid") or (d or {}).get("pci") or (d or {}).get("device_index") for d in comm_ranks}),
                "message_bytes": int(P * C * 4) if not fwd_only else 0, "exchange": "rs-ag" if args.rs_ag else "allreduce",
                "ms_per_step_with_exchange": round(with_ms, 4), "ms_per_step_without": round(solo_ms, 4),
                "exposed_ms": round(with_ms - solo_ms, 4),
                "note": "exposed_ms = step time - the same ranks' step time with the collective skipped (ten steps each, max over ranks); "
                        "the exchange overlaps the next view's geometry stages (features-ready event)"}
    if rank == 0:
        print(f"[bench] {len(block_ms)} settling blocks of {nb} steps, ms/step: first {block_ms[:8]} min {min(block_ms)} "
              f"last {block_ms[-5:]}; timed region {ms_per_step:.3f} ms/step", file=sys.stderr)
    if host_trace is not None and rank == 0:
        f = sorted(1e3 * (b - a) for a, b, c in host_trace)
        g = sorted(1e3 * (c - b) for a, b, c in host_trace)
        print(f"[bench] host ms per call over {len(f)} steps: forward median {f[len(f) // 2]:.3f} max {f[-1]:.1f}; "
              f"backward median {g[len(g) // 2]:.3f} max {g[-1]:.1f}", file=sys.stderr)
        t00 = host_trace[0][0]
        for i, (a, b, c) in enumerate(host_trace):
            if c - a > 0.01:
                gap = 1e3 * (a - host_trace[i - 1][2]) if i else 0.0
                print(f"[bench]   step {i} at {a - t00:.3f} s: forward call {1e3 * (b - a):.1f} ms, backward call {1e3 * (c - b):.1f} ms, "
                      f"gap before {gap:.1f} ms", file=sys.stderr)
    if os.environ.get("MI_BENCH_STEP_TRACE"):
        # diagnosis aid, after the timed region: wall time of single synchronised steps
        per = []
        for _ in range(20):
            barrier()
            ta = time.perf_counter()
            step()
            barrier()
            per.append(round(1e3 * (time.perf_counter() - ta), 3))
        if rank == 0:
            print(f"[bench] single synchronised steps, ms: {per}", file=sys.stderr)

    # ---- counters + live per-stage HIP-event timing (separate, un-timed steps) -------------------
    roofline = None
    stages_ms = {}
    counters = {}
    # From here to the --dump-grads step only rank 0 works (per-stage timing, counters, parity): its steps must not enter collectives
    # the other ranks never join -- they wait for rank 0 in the dump step's all-reduce / in destroy_process_group.  (Until round 4 these
    # steps did start the gradient all-reduce: harmless on the one-rank group of --dist-single, a deadlock with a real second rank.)
    state["solo"] = True
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
        stages_ms = {k: v / nprof for k, v in acc.items() if not (fwd_only and k in ("blend_bwd", "geom_bwd"))}
        # counters: one extra (un-timed) native forward whose opaque buffers we can inspect
        e = torch.empty(0)
        # E and L are counters of the REFERENCE algorithm (positions in its full tile lists): full-list mode for this call
        with torch.no_grad(), R.forward_flags(full_lists=True):
            res = R.rasterize_gaussians_native(
                C, fwd_only, settings.bg, means3D, e if fwd_only else feats, opac, mask if fwd_only else None, scales, rots,
                1.0, e, settings.viewmatrix, settings.projmatrix, cam.tanfovx, cam.tanfovy, H, W,
                shs if fwd_only else e, 3 if fwd_only else 0, settings.campos, False, False)
        num_rendered, imgbuf = res[0], res[-1]
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
        # (quadrant, record) rows the backward walks = what its gradient atomics are counted in (lean lists, the product's)
        rows = None
        if not fwd_only:
            with torch.no_grad():
                res2 = R.rasterize_gaussians_native(C, False, settings.bg, means3D, feats, opac, None, scales, rots, 1.0, e, settings.viewmatrix,
                                                    settings.projmatrix, cam.tanfovx, cam.tanfovy, H, W, e, 0, settings.campos, False, False)
            bbuf, ibuf = res2[-2], res2[-1]
            ntl = tiles_x * tiles_y
            rg = ibuf[ioff["ranges"]:ioff["ranges"] + 8 * ntl].view(torch.int32).reshape(ntl, 2).long()
            lens = rg[:, 1] - rg[:, 0]
            n_list = int(lens.sum().item())
            if n_list > 0:
                bl = bbuf[:4 * n_list].view(torch.int32).long() & 0xFFFFFFFF
                tile_of = torch.repeat_interleave(torch.arange(ntl, device=dev), lens)
                pos = torch.arange(n_list, device=dev) - rg[tile_of, 0]
                nc2 = ibuf[ioff["n_contrib"]:ioff["n_contrib"] + 4 * W * H].view(torch.int32).reshape(H, W)
                pad2 = torch.zeros((tiles_y * 16, tiles_x * 16), dtype=torch.int32, device=dev)
                pad2[:H, :W] = nc2
                lq = pad2.reshape(tiles_y, 2, 8, tiles_x, 2, 8).amax(dim=(2, 5)).permute(0, 2, 1, 3).reshape(ntl, 4).long()   # [tile, qy*2+qx]
                rows = 0
                for q in range(4):
                    rows += int(((((bl >> 28) >> q) & 1).bool() & (pos < lq[tile_of, q])).sum().item())
        ab = algorithmic_bytes(counters, C, forward_only=fwd_only, sh_coeffs=16 if fwd_only else 0, extra=2 if fwd_only else 0)
        dom = max((k for k in stages_ms), key=lambda k: stages_ms[k])
        achieved = ab[dom] / (stages_ms[dom] * 1e-3) / 1e9 if stages_ms[dom] > 0 else 0.0
        # HBM bytes / pipe-busy fractions the PMC passes saw (tools/collect_profiles.sh -> profiles/traffic_<cfg>.json, alu_<cfg>.json).
        # They are measurements of ONE build: each file carries the stamp of the library it was taken with
        # (mi_rast_version(): a hash of the sources, headers and flags), and a file whose stamp is not the loaded library's is
        # ignored -- the line then says null, never a number from another build.
        lib_version = _lib.load().mi_rast_version().decode()
        traffic_all, alu, pmc_note = {}, None, None

        def stamped(name):
            try:
                d = json.load(open(os.path.join(ROOT, "profiles", f"{name}_{args.config}.json")))
            except Exception:  # noqa: BLE001
                return None, "no PMC summary for this configuration"
            if d.get("_stamp") != lib_version:
                return None, f"PMC summary is of another build ({d.get('_stamp')}): ignored"
            return d, d.get("_source")
        traffic_all, pmc_note = stamped("traffic")
        traffic_all = traffic_all or {}
        alu_all, _ = stamped("alu")
        alu = (alu_all or {}).get(dom)
        tr = lambda k: (traffic_all.get(k) or {}).get("bytes_per_launch")
        stage_rows = {}
        for k, ms_k in stages_ms.items():
            a = ab.get(k, 0)
            f = a / (ms_k * 1e-3) / 1e9 / HBM_PEAK_GBPS if ms_k > 0 else 0.0
            row = {"ms": round(ms_k, 4), "algorithmic_bytes": a, "frac": round(f, 4), "traffic": tr(k),
                   "frac_traffic": round(tr(k) / (ms_k * 1e-3) / 1e9 / HBM_PEAK_GBPS, 4) if tr(k) and ms_k > 0 else None}
            if f > 1.0:
                row["note"] = "algorithm replaced: the byte count is the reference's pass, which this pipeline does not perform"
            if a == 0:
                row["note"] = "no counterpart in the reference's byte table (its 64-bit sort covers the depth order)"
            stage_rows[k] = row
        # every kernel of the PMC table that belongs to a stage (tools/summarize_profiles.py fails on a kernel it cannot place)
        total_traffic = (traffic_all.get("_sum_of_stages") or sum(tr(k) for k in stages_ms if tr(k))) if traffic_all else None
        wv_ach = ab["total"] / (ms_per_step * 1e-3) / 1e9
        # the same without the stages whose byte count is the reference's pass that this pipeline replaces (frac > 1 above)
        kept = sum(a for k, a in ab.items() if k != "total" and not (k in stage_rows and stage_rows[k]["frac"] > 1.0))
        wv_kept = kept / (ms_per_step * 1e-3) / 1e9
        # What the two blend kernels are actually bound by, side by side with the contract's byte fraction: HBM bytes the counters
        # saw, busy fractions of the vector / matrix pipes (stamped PMC files), and -- backward -- the L2's atomic units: 64-byte
        # segment requests per launch (per (quadrant, record) row: one per 64 bytes of the feature-gradient row + one for the
        # geometry record, per channel block) against the ~20 G segment-ops/s tools/atomic_bench.hip measures on this part.
        ATOMIC_SEGMENT_RATE = 20.0e9

        def limiter_of(stage):
            if stage not in stages_ms or stages_ms[stage] <= 0:
                return None
            sec = stages_ms[stage] * 1e-3
            al = (alu_all or {}).get(stage) or {}
            d = {"kernel_ms": round(stages_ms[stage], 4), "hbm_frac_algorithmic": round(ab.get(stage, 0) / sec / 1e9 / HBM_PEAK_GBPS, 4),
                 "hbm_frac_traffic": round(tr(stage) / sec / 1e9 / HBM_PEAK_GBPS, 4) if tr(stage) else None,
                 "valu_busy": al.get("valu_busy_frac"), "mfma_busy": al.get("mfma_busy_frac"), "lds_bank_conflict": al.get("lds_bank_conflict_frac")}
            if stage == "blend_bwd" and rows is not None:
                segs, rem = 0, C
                while rem > 0:
                    cb = 64 if rem >= 64 else (32 if rem >= 32 else 16)
                    segs += (min(cb, rem) * 4 + 63) // 64 + 1
                    rem -= cb
                d.update({"rows": rows, "atomic_segment_requests": segs * rows,
                          "atomic_unit_frac": round(segs * rows / sec / ATOMIC_SEGMENT_RATE, 4),
                          "atomic_segment_rate_peak": ATOMIC_SEGMENT_RATE,
                          # the rate this launch ran at, beside what a form of this kernel with nothing but its atomics left reaches
                          # with conflict-free random rows (profiles/r06_bwd_atomics.md: 15.5 G/s; real rows 13.4 - 14.5) -- the
                          # 20 G/s above is that of waves that issue atomics and do nothing else (tools/atomic_bench.hip)
                          "atomic_segment_rate": round(segs * rows / sec, 0), "atomic_segment_rate_in_kernel_ceiling": 15.5e9,
                          "atomic_unit_frac_of_in_kernel_ceiling": round(segs * rows / sec / 15.5e9, 4)})
            return d
        roofline = {"bound": "hbm", "kernel": dom, "achieved": round(achieved, 1), "peak": HBM_PEAK_GBPS,
                    "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBPS, 4), "traffic": tr(dom),
                    "algorithmic_bytes": ab[dom], "kernel_ms": round(stages_ms[dom], 4),
                    "alu": alu, "pmc": pmc_note, "library": lib_version,
                    "limiter": {"note": "`bound`/`frac` are the contract's HBM byte fraction; what limits the kernels in practice: the atomic "
                                        "units (backward: atomic_unit_frac) and VALU issue (forward: valu_busy) -- neither kernel is near HBM peak by traffic",
                                "blend_bwd": limiter_of("blend_bwd"), "blend_fwd": limiter_of("blend_fwd")},
                    "stages": stage_rows,
                    # whole_view.frac: the bytes of the stages this pipeline really performs (the reference's 45-bit global sort,
                    # which it replaces, left out) / the step time -- the number to read.  frac_survey_bytes: SURVEY.md 8(d)'s
                    # literal byte table, replaced sort included: a byte count, not bandwidth.
                    "whole_view": {"algorithmic_bytes": kept, "achieved": round(wv_kept, 1),
                                   "frac": round(wv_kept / HBM_PEAK_GBPS, 4),
                                   "frac_survey_bytes": round(wv_ach / HBM_PEAK_GBPS, 4),
                                   "survey_bytes": ab["total"],
                                   # also against what a float4 copy reaches on this part (6.3 TB/s)
                                   "frac_of_copy_rate": round(wv_kept / 6300.0, 4),
                                   "traffic": total_traffic,
                                   "frac_traffic": round(total_traffic / (ms_per_step * 1e-3) / 1e9 / HBM_PEAK_GBPS, 4)
                                   if total_traffic else None,
                                   "note": "frac leaves out the bytes of the stages whose algorithm this pipeline replaces (the "
                                           "reference's 45-bit global sort: rows flagged 'algorithm replaced' in stages); "
                                           "frac_survey_bytes counts SURVEY.md 8(d)'s whole table (same time): a byte count, not "
                                           "bandwidth; frac_traffic counts the HBM bytes the PMC counters saw"}}

    # ---- CPU baseline: the oracle on this box's host cores (rank 0, N == 1 only) + parity of the benchmarked run ------
    cpu_baseline = None
    parity = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline and not args.dist_single:
        from oracle import saga_oracle as so
        native = so.use_native_build()   # -O3 -march=native, compiled on THIS machine; same results
        if args.cpu_threads > 0:
            so.set_num_threads(args.cpu_threads)
        inp = so.Inputs(means3D=scene.means3D, opacities=scene.opacities, viewmatrix=cam.viewmatrix,
                        projmatrix=cam.projmatrix, campos=cam.campos, bg=np.zeros(C, np.float32), image_width=W,
                        image_height=H, tanfovx=cam.tanfovx, tanfovy=cam.tanfovy, channels=C,
                        colors_precomp=None if fwd_only else scene.features, shs=scene.shs if fwd_only else None,
                        sh_degree=3 if fwd_only else 0, mask=np.ones(P, np.float32) if fwd_only else None,
                        scales=scene.scales, rotations=scene.rotations)
        dLn = dL.cpu().numpy()
        nviews = max(1, args.cpu_views)
        view_s = []
        for _ in range(nviews):
            c0 = time.perf_counter()
            fo = so.forward(inp)
            bo = None if fwd_only else so.backward(inp, fo, dLn)
            view_s.append(time.perf_counter() - c0)
        cpu_s = sum(view_s) / nviews
        # cores = the OpenMP threads that ran (omp_get_max_threads() of the oracle library); the affinity mask and the host's CPU
        # count are reported beside it, not mixed into it
        cpu_baseline = {"value": round(1.0 / cpu_s, 4), "unit": "views/s", "cores": so.num_threads(), "kind": "port",
                        "affinity_cpus": len(os.sched_getaffinity(0)), "host_cpus": os.cpu_count(),
                        "sample": f"{nviews} full views {'fwd' if fwd_only else 'fwd+bwd'} of {args.config} (P={P}, {W}x{H}, C={C}), "
                                  f"{', '.join(f'{v:.2f}' for v in view_s)} s each, OpenMP over Gaussians/tiles with "
                                  f"{so.num_threads()} threads (omp_get_max_threads), gcc -O3 {'-march=native' if native else '(portable build)'}"}
        # parity of what was just benchmarked (product default lists) against that oracle run
        step()
        torch.cuda.synchronize(dev)
        pr = {"radii_equal": bool(np.array_equal(state["radii"].cpu().numpy(), fo.radii)), "rtol": 1e-4,
              "criterion": "|got-want| <= rtol*|want| + rtol*max|want| (tests/helpers.py); norm = ||got-want||2/||want||2"}
        pairs = [("image", state["color"].cpu().numpy(), fo.color)]
        if fwd_only:
            pairs += [("mask", state["mask"].cpu().numpy(), fo.mask), ("depth", state["depth"].cpu().numpy(), fo.depth)]
        else:
            pairs += [("dL_dfeatures", feats.grad.cpu().numpy(), bo.dL_dcolors),
                      ("dL_dmeans3D", means3D.grad.cpu().numpy(), bo.dL_dmeans3D),
                      ("dL_dmeans2D", state["means2D"].grad.cpu().numpy(), bo.dL_dmeans2D),
                      ("dL_dopacity", opac.grad.cpu().numpy(), bo.dL_dopacity),
                      ("dL_dscales", scales.grad.cpu().numpy(), bo.dL_dscales),
                      ("dL_drotations", rots.grad.cpu().numpy(), bo.dL_drotations)]
        pr["rows"] = ("rows_outside = share of non-zero rows (one Gaussian's gradient / one pixel's channels) with max|got-want| > "
                      "rtol*(max|want_row| + MEDIAN row magnitude): a relative criterion whose floor small rows cannot hide under")
        worst_frac = worst_rows = 0.0
        for name, got, want in pairs:
            want = np.asarray(want).reshape(got.shape)
            frac, nrm = _close(got, want)
            # rows: Gaussians for the gradients; pixels (all channels of one pixel) for the images
            g2, w2 = (got, want) if not name in ("image", "mask", "depth") else (got.reshape(got.shape[0], -1).T, want.reshape(want.shape[0], -1).T)
            rows = _rows_outside(g2, w2)
            pr[name] = {"frac_outside": float(f"{frac:.3g}"), "norm": float(f"{nrm:.3g}"), "rows_outside": float(f"{rows:.3g}")}
            worst_frac = max(worst_frac, frac)
            worst_rows = max(worst_rows, rows)
        pr["frac_outside_max"] = float(f"{worst_frac:.3g}")
        pr["rows_outside_max"] = float(f"{worst_rows:.3g}")
        pr["ok"] = bool(pr["radii_equal"] and worst_frac <= 2e-4)
        parity = pr

    ref_on_gpu = None
    if rank == 0 and world == 1 and args.ref_on_gpu and not fwd_only:
        from oracle import saga_oracle as so
        from oracle import saga_ref as sr
        inp = so.Inputs(means3D=scene.means3D, opacities=scene.opacities, viewmatrix=cam.viewmatrix,
                        projmatrix=cam.projmatrix, campos=cam.campos, bg=np.zeros(C, np.float32), image_width=W,
                        image_height=H, tanfovx=cam.tanfovx, tanfovy=cam.tanfovy, channels=C,
                        colors_precomp=scene.features, scales=scene.scales, rotations=scene.rotations)
        ms_ref = sr.RefRun(inp).time_fwd_bwd(dL.cpu().numpy(), steps=5, warmup=2)
        ref_on_gpu = {"ms_per_view": round(ms_ref, 3), "views_per_s": round(1e3 / ms_ref, 2),
                      "speedup_of_this_repo": round(ms_ref / ms_per_step, 2),
                      "what": "oracle/_ref: the reference's own CUDA kernels translated test-only with hipify-perl "
                              "(-ffp-contract=off), same workload, same GPU, incl. its allocations and zero fills"}

    # ---- frozen-geometry reuse: a SECOND, separately labelled figure (never `value`) ---------------------------------------------
    frozen = None
    if args.frozen_geometry and rank == 0 and world == 1 and not fwd_only:
        # SAGA's feature training (train_contrastive_feature.py:231) renders one of ~200 cameras per iteration, ~50 visits each, with
        # the geometry frozen: --views cameras of an orbit, --frozen-passes visits each, in cyclic order; the first pass over the
        # cameras fills the cache INSIDE the timed region.  Same drop-in calls as the headline, with the opt-in switched on.
        nv, npass = max(1, args.views), max(2, args.frozen_passes)
        cams_f = [scenes.orbit_camera(W, H, cfg["focal"], 0.02 * k, 0.01 * k) for k in range(nv)]
        rasts = [GaussianRasterizer(settings._replace(viewmatrix=t(c_.viewmatrix), projmatrix=t(c_.projmatrix), campos=t(c_.campos),
                                                      tanfovx=c_.tanfovx, tanfovy=c_.tanfovy)) for c_ in cams_f]

        def fstep(rz):
            for l in leaves:
                l.grad = None
            m2 = torch.zeros_like(means3D, requires_grad=True)
            col_, _ = rz(means3D=means3D, means2D=m2, shs=None, colors_precomp=feats, opacities=opac, scales=scales, rotations=rots,
                         cov3D_precomp=None)
            torch.autograd.backward(col_, grad_tensors=dL)

        def run(passes):
            torch.cuda.synchronize(dev)
            t0_ = time.perf_counter()
            for _ in range(passes):
                for rz in rasts:
                    fstep(rz)
            torch.cuda.synchronize(dev)
            return time.perf_counter() - t0_
        run(1)                                     # uncached warm-up of the cameras' code paths
        t_plain = run(npass)                       # the same sequence without the cache
        gc_ = R.enable_geometry_cache()
        gc_.clear()
        t_cached = run(npass)                      # first visits (misses that fill the cache) + revisits, all timed
        st = gc_.stats()
        t_steady = run(npass)                      # every visit a hit
        R.disable_geometry_cache(drop=True)
        nsteps = nv * npass
        frozen = {"views_per_s": round(nsteps / t_cached, 3), "ms_per_step": round(1e3 * t_cached / nsteps, 4),
                  "views_per_s_all_hits": round(nsteps / t_steady, 3), "views_per_s_without_cache_same_sequence": round(nsteps / t_plain, 3),
                  "hit_rate": round(st["hit_rate"], 4), "hits": st["hits"], "misses": st["misses"], "views": nv, "visits_per_view": npass,
                  "bytes_cached": st["bytes_cached"], "bytes_cached_per_view": st["bytes_cached"] // max(1, st["views"]),
                  "what": "opt-in per-camera reuse of the geometry-only stages (preprocess, binning, per-tile sort) while the geometry "
                          "tensors and the camera are unchanged (content fingerprints; seganygaussians_amd/rasterizer.py: GeometryCache, "
                          "mi_rast_forward_reuse); NOT the headline: `value` recomputes everything every step as the reference does"}

    # ---- features-only backward: a separately labelled figure (never `value`) -----------------------------------------------------
    feat_only = None
    if args.features_only_grad and rank == 0 and world == 1 and not fwd_only:
        def fo_step(rz=rasterizer):
            for l in leaves:
                l.grad = None
            m2 = torch.zeros_like(means3D, requires_grad=True)
            col_, _ = rz(means3D=means3D, means2D=m2, shs=None, colors_precomp=feats, opacities=opac, scales=scales, rotations=rots,
                         cov3D_precomp=None)
            torch.autograd.backward(col_, grad_tensors=dL)

        def fo_run(n, rzs=(rasterizer,)):
            torch.cuda.synchronize(dev)
            t0_ = time.perf_counter()
            for k in range(n):
                fo_step(rzs[k % len(rzs)])
            torch.cuda.synchronize(dev)
            return time.perf_counter() - t0_
        nfo = max(args.steps, 50)
        fo_run(5)
        t_full = fo_run(nfo)                         # the headline's step again, through the same loop
        g_full = feats.grad.detach().clone()
        prev_fo = R.enable_features_only_backward(True)
        try:
            fo_run(5)
            t_fo = fo_run(nfo)
            g_fo = feats.grad.detach().clone()
            others_unset = all(l.grad is None for l in leaves if l is not feats)
            _lib.profile_enable(True)   # per-stage HIP events of three un-timed steps (the geometry backward does not run: no stage)
            st_fo = {k: 0.0 for k in _lib.MI_STAGES if k != "geom_bwd"}
            for _ in range(3):
                fo_step()
                torch.cuda.synchronize(dev)
                ms_ = _lib.profile_read()
                for k in st_fo:
                    st_fo[k] += ms_[k] / 3
            _lib.profile_enable(False)
            both = None
            if args.frozen_geometry:
                nv = max(1, args.views)
                cams_f = [scenes.orbit_camera(W, H, cfg["focal"], 0.02 * k, 0.01 * k) for k in range(nv)]
                rasts = [GaussianRasterizer(settings._replace(viewmatrix=t(c_.viewmatrix), projmatrix=t(c_.projmatrix), campos=t(c_.campos),
                                                              tanfovx=c_.tanfovx, tanfovy=c_.tanfovy)) for c_ in cams_f]
                gc_ = R.enable_geometry_cache()
                gc_.clear()
                fo_run(nv, rasts)                    # first visits: fill the cache
                t_both = fo_run(nv * max(2, args.frozen_passes), rasts)
                R.disable_geometry_cache(drop=True)
                both = round(nv * max(2, args.frozen_passes) / t_both, 3)
        finally:
            R.enable_features_only_backward(prev_fo)
        scale_ = float(g_full.abs().max())
        feat_only = {"views_per_s": round(nfo / t_fo, 3), "ms_per_step": round(1e3 * t_fo / nfo, 4),
                     "views_per_s_default_backward_same_loop": round(nfo / t_full, 3), "steps": nfo,
                     "dL_dfeatures_max_abs_diff_vs_default": float(f"{float((g_fo - g_full).abs().max()):.3g}"),
                     "dL_dfeatures_max_abs": float(f"{scale_:.3g}"), "other_gradients_unset": bool(others_unset),
                     "stages_ms": ({k: round(v, 4) for k, v in st_fo.items()} if isinstance(st_fo, dict) else None),
                     "views_per_s_with_frozen_geometry_all_hits": both,
                     "what": "opt-in (rasterizer.enable_features_only_backward / MI_RAST_FEATURES_ONLY_BACKWARD=1; automatic when autograd asks "
                             "for the colour gradient alone): the backward blend computes dL_dcolors_precomp only -- no feature rows read, no "
                             "dL/dalpha, no moments, no packed-field atomics -- and the geometry backward does not run (include/mi_rast.h: "
                             "MI_RAST_BWD_FEATURES_ONLY); NOT the headline: `value` computes all eight gradients every step as the reference does"}

    if rank == 0:
        names = {"cfg3": "train views/sec (fwd+bwd), 1080p, 1M Gaussians, 32-D features",
                 "cfg3s": "train views/sec (fwd+bwd), 1080p, 1M Gaussians, 32-D features",
                 "cfg2": "forward views/sec, 1080p, 1M Gaussians, SH-3 RGB + mask + depth"}
        law = "" if args.ply else (", second synthetic law (surfaces of flat opaque disks + faint floaters, scenes.make_surface_scene)"
                                   if cfg.get("law") == "surface" else "")
        what = (f"BASELINE {args.config.rstrip('s')}{' on ' + data if args.ply else ''}: {P} Gaussians, {W}x{H}, " +
                ("SH degree 3 RGB + mask + depth, forward only (diff_gaussian_rasterization_depth)" if fwd_only
                 else f"{C}-D features, fwd+bwd") + law + ", 1 view/GPU/step" +
                (", RCCL all-reduce of (P,C) feature grads" if world > 1 else ""))
        out = {
            "metric": names.get(args.config, f"train views/sec (fwd+bwd), {args.config}"),
            "value": round(value, 3), "unit": "views/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 4), "sustained": sustained, "timing": timing, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": data,
            "config": {"workload": what, "parallelism": f"view-sharded x{world}" + (" rs-ag" if args.rs_ag and world > 1 else ""), "counters": counters,
                       "lists": "lean (product default: only overlaps that pass the exact-conservative cull are listed; "
                                "counters E/L from one full-list call)",
                       "arithmetic": "f32 throughout; C=32/64 forward accumulation = exact 3-way bf16 split of f32 operands, "
                                     "six partial products on the bf16 matrix pipe, f32 accumulate (f32 rounding level)"
                                     + ("; exp() = v_exp_f32(x*log2e) (MI_RAST_FAST_EXP)" if args.fast_exp else
                                        "; exp() = expf, as the reference" if args.exact_exp else
                                        "; exp(): every alpha >= 1/255 decision by expf as in the reference, values by v_exp_f32 away from the cut (<= 1e-6 rel; csrc/common.h hybrid form), backward expf"),
                       "stages_ms": {k: round(v, 4) for k, v in stages_ms.items()}},
            "roofline": roofline, "cpu_baseline": cpu_baseline, "parity": parity,
        }
        if comm is not None:
            out["config"]["comm"] = comm
        if ref_on_gpu:
            out["reference_on_gpu"] = ref_on_gpu
        if frozen is not None:
            out["frozen_geometry"] = frozen
        if feat_only is not None:
            out["features_only_backward"] = feat_only
    phase("reporting")
    state["solo"] = False
    if args.dump_grads and not fwd_only:
        os.makedirs(args.dump_grads, exist_ok=True)
        barrier()
        if args.rs_ag:   # the features this step starts from (updated by every step so far: the same on every rank)
            np.save(os.path.join(args.dump_grads, f"features_before_rank{rank}.npy"), feats.detach().cpu().numpy())
        step()
        barrier()   # waits for this step's exchange: feats.grad holds the sum over the ranks' views (all-reduce) / feats the update (--rs-ag)
        np.save(os.path.join(args.dump_grads, f"camera_{rank}.npy"), np.concatenate([cam.viewmatrix.ravel(), cam.projmatrix.ravel()]))
        if args.rs_ag:   # every rank: its LOCAL gradient (the sum only ever exists in shards) and the features after the sharded update
            np.save(os.path.join(args.dump_grads, f"feature_grad_local_rank{rank}.npy"), feats.grad.detach().cpu().numpy())
            np.save(os.path.join(args.dump_grads, f"features_rank{rank}.npy"), feats.detach().cpu().numpy())
        elif rank == 0:
            np.save(os.path.join(args.dump_grads, "feature_grad_rank0.npy"), feats.grad.detach().cpu().numpy())
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
