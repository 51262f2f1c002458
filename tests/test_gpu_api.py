"""GPU tests through the drop-in Python packages (the reference's extension API): autograd plumbing, gradient
tuple order (CF/diff_gaussian_rasterization_contrastive_f/__init__.py:142-152), the depth package's extra
mask argument / forward_mask, markVisible, and the debug snapshot path."""
import os

import numpy as np
import pytest
import torch

import seganygaussians_amd
from oracle import saga_oracle as so
from seganygaussians_amd import scenes
from tests import helpers as hp

seganygaussians_amd.install_dropin()
pytestmark = pytest.mark.gpu


def _settings(mod, inp, dev, debug=False):
    t = lambda a: torch.as_tensor(np.ascontiguousarray(a, np.float32)).to(dev)
    return mod.GaussianRasterizationSettings(
        image_height=inp.image_height, image_width=inp.image_width, tanfovx=inp.tanfovx, tanfovy=inp.tanfovy,
        bg=t(inp.bg), scale_modifier=inp.scale_modifier, viewmatrix=t(inp.viewmatrix), projmatrix=t(inp.projmatrix),
        sh_degree=inp.sh_degree, campos=t(inp.campos), prefiltered=False, debug=debug)


def _leaf(a, dev):
    return torch.as_tensor(np.ascontiguousarray(a, np.float32)).to(dev).requires_grad_(True)


@pytest.mark.parametrize("pkg,C", [("diff_gaussian_rasterization", 3), ("diff_gaussian_rasterization_contrastive_f", 32)])
def test_autograd_through_dropin(pkg, C):
    mod = __import__(pkg)
    dev = torch.device("cuda:0")
    inp = hp.make_inputs(6000, 208, 144, C, seed=31, bg="random", camera="orbit")
    means3D, feats = _leaf(inp.means3D, dev), _leaf(inp.colors_precomp, dev)
    opac, scales, rots = _leaf(inp.opacities, dev), _leaf(inp.scales, dev), _leaf(inp.rotations, dev)
    means2D = torch.zeros_like(means3D, requires_grad=True)
    rast = mod.GaussianRasterizer(raster_settings=_settings(mod, inp, dev))
    color, radii = rast(means3D=means3D, means2D=means2D, shs=None, colors_precomp=feats, opacities=opac,
                        scales=scales, rotations=rots, cov3D_precomp=None)
    assert color.shape == (C, 144, 208) and radii.dtype == torch.int32 and not radii.requires_grad
    dL = scenes.make_grad_image(C, 144, 208, seed=2)
    (color * torch.as_tensor(dL).to(dev)).sum().backward()
    fwd = so.forward(inp)
    bwd = so.backward(inp, fwd, dL)
    np.testing.assert_array_equal(radii.cpu().numpy(), fwd.radii)
    hp.assert_close("color", color.detach().cpu().numpy(), fwd.color, flip_frac=hp.FLIP_FRAC)
    for name, leaf, want in [("means3D", means3D, bwd.dL_dmeans3D), ("means2D", means2D, bwd.dL_dmeans2D),
                             ("colors_precomp", feats, bwd.dL_dcolors), ("opacities", opac, bwd.dL_dopacity),
                             ("scales", scales, bwd.dL_dscales), ("rotations", rots, bwd.dL_drotations)]:
        hp.assert_close(name, leaf.grad.cpu().numpy(), np.asarray(want).reshape(leaf.shape), flip_frac=hp.GRAD_FLIP_FRAC)
    vis = rast.markVisible(means3D.detach())
    np.testing.assert_array_equal(vis.cpu().numpy(), so.mark_visible(inp.means3D, inp.viewmatrix, inp.projmatrix))
    # no_grad forward (render.py path)
    with torch.no_grad():
        c2, _ = rast(means3D=means3D, means2D=means2D, shs=None, colors_precomp=feats, opacities=opac, scales=scales,
                     rotations=rots, cov3D_precomp=None)
    assert torch.equal(c2, color.detach())


@pytest.mark.parametrize("W,H", [(160, 112), (48, 32)])
def test_second_backward_through_a_retained_graph(W, H):
    """The forward hands its pre-zeroed accumulators (rasterizer.py: prezero) to ONE backward; a second backward through a
    retained graph must fill its own and give the same gradients.  48 x 32: more Gaussians than pixels -- only the packed
    field gradients are pre-zeroed, dL_dcolors is the backward's own torch.zeros."""
    import diff_gaussian_rasterization_contrastive_f as mod
    dev = torch.device("cuda:0")
    inp = hp.make_inputs(5000, W, H, 32, seed=33, camera="orbit")
    means3D, feats = _leaf(inp.means3D, dev), _leaf(inp.colors_precomp, dev)
    opac, scales, rots = _leaf(inp.opacities, dev), _leaf(inp.scales, dev), _leaf(inp.rotations, dev)
    means2D = torch.zeros_like(means3D, requires_grad=True)
    rast = mod.GaussianRasterizer(raster_settings=_settings(mod, inp, dev))
    color, _ = rast(means3D=means3D, means2D=means2D, shs=None, colors_precomp=feats, opacities=opac, scales=scales,
                    rotations=rots, cov3D_precomp=None)
    dL = torch.as_tensor(scenes.make_grad_image(32, H, W, seed=6)).to(dev)
    color.backward(dL, retain_graph=True)
    first = {k: v.grad.clone() for k, v in (("feats", feats), ("means3D", means3D), ("opac", opac))}
    color.backward(dL)
    for k, v in (("feats", feats), ("means3D", means3D), ("opac", opac)):
        hp.assert_close(k + " (second backward)", (v.grad - first[k]).cpu().numpy(), first[k].cpu().numpy(), rtol=2e-4,
                        flip_frac=hp.GRAD_FLIP_FRAC)
    bwd = so.backward(inp, so.forward(inp), dL.cpu().numpy())
    hp.assert_close("colors_precomp", first["feats"].cpu().numpy(), np.asarray(bwd.dL_dcolors).reshape(first["feats"].shape),
                    flip_frac=hp.GRAD_FLIP_FRAC)


def test_depth_package_and_forward_mask():
    import diff_gaussian_rasterization_depth as mod
    dev = torch.device("cuda:0")
    inp = hp.make_inputs(5000, 176, 128, 3, seed=32, with_shs=True, sh_degree=2, use_mask=True, bg="random")
    means3D, shs = _leaf(inp.means3D, dev), _leaf(inp.shs, dev)
    opac, scales, rots, mask = _leaf(inp.opacities, dev), _leaf(inp.scales, dev), _leaf(inp.rotations, dev), _leaf(inp.mask, dev)
    means2D = torch.zeros_like(means3D, requires_grad=True)
    rast = mod.GaussianRasterizer(raster_settings=_settings(mod, inp, dev))
    color, omask, depth, radii = rast(means3D=means3D, means2D=means2D, shs=shs, colors_precomp=None, opacities=opac,
                                      mask=mask, scales=scales, rotations=rots, cov3D_precomp=None)
    assert omask.shape == (1, 128, 176) and depth.shape == (1, 128, 176)
    dL = scenes.make_grad_image(3, 128, 176, seed=3)
    dLm = (np.random.default_rng(4).normal(0, 1, (1, 128, 176)) / (128 * 176)).astype(np.float32)
    ((color * torch.as_tensor(dL).to(dev)).sum() + (omask * torch.as_tensor(dLm).to(dev)).sum()).backward()
    fwd = so.forward(inp)
    bwd = so.backward(inp, fwd, dL, dLm[0])
    hp.assert_close("color", color.detach().cpu().numpy(), fwd.color, flip_frac=hp.FLIP_FRAC)
    hp.assert_close("mask", omask.detach().cpu().numpy(), fwd.mask, flip_frac=hp.FLIP_FRAC)
    hp.assert_close("depth", depth.detach().cpu().numpy(), fwd.depth, flip_frac=hp.FLIP_FRAC)
    for name, leaf, want in [("means3D", means3D, bwd.dL_dmeans3D), ("shs", shs, bwd.dL_dsh), ("mask", mask, bwd.dL_dmask),
                             ("opacities", opac, bwd.dL_dopacity), ("scales", scales, bwd.dL_dscales),
                             ("rotations", rots, bwd.dL_drotations)]:
        hp.assert_close(name, leaf.grad.cpu().numpy(), np.asarray(want).reshape(leaf.shape), flip_frac=hp.GRAD_FLIP_FRAC)
    # mask-only pair: only the mask receives a gradient
    mask2 = _leaf(inp.mask, dev)
    m_only, radii2 = rast.forward_mask(means3D=means3D.detach(), means2D=means2D.detach(), opacities=opac.detach(),
                                       mask=mask2, scales=scales.detach(), rotations=rots.detach())
    (m_only * torch.as_tensor(dLm).to(dev)).sum().backward()
    fm = so.mask_forward(inp)
    hp.assert_close("mask_only", m_only.detach().cpu().numpy(), fm.mask, flip_frac=hp.FLIP_FRAC)
    hp.assert_close("mask_only grad", mask2.grad.cpu().numpy(), so.mask_backward(inp, fm, dLm[0]), flip_frac=hp.GRAD_FLIP_FRAC)


def test_debug_flag_and_error_snapshot(tmp_path, monkeypatch):
    import diff_gaussian_rasterization_contrastive_f as mod
    dev = torch.device("cuda:0")
    inp = hp.make_inputs(500, 64, 48, 32, seed=33)
    monkeypatch.chdir(tmp_path)
    rast = mod.GaussianRasterizer(raster_settings=_settings(mod, inp, dev, debug=True))
    t = lambda a: torch.as_tensor(np.ascontiguousarray(a, np.float32)).to(dev)
    color, _ = rast(means3D=t(inp.means3D), means2D=t(inp.means3D) * 0, shs=None, colors_precomp=t(inp.colors_precomp),
                    opacities=t(inp.opacities), scales=t(inp.scales), rotations=t(inp.rotations), cov3D_precomp=None)
    hp.assert_close("color(debug)", color.cpu().numpy(), so.forward(inp).color, flip_frac=hp.FLIP_FRAC)
    # SH input to the 32-channel package: the reference's runtime_error, plus the debug snapshot file
    with pytest.raises(RuntimeError, match="For non-RGB, provide precomputed Gaussian colors!"):
        rast(means3D=t(inp.means3D), means2D=t(inp.means3D) * 0, shs=torch.zeros(500, 1, 3, device=dev), colors_precomp=None,
             opacities=t(inp.opacities), scales=t(inp.scales), rotations=t(inp.rotations), cov3D_precomp=None)
    assert os.path.exists("snapshot_fw.dump")
    dump = torch.load("snapshot_fw.dump", weights_only=False)
    assert len(dump) == 19 and dump[1].shape == (500, 3)


def test_features_ready_event_orders_the_blend_stage():
    """set_features_ready_event: the next forward waits for the event right before its blend stage (and only that
    forward).  The features are written on a side stream AFTER the forward has been queued; the image must see them."""
    from seganygaussians_amd.rasterizer import set_features_ready_event
    mod = __import__("diff_gaussian_rasterization_contrastive_f")
    dev = torch.device("cuda:0")
    inp = hp.make_inputs(6000, 208, 144, 32, seed=33, camera="orbit")
    t = lambda a: torch.as_tensor(np.ascontiguousarray(a, np.float32)).to(dev)
    means3D, opac, scales, rots = t(inp.means3D), t(inp.opacities), t(inp.scales), t(inp.rotations)
    final = t(inp.colors_precomp)
    rast = mod.GaussianRasterizer(raster_settings=_settings(mod, inp, dev))
    kw = dict(means3D=means3D, means2D=torch.zeros_like(means3D), shs=None, opacities=opac, scales=scales,
              rotations=rots, cov3D_precomp=None)
    with torch.no_grad():
        want, _ = rast(colors_precomp=final, **kw)
        feats = torch.zeros_like(final)          # not ready yet
        side = torch.cuda.Stream(device=dev)
        side.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(side):
            big = torch.empty(64 << 20, device=dev)
            for _ in range(8):                    # keep the side stream busy for a while before the copy
                big.normal_()
            feats.copy_(final)
            ev = torch.cuda.Event()
            ev.record(side)
        set_features_ready_event(ev)
        got, _ = rast(colors_precomp=feats, **kw)
        again, _ = rast(colors_precomp=feats, **kw)   # the event was consumed: plain call
    torch.cuda.synchronize(dev)
    assert torch.equal(got, want) and torch.equal(again, want)


def test_reentrant_two_threads_two_streams():
    """The C-ABI keeps no state between calls: two host threads, each with its own stream, its own scene, its own list mode
    (one full, one lean with the f32 blend) and its own features-ready event, run forwards + backwards concurrently and get
    exactly what they get alone."""
    import threading
    from seganygaussians_amd import rasterizer as R
    mod = __import__("diff_gaussian_rasterization_contrastive_f")
    dev = torch.device("cuda:0")
    t = lambda a: torch.as_tensor(np.ascontiguousarray(a, np.float32)).to(dev)
    jobs = []
    for seed, (W, H), flags in ((41, (304, 208), dict(full_lists=True)), (42, (208, 160), dict(full_lists=False, exact_exp=True))):
        inp = hp.make_inputs(30_000, W, H, 32, seed=seed, camera="orbit")
        jobs.append(dict(inp=inp, flags=flags, dL=t(scenes.make_grad_image(32, H, W, seed=seed))))

    def run(job, stream, use_event, rounds, out):
        inp = job["inp"]
        rast = mod.GaussianRasterizer(raster_settings=_settings(mod, inp, dev))
        res = []
        with torch.cuda.stream(stream), R.forward_flags(**job["flags"]):
            for _ in range(rounds):
                feats = t(inp.colors_precomp).requires_grad_(True)
                m3 = t(inp.means3D).requires_grad_(True)
                if use_event:
                    ev = torch.cuda.Event()
                    ev.record(stream)
                    R.set_features_ready_event(ev)
                color, radii = rast(means3D=m3, means2D=torch.zeros_like(m3), shs=None, colors_precomp=feats,
                                    opacities=t(inp.opacities), scales=t(inp.scales), rotations=t(inp.rotations),
                                    cov3D_precomp=None)
                (color * job["dL"]).sum().backward()
                res.append((color.detach().clone(), radii.clone(), feats.grad.clone(), m3.grad.clone()))
            stream.synchronize()
        out.append(res)

    alone = []
    for job in jobs:
        run(job, torch.cuda.Stream(device=dev), False, 1, alone)
    outs = [[], []]
    threads = [threading.Thread(target=run, args=(job, torch.cuda.Stream(device=dev), True, 6, o)) for job, o in zip(jobs, outs)]
    for th in threads:
        th.start()
    for th in threads:
        th.join()
    for base, o in zip(alone, outs):
        assert len(o) == 1 and len(o[0]) == 6
        c0, r0, g0, m0 = base[0]
        for c, r, g, m in o[0]:
            assert torch.equal(c, c0) and torch.equal(r, r0)            # forward: bit-identical
            hp.assert_close("dL_dfeatures (concurrent)", g.cpu().numpy(), g0.cpu().numpy(), rtol=2e-4, flip_frac=hp.GRAD_FLIP_FRAC)
            hp.assert_close("dL_dmeans3D (concurrent)", m.cpu().numpy(), m0.cpu().numpy(), rtol=2e-4, flip_frac=hp.GRAD_FLIP_FRAC)


def test_depth_package_mask_contract():
    """DEPTH package: a missing mask raises instead of leaving out_mask / out_depth unwritten; the mask gradient comes back
    in the shape of the mask that went in -- (P,1) as the reference allocates it (DEPTH/rasterize_points.cu:167), and (P,)."""
    mod = __import__("diff_gaussian_rasterization_depth")
    dev = torch.device("cuda:0")
    inp = hp.make_inputs(3000, 96, 64, 3, seed=51, use_mask=True)
    t = lambda a: torch.as_tensor(np.ascontiguousarray(a, np.float32)).to(dev)
    rast = mod.GaussianRasterizer(raster_settings=_settings(mod, inp, dev))
    kw = dict(means3D=t(inp.means3D), means2D=t(inp.means3D) * 0, shs=None, colors_precomp=t(inp.colors_precomp),
              opacities=t(inp.opacities), scales=t(inp.scales), rotations=t(inp.rotations), cov3D_precomp=None)
    with pytest.raises(RuntimeError, match="mask must hold one float32 per Gaussian"):
        rast(mask=torch.empty(0, device=dev), **kw)
    grads = []
    for shape in ((3000, 1), (3000,)):
        mask = t(inp.mask).reshape(shape).requires_grad_(True)
        color, omask, depth, radii = rast(mask=mask, **kw)
        (omask.sum() * 0.5 + color.sum() * 0.0).backward()
        assert mask.grad.shape == mask.shape
        grads.append(mask.grad.reshape(-1).clone())
    hp.assert_close("dL_dmask (P,1) vs (P,)", grads[1].cpu().numpy(), grads[0].cpu().numpy(), rtol=2e-4, flip_frac=hp.GRAD_FLIP_FRAC)
    assert float(grads[0].abs().max()) > 0
    # outputs the loss does not use arrive in backward as None (the Functions switch autograd's zero materialisation off: it
    # would fill an int32 (P,) "gradient" for radii on every call): a loss on out_mask alone / on the image alone still works
    mask = t(inp.mask).reshape(3000, 1).requires_grad_(True)
    color, omask, depth, radii = rast(mask=mask, **kw)
    (omask.sum() * 0.5).backward()
    hp.assert_close("dL_dmask, image unused", mask.grad.reshape(-1).cpu().numpy(), grads[0].cpu().numpy(), rtol=2e-4, flip_frac=hp.GRAD_FLIP_FRAC)
    mask = t(inp.mask).reshape(3000, 1).requires_grad_(True)
    cols = t(inp.colors_precomp).requires_grad_(True)
    color, omask, depth, radii = rast(mask=mask, **dict(kw, colors_precomp=cols))
    color.sum().backward()
    assert float(cols.grad.abs().max()) > 0 and not bool(mask.grad.any())


def test_bench_single_rank_rccl_step():
    """The N > 1 step of bench.py -- asynchronous RCCL all-reduce of the feature gradients, the next view's geometry stages
    running ahead of it behind the features-ready event -- on a single-rank process group (the only multi-GPU path a 1-GPU
    box can execute); reduced Gaussian count, parity of the run checked by the JSON line's own fields."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29541", RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--dist-single", "--steps", "4", "--warmup", "2",
                          "--points", "200000", "--no-cpu-baseline"], capture_output=True, text=True, timeout=600, env=env, cwd=root)
    assert out.returncode == 0, out.stderr[-2000:]
    line = json.loads(out.stdout.strip().splitlines()[-1])
    assert line["n_gpus"] == 1 and line["value"] > 0 and line["config"]["counters"]["P"] == 200000
    assert set(line["roofline"]["stages"]) >= {"preprocess", "blend_fwd", "blend_bwd", "geom_bwd"}


@pytest.mark.parametrize("C", [64, 48])
def test_contrastive_dropin_takes_the_channel_count_from_the_call(C):
    """The reference fixes NUM_CHANNELS when its extension is compiled; the drop-in reads it off `colors_precomp`: 64-D (one pass)
    and 48-D (channel blocks 32 + 16) features through the unchanged import name, fwd + bwd against the oracle."""
    import torch
    import seganygaussians_amd
    seganygaussians_amd.install_dropin()
    from diff_gaussian_rasterization_contrastive_f import GaussianRasterizationSettings, GaussianRasterizer
    inp = hp.make_inputs(3000, 160, 112, C, seed=70 + C, camera="orbit")
    t = lambda a: torch.as_tensor(np.ascontiguousarray(a, np.float32)).cuda()
    st = GaussianRasterizationSettings(image_height=inp.image_height, image_width=inp.image_width, tanfovx=inp.tanfovx, tanfovy=inp.tanfovy,
                                       bg=t(inp.bg), scale_modifier=1.0, viewmatrix=t(inp.viewmatrix), projmatrix=t(inp.projmatrix),
                                       sh_degree=0, campos=t(inp.campos), prefiltered=False, debug=False)
    feats = t(inp.colors_precomp).requires_grad_(True)
    means3D = t(inp.means3D)
    color, radii = GaussianRasterizer(st)(means3D=means3D, means2D=torch.zeros_like(means3D), shs=None, colors_precomp=feats,
                                          opacities=t(inp.opacities), scales=t(inp.scales), rotations=t(inp.rotations), cov3D_precomp=None)
    assert color.shape == (C, inp.image_height, inp.image_width)
    dL = scenes.make_grad_image(C, inp.image_height, inp.image_width, seed=4)
    (color * t(dL)).sum().backward()
    fwd = so.forward(inp)
    hp.assert_close("color", color.detach().cpu().numpy(), fwd.color, flip_frac=hp.FLIP_FRAC)
    hp.assert_close("dL_dfeatures", feats.grad.cpu().numpy(), so.backward(inp, fwd, dL).dL_dcolors, flip_frac=hp.GRAD_FLIP_FRAC)


def test_three_steps_on_single_rank_rccl_group():
    """ViewShardedStep / allreduce_grads_async / set_features_ready_event through three consecutive steps with a feature update
    in between, on a single-rank RCCL group (ordering bugs -- a blend stage that reads features before the update behind the
    collective has landed, an event consumed by the wrong call -- show up at one rank too): the gradients of every step and the
    final features equal those of a plain single-stream run.  tests/dist_steps_check.py does the work in a subprocess."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, os.path.join(root, "tests", "dist_steps_check.py")], capture_output=True, text=True,
                         timeout=600, cwd=root)
    assert out.returncode == 0, out.stderr[-3000:]
    r = json.loads(out.stdout.strip().splitlines()[-1])
    assert r["moved"] > 1e-3 and min(r["stale_image_rel"]) > 1e-3, r   # the updates matter: a stale read would be visible
    assert max(r["grad_rel"]) < 1e-5 and r["feat_rel"] < 1e-5, r
    assert max(r["image_rel"]) < 1e-6 and max(r["dopacity_rel"]) < 1e-4, r


@pytest.mark.parametrize("exchange", ["allreduce", "rs-ag"])
def test_bench_two_ranks_share_the_one_gpu(tmp_path, exchange):
    """The N > 1 code path of bench.py itself with a REAL second rank on the hardware that is there: `python -m
    torch.distributed.run --nproc-per-node 2 bench.py --gpus 2` with both ranks on cuda:0 and a gloo group (MI_BENCH_SHARE_GPU /
    MI_BENCH_DIST_BACKEND: RCCL refuses two ranks on one device) -- the settle-flag all-reduce, the per-block MAX over ranks,
    the asynchronous exchange + the features-ready event handed to the next forward, every rank rendering its own camera, the
    sustained region's agreed step count.  Checked: one JSON line from rank 0 with n_gpus 2, and
      allreduce : the feature gradient rank 0 holds after a step == the SUM of the two cameras' single-rank gradients;
      rs-ag     : (--rs-ag: reduce-scatter -> rank-local SGD update of 1/N of the rows -> all-gather, dist.sharded_update_async) both ranks
                  hold the same features before and after the step, after == before - lr * (sum of the two local gradients), and each
                  local gradient is the single-rank gradient of that rank's camera at the features the step started from."""
    import json
    import socket
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    Pn, lr = 30_000, 50.0
    env = dict(os.environ, MI_BENCH_SHARE_GPU="1", MI_BENCH_DIST_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0", MI_BENCH_TRACE="1",
               GLOO_SOCKET_IFNAME="lo")   # (the box's hostname need not resolve)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "4", "--warmup", "2", "--points", str(Pn),
           "--settle", "0.2", "--dist-blocks", "2", "--sustained-seconds", "0.05", "--no-cpu-baseline", "--dump-grads", str(tmp_path)]
    if exchange == "rs-ag":
        cmd += ["--rs-ag", "--rs-ag-lr", str(lr)]   # (a step large enough to see: gradients are ~1e-6)
    import signal
    proc = subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, cwd=root, env=env, start_new_session=True)
    try:
        so_, se_ = proc.communicate(timeout=150)
    except subprocess.TimeoutExpired:   # MI_BENCH_TRACE says where the ranks were
        # torchrun forwards SIGTERM to its workers (they run in sessions of their own: a SIGKILL of the launcher's process group alone
        # leaves them alive, holding the pipes open -- round 4 lost ten minutes of a GPU box to that)
        proc.terminate()
        try:
            so_, se_ = proc.communicate(timeout=20)
        except subprocess.TimeoutExpired:
            os.killpg(proc.pid, signal.SIGKILL)
            try:
                so_, se_ = proc.communicate(timeout=10)
            except subprocess.TimeoutExpired:
                so_, se_ = "", "(workers still hold the pipes)"
        raise AssertionError("bench.py --gpus 2 did not finish within 150 s:\n" + (se_ or "")[-6000:]) from None
    out = subprocess.CompletedProcess(cmd, proc.returncode, so_, se_)
    assert out.returncode == 0, out.stderr[-3000:]
    line = json.loads([l for l in out.stdout.strip().splitlines() if l.startswith("{")][-1])
    assert line["n_gpus"] == 2 and line["steps"] == 4 and line["value"] > 0 and line["scaling"] == "weak"
    assert "x2" in line["config"]["parallelism"] and line["sustained"]["steps"] >= 4
    assert ("rs-ag" in line["config"]["parallelism"]) == (exchange == "rs-ag")
    # the line proves who took part: the process group's own view of the ranks and every rank's device (config.comm)
    comm = line["config"]["comm"]
    assert comm["nranks"] == 2 and comm["backend"] == "gloo" and comm["exchange"] == exchange and len(comm["devices"]) == 2
    assert sorted(d["rank"] for d in comm["devices"]) == [0, 1] and len({d["pid"] for d in comm["devices"]}) == 2
    assert comm["distinct_devices"] == 1            # (both ranks share the one GPU here; N distinct devices on a real node)
    assert comm["message_bytes"] == Pn * 32 * 4 and isinstance(comm["exposed_ms"], float)
    # the same two views, one after the other, on this process: bench.py's scene, cameras (rank 0 front, rank 1 orbit) and dL
    c = scenes.CONFIGS["cfg3"]
    sc = scenes.scene_of_config("cfg3", seed=0, P=Pn)
    dL = torch.as_tensor(scenes.make_grad_image(c["C"], c["H"], c["W"], seed=1)).cuda()
    from seganygaussians_amd import rasterizer as R
    _, _, Rz = R.make_rasterizer(c["C"])
    from diff_gaussian_rasterization_contrastive_f import GaussianRasterizationSettings
    t = lambda a: torch.as_tensor(np.ascontiguousarray(a, np.float32)).cuda()
    if exchange == "rs-ag":
        before = [np.load(tmp_path / f"features_before_rank{r}.npy") for r in range(2)]
        after = [np.load(tmp_path / f"features_rank{r}.npy") for r in range(2)]
        np.testing.assert_array_equal(before[0], before[1], err_msg="features before the step differ between the ranks")
        np.testing.assert_array_equal(after[0], after[1], err_msg="features after the sharded update differ between the ranks")
        start = before[0]
    else:
        start = sc.features
    single = []
    for rank in range(2):
        cam = scenes.look_at_camera(c["W"], c["H"], c["focal"]) if rank == 0 else scenes.orbit_camera(c["W"], c["H"], c["focal"], 0.05, 0.02)
        dumped = np.load(tmp_path / f"camera_{rank}.npy")
        np.testing.assert_array_equal(dumped, np.concatenate([cam.viewmatrix.ravel(), cam.projmatrix.ravel()]))
        feats = t(start).requires_grad_(True)
        st = GaussianRasterizationSettings(image_height=c["H"], image_width=c["W"], tanfovx=cam.tanfovx, tanfovy=cam.tanfovy,
                                           bg=torch.zeros(c["C"]).cuda(), scale_modifier=1.0, viewmatrix=t(cam.viewmatrix),
                                           projmatrix=t(cam.projmatrix), sh_degree=0, campos=t(cam.campos), prefiltered=False, debug=False)
        m3 = t(sc.means3D)
        color, _ = Rz(st)(means3D=m3, means2D=torch.zeros_like(m3), shs=None, colors_precomp=feats, opacities=t(sc.opacities),
                          scales=t(sc.scales), rotations=t(sc.rotations), cov3D_precomp=None)
        torch.autograd.backward(color, grad_tensors=dL)
        single.append(feats.grad.detach().cpu().numpy().astype(np.float64))
    want = single[0] + single[1]
    assert np.abs(want).max() > 0
    if exchange == "rs-ag":
        local = [np.load(tmp_path / f"feature_grad_local_rank{r}.npy").astype(np.float64) for r in range(2)]
        for r in range(2):
            hp.assert_close(f"local feature gradient of rank {r}", local[r], single[r], rtol=2e-4, flip_frac=hp.GRAD_FLIP_FRAC)
        moved = after[0].astype(np.float64) - start.astype(np.float64)
        assert np.abs(moved).max() > 0
        # the update is exactly -lr * (sum of the dumped local gradients), up to float32 rounding of the two operations
        np.testing.assert_allclose(moved, -lr * (local[0] + local[1]), rtol=1e-4, atol=2e-6 * np.abs(start).max())
    else:
        got = np.load(tmp_path / "feature_grad_rank0.npy").astype(np.float64)
        hp.assert_close("all-reduced feature gradient of two ranks", got, want, rtol=2e-4, flip_frac=hp.GRAD_FLIP_FRAC)


def test_bench_on_a_ply_scene_with_colmap_cameras(tmp_path):
    """`bench.py --ply <3DGS point_cloud.ply> --cameras <COLMAP scene>` end to end (BASELINE configs 2-5 name garden / bicycle: where the
    data exists the headline workload runs on it): a synthetic scene written in the reference's 3DGS PLY layout (raw parameters: log
    scales, logit opacities, scene/gaussian_model.py:271-322) and a binary COLMAP model with two PINHOLE cameras.  Checked: the line
    says what it ran on, the counters are the file's, the second camera (sorted by image name) is the one rendered with
    --camera-index 1, and the in-run parity of the loaded scene against the CPU oracle is inside the contract."""
    import json
    import struct
    import subprocess
    import sys
    from seganygaussians_amd import ply_io
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    P, W, H, focal = 20_000, 640, 368, 500.0
    sc = scenes.make_scene(P, W, H, focal, 32, np.log(0.05), 0.6, seed=21, with_shs=True)
    names = (["x", "y", "z", "nx", "ny", "nz"] + [f"f_dc_{i}" for i in range(3)] + [f"f_rest_{i}" for i in range(45)] + ["opacity"] +
             [f"scale_{i}" for i in range(3)] + [f"rot_{i}" for i in range(4)])
    shs = np.asarray(sc.shs, np.float32).reshape(P, 16, 3)
    op = np.clip(np.asarray(sc.opacities, np.float64).reshape(P), 1e-6, 1 - 1e-6)
    cols = np.concatenate([np.asarray(sc.means3D, np.float32), np.zeros((P, 3), np.float32), shs[:, 0, :],
                           shs[:, 1:, :].transpose(0, 2, 1).reshape(P, 45),           # channel-major on disk
                           np.log(op / (1 - op)).astype(np.float32)[:, None], np.log(np.asarray(sc.scales, np.float64)).astype(np.float32),
                           np.asarray(sc.rotations, np.float32)], axis=1).astype(np.float32)
    ply = str(tmp_path / "point_cloud.ply")
    ply_io.write_vertex_ply(ply, names, cols)
    colmap = str(tmp_path / "scene")
    os.makedirs(os.path.join(colmap, "sparse", "0"))
    with open(os.path.join(colmap, "sparse/0/cameras.bin"), "wb") as f:   # one PINHOLE camera: id, model 1, width, height, fx fy cx cy
        f.write(struct.pack("<Q", 1) + struct.pack("<iiQQ", 1, 1, W, H) + struct.pack("<dddd", focal, focal, W / 2, H / 2))
    ang = 0.1
    poses = [(1, (1.0, 0.0, 0.0, 0.0), (0.0, 0.0, 0.0), "b_second.jpg"),                   # camera at the origin looking down +z
             (2, (float(np.cos(ang / 2)), 0.0, float(np.sin(ang / 2)), 0.0), (0.3, 0.0, 0.1), "a_first.jpg")]
    with open(os.path.join(colmap, "sparse/0/images.bin"), "wb") as f:
        f.write(struct.pack("<Q", len(poses)))
        for iid, q, t, name in poses:
            f.write(struct.pack("<idddddddi", iid, *q, *t, 1) + name.encode() + b"\x00" + struct.pack("<Q", 0))
    cmd = [sys.executable, os.path.join(root, "bench.py"), "--ply", ply, "--cameras", colmap, "--camera-index", "1", "--config", "cfg3",
           "--steps", "3", "--warmup", "1", "--settle", "0", "--dist-blocks", "0", "--sustained-seconds", "0", "--cpu-views", "1"]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=600, cwd=root)
    assert out.returncode == 0, out.stderr[-3000:]
    line = json.loads([l for l in out.stdout.strip().splitlines() if l.startswith("{")][-1])
    assert line["data"].startswith("file:") and "b_second.jpg" in line["data"] and f"{P} Gaussians" in line["data"], line["data"]
    c = line["config"]["counters"]
    assert c["P"] == P and c["N"] == W * H and c["V"] > P // 2 and c["R"] > c["V"], c
    assert line["value"] > 0 and line["parity"]["ok"] and line["parity"]["radii_equal"], line["parity"]
    assert f"{W}x{H}" in line["config"]["workload"] and "32-D features" in line["config"]["workload"], line["config"]["workload"]


@pytest.mark.parametrize("use_cov", [False, True])
def test_backward_writes_every_gradient_row(use_cov):
    """include/mi_rast.h: only dL_dcolor / dL_dsh must be cleared by the caller; every other gradient output is written in
    full by mi_rast_backward -- zeros for Gaussians that were not rendered (behind the camera, outside the frustum, zero-area
    rect), zeros for scales / rotations when cov3D_precomp is given.  With `debug` the glue hands the library a NaN-poisoned
    block, so a row it failed to write shows up; values are compared with the oracle's (the reference's zero-filled tensors)."""
    inp = hp.make_inputs(4000, 160, 112, 32, seed=5, camera="orbit", use_cov=use_cov)
    g = hp.GpuRun(inp).forward(debug=True)
    radii = g.radii.cpu().numpy()
    assert (radii == 0).sum() > 100 and (radii > 0).sum() > 1000   # the scene has both kinds
    dL = scenes.make_grad_image(32, 112, 160, seed=3)
    grads = g.backward(dL, debug=True)
    of = so.forward(inp)
    ob = so.backward(inp, of, dL)
    for k, v in grads.items():
        assert np.isfinite(v).all(), k
        if k != "dL_dcolors":
            assert not v[radii == 0].any(), k
    hp.compare_gradients(grads, ob)
    if use_cov:
        assert not grads["dL_dscales"].any() and not grads["dL_drotations"].any()
