"""N > 1 path on CPU: world_size-2 `gloo` processes run the view-sharded step (seganygaussians_amd/dist.py).
Per-view gradients come from the CPU oracle (test infrastructure) standing in for the HIP rasterizer; the
check is the one SURVEY.md 8(e) asks for: all-reduced gradient == sum of the per-view gradients."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from seganygaussians_amd import scenes
from seganygaussians_amd.dist import (ShardedAdam, ViewShardedStep, allreduce_grads, allreduce_grads_async, shard_range, sharded_update_async,
                                      views_for_rank)

NUM_VIEWS, P, W, H, C = 5, 400, 64, 48, 32


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _view_grad(view, sc):
    """Oracle dL/dfeatures and dL/dopacity of one orbit view (deterministic function of the view index)."""
    from oracle import saga_oracle as so
    cam = scenes.orbit_camera(W, H, 0.9 * W, 0.07 * view, 0.03 * view)
    inp = so.Inputs(means3D=sc.means3D, opacities=sc.opacities, viewmatrix=cam.viewmatrix, projmatrix=cam.projmatrix,
                    campos=cam.campos, bg=np.zeros(C, np.float32), image_width=W, image_height=H, tanfovx=cam.tanfovx,
                    tanfovy=cam.tanfovy, channels=C, colors_precomp=sc.features, scales=sc.scales, rotations=sc.rotations)
    fwd = so.forward(inp)
    bwd = so.backward(inp, fwd, scenes.make_grad_image(C, H, W, seed=100 + view))
    return bwd.dL_dcolors, bwd.dL_dopacity


def _worker(rank, world, port, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from oracle import saga_oracle as so
        so.set_num_threads(2)
        sc = scenes.make_scene(P, W, H, 0.9 * W, C, np.log(0.15), 0.5, seed=5, z_range=(3.0, 9.0))
        feats = torch.nn.Parameter(torch.tensor(sc.features))
        opac = torch.nn.Parameter(torch.tensor(sc.opacities))
        rendered = []

        def render_backward(v):
            gf, go = _view_grad(v, sc)
            rendered.append(v)
            for p, g in ((feats, gf), (opac, go)):
                g = torch.tensor(np.asarray(g)).reshape(p.shape)
                p.grad = g if p.grad is None else p.grad + g

        step = ViewShardedStep([feats, opac])
        mine = step(NUM_VIEWS, render_backward)
        assert mine == rendered == views_for_rank(NUM_VIEWS, rank, world)
        np.save(os.path.join(out_dir, f"feats_grad_{rank}.npy"), feats.grad.numpy())
        np.save(os.path.join(out_dir, f"opac_grad_{rank}.npy"), opac.grad.numpy())
        # a lone contiguous tensor takes the un-bucketed path; average=True divides by the world size
        t = torch.full((3, 2), float(rank + 1))
        allreduce_grads([t], average=True)
        assert torch.allclose(t, torch.full((3, 2), sum(range(1, world + 1)) / world))
        # the asynchronous variant (what bench.py overlaps with the next view's geometry stages): on CPU tensors the sum is
        # complete on return and there is no event to wait for
        u = torch.full((5,), float(rank + 1))
        ev, keep = allreduce_grads_async([u, None])
        assert ev is None and torch.equal(u, torch.full((5,), float(sum(range(1, world + 1)))))
        # reduce-scatter -> rank-local update -> all-gather (sharded_update_async) against all-reduce + the same update everywhere:
        # a momentum step whose state every rank keeps for ITS rows only; odd sizes take the padded path
        for shape in ((P, C), (7, 3)):
            g0 = torch.Generator().manual_seed(17)
            param0 = torch.randn(shape, generator=g0)
            grads = [torch.randn(shape, generator=g0) for _ in range(world)]    # rank r's local gradient: grads[r] (same on every rank)
            lo, hi = shard_range(param0.numel(), rank, world)
            mom = torch.zeros(hi - lo)
            touched = []

            def update(prow, grow, a, b):
                assert (a, b) == (lo, hi) and prow.numel() == grow.numel() == b - a
                touched.append((a, b))
                mom.mul_(0.9).add_(grow)
                prow.add_(mom, alpha=-0.1)

            param = param0.clone()
            for _ in range(2):   # two steps: the sharded momentum carries over
                ev, keep = sharded_update_async(param, grads[rank].clone(), update)
                assert ev is None
            want, m = param0.clone().view(-1), torch.zeros(param0.numel())
            for _ in range(2):
                m.mul_(0.9).add_(sum(grads).view(-1))
                want.add_(m, alpha=-0.1)
            assert len(touched) == 2 and torch.allclose(param.view(-1), want, rtol=1e-6, atol=1e-6), shape
            np.save(os.path.join(out_dir, f"sharded_{shape[0]}_{rank}.npy"), param.numpy())
        # ShardedAdam (moments for this rank's rows only) against torch.optim.Adam on the whole tensor with the summed gradients
        g0 = torch.Generator().manual_seed(23)
        p0 = torch.randn((P, C), generator=g0)
        steps = [[torch.randn((P, C), generator=g0) * 1e-3 for _ in range(world)] for _ in range(3)]
        mine_p, opt = p0.clone(), ShardedAdam(lr=0.0025)
        ref_p = torch.nn.Parameter(p0.clone())
        ref_opt = torch.optim.Adam([ref_p], lr=0.0025)
        for gs in steps:
            sharded_update_async(mine_p, gs[rank].clone(), opt)
            ref_p.grad = sum(gs)
            ref_opt.step()
        lo, hi = shard_range(p0.numel(), rank, world)
        assert opt.exp_avg.numel() == hi - lo and opt.step == 3
        assert torch.allclose(mine_p, ref_p.detach(), rtol=1e-5, atol=1e-7), float((mine_p - ref_p.detach()).abs().max())
        # checkpoint / resume: the gathered state is torch.optim.Adam's state of the parameter, and a fresh ShardedAdam loaded from
        # EITHER continues like the reference optimizer (FeatureGaussianModel.capture / restore round-trips optimizer.state_dict())
        sd = opt.state_dict(p0.numel(), shape=(P, C))
        import copy
        ref_state = copy.deepcopy(ref_opt.state_dict()["state"][0])   # (the optimizer's own tensors move with its next step)
        assert float(sd["step"]) == float(ref_state["step"]) == 3.0
        assert torch.allclose(sd["exp_avg"], ref_state["exp_avg"], rtol=1e-5, atol=1e-9)
        assert torch.allclose(sd["exp_avg_sq"], ref_state["exp_avg_sq"], rtol=1e-5, atol=1e-12)
        more = [torch.randn((P, C), generator=g0) * 1e-3 for _ in range(world)]
        ref_p.grad = sum(more)
        ref_opt.step()
        for source in (sd, ref_state):
            resumed, p_res = ShardedAdam(lr=0.0025), mine_p.clone()
            resumed.load_state_dict(source)
            assert resumed.step == 3 and resumed.exp_avg.numel() == hi - lo
            sharded_update_async(p_res, more[rank].clone(), resumed)
            assert torch.allclose(p_res, ref_p.detach(), rtol=1e-5, atol=1e-7)
        with pytest.raises(ValueError):
            resumed.load_state_dict({"step": 1, "exp_avg": torch.zeros(5), "exp_avg_sq": torch.zeros(5)})   # another tensor's state
        # average=True: the summed gradient divided by the world size before the update
        pa, ga = torch.zeros(6), torch.full((6,), float(rank + 1))
        sharded_update_async(pa, ga, lambda prow, grow, a, b: prow.add_(grow), average=True)
        assert torch.allclose(pa, torch.full((6,), sum(range(1, world + 1)) / world))
    finally:
        dist.destroy_process_group()


def test_views_for_rank_partition():
    for world in (1, 2, 4, 8):
        seen = sorted(v for r in range(world) for v in views_for_rank(13, r, world))
        assert seen == list(range(13))
    assert views_for_rank(8, 3, 8) == [3] and views_for_rank(2, 5, 8) == []
    with pytest.raises(ValueError):
        views_for_rank(4, 4, 4)


def test_allreduce_is_noop_without_process_group():
    t = torch.ones(4)
    allreduce_grads([t, None])
    assert torch.equal(t, torch.ones(4))
    assert allreduce_grads_async([t]) == (None, None) and torch.equal(t, torch.ones(4))
    # the sharded update without a process group: one "shard", the whole tensor
    p, g = torch.ones(3, 2), torch.full((3, 2), 2.0)
    assert sharded_update_async(p, g, lambda prow, grow, lo, hi: prow.sub_(grow)) == (None, None)
    assert torch.equal(p, torch.full((3, 2), -1.0))
    assert [shard_range(10, r, 4) for r in range(4)] == [(0, 3), (3, 6), (6, 9), (9, 10)] and shard_range(2, 3, 4) == (2, 2)


@pytest.mark.timeout(300)
def test_view_sharded_step_world2_gloo(tmp_path):
    world = 2
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    sc = scenes.make_scene(P, W, H, 0.9 * W, C, np.log(0.15), 0.5, seed=5, z_range=(3.0, 9.0))
    want_f = sum(_view_grad(v, sc)[0].astype(np.float64) for v in range(NUM_VIEWS))
    want_o = sum(_view_grad(v, sc)[1].astype(np.float64) for v in range(NUM_VIEWS))
    assert np.abs(want_f).max() > 0
    for r in range(world):
        got_f = np.load(tmp_path / f"feats_grad_{r}.npy")
        got_o = np.load(tmp_path / f"opac_grad_{r}.npy")
        np.testing.assert_allclose(got_f, want_f, rtol=1e-5, atol=1e-6 * np.abs(want_f).max())
        np.testing.assert_allclose(got_o.reshape(-1), want_o.reshape(-1), rtol=1e-5, atol=1e-6 * np.abs(want_o).max())
    # the sharded update left every rank with the same, fully updated parameter
    for n in (P, 7):
        np.testing.assert_array_equal(np.load(tmp_path / f"sharded_{n}_0.npy"), np.load(tmp_path / f"sharded_{n}_1.npy"))
