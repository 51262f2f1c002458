"""north_star: "train_contrastive_feature.py and render.py call it unchanged".  The reference's REAL scripts -- Scene (COLMAP
loader, cameras), GaussianModel / FeatureGaussianModel (create_from_pcd, save_ply, load_ply, load_ply_from_3dgs, Adam),
train_contrastive_feature.training() and render.render_sets() -- run end to end on a synthetic COLMAP-format dataset written to a
temporary directory, on top of the drop-in packages (rasterizer, simple_knn, pytorch3d.ops = this repository; plyfile / torchvision =
small stand-ins, tests/ref_env.py).  Checks: the loop trains (finite losses, features and scale gate move), every file the scripts
write is there, the rendered feature image equals the CPU oracle's, and the PLY files written / read by the reference's code agree
bit for bit with seganygaussians_amd/ply_io.py.  File name sorts last (a missing oracle/_ref fails here)."""
import math
import os
from argparse import ArgumentParser

import numpy as np
import pytest
import torch

from oracle import saga_oracle as so
from seganygaussians_amd import ply_io
from tests import helpers as hp
from tests.ref_env import ReferenceEnv

pytestmark = pytest.mark.gpu

W, H, FOCAL, NCAM, NPTS, NMASK = 160, 100, 140.0, 6, 4000, 6


def _qvec(R):
    """COLMAP quaternion (w, x, y, z) of a rotation matrix."""
    t = np.trace(R)
    w = math.sqrt(max(0.0, 1 + t)) / 2
    x = math.copysign(math.sqrt(max(0.0, 1 + R[0, 0] - R[1, 1] - R[2, 2])) / 2, R[2, 1] - R[1, 2])
    y = math.copysign(math.sqrt(max(0.0, 1 - R[0, 0] + R[1, 1] - R[2, 2])) / 2, R[0, 2] - R[2, 0])
    z = math.copysign(math.sqrt(max(0.0, 1 - R[0, 0] - R[1, 1] + R[2, 2])) / 2, R[1, 0] - R[0, 1])
    return w, x, y, z


def _write_dataset(root, seed=0):
    from PIL import Image
    rng = np.random.default_rng(seed)
    for d in ("sparse/0", "images", "sam_masks", "mask_scales"):
        os.makedirs(os.path.join(root, d), exist_ok=True)
    with open(os.path.join(root, "sparse/0/cameras.txt"), "w") as f:
        f.write(f"# synthetic\n1 PINHOLE {W} {H} {FOCAL} {FOCAL} {W / 2} {H / 2}\n")
    with open(os.path.join(root, "sparse/0/images.txt"), "w") as f:
        for i in range(NCAM):
            a = 0.25 * (i - NCAM / 2)
            Rc2w = np.array([[math.cos(a), 0, math.sin(a)], [0, 1, 0], [-math.sin(a), 0, math.cos(a)]])
            centre = np.array([0.0, 0.0, 6.0])
            pos = centre - Rc2w @ np.array([0.0, 0.0, 6.0])
            Rw2c = Rc2w.T
            t = -Rw2c @ pos
            q = _qvec(Rw2c)
            f.write(f"{i + 1} {q[0]} {q[1]} {q[2]} {q[3]} {t[0]} {t[1]} {t[2]} 1 view_{i:02d}.png\n0.0 0.0 -1\n")
            Image.fromarray(rng.integers(0, 255, (H, W, 3), dtype=np.uint8)).save(os.path.join(root, "images", f"view_{i:02d}.png"))
            # SAM-like masks at several scales: blocky partitions of the image, coarse to fine
            masks = torch.zeros(NMASK, H, W)
            for m in range(NMASK):
                y0, x0 = rng.integers(0, H // 2), rng.integers(0, W // 2)
                hh, ww = rng.integers(H // 6, H // 2), rng.integers(W // 6, W // 2)
                masks[m, y0:y0 + hh, x0:x0 + ww] = 1
            torch.save(masks.bool(), os.path.join(root, "sam_masks", f"view_{i:02d}.pt"))
            torch.save(torch.as_tensor(rng.uniform(0.2, 3.0, NMASK), dtype=torch.float32), os.path.join(root, "mask_scales", f"view_{i:02d}.pt"))
    pts = np.stack([rng.uniform(-3, 3, NPTS), rng.uniform(-2, 2, NPTS), rng.uniform(4, 8, NPTS)], 1)
    with open(os.path.join(root, "sparse/0/points3D.txt"), "w") as f:
        for i, p in enumerate(pts):
            c = rng.integers(0, 255, 3)
            f.write(f"{i + 1} {p[0]} {p[1]} {p[2]} {c[0]} {c[1]} {c[2]} 0.5 1 0\n")


@pytest.mark.parametrize("geometry_cache,features_only", [(False, False), (True, False), (True, True)])
def test_reference_training_and_rendering_scripts_end_to_end(tmp_path, geometry_cache, features_only):
    """geometry_cache=True: the same scripts with the opt-in frozen-geometry reuse switched on (rasterizer.GeometryCache): the
    training loop revisits its cameras with unchanged geometry -- activation outputs, new tensors per call: hits by content --,
    and every check below (losses, files, the rendered feature image against the oracle) must hold unchanged.
    features_only=True: also the opt-in features-only backward (rasterizer.enable_features_only_backward): the loop optimises
    `_point_features` alone (scene/gaussian_model_ff.py:154-162) -- every backward of the training must take that form, and the
    same checks hold."""
    from seganygaussians_amd import rasterizer as R_
    cache = None
    if geometry_cache:
        cache = R_.enable_geometry_cache(4 << 30)
        cache.clear()
    prev_fo = R_.enable_features_only_backward(features_only)
    calls, orig = [], R_.rasterize_gaussians_backward_native

    def spy(*a, **k):
        calls.append(bool(k.get("features_only")))
        return orig(*a, **k)
    R_.rasterize_gaussians_backward_native = spy
    try:
        _run_reference_scripts(tmp_path, cache)
    finally:
        R_.rasterize_gaussians_backward_native = orig
        R_.enable_features_only_backward(prev_fo)
        R_.disable_geometry_cache(drop=True)
    assert len(calls) >= 20 and all(c == features_only for c in calls), calls


def _run_reference_scripts(tmp_path, cache):
    src, model = str(tmp_path / "data"), str(tmp_path / "model")
    _write_dataset(src)
    os.makedirs(model)
    with ReferenceEnv() as ref:
        A, S = ref.mod["arguments"], ref.mod["scene"]
        tr, rd = ref.mod["train_contrastive_feature"], ref.mod["render"]
        import diff_gaussian_rasterization_contrastive_f
        assert ref.mod["gaussian_renderer"].GaussianRasterizerContrastiveF is diff_gaussian_rasterization_contrastive_f.GaussianRasterizer
        parser = ArgumentParser()
        lp, op, pp = A.ModelParams(parser), A.OptimizationParams(parser), A.PipelineParams(parser)
        args = parser.parse_args(["-s", src, "-m", model, "--iterations", "20", "--num_sampled_rays", "400"])
        dataset, opt, pipe = lp.extract(args), op.extract(args), pp.extract(args)

        # ---- the 3DGS scene the feature training starts from: COLMAP points -> create_from_pcd (distCUDA2 = ours) -> save_ply
        g0 = S.GaussianModel(dataset.sh_degree)
        scene0 = S.Scene(dataset, g0, None, load_iteration=None, shuffle=False, target="scene", mode="train")
        assert len(scene0.getTrainCameras()) == NCAM and g0.get_xyz.shape == (NPTS, 3)
        with torch.no_grad():   # make the initial blobs visible enough for a meaningful render
            g0._opacity.data.fill_(2.0)
        scene0.save(7, target="scene")
        ply3 = os.path.join(model, "point_cloud/iteration_7/scene_point_cloud.ply")
        mine = ply_io.load_3dgs_ply(ply3, dataset.sh_degree)
        np.testing.assert_array_equal(mine["xyz"], g0._xyz.detach().cpu().numpy())
        np.testing.assert_array_equal(mine["features_dc"], g0._features_dc.detach().cpu().numpy())
        np.testing.assert_array_equal(mine["features_rest"], g0._features_rest.detach().cpu().numpy())
        np.testing.assert_array_equal(mine["scaling"], g0._scaling.detach().cpu().numpy())
        g1 = S.GaussianModel(dataset.sh_degree)
        g1.load_ply(ply3)                                      # the reference's reader on the reference's file
        np.testing.assert_array_equal(g1._features_rest.detach().cpu().numpy(), mine["features_rest"])
        np.testing.assert_array_equal(g1._rotation.detach().cpu().numpy(), mine["rotation"])

        # ---- train_contrastive_feature.training(): 20 iterations of the real loop
        losses, seen = [], {}

        class Bar:   # stands in for tqdm: records what the loop reports
            def __init__(self, *a, **k): pass
            def set_postfix(self, d): losses.append({k: float(v) for k, v in d.items()})
            def update(self, n): pass
            def close(self): pass
        tr.tqdm = Bar
        real_save = S.Scene.save_feature

        def spy_save(self, iteration, **kw):
            seen["features"] = self.feature_gaussians._point_features.detach().clone()
            seen["model"] = self.feature_gaussians
            return real_save(self, iteration, **kw)
        S.Scene.save_feature = spy_save
        torch.manual_seed(0)
        tr.training(dataset, opt, pipe, 7, [], [], -1)
        S.Scene.save_feature = real_save
        if cache is not None:   # 20 iterations over NCAM cameras: every camera is revisited, the geometry never changes
            st = cache.stats()
            assert st["hits"] >= 20 - 2 * NCAM and st["views"] <= NCAM + 1 and st["hits"] + st["misses"] >= 20, st
        assert len(losses) == 2 and all(math.isfinite(v) for d in losses for v in d.values()), losses
        feats = seen["features"]
        assert torch.isfinite(feats).all() and float(feats.abs().max()) > 2e-2      # moved away from the 1e-2 randn start
        out_dir = os.path.join(model, "point_cloud/iteration_20")
        fply = os.path.join(out_dir, "contrastive_feature_point_cloud.ply")
        assert os.path.exists(fply) and os.path.exists(os.path.join(out_dir, "scale_gate.pt"))
        gate = torch.load(os.path.join(out_dir, "scale_gate.pt"))
        assert set(gate) == {"0.weight", "0.bias"} and all(torch.isfinite(v).all() for v in gate.values())

        # ---- the feature PLY: reference writer (smoothed features, save_ply :567-592) vs ply_io, both directions, bit for bit
        fg = S.FeatureGaussianModel(dataset.feature_dim)
        fg.load_ply(fply)
        mine = ply_io.load_feature_ply(fply, dataset.feature_dim)
        for k, t in dict(xyz=fg._xyz, point_features=fg._point_features, opacity=fg._opacity, scaling=fg._scaling, rotation=fg._rotation).items():
            np.testing.assert_array_equal(mine[k], t.detach().cpu().numpy())
        again = str(tmp_path / "again.ply")
        ply_io.save_feature_ply(again, mine["xyz"], mine["point_features"], mine["opacity"], mine["scaling"], mine["rotation"])
        assert open(again, "rb").read() == open(fply, "rb").read()
        fg2 = S.FeatureGaussianModel(dataset.feature_dim)
        fg2.load_ply(again)                                    # the reference's reader on OUR file
        assert torch.equal(fg2._point_features, fg._point_features) and torch.equal(fg2._rotation, fg._rotation)

        # ---- render.py: render_sets(target='contrastive_feature') loads the feature PLY and writes one .pt per view
        rd.tqdm = lambda it, **k: it
        rd.render_sets(dataset, -1, pipe, False, True, False, "contrastive_feature")
        rdir = os.path.join(model, "train/ours_-1/renders")      # scene.loaded_iter is -1 when only the feature model is loaded (scene/__init__.py:64)
        files = sorted(os.listdir(rdir))
        assert files == [f"{i:05d}.pt" for i in range(NCAM)]
        img = torch.load(os.path.join(rdir, files[0]))
        cam = S.Scene(dataset, None, S.FeatureGaussianModel(dataset.feature_dim), load_iteration=-1, shuffle=False, mode="eval",
                      target="contrastive_feature").getTrainCameras()[0]
        fh, fw = cam.feature_height, cam.feature_width
        assert img.shape == (dataset.feature_dim, fh, fw) and torch.isfinite(img).all() and float(img.abs().max()) > 0
        n = lambda t: t.detach().cpu().numpy()
        inp = so.Inputs(means3D=n(fg.get_xyz), opacities=n(fg.get_opacity), viewmatrix=n(cam.world_view_transform),
                        projmatrix=n(cam.full_proj_transform), campos=n(cam.camera_center), bg=np.zeros(dataset.feature_dim, np.float32),
                        image_width=fw, image_height=fh, tanfovx=math.tan(cam.FoVx * 0.5), tanfovy=math.tan(cam.FoVy * 0.5),
                        channels=dataset.feature_dim, sh_degree=0, shs=None, colors_precomp=n(fg.get_point_features),
                        scales=n(fg.get_scaling), rotations=n(fg.get_rotation), mask=None)
        hp.assert_close("render.py feature image", n(img), so.forward(inp).color, flip_frac=hp.FLIP_FRAC)
