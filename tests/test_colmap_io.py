"""seganygaussians_amd/colmap_io.py (the COLMAP camera files `bench.py --cameras` reads) against the REFERENCE's own loader:
scene/colmap_loader.py's readers and qvec2rotmat, utils/graphics_utils.py's focal2fov / getWorld2View2 / getProjectionMatrix,
byte-compiled where they lie (oracle/_ref/pyref, test infrastructure).  CPU only."""
import math
import os
import struct
import sys
import types

import numpy as np
import pytest

from oracle import build_ref
from seganygaussians_amd import colmap_io, scenes


def _ref_modules():
    from tests import ref_env
    for name in ("scene.colmap_loader", "utils.graphics_utils"):
        if not os.path.exists(build_ref.pyref_path(name)):
            pytest.skip("oracle/_ref/pyref missing (python oracle/build_ref.py builds it where /root/reference exists)")
    saved = {k: sys.modules.get(k) for k in ("scene", "utils", "scene.colmap_loader", "utils.graphics_utils")}
    try:
        for pkg in ("utils", "scene"):
            m = types.ModuleType(pkg)
            m.__path__ = []
            sys.modules[pkg] = m
        return ref_env._exec_pyc("scene.colmap_loader"), ref_env._exec_pyc("utils.graphics_utils")
    finally:
        for k, v in saved.items():
            if v is None:
                sys.modules.pop(k, None)
            else:
                sys.modules[k] = v


def _poses(n, seed=0):
    rng = np.random.default_rng(seed)
    out = []
    for i in range(n):
        q = rng.normal(0, 1, 4)
        q /= np.linalg.norm(q)
        out.append((i + 1, tuple(q), tuple(rng.uniform(-3, 3, 3)), 1 + i % 2, f"DSC{90 - i:04d}.JPG"))
    return out


CAMS = {1: ("PINHOLE", 5187, 3361, (4000.5, 3990.25, 2593.5, 1680.5)), 2: ("SIMPLE_PINHOLE", 1296, 840, (1100.0, 648.0, 420.0))}
CAMS_TXT = {1: CAMS[1], 2: ("PINHOLE", 1296, 840, (1100.0, 1090.0, 648.0, 420.0))}   # the reference's text reader takes PINHOLE only
MODEL_ID = {"SIMPLE_PINHOLE": 0, "PINHOLE": 1}


def _write_txt(root, poses):
    os.makedirs(os.path.join(root, "sparse", "0"))
    with open(os.path.join(root, "sparse/0/cameras.txt"), "w") as f:
        f.write("# Camera list with one line of data per camera:\n")
        for cid, (m, w, h, p) in CAMS_TXT.items():
            f.write(f"{cid} {m} {w} {h} " + " ".join(repr(v) for v in p) + "\n")
    with open(os.path.join(root, "sparse/0/images.txt"), "w") as f:
        f.write("# Image list with two lines of data per image:\n")
        for iid, q, t, cid, name in poses:
            f.write(f"{iid} " + " ".join(repr(float(v)) for v in q + t) + f" {cid} {name}\n1.0 2.0 -1 3.5 4.5 17\n")


def _write_bin(root, poses):
    os.makedirs(os.path.join(root, "sparse", "0"))
    with open(os.path.join(root, "sparse/0/cameras.bin"), "wb") as f:
        f.write(struct.pack("<Q", len(CAMS)))
        for cid, (m, w, h, p) in CAMS.items():
            f.write(struct.pack("<iiQQ", cid, MODEL_ID[m], w, h) + struct.pack("<" + "d" * len(p), *p))
    with open(os.path.join(root, "sparse/0/images.bin"), "wb") as f:
        f.write(struct.pack("<Q", len(poses)))
        for iid, q, t, cid, name in poses:
            f.write(struct.pack("<idddddddi", iid, *q, *t, cid) + name.encode() + b"\x00")
            f.write(struct.pack("<Q", 2) + struct.pack("<ddq", 1.0, 2.0, -1) + struct.pack("<ddq", 3.5, 4.5, 17))


@pytest.mark.parametrize("fmt", ["txt", "bin"])
def test_cameras_match_the_reference_loader(tmp_path, fmt):
    cl, gu = _ref_modules()
    poses = _poses(7)
    root = str(tmp_path / fmt)
    (_write_txt if fmt == "txt" else _write_bin)(root, poses)
    mine = colmap_io.read_colmap_cameras(root)
    if fmt == "txt":
        extr, intr = cl.read_extrinsics_text(root + "/sparse/0/images.txt"), cl.read_intrinsics_text(root + "/sparse/0/cameras.txt")
    else:
        extr, intr = cl.read_extrinsics_binary(root + "/sparse/0/images.bin"), cl.read_intrinsics_binary(root + "/sparse/0/cameras.bin")
    theirs = sorted(extr.values(), key=lambda e: os.path.basename(e.name).split(".")[0])   # dataset_readers.py:169
    assert [c.name for c in mine] == [os.path.basename(e.name) for e in theirs]
    for c, e in zip(mine, theirs):
        it = intr[e.camera_id]
        R = np.transpose(cl.qvec2rotmat(e.qvec))                    # dataset_readers.py:87-88
        T = np.array(e.tvec)
        fx = it.params[0]
        fy = it.params[1] if it.model == "PINHOLE" else it.params[0]
        np.testing.assert_array_equal(c.R, R)
        np.testing.assert_array_equal(c.T, T)
        assert (c.width, c.height) == (it.width, it.height)
        assert c.fovx == gu.focal2fov(fx, it.width) and c.fovy == gu.focal2fov(fy, it.height)
        # matrices: scene/cameras.py:56-66
        import torch
        cam = colmap_io.to_camera(c)
        view = torch.tensor(gu.getWorld2View2(R, T, np.array([0.0, 0.0, 0.0]), 1.0)).transpose(0, 1)
        proj = gu.getProjectionMatrix(znear=0.01, zfar=100.0, fovX=c.fovx, fovY=c.fovy).transpose(0, 1)
        full = view.unsqueeze(0).bmm(proj.unsqueeze(0)).squeeze(0)
        np.testing.assert_array_equal(cam.viewmatrix, view.numpy())
        np.testing.assert_allclose(cam.projmatrix, full.numpy(), rtol=2e-6, atol=1e-7)
        np.testing.assert_allclose(cam.campos, view.inverse()[3, :3].numpy(), rtol=1e-5, atol=1e-6)
        assert cam.tanfovx == math.tan(c.fovx * 0.5) and cam.tanfovy == math.tan(c.fovy * 0.5)
        # image size: utils/camera_utils.py:20-40 with --resolution -1 (width capped at 1600, int() of both extents)
        if it.width > 1600:
            down = it.width / 1600
            assert (cam.image_width, cam.image_height) == (int(it.width / down), int(it.height / down))
        else:
            assert (cam.image_width, cam.image_height) == (it.width, it.height)


def test_render_size_rules():
    assert colmap_io.render_size(4946, 3286) == (1600, 1063)       # bicycle: SURVEY.md section 8 (cfg5)
    assert colmap_io.render_size(5187, 3361, 4) == (1297, 840)
    assert colmap_io.render_size(5187, 3361, 1920) == (1920, 1244)
    assert colmap_io.render_size(1296, 840) == (1296, 840)


def test_unsupported_camera_model_is_refused(tmp_path):
    root = str(tmp_path / "m")
    os.makedirs(root + "/sparse/0")
    open(root + "/sparse/0/cameras.txt", "w").write("1 OPENCV 100 80 90 90 50 40 0 0 0 0\n")
    open(root + "/sparse/0/images.txt", "w").write("1 1 0 0 0 0 0 0 1 a.png\n\n")
    with pytest.raises(ValueError, match="only undistorted datasets"):
        colmap_io.read_colmap_cameras(root)


def test_load_3dgs_scene_applies_the_models_activations(tmp_path):
    """ply -> rasterizer inputs: exp / normalise / sigmoid as GaussianModel's getters (scene/gaussian_model.py:74-96)."""
    from seganygaussians_amd import ply_io
    rng = np.random.default_rng(3)
    P = 50
    names = (["x", "y", "z", "nx", "ny", "nz"] + [f"f_dc_{i}" for i in range(3)] + [f"f_rest_{i}" for i in range(45)] + ["opacity"] +
             [f"scale_{i}" for i in range(3)] + [f"rot_{i}" for i in range(4)])
    cols = rng.normal(0, 1, (P, len(names))).astype(np.float32)
    path = str(tmp_path / "point_cloud.ply")
    ply_io.write_vertex_ply(path, names, cols)
    sc = colmap_io.load_3dgs_scene(path, 32, seed=1)
    col = {n: cols[:, i] for i, n in enumerate(names)}
    np.testing.assert_array_equal(sc.means3D, cols[:, :3])
    np.testing.assert_allclose(sc.scales, np.exp(np.stack([col[f"scale_{i}"] for i in range(3)], 1)), rtol=1e-6)
    np.testing.assert_allclose(sc.opacities[:, 0], 1 / (1 + np.exp(-col["opacity"].astype(np.float64))), rtol=1e-6)
    q = np.stack([col[f"rot_{i}"] for i in range(4)], 1).astype(np.float64)
    np.testing.assert_allclose(sc.rotations, q / np.linalg.norm(q, axis=1, keepdims=True), rtol=1e-6)
    assert sc.shs.shape == (P, 16, 3) and sc.features.shape == (P, 32)
    np.testing.assert_array_equal(sc.shs[:, 0, :], np.stack([col[f"f_dc_{i}"] for i in range(3)], 1))
    np.testing.assert_array_equal(sc.shs[:, 1:, 0], np.stack([col[f"f_rest_{i}"] for i in range(15)], 1))   # channel-major on disk
    np.testing.assert_allclose(np.linalg.norm(sc.features, axis=1), 1.0, rtol=1e-5)


def test_render_size_follows_the_image_file(tmp_path):
    """The reference's loadCam takes `orig_w, orig_h` from the loaded image FILE (utils/camera_utils.py:21), not from the COLMAP camera
    model: with `-i images_4` the render is a quarter of the model's size.  image_size_of reads that size from a PNG / JPEG header
    (no image library), to_camera(image_size=) applies the resolution rule to it."""
    import struct
    import zlib

    def chunk(t, d):
        return struct.pack(">I", len(d)) + t + d + struct.pack(">I", zlib.crc32(t + d))
    png = (b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", 1297, 840, 8, 0, 0, 0, 0))
           + chunk(b"IDAT", zlib.compress(b"\x00" * (1298 * 840))) + chunk(b"IEND", b""))
    (tmp_path / "a.png").write_bytes(png)
    jpg = (b"\xff\xd8" + b"\xff\xe0" + struct.pack(">H", 16) + b"JFIF\x00" + b"\x00" * 9 + b"\xff\xdb" + struct.pack(">H", 4) + b"\x00\x00"
           + b"\xff\xc2" + struct.pack(">H", 11) + b"\x08" + struct.pack(">HH", 3361, 5187) + b"\x01\x01\x11\x00")
    (tmp_path / "b.jpg").write_bytes(jpg)
    assert colmap_io.image_size_of(str(tmp_path / "a.png")) == (1297, 840)
    assert colmap_io.image_size_of(str(tmp_path / "b.jpg")) == (5187, 3361)
    # truncated / frame-less files: the ValueError the docstring promises, not a struct.error from a short read
    whole = (tmp_path / "b.jpg").read_bytes()
    for k, blob in enumerate((whole[:12], whole[:4], b"\xff\xd8\xff\xd9", b"\xff\xd8\xff\xe0\x00", b"\x89PNG\r\n\x1a\n\x00\x00")):
        (tmp_path / f"bad{k}.bin").write_bytes(blob)
        with pytest.raises(ValueError):
            colmap_io.image_size_of(str(tmp_path / f"bad{k}.bin"))
    c = colmap_io.ColmapCamera(name="a.png", width=5187, height=3361, fovx=1.0, fovy=0.7, R=np.eye(3), T=np.zeros(3))
    assert (colmap_io.to_camera(c).image_width, colmap_io.to_camera(c).image_height) == (1600, 1036)           # camera model, capped
    cam = colmap_io.to_camera(c, image_size=(1297, 840))                                                       # images_4: as loaded
    assert (cam.image_width, cam.image_height) == (1297, 840)
