"""COLMAP camera files -> the cameras the rasterizer is called with (formats at the hot path's edges).

What the reference does between a 360_v2 scene directory and `GaussianRasterizationSettings` (restated here with numpy only, so
that `bench.py --ply <3dgs.ply> --cameras <scene dir>` runs the headline on garden / bicycle where the data exists):

* `sparse/0/{cameras,images}.bin` (or `.txt`) in COLMAP's published model format -- scene/colmap_loader.py:144-259;
* per registered image (scene/dataset_readers.py:73-104): R = qvec2rotmat(qvec)^T, T = tvec, FovX / FovY from the focal lengths
  of a SIMPLE_PINHOLE / PINHOLE / SIMPLE_RADIAL camera (others are refused there as well); cameras sorted by image name
  (dataset_readers.py:169);
* image size (utils/camera_utils.py:20-40, `--resolution -1`): the camera's own size, scaled down to 1600 pixels of width when
  wider (int() of both extents);
* matrices (scene/cameras.py:56-66, utils/graphics_utils.py:38-98): seganygaussians_amd.scenes.camera_from_fov.

Host-only; nothing here touches the GPU.
"""
from __future__ import annotations

import math
import os
import struct
from dataclasses import dataclass
from typing import Dict, List, Optional

import numpy as np

from . import scenes

# model id -> (name, number of parameters)  (COLMAP src/base/camera_models.h; colmap_loader.py:24-36)
_MODELS = {0: ("SIMPLE_PINHOLE", 3), 1: ("PINHOLE", 4), 2: ("SIMPLE_RADIAL", 4), 3: ("RADIAL", 5), 4: ("OPENCV", 8),
           5: ("OPENCV_FISHEYE", 8), 6: ("FULL_OPENCV", 12), 7: ("FOV", 5), 8: ("SIMPLE_RADIAL_FISHEYE", 4),
           9: ("RADIAL_FISHEYE", 5), 10: ("THIN_PRISM_FISHEYE", 12)}


@dataclass
class ColmapCamera:
    name: str            # image file name
    width: int           # of the camera model (the full-resolution image)
    height: int
    fovx: float
    fovy: float
    R: np.ndarray        # (3,3) camera-to-world rotation, as scene/cameras.py stores it
    T: np.ndarray        # (3,)  world-to-camera translation


def qvec2rotmat(q) -> np.ndarray:
    """COLMAP quaternion (w, x, y, z) -> rotation matrix (colmap_loader.py:43-53)."""
    w, x, y, z = [float(v) for v in q]
    return np.array([[1 - 2 * y * y - 2 * z * z, 2 * x * y - 2 * w * z, 2 * z * x + 2 * w * y],
                     [2 * x * y + 2 * w * z, 1 - 2 * x * x - 2 * z * z, 2 * y * z - 2 * w * x],
                     [2 * z * x - 2 * w * y, 2 * y * z + 2 * w * x, 1 - 2 * x * x - 2 * y * y]], np.float64)


def focal2fov(focal: float, pixels: float) -> float:
    return 2 * math.atan(pixels / (2 * focal))   # utils/graphics_utils.py:100-101


def _read_intrinsics_bin(path) -> Dict[int, tuple]:
    out = {}
    with open(path, "rb") as f:
        (n,) = struct.unpack("<Q", f.read(8))
        for _ in range(n):
            cam_id, model_id, w, h = struct.unpack("<iiQQ", f.read(24))
            name, npar = _MODELS[model_id]
            params = struct.unpack("<" + "d" * npar, f.read(8 * npar))
            out[cam_id] = (name, int(w), int(h), params)
    return out


def _read_extrinsics_bin(path) -> List[tuple]:
    out = []
    with open(path, "rb") as f:
        (n,) = struct.unpack("<Q", f.read(8))
        for _ in range(n):
            vals = struct.unpack("<idddddddi", f.read(64))
            name = b""
            while True:
                c = f.read(1)
                if c == b"\x00" or not c:
                    break
                name += c
            (npts,) = struct.unpack("<Q", f.read(8))
            f.seek(24 * npts, os.SEEK_CUR)   # (x, y, point3D id) per observation: not needed
            out.append((vals[0], vals[1:5], vals[5:8], vals[8], name.decode("utf-8")))
    return out


def _read_intrinsics_txt(path) -> Dict[int, tuple]:
    out = {}
    with open(path) as f:
        for line in f:
            line = line.strip()
            if not line or line.startswith("#"):
                continue
            tok = line.split()
            out[int(tok[0])] = (tok[1], int(tok[2]), int(tok[3]), tuple(float(v) for v in tok[4:]))
    return out


def _read_extrinsics_txt(path) -> List[tuple]:
    out = []
    with open(path) as f:
        lines = [l.rstrip("\n") for l in f]
    i = 0
    while i < len(lines):
        line = lines[i].strip()
        i += 1
        if not line or line.startswith("#"):
            continue
        tok = line.split()
        out.append((int(tok[0]), tuple(float(v) for v in tok[1:5]), tuple(float(v) for v in tok[5:8]), int(tok[8]), tok[9]))
        i += 1   # the image's 2D points line (may be empty)
    return out


def read_colmap_cameras(scene_dir: str) -> List[ColmapCamera]:
    """Every registered image of `<scene_dir>/sparse/0` (or of `scene_dir` itself when it holds the model files), sorted by
    image name like the reference's training camera list."""
    model = os.path.join(scene_dir, "sparse", "0")
    if not os.path.isdir(model):
        model = scene_dir
    if os.path.exists(os.path.join(model, "images.bin")):
        extr = _read_extrinsics_bin(os.path.join(model, "images.bin"))
        intr = _read_intrinsics_bin(os.path.join(model, "cameras.bin"))
    elif os.path.exists(os.path.join(model, "images.txt")):
        extr = _read_extrinsics_txt(os.path.join(model, "images.txt"))
        intr = _read_intrinsics_txt(os.path.join(model, "cameras.txt"))
    else:
        raise FileNotFoundError(f"no COLMAP model (images.bin / images.txt) under {model}")
    cams = []
    for _img_id, qvec, tvec, cam_id, name in extr:
        model_name, w, h, params = intr[cam_id]
        if model_name in ("SIMPLE_PINHOLE", "SIMPLE_RADIAL"):
            fx = fy = params[0]
        elif model_name == "PINHOLE":
            fx, fy = params[0], params[1]
        else:
            raise ValueError(f"Colmap camera model {model_name} not handled: only undistorted datasets (PINHOLE or "
                             f"SIMPLE_PINHOLE cameras) supported!")
        cams.append(ColmapCamera(os.path.basename(name), w, h, focal2fov(fx, w), focal2fov(fy, h),
                                 np.transpose(qvec2rotmat(qvec)), np.asarray(tvec, np.float64)))
    cams.sort(key=lambda c: c.name.split(".")[0])
    return cams


def render_size(width: int, height: int, resolution: float = -1) -> tuple:
    """loadCam's image size (utils/camera_utils.py:20-40): resolution 1 / 2 / 4 / 8 divides and rounds; -1 keeps the size but caps the
    width at 1600; any other value is the target width."""
    if resolution in (1, 2, 4, 8):
        return round(width / resolution), round(height / resolution)
    down = (width / 1600 if width > 1600 else 1) if resolution == -1 else width / resolution
    return int(width / float(down)), int(height / float(down))


def image_size_of(path: str) -> tuple:
    """(width, height) of a PNG or JPEG file from its header (no image library needed) -- what PIL's Image.size gives the reference's
    loadCam (utils/camera_utils.py:21: `orig_w, orig_h = cam_info.image.size`)."""
    import struct
    with open(path, "rb") as f:
        head = f.read(26)
        if head[:8] == b"\x89PNG\r\n\x1a\n":
            if len(head) < 24:
                raise ValueError(f"{path}: truncated PNG header")
            return struct.unpack(">II", head[16:24])
        if head[:2] != b"\xff\xd8":
            raise ValueError(f"{path}: neither PNG nor JPEG")
        f.seek(2)
        while True:
            b = f.read(1)
            while b and b != b"\xff":
                b = f.read(1)
            while b == b"\xff":
                b = f.read(1)
            if not b:
                raise ValueError(f"{path}: no JPEG frame header")
            m = b[0]
            if 0xC0 <= m <= 0xCF and m not in (0xC4, 0xC8, 0xCC):   # SOFn
                seg = f.read(7)
                if len(seg) < 7:   # truncated file
                    raise ValueError(f"{path}: no JPEG frame header")
                h, w = struct.unpack(">HH", seg[3:7])
                return w, h
            if m in (0xD8, 0x01) or 0xD0 <= m <= 0xD7:
                continue
            if m in (0xD9, 0xDA):   # end of image / start of scan before any frame header
                raise ValueError(f"{path}: no JPEG frame header")
            ln = f.read(2)
            if len(ln) < 2 or struct.unpack(">H", ln)[0] < 2:
                raise ValueError(f"{path}: no JPEG frame header")
            f.seek(struct.unpack(">H", ln)[0] - 2, 1)


def to_camera(c: ColmapCamera, resolution: float = -1, image_size: Optional[tuple] = None) -> scenes.Camera:
    """`image_size`: (width, height) of the image FILE the reference would load for this camera.  Its loadCam sizes the render from
    the loaded image (utils/camera_utils.py:21), not from the COLMAP camera model: a 360_v2 run with `-i images_4` renders a quarter
    of the model's size (before --resolution applies).  Without it the camera model's size is used, which is the reference's size
    only for the full-resolution `images` folder."""
    w0, h0 = image_size if image_size is not None else (c.width, c.height)
    w, h = render_size(w0, h0, resolution)
    return scenes.camera_from_fov(w, h, c.fovx, c.fovy, c.R, c.T)


def load_3dgs_scene(ply_path: str, channels: int, seed: int = 0, feature_ply: Optional[str] = None) -> scenes.Scene:
    """A trained 3DGS point cloud as rasterizer inputs: activations applied as the model's getters do (scene/gaussian_model.py:
    74-96: exp scales, normalised rotations, sigmoid opacities), SH coefficients as (P, 16, 3).  Feature rows: those of
    `feature_ply` (a FeatureGaussianModel.save_ply file) when given, else seeded N(0,1) rows, L2-normalised -- the reference
    starts training from zeros (scene/gaussian_model_ff.py:665), which exercises nothing."""
    from . import ply_io
    g = ply_io.load_3dgs_ply(ply_path)
    P = g["xyz"].shape[0]
    rot = g["rotation"].astype(np.float64)
    rot /= np.maximum(np.linalg.norm(rot, axis=1, keepdims=True), 1e-12)
    shs = np.concatenate([g["features_dc"], g["features_rest"]], axis=1).astype(np.float32)
    if feature_ply:
        feats = ply_io.load_feature_ply(feature_ply, channels)["point_features"]
        if feats.shape[0] != P:
            raise ValueError(f"{feature_ply} holds {feats.shape[0]} rows, {ply_path} {P}")
    else:
        f = np.random.default_rng(seed).normal(0, 1, (P, channels))
        feats = (f / (np.linalg.norm(f, axis=1, keepdims=True) + 1e-9)).astype(np.float32)
    return scenes.Scene(means3D=g["xyz"], scales=np.exp(g["scaling"]).astype(np.float32), rotations=rot.astype(np.float32),
                        opacities=(1.0 / (1.0 + np.exp(-g["opacity"].astype(np.float64)))).astype(np.float32),
                        features=feats, shs=shs)
