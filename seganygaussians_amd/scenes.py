"""Seeded synthetic scenes for parity tests and bench.py.

No real scene data exists offline, so every benchmark / parity input is generated here with
numpy's default_rng.  Camera conventions follow the reference exactly:

* world->view matrix is stored transposed, i.e. the flat buffer is column-major for the kernels
  (scene/cameras.py:62, CF/cuda_rasterizer/auxiliary.h:58-77);
* projection follows getProjectionMatrix (utils/graphics_utils.py:51-98) with znear=0.01,
  zfar=100 (scene/cameras.py:56-57), also stored transposed; full_proj = view_T @ proj_T
  (scene/cameras.py:64).

Host-only numpy; nothing here touches the oracle or the GPU.
"""
from __future__ import annotations

import math
from dataclasses import dataclass
from typing import Optional

import numpy as np


@dataclass
class Camera:
    image_width: int
    image_height: int
    tanfovx: float
    tanfovy: float
    viewmatrix: np.ndarray   # (4,4) float32, == world_view_transform (W2C transposed)
    projmatrix: np.ndarray   # (4,4) float32, == full_proj_transform
    campos: np.ndarray       # (3,) float32
    R: Optional[np.ndarray] = None   # (3,3) camera-to-world rotation and T as scene/cameras.py stores them
    T: Optional[np.ndarray] = None


def projection_matrix(znear: float, zfar: float, fovx: float, fovy: float) -> np.ndarray:
    """getProjectionMatrix without principal-point shift (utils/graphics_utils.py:51-98)."""
    t = math.tan(fovy / 2) * znear
    r = math.tan(fovx / 2) * znear
    P = np.zeros((4, 4), np.float32)
    P[0, 0] = 2.0 * znear / (2 * r)
    P[1, 1] = 2.0 * znear / (2 * t)
    P[3, 2] = 1.0
    P[2, 2] = zfar / (zfar - znear)
    P[2, 3] = -(zfar * znear) / (zfar - znear)
    return P


def look_at_camera(width: int, height: int, focal_px: float, R: Optional[np.ndarray] = None,
                   T: Optional[np.ndarray] = None) -> Camera:
    """Camera with world->camera rotation R (3,3, stored like the reference: R is C2W rotation,
    getWorld2View2 uses R.T) and translation T."""
    fovx = 2 * math.atan(width / (2 * focal_px))
    fovy = 2 * math.atan(height / (2 * focal_px))
    Rt = np.zeros((4, 4), np.float64)
    Rt[:3, :3] = np.eye(3) if R is None else np.asarray(R, np.float64).T
    Rt[:3, 3] = np.zeros(3) if T is None else np.asarray(T, np.float64)
    Rt[3, 3] = 1.0
    w2c = np.float32(Rt)
    view_t = np.ascontiguousarray(w2c.T)                      # world_view_transform
    proj_t = np.ascontiguousarray(projection_matrix(0.01, 100.0, fovx, fovy).T)
    full = (view_t @ proj_t).astype(np.float32)               # full_proj_transform
    campos = np.linalg.inv(view_t.astype(np.float64))[3, :3].astype(np.float32)
    return Camera(width, height, math.tan(fovx * 0.5), math.tan(fovy * 0.5), view_t,
                  np.ascontiguousarray(full), campos,
                  np.eye(3) if R is None else np.asarray(R, np.float64), np.zeros(3) if T is None else np.asarray(T, np.float64))


def orbit_camera(width: int, height: int, focal_px: float, angle_rad: float, tilt_rad: float = 0.0) -> Camera:
    """Camera rotated about the world y axis (and tilted about x) around the scene centre at
    z ~ 6.75: used for the 8 view-sharded poses of BASELINE config 4."""
    ca, sa = math.cos(angle_rad), math.sin(angle_rad)
    ct, st = math.cos(tilt_rad), math.sin(tilt_rad)
    Ry = np.array([[ca, 0, sa], [0, 1, 0], [-sa, 0, ca]], np.float64)
    Rx = np.array([[1, 0, 0], [0, ct, -st], [0, st, ct]], np.float64)
    Rc2w = Ry @ Rx
    centre = np.array([0.0, 0.0, 6.75])
    cam_pos = centre - Rc2w @ np.array([0.0, 0.0, 6.75])
    # world->cam: x_c = R^T (x_w - c)
    T = -Rc2w.T @ cam_pos
    return look_at_camera(width, height, focal_px, R=Rc2w, T=T)


@dataclass
class Scene:
    means3D: np.ndarray     # (P,3)
    scales: np.ndarray      # (P,3) already exp-ed
    rotations: np.ndarray   # (P,4) unit quaternions (r,x,y,z)
    opacities: np.ndarray   # (P,1) already sigmoid-ed
    features: Optional[np.ndarray]  # (P,C) colours_precomp
    shs: Optional[np.ndarray]       # (P,16,3)


def make_scene(P: int, width: int, height: int, focal_px: float, channels: int, log_scale_mean: float,
               log_scale_std: float, seed: int = 0, with_shs: bool = False, normalize_features: bool = True,
               z_range=(1.5, 12.0)) -> Scene:
    """Synthetic law of SURVEY.md 8(d): z~U[1.5,12]; x,y = U[-1,1]*1.15*tanfov*z; per-axis
    log-scales N(log_scale_mean, std^2); random unit quaternions; opacity sigmoid(N(0,2^2))."""
    rng = np.random.default_rng(seed)
    tanx = width / (2 * focal_px)
    tany = height / (2 * focal_px)
    z = rng.uniform(z_range[0], z_range[1], P)
    x = rng.uniform(-1, 1, P) * 1.15 * tanx * z
    y = rng.uniform(-1, 1, P) * 1.15 * tany * z
    means = np.stack([x, y, z], 1).astype(np.float32)
    scales = np.exp(rng.normal(log_scale_mean, log_scale_std, (P, 3))).astype(np.float32)
    q = rng.normal(0, 1, (P, 4))
    q /= np.linalg.norm(q, axis=1, keepdims=True)
    opac = (1.0 / (1.0 + np.exp(-rng.normal(0, 2, (P, 1))))).astype(np.float32)
    feats = None
    shs = None
    if with_shs:
        shs = rng.normal(0, 0.3, (P, 16, 3)).astype(np.float32)
    else:
        if channels == 3:
            feats = rng.uniform(0, 1, (P, 3)).astype(np.float32)
        else:
            f = rng.normal(0, 1, (P, channels))
            if normalize_features:
                f /= (np.linalg.norm(f, axis=1, keepdims=True) + 1e-9)
            feats = f.astype(np.float32)
    return Scene(means, scales, q.astype(np.float32), opac, feats, shs)


def make_grad_image(channels: int, height: int, width: int, seed: int = 1) -> np.ndarray:
    """dL/dout: iid N(0,1)/(W*H), CHW."""
    rng = np.random.default_rng(seed)
    return (rng.normal(0, 1, (channels, height, width)) / (width * height)).astype(np.float32)


# BASELINE.json configs.  cfg 4 == cfg 3 with 8 orbit poses.
CONFIGS = {
    "cfg1": dict(P=10_000, W=256, H=256, focal=220.0, C=3, ls_mean=math.log(0.05), ls_std=0.6),
    "cfg2": dict(P=1_000_000, W=1920, H=1080, focal=1421.0, C=3, ls_mean=math.log(0.01), ls_std=0.8),
    "cfg3": dict(P=1_000_000, W=1920, H=1080, focal=1421.0, C=32, ls_mean=math.log(0.01), ls_std=0.8),
    "cfg5": dict(P=5_000_000, W=1600, H=1063, focal=1160.0, C=64, ls_mean=math.log(0.008), ls_std=0.8),
}


def make_config(name: str, seed: int = 0, P: Optional[int] = None, with_shs: bool = False):
    """Returns (scene, camera, channels) for a BASELINE config; P may be overridden for
    reduced-size parity cases (same law, fewer Gaussians)."""
    c = CONFIGS[name]
    n = c["P"] if P is None else P
    scene = make_scene(n, c["W"], c["H"], c["focal"], c["C"], c["ls_mean"], c["ls_std"], seed=seed,
                       with_shs=with_shs)
    cam = look_at_camera(c["W"], c["H"], c["focal"])
    return scene, cam, c["C"]
