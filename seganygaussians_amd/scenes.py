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


def camera_from_fov(width: int, height: int, fovx: float, fovy: float, R: Optional[np.ndarray] = None,
                    T: Optional[np.ndarray] = None) -> Camera:
    """The matrices scene/cameras.py:56-66 builds for a camera with fields of view (fovx, fovy), camera-to-world rotation R
    (getWorld2View2 uses R^T, utils/graphics_utils.py:38-49) and world-to-camera translation T; znear 0.01, zfar 100."""
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


def look_at_camera(width: int, height: int, focal_px: float, R: Optional[np.ndarray] = None,
                   T: Optional[np.ndarray] = None) -> Camera:
    """Camera with world->camera rotation R (3,3, stored like the reference: R is C2W rotation,
    getWorld2View2 uses R.T) and translation T."""
    fovx = 2 * math.atan(width / (2 * focal_px))
    fovy = 2 * math.atan(height / (2 * focal_px))
    return camera_from_fov(width, height, fovx, fovy, R, T)


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


def _quat_from_matrix(R: np.ndarray) -> np.ndarray:
    """(P,3,3) rotation matrices -> (P,4) unit quaternions (r, x, y, z), the layout of build_rotation (utils/general_utils.py:78-99)."""
    m00, m11, m22 = R[:, 0, 0], R[:, 1, 1], R[:, 2, 2]
    q = np.empty((R.shape[0], 4))
    q[:, 0] = np.sqrt(np.maximum(0.0, 1 + m00 + m11 + m22)) / 2
    q[:, 1] = np.copysign(np.sqrt(np.maximum(0.0, 1 + m00 - m11 - m22)) / 2, R[:, 2, 1] - R[:, 1, 2])
    q[:, 2] = np.copysign(np.sqrt(np.maximum(0.0, 1 - m00 + m11 - m22)) / 2, R[:, 0, 2] - R[:, 2, 0])
    q[:, 3] = np.copysign(np.sqrt(np.maximum(0.0, 1 - m00 - m11 + m22)) / 2, R[:, 1, 0] - R[:, 0, 1])
    return q / np.linalg.norm(q, axis=1, keepdims=True)


def make_surface_scene(P: int, width: int, height: int, focal_px: float, channels: int, log_scale_mean: float,
                       log_scale_std: float, seed: int = 0, with_shs: bool = False, normalize_features: bool = True) -> Scene:
    """A second synthetic law, shaped like a TRAINED scene instead of a uniform cloud (the `s` configurations, e.g. cfg3s): most
    Gaussians lie ON a few undulating surfaces as flat, anisotropic, mostly opaque disks (thin axis along the surface normal),
    the rest are faint floaters of the first law.  Pixels saturate behind the first surface they meet, so tile lists are long
    but end early (T < 1e-4, forward.cu:358-362): E / R and the list-length skew differ from make_scene's, which is what the
    XCD run balance and the cull were tuned on.  Counters are reported with every number (bench.py config.counters)."""
    rng = np.random.default_rng(seed)
    tanx = width / (2 * focal_px)
    tany = height / (2 * focal_px)
    n_float = P // 5
    n_surf = P - n_float
    base = np.array([2.5, 4.0, 6.5, 10.0])
    k = rng.choice(4, n_surf, p=[0.35, 0.3, 0.2, 0.15])
    u, v = rng.uniform(-1, 1, n_surf), rng.uniform(-1, 1, n_surf)
    # z = z_k (1 + 0.12 sin(3 pi u + k) cos(2 pi v)) + thin noise; each surface only covers part of the image, so later ones show
    zk = base[k]
    z = zk * (1 + 0.12 * np.sin(3 * np.pi * u + k) * np.cos(2 * np.pi * v)) + rng.normal(0, 0.004, n_surf) * zk
    ext = np.array([0.55, 0.8, 1.0, 1.15])[k]          # half extent of surface k in units of the frustum's half width
    cx = np.array([-0.35, 0.3, 0.0, 0.0])[k]
    x = (cx + ext * u) * tanx * z
    y = (0.1 * (k - 1.5) + ext * v) * tany * z
    means_s = np.stack([x, y, z], 1)
    # surface normal from the height field (finite differences in u, v), thin axis along it
    dzdu = zk * 0.12 * 3 * np.pi * np.cos(3 * np.pi * u + k) * np.cos(2 * np.pi * v)
    dzdv = -zk * 0.12 * 2 * np.pi * np.sin(3 * np.pi * u + k) * np.sin(2 * np.pi * v)
    tu = np.stack([ext * tanx * z, np.zeros(n_surf), dzdu], 1)
    tv = np.stack([np.zeros(n_surf), ext * tany * z, dzdv], 1)
    nrm = np.cross(tu, tv)
    nrm /= np.linalg.norm(nrm, axis=1, keepdims=True)
    t1 = tu / np.linalg.norm(tu, axis=1, keepdims=True)
    t2 = np.cross(nrm, t1)
    spin = rng.uniform(0, 2 * np.pi, n_surf)           # in-plane orientation of the disk's long axis
    a1 = np.cos(spin)[:, None] * t1 + np.sin(spin)[:, None] * t2
    a2 = np.cross(nrm, a1)
    Rm = np.stack([a1, a2, nrm], 2)                    # columns = the Gaussian's axes (build_rotation's R)
    q_s = _quat_from_matrix(Rm)
    s_long = np.exp(rng.normal(log_scale_mean + 0.4, log_scale_std * 0.6, n_surf))
    s_short = s_long * rng.uniform(0.25, 0.8, n_surf)
    s_thin = s_long * rng.uniform(0.03, 0.12, n_surf)
    scales_s = np.stack([s_long, s_short, s_thin], 1)
    opac_s = 1.0 / (1.0 + np.exp(-rng.normal(1.5, 1.5, (n_surf, 1))))
    # floaters: the first law, faint
    zf = rng.uniform(1.5, 12.0, n_float)
    xf = rng.uniform(-1, 1, n_float) * 1.15 * tanx * zf
    yf = rng.uniform(-1, 1, n_float) * 1.15 * tany * zf
    means_f = np.stack([xf, yf, zf], 1)
    scales_f = np.exp(rng.normal(log_scale_mean, log_scale_std, (n_float, 3)))
    q_f = rng.normal(0, 1, (n_float, 4))
    q_f /= np.linalg.norm(q_f, axis=1, keepdims=True)
    opac_f = 1.0 / (1.0 + np.exp(-rng.normal(-2.5, 1.0, (n_float, 1))))
    perm = rng.permutation(P)                          # index order carries no structure (as after densification)
    means = np.concatenate([means_s, means_f])[perm].astype(np.float32)
    scales = np.concatenate([scales_s, scales_f])[perm].astype(np.float32)
    quats = np.concatenate([q_s, q_f])[perm].astype(np.float32)
    opac = np.concatenate([opac_s, opac_f])[perm].astype(np.float32)
    feats = None
    shs = None
    if with_shs:
        shs = rng.normal(0, 0.3, (P, 16, 3)).astype(np.float32)
    elif channels == 3:
        feats = rng.uniform(0, 1, (P, 3)).astype(np.float32)
    else:
        f = rng.normal(0, 1, (P, channels))
        if normalize_features:
            f /= (np.linalg.norm(f, axis=1, keepdims=True) + 1e-9)
        feats = f.astype(np.float32)
    return Scene(means, scales, quats, opac, feats, shs)


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
    # the same sizes under the second synthetic law (make_surface_scene): surfaces of flat opaque disks + faint floaters
    "cfg3s": dict(P=1_000_000, W=1920, H=1080, focal=1421.0, C=32, ls_mean=math.log(0.01), ls_std=0.8, law="surface"),
    "cfg5s": dict(P=5_000_000, W=1600, H=1063, focal=1160.0, C=64, ls_mean=math.log(0.008), ls_std=0.8, law="surface"),
}


def scene_of_config(name: str, seed: int = 0, P: Optional[int] = None, with_shs: bool = False) -> Scene:
    c = CONFIGS[name]
    make = make_surface_scene if c.get("law") == "surface" else make_scene
    return make(c["P"] if P is None else P, c["W"], c["H"], c["focal"], c["C"], c["ls_mean"], c["ls_std"], seed=seed, with_shs=with_shs)


def make_config(name: str, seed: int = 0, P: Optional[int] = None, with_shs: bool = False):
    """Returns (scene, camera, channels) for a BASELINE config; P may be overridden for
    reduced-size parity cases (same law, fewer Gaussians)."""
    c = CONFIGS[name]
    n = c["P"] if P is None else P
    scene = scene_of_config(name, seed=seed, P=n, with_shs=with_shs)
    cam = look_at_camera(c["W"], c["H"], c["focal"])
    return scene, cam, c["C"]
