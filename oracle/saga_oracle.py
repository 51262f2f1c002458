"""ctypes/numpy front-end of the CPU oracle (oracle/saga_rast_oracle.c).

TEST INFRASTRUCTURE ONLY: importable from tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg.  The product package never imports this module.

PARITY STATUS: pinned against the reference implementation itself (oracle/_ref built by oracle/build_ref.py from the
reference's sources; tests/test_zz_reference_pin.py, GPU) -- see saga_rast_oracle.h.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from dataclasses import dataclass, field
from typing import Optional

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "_build", "libsaga_rast_oracle.so")
_lib = None

F_DEPTHS, F_MEANS2D, F_COV3D, F_CONIC_OPACITY, F_RGB, F_CLAMPED, F_TILES_TOUCHED, F_POINT_OFFSETS, \
    F_KEYS_SORTED, F_POINT_LIST, F_RANGES, F_FINAL_T, F_N_CONTRIB, F_RADII, F_KEYS_UNSORTED, \
    F_VALUES_UNSORTED = range(16)
C_P, C_V, C_R, C_E, C_L, C_PAIRS, C_TILES, C_SORT_BITS = range(8)

_FIELD_DTYPES = {
    F_DEPTHS: np.float32, F_MEANS2D: np.float32, F_COV3D: np.float32, F_CONIC_OPACITY: np.float32,
    F_RGB: np.float32, F_CLAMPED: np.uint8, F_TILES_TOUCHED: np.uint32, F_POINT_OFFSETS: np.uint32,
    F_KEYS_SORTED: np.uint64, F_POINT_LIST: np.uint32, F_RANGES: np.uint32, F_FINAL_T: np.float32,
    F_N_CONTRIB: np.uint32, F_RADII: np.int32, F_KEYS_UNSORTED: np.uint64, F_VALUES_UNSORTED: np.uint32,
}


def build(force: bool = False) -> str:
    """Compile the oracle with gcc (oracle/Makefile)."""
    src = os.path.join(_HERE, "saga_rast_oracle.c")
    hdr = os.path.join(_HERE, "saga_rast_oracle.h")
    stale = (not os.path.exists(_LIB_PATH)
             or os.path.getmtime(_LIB_PATH) < max(os.path.getmtime(src), os.path.getmtime(hdr)))
    if force or stale:
        subprocess.check_call(["make", "-C", _HERE, "-B" if force else "-s"], stdout=subprocess.DEVNULL)
    return _LIB_PATH


_NATIVE = False


def use_native_build() -> bool:
    """the CPU BASELINE (bench.py's cpu_baseline leg) is timed with an `-O3 -march=native` build.  The portable
    library under _build/ is compiled in the build container, whose CPU is not the GPU box's, so this compiles a second copy
    ON THE MACHINE IT RUNS ON (same flags otherwise: no FMA contraction, no fast-math -- the results are the same), keyed by the
    CPU's flag set.  Call before the first oracle call; returns False (and keeps the portable build) if gcc is unavailable."""
    global _NATIVE, _LIB_PATH, _lib
    import hashlib
    try:
        flags = next(l for l in open("/proc/cpuinfo") if l.startswith("flags"))
    except Exception:  # noqa: BLE001
        flags = "unknown"
    tag = hashlib.sha1(flags.encode()).hexdigest()[:10]
    path = os.path.join(_HERE, "_build", f"libsaga_rast_oracle_native_{tag}.so")
    src = os.path.join(_HERE, "saga_rast_oracle.c")
    try:
        if not os.path.exists(path) or os.path.getmtime(path) < os.path.getmtime(src):
            os.makedirs(os.path.dirname(path), exist_ok=True)
            subprocess.check_call(["gcc", "-O3", "-march=native", "-std=c11", "-fPIC", "-ffp-contract=off", "-fno-fast-math", "-fopenmp",
                                   "-shared", "-o", path, src, "-lm"], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    except Exception:  # noqa: BLE001
        return False
    if _lib is not None and not _NATIVE:
        _lib = None
    _LIB_PATH, _NATIVE = path, True
    return True


def lib():
    global _lib
    if _lib is None:
        if not _NATIVE:
            build()
        L = C.CDLL(_LIB_PATH)
        fp, ip, vp = C.POINTER(C.c_float), C.POINTER(C.c_int), C.c_void_p
        L.saga_oracle_forward.restype = vp
        L.saga_oracle_forward.argtypes = [C.c_int] * 4 + [vp, C.c_int, C.c_int] + [vp] * 4 + [vp, C.c_float] + \
            [vp] * 5 + [C.c_float, C.c_float, C.c_int] + [vp] * 5 + [ip]
        L.saga_oracle_mask_forward.restype = vp
        L.saga_oracle_mask_forward.argtypes = [C.c_int] * 3 + [vp] * 4 + [C.c_float] + [vp] * 4 + \
            [C.c_float, C.c_float, C.c_int, vp, vp, ip]
        L.saga_oracle_backward.restype = None
        L.saga_oracle_backward.argtypes = [vp] + [C.c_int] * 4 + [vp, C.c_int, C.c_int] + [vp] * 4 + \
            [C.c_float] + [vp] * 5 + [C.c_float, C.c_float] + [vp] * 13 + [C.c_int]
        L.saga_oracle_mask_backward.restype = None
        L.saga_oracle_mask_backward.argtypes = [vp, C.c_int, C.c_int, C.c_int, vp, vp, C.c_int]
        L.saga_oracle_mark_visible.restype = None
        L.saga_oracle_mark_visible.argtypes = [C.c_int, vp, vp, vp, vp]
        L.saga_oracle_get_higher_msb.restype = C.c_uint32
        L.saga_oracle_get_higher_msb.argtypes = [C.c_uint32]
        L.saga_oracle_field.restype = vp
        L.saga_oracle_field.argtypes = [vp, C.c_int, C.POINTER(C.c_size_t)]
        L.saga_oracle_counter.restype = C.c_int64
        L.saga_oracle_counter.argtypes = [vp, C.c_int]
        L.saga_oracle_free.restype = None
        L.saga_oracle_free.argtypes = [vp]
        L.saga_oracle_num_threads.restype = C.c_int
        L.saga_oracle_set_num_threads.argtypes = [C.c_int]
        _lib = L
    return _lib


def _f32(a, shape=None) -> Optional[np.ndarray]:
    if a is None:
        return None
    if hasattr(a, "detach"):
        a = a.detach().cpu().numpy()
    a = np.ascontiguousarray(a, dtype=np.float32)
    if a.size == 0:
        return None
    if shape is not None:
        a = a.reshape(shape)
    return a


def _ptr(a: Optional[np.ndarray]):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


class State:
    """Owns a saga_oracle_state*; exposes intermediate arrays as numpy copies."""

    def __init__(self, handle):
        self._h = handle

    def field(self, fid: int) -> np.ndarray:
        n = C.c_size_t(0)
        p = lib().saga_oracle_field(self._h, fid, C.byref(n))
        dt = np.dtype(_FIELD_DTYPES[fid])
        if not p or n.value == 0:
            return np.zeros(0, dtype=dt)
        buf = (C.c_char * (n.value * dt.itemsize)).from_address(p)
        return np.frombuffer(buf, dtype=dt).copy()

    def counter(self, cid: int) -> int:
        return int(lib().saga_oracle_counter(self._h, cid))

    def counters(self) -> dict:
        return {k: self.counter(v) for k, v in
                dict(P=C_P, V=C_V, R=C_R, E=C_E, L=C_L, pairs=C_PAIRS, tiles=C_TILES, sort_bits=C_SORT_BITS).items()}

    def __del__(self):
        if getattr(self, "_h", None):
            lib().saga_oracle_free(self._h)
            self._h = None


@dataclass
class Inputs:
    """Host-side bundle of one rasterizer call (mirrors the 19 positional args of
    _C.rasterize_gaussians, CF/diff_gaussian_rasterization_contrastive_f/__init__.py:62-81)."""
    means3D: np.ndarray
    opacities: np.ndarray
    viewmatrix: np.ndarray
    projmatrix: np.ndarray
    campos: np.ndarray
    bg: np.ndarray
    image_width: int
    image_height: int
    tanfovx: float
    tanfovy: float
    channels: int
    scale_modifier: float = 1.0
    sh_degree: int = 0
    shs: Optional[np.ndarray] = None
    colors_precomp: Optional[np.ndarray] = None
    scales: Optional[np.ndarray] = None
    rotations: Optional[np.ndarray] = None
    cov3D_precomp: Optional[np.ndarray] = None
    mask: Optional[np.ndarray] = None
    prefiltered: bool = False


@dataclass
class ForwardOut:
    color: np.ndarray
    radii: np.ndarray
    state: State
    rc: int
    mask: Optional[np.ndarray] = None
    depth: Optional[np.ndarray] = None
    num_rendered: int = 0


def forward(inp: Inputs) -> ForwardOut:
    L = lib()
    means3D = _f32(inp.means3D)
    P = 0 if means3D is None else means3D.reshape(-1, 3).shape[0]
    Cn, W, H = int(inp.channels), int(inp.image_width), int(inp.image_height)
    shs = _f32(inp.shs)
    M = 0 if shs is None else shs.reshape(P, -1, 3).shape[1]
    arrs = dict(bg=_f32(inp.bg), means3D=means3D, shs=shs, colors=_f32(inp.colors_precomp),
                opac=_f32(inp.opacities), scales=_f32(inp.scales), rots=_f32(inp.rotations),
                cov=_f32(inp.cov3D_precomp), view=_f32(inp.viewmatrix), proj=_f32(inp.projmatrix),
                campos=_f32(inp.campos), mask=_f32(inp.mask))
    out_color = np.zeros((Cn, H, W), np.float32)
    has_mask = arrs["mask"] is not None
    out_mask = np.zeros((1, H, W), np.float32) if has_mask else None
    out_depth = np.zeros((1, H, W), np.float32) if has_mask else None
    radii = np.zeros(max(P, 1), np.int32)
    rc = C.c_int(0)
    h = L.saga_oracle_forward(P, int(inp.sh_degree), M, Cn, _ptr(arrs["bg"]), W, H, _ptr(arrs["means3D"]),
                              _ptr(arrs["shs"]), _ptr(arrs["colors"]), _ptr(arrs["opac"]), _ptr(arrs["scales"]),
                              float(inp.scale_modifier), _ptr(arrs["rots"]), _ptr(arrs["cov"]), _ptr(arrs["view"]),
                              _ptr(arrs["proj"]), _ptr(arrs["campos"]), float(inp.tanfovx), float(inp.tanfovy),
                              int(bool(inp.prefiltered)), _ptr(arrs["mask"]), _ptr(out_color), _ptr(out_mask),
                              _ptr(out_depth), _ptr(radii), C.byref(rc))
    st = State(h)
    return ForwardOut(color=out_color, radii=radii[:P], state=st, rc=rc.value, mask=out_mask, depth=out_depth,
                      num_rendered=st.counter(C_R))


@dataclass
class BackwardOut:
    dL_dmeans2D: np.ndarray
    dL_dconic: np.ndarray
    dL_dopacity: np.ndarray
    dL_dcolors: np.ndarray
    dL_dmeans3D: np.ndarray
    dL_dcov3D: np.ndarray
    dL_dsh: np.ndarray
    dL_dscales: np.ndarray
    dL_drotations: np.ndarray
    dL_dmask: Optional[np.ndarray] = None


def backward(inp: Inputs, fwd: ForwardOut, dL_dout_color, dL_dout_mask=None, exact_pairs=False) -> BackwardOut:
    """exact_pairs=False: binary64 sums of per-pair terms rounded to binary32 exactly like the reference's kernel (the parity
    checker).  exact_pairs=True: the per-pair values in binary64 as well (decisions still binary32): the yardstick for
    comparing two implementations' rounding noise -- it shares neither's (saga_rast_oracle.c: render_backward)."""
    L = lib()
    means3D = _f32(inp.means3D)
    P = 0 if means3D is None else means3D.reshape(-1, 3).shape[0]
    Cn, W, H = int(inp.channels), int(inp.image_width), int(inp.image_height)
    shs = _f32(inp.shs)
    M = 0 if shs is None else shs.reshape(P, -1, 3).shape[1]
    a = dict(bg=_f32(inp.bg), means3D=means3D, shs=shs, colors=_f32(inp.colors_precomp),
             scales=_f32(inp.scales), rots=_f32(inp.rotations), cov=_f32(inp.cov3D_precomp),
             view=_f32(inp.viewmatrix), proj=_f32(inp.projmatrix), campos=_f32(inp.campos),
             mask=_f32(inp.mask), dpix=_f32(dL_dout_color), dmask=_f32(dL_dout_mask))
    has_mask = a["mask"] is not None
    if has_mask and a["dmask"] is None:
        a["dmask"] = np.zeros((H, W), np.float32)
    n = max(P, 1)
    o = BackwardOut(
        dL_dmeans2D=np.zeros((n, 3), np.float32), dL_dconic=np.zeros((n, 2, 2), np.float32),
        dL_dopacity=np.zeros((n, 1), np.float32), dL_dcolors=np.zeros((n, Cn), np.float32),
        dL_dmeans3D=np.zeros((n, 3), np.float32), dL_dcov3D=np.zeros((n, 6), np.float32),
        dL_dsh=np.zeros((n, max(M, 1), 3), np.float32), dL_dscales=np.zeros((n, 3), np.float32),
        dL_drotations=np.zeros((n, 4), np.float32),
        dL_dmask=np.zeros((n,), np.float32) if has_mask else None)
    L.saga_oracle_backward(fwd.state._h, P, int(inp.sh_degree), M, Cn, _ptr(a["bg"]), W, H, _ptr(a["means3D"]),
                           _ptr(a["shs"]), _ptr(a["colors"]), _ptr(a["scales"]), float(inp.scale_modifier),
                           _ptr(a["rots"]), _ptr(a["cov"]), _ptr(a["view"]), _ptr(a["proj"]), _ptr(a["campos"]),
                           float(inp.tanfovx), float(inp.tanfovy), _ptr(a["dpix"]), _ptr(a["dmask"]),
                           _ptr(a["mask"]), _ptr(o.dL_dmeans2D), _ptr(o.dL_dconic), _ptr(o.dL_dopacity),
                           _ptr(o.dL_dcolors), _ptr(o.dL_dmask), _ptr(o.dL_dmeans3D), _ptr(o.dL_dcov3D),
                           _ptr(o.dL_dsh), _ptr(o.dL_dscales), _ptr(o.dL_drotations), 2 if exact_pairs else 1)
    for k, v in list(o.__dict__.items()):
        if v is not None:
            setattr(o, k, v[:P] if k != "dL_dsh" else v[:P, :M])
    return o


def mask_forward(inp: Inputs):
    L = lib()
    means3D = _f32(inp.means3D)
    P = means3D.reshape(-1, 3).shape[0]
    W, H = int(inp.image_width), int(inp.image_height)
    a = dict(means3D=means3D, opac=_f32(inp.opacities), mask=_f32(inp.mask), scales=_f32(inp.scales),
             rots=_f32(inp.rotations), cov=_f32(inp.cov3D_precomp), view=_f32(inp.viewmatrix),
             proj=_f32(inp.projmatrix))
    out_mask = np.zeros((1, H, W), np.float32)
    radii = np.zeros(max(P, 1), np.int32)
    rc = C.c_int(0)
    h = L.saga_oracle_mask_forward(P, W, H, _ptr(a["means3D"]), _ptr(a["opac"]), _ptr(a["mask"]), _ptr(a["scales"]),
                                   float(inp.scale_modifier), _ptr(a["rots"]), _ptr(a["cov"]), _ptr(a["view"]),
                                   _ptr(a["proj"]), float(inp.tanfovx), float(inp.tanfovy),
                                   int(bool(inp.prefiltered)), _ptr(out_mask), _ptr(radii), C.byref(rc))
    st = State(h)
    return ForwardOut(color=np.zeros((0, H, W), np.float32), radii=radii[:P], state=st, rc=rc.value, mask=out_mask,
                      num_rendered=st.counter(C_R))


def mask_backward(inp: Inputs, fwd: ForwardOut, dL_dout_mask) -> np.ndarray:
    L = lib()
    P = _f32(inp.means3D).reshape(-1, 3).shape[0]
    d = _f32(dL_dout_mask)
    out = np.zeros(max(P, 1), np.float32)
    L.saga_oracle_mask_backward(fwd.state._h, P, int(inp.image_width), int(inp.image_height), _ptr(d), _ptr(out), 1)
    return out[:P]


def mark_visible(means3D, viewmatrix, projmatrix) -> np.ndarray:
    m = _f32(means3D)
    P = 0 if m is None else m.reshape(-1, 3).shape[0]
    out = np.zeros(max(P, 1), np.uint8)
    if P:
        lib().saga_oracle_mark_visible(P, _ptr(m), _ptr(_f32(viewmatrix)), _ptr(_f32(projmatrix)), _ptr(out))
    return out[:P].astype(bool)


def get_higher_msb(n: int) -> int:
    return int(lib().saga_oracle_get_higher_msb(n))


def num_threads() -> int:
    return int(lib().saga_oracle_num_threads())


def set_num_threads(n: int) -> None:
    lib().saga_oracle_set_num_threads(int(n))
