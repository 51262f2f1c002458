"""ctypes front-end of oracle/_ref/ -- the REFERENCE's own rasterizer core built for the test box (oracle/build_ref.py).

TEST INFRASTRUCTURE ONLY: importable from tests/, tools/ and bench.py's reporting legs; the product package never
imports this module.  It runs on the GPU (the reference has no CPU path), uses torch only for device memory, and returns
the same `ForwardOut` / `BackwardOut` bundles as the CPU oracle (oracle/saga_oracle.py) so that one set of comparison
helpers serves "oracle vs reference" (the pin) and "product vs reference".
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

from . import build_ref
from . import saga_oracle as so

RESIZE_FN = C.CFUNCTYPE(C.c_void_p, C.c_size_t, C.c_void_p)
_libs = {}


def available(variant: str) -> bool:
    return os.path.exists(build_ref.lib_path(variant))


def variant_for(channels: int, with_mask: bool) -> str:
    if with_mask:
        return "depth3"
    return {3: "base3", 8: "cf8", 16: "cf16", 32: "cf32", 40: "cf40", 64: "cf64", 100: "cf100", 128: "cf128"}[channels]


def lib(variant: str):
    if variant in _libs:
        return _libs[variant]
    import torch  # noqa: F401  (binds the process to torch's HIP runtime first, as seganygaussians_amd/_lib.py does)
    path = build_ref.lib_path(variant)
    if not os.path.exists(path):
        raise FileNotFoundError(f"{path} is missing: run `python oracle/build_ref.py` where /root/reference exists")
    L = C.CDLL(path)
    vp, i, f = C.c_void_p, C.c_int, C.c_float
    if variant == "knn":
        L.saga_ref_knn.restype = i
        L.saga_ref_knn.argtypes = [i, vp, vp]
        L.saga_ref_knn_last_error.restype = C.c_char_p
        _libs[variant] = L
        return L
    L.saga_ref_channels.restype = i
    L.saga_ref_is_depth.restype = i
    L.saga_ref_last_error.restype = C.c_char_p
    L.saga_ref_layout.restype = None
    L.saga_ref_layout.argtypes = [C.c_size_t] * 3 + [C.POINTER(C.c_size_t)] * 3
    L.saga_ref_mark_visible.restype = i
    L.saga_ref_mark_visible.argtypes = [i, vp, vp, vp, vp]
    L.saga_ref_forward.restype = i
    L.saga_ref_forward.argtypes = [RESIZE_FN, vp, RESIZE_FN, vp, RESIZE_FN, vp, i, i, i, vp, i, i, vp, vp, vp, vp, vp, vp,
                                   f, vp, vp, vp, vp, vp, f, f, i, vp, vp, vp, vp, i]
    L.saga_ref_backward.restype = i
    L.saga_ref_backward.argtypes = [i, i, i, i, vp, i, i, vp, vp, vp, vp, f, vp, vp, vp, vp, vp, f, f, vp, vp, vp, vp,
                                    vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, i]
    L.saga_ref_mask_forward.restype = i
    L.saga_ref_mask_forward.argtypes = [RESIZE_FN, vp, RESIZE_FN, vp, RESIZE_FN, vp, i, i, vp, i, i, vp, vp, vp, vp, f,
                                        vp, vp, vp, vp, vp, f, f, i, vp, vp, i]
    L.saga_ref_mask_backward.restype = i
    L.saga_ref_mask_backward.argtypes = [i, i, i, vp, i, i, vp, vp, f, vp, vp, vp, vp, vp, f, f, vp, vp, vp, vp, vp, vp, i]
    _libs[variant] = L
    return L


class _Buf:
    def __init__(self, torch, dev):
        self.torch, self.dev = torch, dev
        self.tensor = torch.empty(0, dtype=torch.uint8, device=dev)
        self.cb = RESIZE_FN(self._resize)

    def _resize(self, n, _ctx):
        self.tensor = self.torch.empty(int(n) + 256, dtype=self.torch.uint8, device=self.dev)
        p = self.tensor.data_ptr()
        self.base = (p + 127) // 128 * 128    # saga_ref_layout's offsets assume a 128-byte aligned base
        return self.base


class RefState:
    """Duck-type of saga_oracle.State on top of the reference's three byte buffers."""

    def __init__(self, run):
        self._r = run

    def _read(self, buf, off, count, dtype):
        dt = np.dtype(dtype)
        start = buf.base - buf.tensor.data_ptr() + off
        raw = buf.tensor[start:start + count * dt.itemsize].cpu().numpy()
        return raw.view(dt).copy()

    def field(self, fid):
        r = self._r
        P, N, R, T = r.P, r.W * r.H, r.num_rendered, r.tiles
        g, im, b = r.geom_off, r.img_off, r.bin_off
        G, I, B = r.geom, r.img, r.binning
        if fid == so.F_DEPTHS: return self._read(G, g[0], P, np.float32)
        if fid == so.F_CLAMPED: return self._read(G, g[1], 3 * P, np.uint8)
        if fid == so.F_MEANS2D: return self._read(G, g[3], 2 * P, np.float32)
        if fid == so.F_COV3D: return self._read(G, g[4], 6 * P, np.float32)
        if fid == so.F_CONIC_OPACITY: return self._read(G, g[5], 4 * P, np.float32)
        if fid == so.F_RGB: return self._read(G, g[6], 3 * P, np.float32)
        if fid == so.F_TILES_TOUCHED: return self._read(G, g[7], P, np.uint32)
        if fid == so.F_POINT_OFFSETS: return self._read(G, g[9], P, np.uint32)
        if fid == so.F_FINAL_T: return self._read(I, im[0], N, np.float32)
        if fid == so.F_N_CONTRIB: return self._read(I, im[1], N, np.uint32)
        if fid == so.F_RANGES: return self._read(I, im[2], 2 * T, np.uint32)
        if R == 0:
            return np.zeros(0, so._FIELD_DTYPES[fid])
        if fid == so.F_POINT_LIST: return self._read(B, b[0], R, np.uint32)
        if fid == so.F_VALUES_UNSORTED: return self._read(B, b[1], R, np.uint32)
        if fid == so.F_KEYS_SORTED: return self._read(B, b[2], R, np.uint64)
        if fid == so.F_KEYS_UNSORTED: return self._read(B, b[3], R, np.uint64)
        raise KeyError(fid)

    def counter(self, cid):
        r = self._r
        if cid == so.C_P: return r.P
        if cid == so.C_R: return r.num_rendered
        if cid == so.C_V: return int((r.radii > 0).sum().item())
        if cid == so.C_TILES: return r.tiles
        raise KeyError(cid)


class RefRun:
    """One rasterizer call through the reference's core on the GPU; mirrors tests/helpers.GpuRun."""

    def __init__(self, inp: so.Inputs, variant: str | None = None, device="cuda:0"):
        import torch
        self.torch, self.inp, self.dev = torch, inp, torch.device(device)
        self.with_mask = inp.mask is not None
        self.variant = variant or variant_for(int(inp.channels), self.with_mask)
        self.L = lib(self.variant)
        assert self.L.saga_ref_channels() == int(inp.channels), (self.variant, inp.channels)
        t = lambda a: None if a is None else torch.as_tensor(np.ascontiguousarray(a, np.float32)).to(self.dev)
        P = np.asarray(inp.means3D).reshape(-1, 3).shape[0]
        self.P, self.W, self.H = P, int(inp.image_width), int(inp.image_height)
        self.tiles = ((self.W + 15) // 16) * ((self.H + 15) // 16)
        self.means3D = t(np.asarray(inp.means3D).reshape(-1, 3))
        self.opac = t(np.asarray(inp.opacities).reshape(-1, 1))
        self.shs = t(None if inp.shs is None else np.asarray(inp.shs).reshape(P, -1, 3))
        self.M = 0 if self.shs is None else self.shs.shape[1]
        self.colors, self.scales, self.rots = t(inp.colors_precomp), t(inp.scales), t(inp.rotations)
        self.cov, self.mask = t(inp.cov3D_precomp), t(inp.mask)
        self.view, self.proj, self.campos, self.bg = t(inp.viewmatrix), t(inp.projmatrix), t(inp.campos), t(inp.bg)
        self.num_rendered = 0

    @staticmethod
    def _p(x):
        return None if x is None else x.data_ptr()

    def _layout(self):
        g, im, b = (C.c_size_t * 10)(), (C.c_size_t * 3)(), (C.c_size_t * 5)()
        self.L.saga_ref_layout(self.P, self.W * self.H, max(self.num_rendered, 0), g, im, b)
        self.geom_off, self.img_off, self.bin_off = list(g), list(im), list(b)

    def forward(self, debug=False) -> so.ForwardOut:
        torch, i, p = self.torch, self.inp, self._p
        self.geom, self.binning, self.img = (_Buf(torch, self.dev) for _ in range(3))
        # rasterize_points.cu:68-69: torch::full(0.0) colour, full(0) radii
        self.color = torch.zeros((int(i.channels), self.H, self.W), dtype=torch.float32, device=self.dev)
        self.radii = torch.zeros(self.P, dtype=torch.int32, device=self.dev)
        self.out_mask = torch.zeros((1, self.H, self.W), dtype=torch.float32, device=self.dev) if self.with_mask else None
        self.out_depth = torch.zeros((1, self.H, self.W), dtype=torch.float32, device=self.dev) if self.with_mask else None
        with torch.cuda.device(self.dev):
            n = self.L.saga_ref_forward(
                self.geom.cb, None, self.binning.cb, None, self.img.cb, None, self.P, int(i.sh_degree), self.M,
                p(self.bg), self.W, self.H, p(self.means3D), p(self.shs), p(self.colors), p(self.opac), p(self.mask),
                p(self.scales), float(i.scale_modifier), p(self.rots), p(self.cov), p(self.view), p(self.proj),
                p(self.campos), float(i.tanfovx), float(i.tanfovy), int(bool(i.prefiltered)), p(self.color),
                p(self.out_mask), p(self.out_depth), p(self.radii), int(bool(debug)))
        if n < 0:
            raise RuntimeError(self.L.saga_ref_last_error().decode())
        torch.cuda.synchronize(self.dev)
        self.num_rendered = n
        self._layout()
        self.fwd = so.ForwardOut(color=self.color.cpu().numpy(), radii=self.radii.cpu().numpy(), state=RefState(self), rc=0,
                                 mask=None if self.out_mask is None else self.out_mask.cpu().numpy(),
                                 depth=None if self.out_depth is None else self.out_depth.cpu().numpy(), num_rendered=n)
        return self.fwd

    def _grad_tensors(self):
        torch, P, Cn, M = self.torch, self.P, int(self.inp.channels), self.M
        z = lambda *s: torch.zeros(s, dtype=torch.float32, device=self.dev)
        return dict(dL_dmeans2D=z(P, 3), dL_dconic=z(P, 2, 2), dL_dopacity=z(P, 1), dL_dcolors=z(P, Cn), dL_dmeans3D=z(P, 3),
                    dL_dcov3D=z(P, 6), dL_dsh=z(P, max(M, 1), 3), dL_dscales=z(P, 3), dL_drotations=z(P, 4),
                    dL_dmask=z(P, 1) if self.with_mask else None)

    def backward(self, dL_dout_color, dL_dout_mask=None, debug=False, to_numpy=True):
        torch, i, p = self.torch, self.inp, self._p
        dpix = torch.as_tensor(np.ascontiguousarray(dL_dout_color, np.float32)).to(self.dev)
        dmask = None
        if self.with_mask:
            dmask = torch.as_tensor(np.ascontiguousarray(
                np.zeros((1, self.H, self.W), np.float32) if dL_dout_mask is None else dL_dout_mask, np.float32)).to(self.dev)
        g = self._grad_tensors()
        with torch.cuda.device(self.dev):
            rc = self.L.saga_ref_backward(
                self.P, int(i.sh_degree), self.M, self.num_rendered, p(self.bg), self.W, self.H, p(self.means3D),
                p(self.shs), p(self.colors), p(self.scales), float(i.scale_modifier), p(self.rots), p(self.cov),
                p(self.view), p(self.proj), p(self.campos), float(i.tanfovx), float(i.tanfovy), p(self.radii),
                self.geom.base, self.binning.base, self.img.base, p(dpix), p(dmask), p(g["dL_dmeans2D"]),
                p(g["dL_dconic"]), p(g["dL_dopacity"]), p(g["dL_dmask"]), p(g["dL_dcolors"]), p(g["dL_dmeans3D"]),
                p(g["dL_dcov3D"]), p(g["dL_dsh"]), p(g["dL_dscales"]), p(g["dL_drotations"]), int(bool(debug)))
        if rc != 0:
            raise RuntimeError(self.L.saga_ref_last_error().decode())
        torch.cuda.synchronize(self.dev)
        if not to_numpy:
            return g
        n = {k: (None if v is None else v.cpu().numpy()) for k, v in g.items()}
        n["dL_dsh"] = n["dL_dsh"][:, :self.M]
        if n["dL_dmask"] is not None:
            n["dL_dmask"] = n["dL_dmask"].reshape(-1)
        return so.BackwardOut(**n)

    def mask_forward(self, debug=False) -> so.ForwardOut:
        torch, i, p = self.torch, self.inp, self._p
        self.geom, self.binning, self.img = (_Buf(torch, self.dev) for _ in range(3))
        self.radii = torch.zeros(self.P, dtype=torch.int32, device=self.dev)
        self.out_mask = torch.zeros((1, self.H, self.W), dtype=torch.float32, device=self.dev)
        with torch.cuda.device(self.dev):
            n = self.L.saga_ref_mask_forward(
                self.geom.cb, None, self.binning.cb, None, self.img.cb, None, self.P, int(i.sh_degree), p(self.bg), self.W,
                self.H, p(self.means3D), p(self.opac), p(self.mask), p(self.scales), float(i.scale_modifier), p(self.rots),
                p(self.cov), p(self.view), p(self.proj), p(self.campos), float(i.tanfovx), float(i.tanfovy),
                int(bool(i.prefiltered)), p(self.out_mask), p(self.radii), int(bool(debug)))
        if n < 0:
            raise RuntimeError(self.L.saga_ref_last_error().decode())
        torch.cuda.synchronize(self.dev)
        self.num_rendered = n
        self._layout()
        return so.ForwardOut(color=np.zeros((0, self.H, self.W), np.float32), radii=self.radii.cpu().numpy(),
                             state=RefState(self), rc=0, mask=self.out_mask.cpu().numpy(), num_rendered=n)

    def mask_backward(self, dL_dout_mask, debug=False) -> np.ndarray:
        torch, i, p = self.torch, self.inp, self._p
        d = torch.as_tensor(np.ascontiguousarray(dL_dout_mask, np.float32)).to(self.dev)
        out = torch.zeros(self.P, dtype=torch.float32, device=self.dev)
        with torch.cuda.device(self.dev):
            rc = self.L.saga_ref_mask_backward(
                self.P, int(i.sh_degree), self.num_rendered, p(self.bg), self.W, self.H, p(self.means3D), p(self.scales),
                float(i.scale_modifier), p(self.rots), p(self.cov), p(self.view), p(self.proj), p(self.campos),
                float(i.tanfovx), float(i.tanfovy), p(self.radii), self.geom.base, self.binning.base, self.img.base,
                p(d), p(out), int(bool(debug)))
        if rc != 0:
            raise RuntimeError(self.L.saga_ref_last_error().decode())
        torch.cuda.synchronize(self.dev)
        return out.cpu().numpy()

    def time_fwd_bwd(self, dL_dout_color, steps=5, warmup=2, backward=True):
        """ms per view of the reference's forward (+ backward) incl. its allocations and zero fills, as its torch
        glue performs them (rasterize_points.cu:68-69,151-159).  Reporting only."""
        torch = self.torch
        dpix = torch.as_tensor(np.ascontiguousarray(dL_dout_color, np.float32)).to(self.dev) if backward else None
        i, p = self.inp, self._p
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)

        def step():
            self.geom, self.binning, self.img = (_Buf(torch, self.dev) for _ in range(3))
            self.color = torch.zeros((int(i.channels), self.H, self.W), dtype=torch.float32, device=self.dev)
            self.radii = torch.zeros(self.P, dtype=torch.int32, device=self.dev)
            n = self.L.saga_ref_forward(
                self.geom.cb, None, self.binning.cb, None, self.img.cb, None, self.P, int(i.sh_degree), self.M,
                p(self.bg), self.W, self.H, p(self.means3D), p(self.shs), p(self.colors), p(self.opac), p(self.mask),
                p(self.scales), float(i.scale_modifier), p(self.rots), p(self.cov), p(self.view), p(self.proj),
                p(self.campos), float(i.tanfovx), float(i.tanfovy), 0, p(self.color), p(self.out_mask),
                p(self.out_depth), p(self.radii), 0)
            assert n >= 0
            if backward:
                g = self._grad_tensors()
                self.L.saga_ref_backward(
                    self.P, int(i.sh_degree), self.M, n, p(self.bg), self.W, self.H, p(self.means3D), p(self.shs),
                    p(self.colors), p(self.scales), float(i.scale_modifier), p(self.rots), p(self.cov), p(self.view),
                    p(self.proj), p(self.campos), float(i.tanfovx), float(i.tanfovy), p(self.radii), self.geom.base,
                    self.binning.base, self.img.base, p(dpix), None, p(g["dL_dmeans2D"]), p(g["dL_dconic"]),
                    p(g["dL_dopacity"]), p(g["dL_dmask"]), p(g["dL_dcolors"]), p(g["dL_dmeans3D"]), p(g["dL_dcov3D"]),
                    p(g["dL_dsh"]), p(g["dL_dscales"]), p(g["dL_drotations"]), 0)

        self.out_mask = torch.zeros((1, self.H, self.W), dtype=torch.float32, device=self.dev) if self.with_mask else None
        self.out_depth = torch.zeros((1, self.H, self.W), dtype=torch.float32, device=self.dev) if self.with_mask else None
        with torch.cuda.device(self.dev):
            for _ in range(warmup):
                step()
            torch.cuda.synchronize(self.dev)
            ev0.record()
            for _ in range(steps):
                step()
            ev1.record()
            torch.cuda.synchronize(self.dev)
        return ev0.elapsed_time(ev1) / steps


def mark_visible(inp: so.Inputs, variant="cf32", device="cuda:0") -> np.ndarray:
    import torch
    L = lib(variant)
    dev = torch.device(device)
    t = lambda a: torch.as_tensor(np.ascontiguousarray(a, np.float32)).to(dev)
    m, v, pr = t(np.asarray(inp.means3D).reshape(-1, 3)), t(inp.viewmatrix), t(inp.projmatrix)
    out = torch.zeros(m.shape[0], dtype=torch.bool, device=dev)
    with torch.cuda.device(dev):
        L.saga_ref_mark_visible(m.shape[0], m.data_ptr(), v.data_ptr(), pr.data_ptr(), out.data_ptr())
    torch.cuda.synchronize(dev)
    return out.cpu().numpy()


def knn_mean_dist2(points: np.ndarray, device="cuda:0") -> np.ndarray:
    """distCUDA2 of the reference (spatial.cu:16-25 -> SimpleKNN::knn): mean squared distance to the 3 nearest neighbours."""
    import torch
    L = lib("knn")
    dev = torch.device(device)
    pts = torch.as_tensor(np.ascontiguousarray(points, np.float32)).to(dev)
    out = torch.zeros(pts.shape[0], dtype=torch.float32, device=dev)
    with torch.cuda.device(dev):
        rc = L.saga_ref_knn(pts.shape[0], pts.data_ptr(), out.data_ptr())
    if rc != 0:
        raise RuntimeError(L.saga_ref_knn_last_error().decode())
    torch.cuda.synchronize(dev)
    return out.cpu().numpy()
