"""On-disk formats at the hot path's edges: the PLY files SAGA reads and writes.

* feature PLY -- FeatureGaussianModel.save_ply / load_ply (scene/gaussian_model_ff.py:552-648): one `vertex` element,
  all properties float32, in the order  x y z nx ny nz f_0..f_{C-1} opacity scale_0..2 rot_0..3  (normals zero);
* 3DGS PLY -- GaussianModel.load_ply (scene/gaussian_model.py:271-322): x y z nx ny nz f_dc_0..2 f_rest_0..(3(d+1)^2-4)
  opacity scale_* rot_*, with f_rest stored channel-major ((P, 3, (d+1)^2-1) -> transposed to (P, (d+1)^2-1, 3)).

The reference goes through the `plyfile` package (absent here); this module reads and writes the same byte layout
(`format binary_little_endian 1.0`, what plyfile emits on little-endian hosts) with numpy only, and also reads ASCII
PLY.  Arrays are returned as numpy float32 in the shapes the reference's nn.Parameters have; moving them to the GPU
stays with the caller (these files are loaded once, not per iteration)."""
from __future__ import annotations

import os
from typing import Dict, List, Tuple

import numpy as np

_PLY_TYPES = {"char": "i1", "int8": "i1", "uchar": "u1", "uint8": "u1", "short": "i2", "int16": "i2",
              "ushort": "u2", "uint16": "u2", "int": "i4", "int32": "i4", "uint": "u4", "uint32": "u4",
              "float": "f4", "float32": "f4", "double": "f8", "float64": "f8"}


def feature_attributes(feature_dim: int, n_scale: int = 3, n_rot: int = 4) -> List[str]:
    """construct_list_of_attributes (gaussian_model_ff.py:552-565)."""
    return (["x", "y", "z", "nx", "ny", "nz"] + [f"f_{i}" for i in range(feature_dim)] + ["opacity"] +
            [f"scale_{i}" for i in range(n_scale)] + [f"rot_{i}" for i in range(n_rot)])


def write_vertex_ply(path: str, names: List[str], columns: np.ndarray) -> None:
    """One `vertex` element with float32 properties `names`; columns (P, len(names))."""
    columns = np.ascontiguousarray(columns, dtype="<f4")
    if columns.ndim != 2 or columns.shape[1] != len(names):
        raise ValueError(f"columns has shape {columns.shape}, expected (P, {len(names)})")
    d = os.path.dirname(path)
    if d:
        os.makedirs(d, exist_ok=True)
    header = ["ply", "format binary_little_endian 1.0", f"element vertex {columns.shape[0]}"]
    header += [f"property float {n}" for n in names] + ["end_header"]
    with open(path, "wb") as f:
        f.write(("\n".join(header) + "\n").encode("ascii"))
        f.write(columns.tobytes())


def read_vertex_ply(path: str) -> Dict[str, np.ndarray]:
    """Properties of the first element of a PLY file (binary little/big endian or ASCII), by name."""
    with open(path, "rb") as f:
        if f.readline().strip() != b"ply":
            raise ValueError(f"{path}: not a PLY file")
        fmt, count, props, first = None, None, [], True
        while True:
            line = f.readline()
            if not line:
                raise ValueError(f"{path}: header without end_header")
            tok = line.decode("ascii").split()
            if not tok or tok[0] == "comment" or tok[0] == "obj_info":
                continue
            if tok[0] == "format":
                fmt = tok[1]
            elif tok[0] == "element":
                if count is None:
                    count = int(tok[2])
                else:
                    first = False           # later elements are not needed
            elif tok[0] == "property" and first:
                if tok[1] == "list":
                    raise ValueError(f"{path}: list properties in the vertex element are not supported")
                props.append((tok[2], _PLY_TYPES[tok[1]]))
            elif tok[0] == "end_header":
                break
        if fmt is None or count is None:
            raise ValueError(f"{path}: incomplete header")
        if fmt == "ascii":
            data = np.loadtxt(f, max_rows=count, ndmin=2)
            return {n: data[:, i].astype(t) for i, (n, t) in enumerate(props)}
        order = "<" if fmt == "binary_little_endian" else ">"
        dt = np.dtype([(n, order + t) for n, t in props])
        rec = np.frombuffer(f.read(count * dt.itemsize), dtype=dt, count=count)
        return {n: np.ascontiguousarray(rec[n]) for n, _ in props}


def _numbered(props: Dict[str, np.ndarray], prefix: str) -> np.ndarray:
    names = sorted((n for n in props if n.startswith(prefix)), key=lambda x: int(x.split("_")[-1]))
    return np.stack([props[n] for n in names], axis=1).astype(np.float32) if names else np.zeros((len(props["x"]), 0), np.float32)


def save_feature_ply(path: str, xyz, features, opacities, scales, rotations) -> None:
    """FeatureGaussianModel.save_ply (gaussian_model_ff.py:567-595): normals are zeros."""
    xyz = np.asarray(xyz, np.float32)
    cols = np.concatenate([xyz, np.zeros_like(xyz), np.asarray(features, np.float32),
                           np.asarray(opacities, np.float32).reshape(len(xyz), -1), np.asarray(scales, np.float32),
                           np.asarray(rotations, np.float32)], axis=1)
    write_vertex_ply(path, feature_attributes(np.asarray(features).shape[1], np.asarray(scales).shape[1],
                                              np.asarray(rotations).shape[1]), cols)


def load_feature_ply(path: str, feature_dim: int) -> Dict[str, np.ndarray]:
    """FeatureGaussianModel.load_ply (gaussian_model_ff.py:606-646)."""
    p = read_vertex_ply(path)
    feats = _numbered(p, "f_")
    assert feats.shape[1] == feature_dim, (feats.shape[1], feature_dim)
    return dict(xyz=np.stack([p["x"], p["y"], p["z"]], axis=1).astype(np.float32), point_features=feats,
                opacity=p["opacity"].astype(np.float32)[:, None], scaling=_numbered(p, "scale_"),
                rotation=_numbered(p, "rot"))


def load_3dgs_ply(path: str, max_sh_degree: int = 3) -> Dict[str, np.ndarray]:
    """GaussianModel.load_ply (gaussian_model.py:271-322): features_dc (P, 1, 3), features_rest (P, (d+1)^2-1, 3)."""
    p = read_vertex_ply(path)
    P = len(p["x"])
    dc = np.stack([p["f_dc_0"], p["f_dc_1"], p["f_dc_2"]], axis=1).astype(np.float32).reshape(P, 3, 1)
    rest = _numbered(p, "f_rest_")
    assert rest.shape[1] == 3 * (max_sh_degree + 1) ** 2 - 3, rest.shape
    rest = rest.reshape(P, 3, (max_sh_degree + 1) ** 2 - 1)
    return dict(xyz=np.stack([p["x"], p["y"], p["z"]], axis=1).astype(np.float32),
                features_dc=np.ascontiguousarray(dc.transpose(0, 2, 1)),
                features_rest=np.ascontiguousarray(rest.transpose(0, 2, 1)),
                opacity=p["opacity"].astype(np.float32)[:, None], scaling=_numbered(p, "scale_"),
                rotation=_numbered(p, "rot"))
