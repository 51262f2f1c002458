"""the PLY layouts of scene/gaussian_model_ff.py:552-648 and scene/gaussian_model.py:271-322."""
import numpy as np

from seganygaussians_amd import ply_io


def test_feature_ply_roundtrip_and_header(tmp_path):
    rng = np.random.default_rng(0)
    P, C = 257, 32
    xyz, f = rng.normal(size=(P, 3)).astype(np.float32), rng.normal(size=(P, C)).astype(np.float32)
    op, sc, rot = rng.normal(size=(P, 1)).astype(np.float32), rng.normal(size=(P, 3)).astype(np.float32), rng.normal(size=(P, 4)).astype(np.float32)
    path = str(tmp_path / "point_cloud" / "iteration_10000" / "feature_point_cloud.ply")
    ply_io.save_feature_ply(path, xyz, f, op, sc, rot)
    raw = open(path, "rb").read()
    head = raw[: raw.index(b"end_header\n") + 11].decode().splitlines()
    # the header plyfile writes for PlyElement.describe(elements, 'vertex') with an all-'f4' dtype
    assert head[:3] == ["ply", "format binary_little_endian 1.0", f"element vertex {P}"]
    want = ["x", "y", "z", "nx", "ny", "nz"] + [f"f_{i}" for i in range(C)] + ["opacity", "scale_0", "scale_1", "scale_2", "rot_0", "rot_1", "rot_2", "rot_3"]
    assert head[3:-1] == [f"property float {n}" for n in want]
    assert len(raw) == raw.index(b"end_header\n") + 11 + P * len(want) * 4
    got = ply_io.load_feature_ply(path, C)
    for k, v in dict(xyz=xyz, point_features=f, opacity=op, scaling=sc, rotation=rot).items():
        np.testing.assert_array_equal(got[k], v)
    assert np.all(ply_io.read_vertex_ply(path)["nx"] == 0)


def test_3dgs_ply_layout(tmp_path):
    """f_rest is stored channel-major; the loader must reproduce gaussian_model.py:286-291 + the transposes at :309-310."""
    rng = np.random.default_rng(1)
    P, deg = 40, 3
    nrest = 3 * (deg + 1) ** 2 - 3
    names = (["x", "y", "z", "nx", "ny", "nz"] + [f"f_dc_{i}" for i in range(3)] + [f"f_rest_{i}" for i in range(nrest)] +
             ["opacity"] + [f"scale_{i}" for i in range(3)] + [f"rot_{i}" for i in range(4)])
    cols = rng.normal(size=(P, len(names))).astype(np.float32)
    path = str(tmp_path / "point_cloud.ply")
    ply_io.write_vertex_ply(path, names, cols)
    got = ply_io.load_3dgs_ply(path, deg)
    col = {n: cols[:, i] for i, n in enumerate(names)}
    dc = np.zeros((P, 3, 1), np.float32)
    for c in range(3):
        dc[:, c, 0] = col[f"f_dc_{c}"]
    extra = np.stack([col[f"f_rest_{i}"] for i in range(nrest)], axis=1).reshape(P, 3, (deg + 1) ** 2 - 1)
    np.testing.assert_array_equal(got["features_dc"], dc.transpose(0, 2, 1))
    np.testing.assert_array_equal(got["features_rest"], extra.transpose(0, 2, 1))
    assert got["features_dc"].shape == (P, 1, 3) and got["features_rest"].shape == (P, 15, 3)
    np.testing.assert_array_equal(got["rotation"], cols[:, -4:])


def test_ascii_ply(tmp_path):
    path = str(tmp_path / "a.ply")
    open(path, "w").write("ply\nformat ascii 1.0\ncomment made by hand\nelement vertex 2\nproperty float x\nproperty float y\n"
                          "property float z\nproperty uchar red\nend_header\n0 1 2 255\n3 4 5 7\n")
    p = ply_io.read_vertex_ply(path)
    np.testing.assert_array_equal(p["z"], np.array([2, 5], np.float32))
    assert p["red"].dtype == np.uint8 and p["red"][1] == 7
