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


def test_ply_io_against_the_plyfile_calls_of_the_reference(tmp_path):
    """The reference writes its PLY files with PlyElement.describe(elements, 'vertex') + PlyData([el]).write(path) on a
    structured all-'f4' array (scene/gaussian_model_ff.py:586-592) and reads them back property by property.  Through the
    plyfile stand-in (tests/plyfile_shim.py: the header and record layout plyfile itself emits) both directions agree with
    ply_io byte for byte."""
    from tests import plyfile_shim as pf
    rng = np.random.default_rng(2)
    P, C = 123, 32
    names = ply_io.feature_attributes(C)
    cols = rng.normal(size=(P, len(names))).astype(np.float32)
    elements = np.empty(P, dtype=[(n, "f4") for n in names])
    elements[:] = list(map(tuple, cols))                        # the reference's own packing idiom
    a, b = str(tmp_path / "ref.ply"), str(tmp_path / "ours.ply")
    pf.PlyData([pf.PlyElement.describe(elements, "vertex")]).write(a)
    ply_io.write_vertex_ply(b, names, cols)
    assert open(a, "rb").read() == open(b, "rb").read()
    got = ply_io.load_feature_ply(a, C)
    np.testing.assert_array_equal(got["point_features"], cols[:, 6:6 + C])
    back = pf.PlyData.read(b)
    assert [p.name for p in back.elements[0].properties] == names and back["vertex"].count == P
    for i, n in enumerate(names):
        np.testing.assert_array_equal(np.asarray(back.elements[0][n]), cols[:, i])
    # storePly's layout (scene/dataset_readers.py:143-160): float positions / normals + uchar colours
    dtype = [("x", "f4"), ("y", "f4"), ("z", "f4"), ("nx", "f4"), ("ny", "f4"), ("nz", "f4"), ("red", "u1"), ("green", "u1"), ("blue", "u1")]
    pts = np.empty(5, dtype=dtype)
    pts[:] = [tuple(list(rng.normal(size=6)) + list(rng.integers(0, 255, 3))) for _ in range(5)]
    c = str(tmp_path / "points3D.ply")
    pf.PlyData([pf.PlyElement.describe(pts, "vertex")]).write(c)
    head = open(c, "rb").read().split(b"end_header\n")[0].decode().splitlines()
    assert head[-3:] == ["property uchar red", "property uchar green", "property uchar blue"]
    p = ply_io.read_vertex_ply(c)
    np.testing.assert_array_equal(p["red"], pts["red"])
    np.testing.assert_array_equal(p["y"], pts["y"])
