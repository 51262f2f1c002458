"""Golden fixtures (tests/golden/*.npz, made by tests/golden/make_golden.py from the independent dense fp64
autograd re-derivation): the CPU oracle must reproduce them (CPU suite), and so must the HIP path through
the C-ABI (GPU suite)."""
import glob
import os

import numpy as np
import pytest

from oracle import saga_oracle as so
from tests import helpers as hp

GOLDEN = sorted(glob.glob(os.path.join(os.path.dirname(__file__), "golden", "*.npz")))
RTOL = 5e-4  # fp32 implementation vs fp64 expectation (geometry gradients go through large multipliers)


def _inputs(z):
    g = lambda k: z[k] if k in z.files else None
    return so.Inputs(means3D=z["means3D"], opacities=z["opacities"], viewmatrix=z["viewmatrix"],
                     projmatrix=z["projmatrix"], campos=z["campos"], bg=z["bg"], image_width=int(z["image_width"]),
                     image_height=int(z["image_height"]), tanfovx=float(z["tanfovx"]), tanfovy=float(z["tanfovy"]),
                     channels=int(z["channels"]), sh_degree=int(z["sh_degree"]), shs=g("shs"),
                     colors_precomp=g("colors_precomp"), scales=z["scales"], rotations=z["rotations"], mask=g("mask"))


def _check(z, color, radii, grads, mask=None, depth=None):
    np.testing.assert_array_equal(np.asarray(radii), z["exp_radii"])
    hp.assert_close("color", color, z["exp_color"], rtol=2e-4)
    if "exp_mask" in z.files:
        hp.assert_close("mask", mask, z["exp_mask"], rtol=2e-4)
        hp.assert_close("depth", depth, z["exp_depth"], rtol=2e-4)
    pairs = [("dL_dmeans3D", "exp_dL_dmeans3D"), ("dL_dopacity", "exp_dL_dopacity"), ("dL_dscales", "exp_dL_dscales"),
             ("dL_drotations", "exp_dL_drotations"), ("dL_dcolors", "exp_dL_dcolors"), ("dL_dsh", "exp_dL_dsh"),
             ("dL_dmask", "exp_dL_dmask")]
    for k, e in pairs:
        if e in z.files:
            got = np.asarray(grads[k], np.float64)
            hp.assert_close(k, got.reshape(z[e].shape), z[e], rtol=RTOL)
    hp.assert_close("dL_dmeans2D", np.asarray(grads["dL_dmeans2D"])[:, :2], z["exp_dL_dmeans2D"][:, :2], rtol=RTOL)


@pytest.mark.parametrize("path", GOLDEN, ids=[os.path.basename(p)[:-4] for p in GOLDEN])
def test_oracle_reproduces_golden(path):
    z = np.load(path)
    inp = _inputs(z)
    fwd = so.forward(inp)
    assert fwd.rc == 0
    bwd = so.backward(inp, fwd, z["dL_dout_color"], z["dL_dout_mask"] if "dL_dout_mask" in z.files else None)
    _check(z, fwd.color, fwd.radii, bwd.__dict__, fwd.mask, fwd.depth)


@pytest.mark.gpu
@pytest.mark.parametrize("path", GOLDEN, ids=[os.path.basename(p)[:-4] for p in GOLDEN])
def test_hip_reproduces_golden(path):
    z = np.load(path)
    inp = _inputs(z)
    gpu = hp.GpuRun(inp).forward()
    dlm = z["dL_dout_mask"][None] if "dL_dout_mask" in z.files else None
    grads = gpu.backward(z["dL_dout_color"], dlm)
    _check(z, gpu.color.cpu().numpy(), gpu.radii.cpu().numpy(), grads,
           None if gpu.out_mask is None else gpu.out_mask.cpu().numpy(),
           None if gpu.out_depth is None else gpu.out_depth.cpu().numpy())


def test_golden_files_present():
    assert len(GOLDEN) >= 3
