import sys, os, math
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from oracle import saga_oracle as so
from seganygaussians_amd import scenes
from tests import helpers as hp
inp = hp.make_inputs(int(os.environ.get("NP", "3000")), 160, 96, 32, seed=3, focal=120.0, log_scale=math.log(0.05), log_scale_std=0.7)
gpu = hp.GpuRun(inp).forward()
fwd = so.forward(inp)
dL = scenes.make_grad_image(32, inp.image_height, inp.image_width, seed=1)
grads = gpu.backward(dL)
bwd = so.backward(inp, fwd, dL)
for k, got in grads.items():
    want = np.asarray(getattr(bwd, k)).reshape(got.shape)
    frac, emax, scale = hp.close_report(got, want)
    print(f"{k:16s} frac_out {frac:.3e} maxerr {emax:.3e} scale {scale:.3e}")
g = grads["dL_dmeans2D"]; w = np.asarray(bwd.dL_dmeans2D)
bad = np.argwhere(np.abs(g - w) > 1e-4 * np.abs(w).max())
print("bad rows", len(bad), bad[:10].tolist())
for i, c in bad[:5]:
    print(i, c, g[i], w[i], "radius", fwd.radii[i])
