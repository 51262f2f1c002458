"""Gradient noise of the product against the fp64-accumulating oracle next to the reference's own (the statistics
tests/test_zz_reference_pin.py asserts), for A/B runs of library variants:  MI_RAST_LIB=... python tools/grad_noise.py [cfg3] [repeat]
The oracle's and the reference's gradients are cached under /tmp for the following variants of the same gpurun call."""
import os
import pickle
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import saga_oracle as so  # noqa: E402
from oracle import saga_ref as sr  # noqa: E402
from seganygaussians_amd import scenes  # noqa: E402
from tests import helpers as hp  # noqa: E402

cfg = sys.argv[1] if len(sys.argv) > 1 else "cfg3"
rep = int(sys.argv[2]) if len(sys.argv) > 2 else 1
inp = hp.inputs_from_config(cfg)
dL = scenes.make_grad_image(inp.channels, inp.image_height, inp.image_width, seed=1)
cache = f"/tmp/grad_noise_{cfg}_{int(os.environ.get('GRAD_NOISE_F32_PAIRS') is None)}.pkl"
if os.path.exists(cache):
    ob, theirs_list = pickle.load(open(cache, "rb"))
else:
    of = so.forward(inp)
    ob = hp.grads_as_dict(so.backward(inp, of, dL, exact_pairs=os.environ.get('GRAD_NOISE_F32_PAIRS') is None))
    theirs_list = []
    for _ in range(3):
        ref = sr.RefRun(inp, None)
        ref.forward()
        theirs_list.append(hp.error_stats(hp.grads_as_dict(ref.backward(dL, None)), ob))
    pickle.dump((ob, theirs_list), open(cache, "wb"))
tag = os.path.basename(os.environ.get("MI_RAST_LIB", "default"))
for r in range(rep):
    gpu = hp.GpuRun(inp).forward(full_lists=False)
    mine = hp.error_stats(gpu.backward(dL, None), ob)
    for k, s in mine.items():
        tn = [t[k]["norm"] for t in theirs_list]
        tr = [t[k]["row_frac"] for t in theirs_list]
        print(f"{tag} {cfg} {k:14s} norm {s['norm']:.2e} rows {s['row_frac']:.2e} | reference norm {min(tn):.2e}..{max(tn):.2e} rows {min(tr):.2e}..{max(tr):.2e}"
              f" | ratio rows {s['row_frac'] / (np.mean(tr) + 1e-30):.2f} norm {s['norm'] / np.mean(tn):.2f}")
