"""Per-XCD finish times of the two blend kernels (is the run model's balance what the kernel times suggest?).
PROFILING library only (MI_RAST_LIB=seganygaussians_amd/libmi_rast_prof.so): every wave of blend_fwd_wave_kernel / blend_bwd_wave_kernel
stores the 100-MHz clock at its start and end into its own two words (csrc/common.h: MI_XCD_STAMP); per XCD the earliest start and the
latest end are reported.
    MI_RAST_LIB=$PWD/seganygaussians_amd/libmi_rast_prof.so python tools/xcd_stamps.py cfg3 cfg3s"""
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from seganygaussians_amd import _lib, scenes  # noqa: E402
from tests import helpers as hp  # noqa: E402

L = _lib.load()
L.mi_rast_xcd_stamps.restype = C.c_int
L.mi_rast_xcd_stamps.argtypes = [C.POINTER(C.c_uint64), C.c_int]


def stamps(reset=True):
    out = (C.c_uint64 * 24)()
    assert L.mi_rast_xcd_stamps(out, int(reset)) == 0
    a = np.array(list(out), np.float64)
    return a[:8], a[8:16], a[16:]


for cfg in (sys.argv[1:] or ["cfg3", "cfg3s"]):
    inp = hp.inputs_from_config(cfg)
    if scenes.CONFIGS[cfg].get("law") == "surface":   # the second synthetic law (bench.py --config cfg3s): its Gaussians, the same camera
        sc = scenes.scene_of_config(cfg, seed=0, P=scenes.CONFIGS[cfg]["P"])
        inp.means3D, inp.opacities, inp.scales, inp.rotations, inp.colors_precomp = sc.means3D, sc.opacities, sc.scales, sc.rotations, sc.features
    dL = scenes.make_grad_image(inp.channels, inp.image_height, inp.image_width, seed=1)
    g = hp.GpuRun(inp)
    for rep in range(3):   # (the last repetition is reported: the first ones load code objects)
        stamps()
        g.forward(full_lists=False)
        fe, fs, fn = stamps()
        g.backward(dL)
        be, bs, bn = stamps()
    print(f"{cfg}: P = {inp.means3D.shape[0]}, {inp.image_width} x {inp.image_height}, C = {inp.channels}")
    for name, e, s, n in (("blend fwd", fe, fs, fn), ("blend bwd", be, bs, bn)):
        t0 = s[s > 0].min()
        fin = (e - t0) * 1e-2   # microseconds (100-MHz ticks)
        print(f"{cfg} {name}: per-XCD finish (us after the first wave's start) {np.round(fin, 1).tolist()}, waves {n.astype(int).tolist()}; "
              f"spread (max - min) / max = {(fin.max() - fin.min()) / fin.max():.3f}")
