"""Builds A/B variants of the C-ABI library: the same sources with extra -D flags, as seganygaussians_amd/libmi_rast_<name>.so
(git-ignored; they travel to the GPU box with the snapshot).  Select one with MI_RAST_LIB=<path> (seganygaussians_amd/_lib.py).

   python tools/build_variants.py base=-DMI_BWD_HYBRID_EXP=0,-DMI_BWD_SEPMOM=0 hyb=-DMI_BWD_SEPMOM=0
"""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root)
from seganygaussians_amd import build as b  # noqa: E402


def one(spec):
    name, _, flags = spec.partition("=")
    out = os.path.join(root, "seganygaussians_amd", f"libmi_rast_{name}.so")
    extra = [f for f in flags.split(",") if f]
    cmd = [b.find_hipcc()] + b.HIPCC_FLAGS + extra + b._hash_flag(extra) + ["-o", out, os.path.join(b.SRC_DIR, "mi_rast.hip")]
    subprocess.check_call(cmd)
    return out


with ThreadPoolExecutor(4) as ex:
    for p in ex.map(one, sys.argv[1:]):
        print("built", p)
