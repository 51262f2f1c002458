"""The wave-level primitives the binning kernels rest on (seganygaussians_amd/csrc/binning.h), checked ON THE GPU against serial
restatements by tools/wave_prims_probe.hip (built by __graft_entry__.build()): the seven-DPP inclusive scan (sum and running
maximum), wave_owner -- the owner search of the count / emit passes' balanced walks: a ballot, one LDS scatter, a DPP scan -- against
a linear search over random item counts, and every lane_xor<M> of the wave-per-tile bitonic sorter.  Round 6's one memory fault was a
wave_owner whose LDS hand-offs had release fences only: the compiler forwarded a lane's own store to its later load across another
lane's store.  This test is what would have caught it before the emit pass scattered entries through memory."""
import os
import subprocess

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_wave_primitives_on_the_gpu():
    probe = os.path.join(ROOT, "tools", "wave_prims_probe.bin")
    assert os.path.exists(probe), "tools/wave_prims_probe.bin missing: run __graft_entry__.build() in the build container"
    out = subprocess.run([probe], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0 and "-> ok" in out.stdout, out.stdout[-2000:] + out.stderr[-2000:]
