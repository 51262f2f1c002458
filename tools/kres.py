"""Resource usage (VGPRs, spills, LDS, occupancy) of every kernel of csrc/mi_rast.hip as hipcc reports it for gfx950.
   python tools/kres.py [substring] [-DMI_RAST_PROFILING]"""
import re, subprocess, sys, os
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
flt = [a for a in sys.argv[1:] if not a.startswith("-")]
extra = [a for a in sys.argv[1:] if a.startswith("-")]
cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-munsafe-fp-atomics",
       "-fno-slp-vectorize", "--cuda-device-only", "-c", "-Rpass-analysis=kernel-resource-usage",
       os.path.join(root, "seganygaussians_amd/csrc/mi_rast.hip"), "-o", "/tmp/kres.o"] + extra
out = subprocess.run(cmd, capture_output=True, text=True).stderr
cur, d = None, {}
for l in out.splitlines():
    m = re.search(r"Name: (\S+)", l)
    if m:
        cur = m.group(1); d[cur] = {}; continue
    m = re.search(r"(VGPRs|AGPRs|VGPRs Spill|SGPRs Spill|ScratchSize \[bytes/lane\]|LDS Size \[bytes/block\]|Occupancy \[waves/SIMD\]): (\d+)", l)
    if m and cur:
        d[cur][m.group(1)] = int(m.group(2))
for k, v in d.items():
    n = re.sub(r"^_ZN6mirast\d+", "", k); n = re.sub(r"EEv.*", "", n)
    if not flt or any(f in k for f in flt):
        print(f"{n:48s} vgpr {v.get('VGPRs')} agpr {v.get('AGPRs')} spill {v.get('VGPRs Spill')} scratch {v.get('ScratchSize [bytes/lane]')} "
              f"lds {v.get('LDS Size [bytes/block]')} occ {v.get('Occupancy [waves/SIMD]')}")
