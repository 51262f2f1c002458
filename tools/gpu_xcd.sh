#!/bin/bash
# XCD balance A/B (round 5): backward runs work-balanced (default) vs equal counts (--equal-runs); forward with m = 1 / 2 / 4 runs per XCD
# (profiling library, MI_RAST_FWD_RUNS).  cfg3 = uniform law, cfg3s = density varying over the image.
out=gpurun_out/${1:-xcd}; mkdir -p $out
line() { python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); s=d['config']['stages_ms']; print('$1', d['value'], 'views/s  fwd', s['blend_fwd'], 'bwd', s['blend_bwd'], 'ms/step', d['ms_per_step'])"; }
B="--no-cpu-baseline --steps 30 --warmup 3 --settle 1 --dist-blocks 0 --sustained-seconds 0"
for cfg in cfg3 cfg3s; do
  for rep in 1 2; do
    timeout 200 python bench.py --config $cfg $B 2>$out/err.log | line "$cfg balanced-bwd  "
    timeout 200 python bench.py --config $cfg $B --equal-runs 2>$out/err.log | line "$cfg equal-runs    "
  done
  for m in 1 2 4; do
    MI_RAST_LIB=$PWD/seganygaussians_amd/libmi_rast_prof.so MI_RAST_FWD_RUNS=$m timeout 200 python bench.py --config $cfg $B 2>$out/err.log | line "$cfg prof fwd m=$m "
  done
done 2>&1 | tee $out/xcd.log
