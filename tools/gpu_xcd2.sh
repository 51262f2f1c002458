#!/bin/bash
# XCD dealing A/B, second matrix (round 5; profiling library, knobs from the environment -- mi_rast.hip: knob()):
#   FWD_RUNS m (0: one run per XCD from the range scan's bounds), RUN_CAP c (0: equal tile counts), BWD_SCAN (1: backward runs from the forward's walks)
out=gpurun_out/${1:-xcd2}; mkdir -p $out
line() { python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); s=d['config']['stages_ms']; print('$1', d['value'], 'views/s  scan', s['tile_scan'], 'fwd', s['blend_fwd'], 'bwd', s['blend_bwd'], 'ms/step', d['ms_per_step'])"; }
B="--no-cpu-baseline --steps 30 --warmup 3 --settle 1 --dist-blocks 0 --sustained-seconds 0"
run() { cfg=$1; m=$2; cap=$3; scan=$4
  MI_RAST_LIB=$PWD/seganygaussians_amd/libmi_rast_prof.so MI_RAST_FWD_RUNS=$m MI_RAST_RUN_CAP=$cap MI_RAST_RUN_FIX=${FIX:-64} MI_RAST_BWD_SCAN=$scan \
    timeout 200 python bench.py --config $cfg $B 2>$out/err.log | line "$cfg m=$m cap=$cap scan=$scan"; }
for cfg in cfg3 cfg3s; do
  run $cfg 1 0 0; run $cfg 1 0 1; run $cfg 4 0 1; run $cfg 0 128 0; run $cfg 0 256 0; run $cfg 0 256 1; run $cfg 4 256 0; run $cfg 0 512 0
done 2>&1 | tee $out/xcd2.log
for cfg in cfg5; do run $cfg 1 0 0; run $cfg 4 0 1; run $cfg 0 256 0; done 2>&1 | tee -a $out/xcd2.log
