#!/bin/bash
# XCD run model sweep (round 5; profiling library): runs of equal sum(min(list length, cap) + fix) for both blend kernels (m = 0, no walk scan)
out=gpurun_out/${1:-xcd3}; mkdir -p $out
line() { python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); s=d['config']['stages_ms']; print('$1', d['value'], 'views/s  scan', s['tile_scan'], 'fwd', s['blend_fwd'], 'bwd', s['blend_bwd'], 'ms/step', d['ms_per_step'])"; }
B="--no-cpu-baseline --steps 30 --warmup 3 --settle 1 --dist-blocks 0 --sustained-seconds 0"
run() { cfg=$1; m=$2; cap=$3; fix=$4; scan=${5:-0}
  MI_RAST_LIB=$PWD/seganygaussians_amd/libmi_rast_prof.so MI_RAST_FWD_RUNS=$m MI_RAST_RUN_CAP=$cap MI_RAST_RUN_FIX=$fix MI_RAST_BWD_SCAN=$scan \
    timeout 200 python bench.py --config $cfg $B 2>$out/err.log | line "$cfg m=$m cap=$cap fix=$fix scan=$scan"; }
for cfg in cfg3s cfg3; do
  run $cfg 1 0 64; run $cfg 0 384 64; run $cfg 0 512 64; run $cfg 0 768 64; run $cfg 0 1024 64; run $cfg 0 512 128; run $cfg 0 768 128; run $cfg 0 768 32; run $cfg 0 100000 64
done 2>&1 | tee $out/xcd3.log
run cfg5 0 768 64 2>&1 | tee -a $out/xcd3.log
run cfg5s 1 0 64 2>&1 | tee -a $out/xcd3.log
run cfg5s 0 768 64 2>&1 | tee -a $out/xcd3.log
