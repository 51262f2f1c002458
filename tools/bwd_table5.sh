#!/bin/bash
# Round 5: the missing column of profiles/r04_bwd_ablation.md -- the three contractions of the backward blend OFF the f32 matrix pipe.
# Timing proxy (wrong results): the profiling library built with -DMI_BWD_NOMFMA=1 (tools/experiments/blend_bwd_wave_lab.h: every
# v_mfma_f32_16x16x4_f32 replaced by ONE v_fma_f32 on the same operands; operand loads and staging stay alive): 80 x 4 cycles of issue per
# chunk instead of 80 x 32 -- LESS than what the exact bf16 form would issue (~24 v_mfma_f32_32x32x16_bf16 for the six partial products of S
# and dF + the per-chunk 3-way splits of w and u), so a lower bound for it.
#   python tools/build_variants.py pbase=-DMI_RAST_PROFILING pnomfma=-DMI_RAST_PROFILING,-DMI_BWD_NOMFMA=1
#   columns = MI_RAST_ABLATE: 0 atomics as shipped | 16384 zero-cost (tile, record) merge | 192 no gradient atomics
CFG=${1:-cfg3}
for rep in 1 2; do
for v in pbase pnomfma; do
for a in 0 16384 192; do
  MI_RAST_LIB=$PWD/seganygaussians_amd/libmi_rast_$v.so MI_RAST_ABLATE=$a python bench.py --config $CFG --no-cpu-baseline --steps 40 --warmup 5 --settle 1 --dist-blocks 0 --sustained-seconds 0 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('bwd_table5 $CFG variant=$v ablate=$a blend_bwd_ms', d['config']['stages_ms']['blend_bwd'], 'step_ms', d['ms_per_step'])"
done
done
done
