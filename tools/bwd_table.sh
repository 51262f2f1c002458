#!/bin/bash
# The backward blend's two limiters removed one at a time AND together (run through gpurun from the repo root
# after `python tools/build_variants.py pbase=-DMI_RAST_PROFILING pboth=-DMI_RAST_PROFILING,-DMI_BWD_SEPMOM=1,-DMI_BWD_HYBRID_EXP=1`):
#   rows    = ALU work: the kernel as shipped | separable moments on 4x4x1 MFMA + hybrid exp (-15 % matrix time, -8 % VALU)
#   columns = MI_RAST_ABLATE: 0 atomics as is | 16384 only the lowest quadrant of a record's mask adds (the request count of a perfect
#             (tile, record) merge, at no cost for the merge) | 192 no gradient atomics at all          (timing only: wrong results)
# Prints one line per cell: blend_bwd stage time (HIP events) of bench.py --config ${1:-cfg3}, twice (box noise).
CFG=${1:-cfg3}
for rep in 1 2; do
for v in pbase pboth; do
for a in 0 16384 192; do
  MI_RAST_LIB=$PWD/seganygaussians_amd/libmi_rast_$v.so MI_RAST_ABLATE=$a python bench.py --config $CFG --no-cpu-baseline --steps 40 --warmup 5 --settle 1 --dist-blocks 0 --sustained-seconds 0 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('bwd_table $CFG variant=$v ablate=$a blend_bwd_ms', d['config']['stages_ms']['blend_bwd'], 'step_ms', d['ms_per_step'])"
done
done
done
