#!/bin/bash
# A/B of backward-blend variants with the profiling build (run through gpurun from the repo root):
#   tools/ab.sh "0 4096 2048"   -> MI_RAST_ABLATE values; prints views/s and the stage times of each
export MI_RAST_LIB=$PWD/seganygaussians_amd/libmi_rast_prof.so
for a in $1; do
  MI_RAST_ABLATE=$a python bench.py --no-cpu-baseline --steps 20 --warmup 5 --sustained-seconds 0 ${2:-} 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('ABLATE=$a', d['value'], d['ms_per_step'], d['config']['stages_ms'])"
done
