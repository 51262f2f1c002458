#!/bin/bash
# A/B of library variants built by tools/build_variants.py (run through gpurun from the repo root):
#   tools/abv.sh "base hyb sep" [extra bench.py args]   -> views/s, ms/step and the stage times of each, twice (box noise)
for rep in 1 2; do
for v in $1; do
  lib=$PWD/seganygaussians_amd/libmi_rast_$v.so
  [ "$v" = "default" ] && lib=$PWD/seganygaussians_amd/libmi_rast.so
  MI_RAST_LIB=$lib python bench.py --no-cpu-baseline --steps 40 --warmup 5 --settle 1 --sustained-seconds 0 ${2:-} 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('variant $v', d['value'], d['ms_per_step'], d['config']['stages_ms'])"
done
done
