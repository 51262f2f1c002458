#!/bin/bash
# A/B timings of the lean count / emit passes with the profiling library (MI_RAST_ABLATE_FWD masks, csrc/binning.h: bin_spans_kernel); wrong results
#   1<<20: no (Gaussian, row) items at all (fixed cost: counter / cursor initialisation, record loads, rect clipping)
#   1<<16: no level 2 (tiles of the spans)      1<<17: level 2 without the cursor atomic and the store      1<<19: without the store only
out=gpurun_out/${1:-abl}; mkdir -p $out
for m in 0 $((1<<19)) $((1<<17)) $((1<<16)) $((1<<20)); do
  MI_RAST_LIB=$PWD/seganygaussians_amd/libmi_rast_prof.so MI_RAST_ABLATE_FWD=$m timeout 120 python bench.py --config ${CFG:-cfg3} --no-cpu-baseline --steps 20 --warmup 3 --settle 1 --dist-blocks 0 --sustained-seconds 0 2>/dev/null | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read()); s=d['config']['stages_ms']; print('mask',$m,'views/s',d['value'],'scan',s['tile_scan'],'emit',s['emit'],'sort',s['tile_sort'])"
done 2>&1 | tee $out/abl_emit.log
