# A/B timings of the forward's binning stages with the profiling library (MI_RAST_ABLATE_FWD masks: see csrc/binning.h,
# csrc/mi_rast.hip; MI_RAST_NWG: number of rank slices).  Usage on the GPU box: bash tools/abl_fwd.sh "0 65536" "512 256"
for n in ${2:-512}; do
for m in ${1:-0}; do
  MI_RAST_LIB=$PWD/seganygaussians_amd/libmi_rast_prof.so MI_RAST_NWG=$n MI_RAST_ABLATE_FWD=$m timeout 120 python bench.py --no-cpu-baseline --steps 20 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.readline()); s=d['config']['stages_ms']; print('nwg $n ablate $m', d['value'], s['tile_scan'], s['emit'], s['tile_sort'])"
done
done
