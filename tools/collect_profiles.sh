#!/bin/bash
# Collects the round's rocprofv3 evidence on the GPU box (run through gpurun from the repo root):
#   tools/collect_profiles.sh <tag> [cfg3|cfg2|cfg5]
#   1. kernel-trace + stats of the bench command of that configuration
#   2. separate PMC passes (FETCH_SIZE; WRITE_SIZE; two SQ groups) restricted to the hot kernels
# Results land in gpurun_out/prof_<tag>[_<cfg>]/ ; tools/summarize_profiles.py <tag> <cfg> turns them into profiles/<tag>_*.
TAG=${1:-r01}
CFG=${2:-cfg3}
SFX=""; [ "$CFG" != "cfg3" ] && SFX="_$CFG"
REPO=$(pwd)
OUT=$REPO/gpurun_out/prof_$TAG$SFX
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
BENCH="python $REPO/bench.py --config $CFG --steps 10 --warmup 2 --settle 0 --dist-blocks 0 --sustained-seconds 0 --no-cpu-baseline"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o trace -- $BENCH > $OUT/trace.log 2>&1
tail -1 $OUT/trace.log > $OUT/bench_line_profiled.json
rocprofv3 --kernel-trace --pmc FETCH_SIZE --kernel-include-regex "mirast" --output-format csv -d $OUT -o fetch -- $BENCH > $OUT/fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --kernel-include-regex "mirast" --output-format csv -d $OUT -o write -- $BENCH > $OUT/write.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY --kernel-include-regex "blend|bin_spans|tile_sort" --output-format csv -d $OUT -o sq1 -- $BENCH > $OUT/sq1.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY --kernel-include-regex "blend|bin_spans|tile_sort" --output-format csv -d $OUT -o sq2 -- $BENCH > $OUT/sq2.log 2>&1
cd $REPO && python bench.py --config $CFG --steps 20 --warmup 3 > $OUT/bench_line.json 2> $OUT/bench.err
ls $OUT
