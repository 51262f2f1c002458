#!/bin/bash
# usage (through gpurun, from the repository root):  bash tools/evidence_run.sh <tag>      e.g. r06
TAG=${1:-r06}
# end-of-round evidence, ALL IN ONE LEASE (one box): the GPU suite, smoke, the store probe, the randomised parity sweep, rocprofv3 passes of
# the three single-GPU configurations (each ends with an un-profiled bench line of the same build on the same box), their summaries
# (so that the bench lines below carry this build's counted traffic), then the bench lines.  The summaries are written on the box's copy of
# profiles/; the raw passes come back under gpurun_out/ and `python tools/summarize_profiles.py <tag> <cfg>` reproduces them here.
mkdir -p gpurun_out
python -m pytest tests -m gpu -q -s > gpurun_out/${TAG}_final_tests.log 2>&1
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/${TAG}_final_smoke.log 2>&1
[ -x tools/write_combine_probe.bin ] && tools/write_combine_probe.bin > gpurun_out/${TAG}_write_combine.log 2>&1
timeout 600 python tools/fuzz_parity.py 120 2027 > gpurun_out/${TAG}_fuzz.log 2>&1
for cfg in cfg3 cfg2 cfg5; do
  bash tools/collect_profiles.sh $TAG $cfg > gpurun_out/${TAG}_collect_$cfg.log 2>&1
  python tools/summarize_profiles.py $TAG $cfg > gpurun_out/${TAG}_summarize_$cfg.log 2>&1
done
bash tools/final_lines.sh $TAG > gpurun_out/${TAG}_final_lines.log 2>&1
tail -3 gpurun_out/${TAG}_final_tests.log; tail -2 gpurun_out/${TAG}_final_smoke.log; tail -2 gpurun_out/${TAG}_fuzz.log; tail -2 gpurun_out/${TAG}_write_combine.log 2>/dev/null
