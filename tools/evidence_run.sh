#!/bin/bash
# usage (through gpurun, from the repository root):  bash tools/evidence_run.sh <tag>      e.g. r04
TAG=${1:-r04}
# end-of-round evidence: the GPU suite, smoke, rocprofv3 passes of the three single-GPU configurations, the bench lines
mkdir -p gpurun_out
python -m pytest tests -m gpu -q -s > gpurun_out/${TAG}_final_tests.log 2>&1
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/${TAG}_final_smoke.log 2>&1
bash tools/collect_profiles.sh $TAG cfg3 > gpurun_out/${TAG}_collect_cfg3.log 2>&1
[ -n "$WITH_CFG2" ] && bash tools/collect_profiles.sh $TAG cfg2 > gpurun_out/${TAG}_collect_cfg2.log 2>&1
bash tools/collect_profiles.sh $TAG cfg5 > gpurun_out/${TAG}_collect_cfg5.log 2>&1
bash tools/final_lines.sh $TAG > gpurun_out/${TAG}_final_lines.log 2>&1
tail -3 gpurun_out/${TAG}_final_tests.log; tail -2 gpurun_out/${TAG}_final_smoke.log
