#!/bin/bash
# end-of-round evidence: the GPU suite, smoke, rocprofv3 passes of the three single-GPU configurations, the bench lines
mkdir -p gpurun_out
python -m pytest tests -m gpu -q -s > gpurun_out/r3_final_tests.log 2>&1
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r3_final_smoke.log 2>&1
bash tools/collect_profiles.sh r03 cfg3 > gpurun_out/r3_collect_cfg3.log 2>&1
bash tools/collect_profiles.sh r03 cfg2 > gpurun_out/r3_collect_cfg2.log 2>&1
bash tools/collect_profiles.sh r03 cfg5 > gpurun_out/r3_collect_cfg5.log 2>&1
bash tools/final_lines.sh r03 > gpurun_out/r3_final_lines.log 2>&1
tail -3 gpurun_out/r3_final_tests.log; tail -2 gpurun_out/r3_final_smoke.log
