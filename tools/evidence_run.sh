#!/bin/bash
# usage (through gpurun, from the repository root):  bash tools/evidence_run.sh <tag>      e.g. r06
TAG=${1:-r06}
# end-of-round evidence, ALL IN ONE LEASE (one box): the GPU suite, smoke, the wave-primitive and store probes, rocprofv3 passes of the
# three single-GPU configurations (each ends with an un-profiled bench line of the same build on the same box), then the bench lines
mkdir -p gpurun_out
python -m pytest tests -m gpu -q -s > gpurun_out/${TAG}_final_tests.log 2>&1
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/${TAG}_final_smoke.log 2>&1
[ -x tools/wave_prims_probe.bin ] && tools/wave_prims_probe.bin > gpurun_out/${TAG}_wave_prims.log 2>&1
[ -x tools/write_combine_probe.bin ] && tools/write_combine_probe.bin > gpurun_out/${TAG}_write_combine.log 2>&1
bash tools/collect_profiles.sh $TAG cfg3 > gpurun_out/${TAG}_collect_cfg3.log 2>&1
[ -z "$SKIP_CFG2" ] && bash tools/collect_profiles.sh $TAG cfg2 > gpurun_out/${TAG}_collect_cfg2.log 2>&1
bash tools/collect_profiles.sh $TAG cfg5 > gpurun_out/${TAG}_collect_cfg5.log 2>&1
bash tools/final_lines.sh $TAG > gpurun_out/${TAG}_final_lines.log 2>&1
tail -3 gpurun_out/${TAG}_final_tests.log; tail -2 gpurun_out/${TAG}_final_smoke.log; cat gpurun_out/${TAG}_wave_prims.log gpurun_out/${TAG}_write_combine.log 2>/dev/null | tail -8
