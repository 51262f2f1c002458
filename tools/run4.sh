#!/bin/bash
mkdir -p gpurun_out
for v in default refine nogroup both; do
  lib=$PWD/seganygaussians_amd/libmi_rast_$v.so; [ "$v" = "default" ] && lib=$PWD/seganygaussians_amd/libmi_rast.so
  MI_RAST_LIB=$lib python tools/grad_noise.py cfg3 2 2>&1 | grep -v "amdgpu.ids"
done > gpurun_out/r3_noise.log 2>&1
tools/abv.sh "default refine nogroup both" > gpurun_out/r3_ab4.log 2>&1
python -m pytest tests/test_zz_reference_training.py -q -x > gpurun_out/r3_t_train.log 2>&1
grep "dL_dopacity\|dL_dmeans2D\|dL_dcolors" gpurun_out/r3_noise.log; cat gpurun_out/r3_ab4.log; tail -30 gpurun_out/r3_t_train.log
