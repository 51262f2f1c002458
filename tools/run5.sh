#!/bin/bash
mkdir -p gpurun_out
for v in default refine; do
  lib=$PWD/seganygaussians_amd/libmi_rast_$v.so; [ "$v" = "default" ] && lib=$PWD/seganygaussians_amd/libmi_rast.so
  MI_RAST_LIB=$lib python tools/grad_noise.py cfg3 2 2>&1 | grep -v "amdgpu.ids"
done > gpurun_out/r3_noise2.log 2>&1
MI_RAST_LIB=$PWD/seganygaussians_amd/libmi_rast.so python tools/grad_noise.py cfg5 1 2>&1 | grep -v "amdgpu.ids" >> gpurun_out/r3_noise2.log
cat gpurun_out/r3_noise2.log
