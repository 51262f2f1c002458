#!/bin/bash
bash tools/collect_profiles.sh r03 cfg3 > gpurun_out/r3_collect_cfg3.log 2>&1
bash tools/collect_profiles.sh r03 cfg2 > gpurun_out/r3_collect_cfg2.log 2>&1
bash tools/collect_profiles.sh r03 cfg5 > gpurun_out/r3_collect_cfg5.log 2>&1
bash tools/final_lines.sh r03 > gpurun_out/r3_final_lines.log 2>&1
ls gpurun_out/prof_r03 gpurun_out/prof_r03_cfg2 gpurun_out/prof_r03_cfg5 gpurun_out/lines_r03
