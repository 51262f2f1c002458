#!/bin/bash
# round-3 GPU call 1: layout probe, backward A/B, parity at size
mkdir -p gpurun_out
./tools/mfma4_probe.bin > gpurun_out/r3_probe.log 2>&1
tools/abv.sh "base hyb sep default" > gpurun_out/r3_ab1.log 2>&1
python -m pytest tests/test_gpu_parity.py -x -q > gpurun_out/r3_t_parity.log 2>&1
python -m pytest tests/test_zz_reference_pin.py -q -s -k "full_size" > gpurun_out/r3_t_full.log 2>&1
tail -3 gpurun_out/r3_t_parity.log gpurun_out/r3_t_full.log; cat gpurun_out/r3_probe.log gpurun_out/r3_ab1.log
