#!/bin/bash
mkdir -p gpurun_out
python -m pytest tests/test_gpu_parity.py tests/test_gpu_api.py tests/test_zz_reference_callers.py -x -q > gpurun_out/r3_t8.log 2>&1
python -m pytest tests/test_zz_reference_pin.py -x -q >> gpurun_out/r3_t8.log 2>&1
tools/abv.sh "default" "--config cfg2" > gpurun_out/r3_ab8.log 2>&1
tools/abv.sh "default" "--config cfg2 --tile-fwd" >> gpurun_out/r3_ab8.log 2>&1
tools/abv.sh "default" "--config cfg1" >> gpurun_out/r3_ab8.log 2>&1
grep -n "passed\|failed\|FAILED\|Error" gpurun_out/r3_t8.log | head; cat gpurun_out/r3_ab8.log
