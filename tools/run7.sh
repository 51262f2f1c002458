#!/bin/bash
mkdir -p gpurun_out
python -m pytest tests/test_gpu_parity.py -x -q > gpurun_out/r3_t7.log 2>&1
python -m pytest tests/test_zz_reference_pin.py -x -q -k "cfg3_product or cfg5 or channel_blocks" >> gpurun_out/r3_t7.log 2>&1
tools/abv.sh "default fw5" > gpurun_out/r3_ab7.log 2>&1
tools/abv.sh "default" "--tile-fwd" >> gpurun_out/r3_ab7.log 2>&1
tools/abv.sh "default fw5" "--config cfg5" >> gpurun_out/r3_ab7.log 2>&1
tools/abv.sh "default" "--config cfg5 --tile-fwd" >> gpurun_out/r3_ab7.log 2>&1
grep -n "passed\|failed\|FAILED\|Error" gpurun_out/r3_t7.log | head; cat gpurun_out/r3_ab7.log
