#!/bin/bash
mkdir -p gpurun_out
./tools/lds_atomic_probe.bin > gpurun_out/r3_ldsprobe.log 2>&1
tools/ab.sh "0 64 128 192 8192 16384 0 16384" > gpurun_out/r3_ab2.log 2>&1
cat gpurun_out/r3_ldsprobe.log gpurun_out/r3_ab2.log
