REPO=$(pwd); OUT=$REPO/gpurun_out/prof_kt; mkdir -p $OUT; cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o trace -- python $REPO/bench.py --steps 10 --warmup 2 --settle 0 --no-cpu-baseline > $OUT/trace.log 2>&1
ls -la $OUT
