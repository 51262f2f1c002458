REPO=$(pwd); OUT=$REPO/gpurun_out/tl; rm -rf $OUT; mkdir -p $OUT; cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $OUT -o tl -- python $REPO/bench.py --steps 10 --warmup 2 --sustained-seconds 0 --no-cpu-baseline $@ > $OUT/log.txt 2>&1
cd $REPO; f=$(find $OUT -name "*kernel_trace.csv" | head -1); python tools/timeline.py $f
