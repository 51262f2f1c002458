# Kernel trace + stats of one bench configuration (run through gpurun from the repo root):  bash tools/kt.sh cfg5
CFG=${1:-cfg3}
REPO=$(pwd); OUT=$REPO/gpurun_out/prof_kt_$CFG; mkdir -p $OUT; cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o trace -- python $REPO/bench.py --config $CFG --steps 10 --warmup 2 --settle 0 --sustained-seconds 0 --no-cpu-baseline > $OUT/trace.log 2>&1
ls $OUT
