# Bench lines of the current build for profiles/<tag>_bench_lines.md (run on the GPU box through gpurun).
TAG=${1:-r02f}
OUT=gpurun_out/lines_$TAG
mkdir -p $OUT
python bench.py > $OUT/cfg3_default.log 2>&1
python bench.py --ref-on-gpu --steps 30 > $OUT/cfg3.log 2>&1
python bench.py --config cfg2 > $OUT/cfg2.log 2>&1
python bench.py --config cfg3s > $OUT/cfg3s.log 2>&1
python bench.py --config cfg5 --ref-on-gpu > $OUT/cfg5.log 2>&1
python bench.py --config cfg1 > $OUT/cfg1.log 2>&1
python bench.py --dist-single --steps 10 --warmup 3 > $OUT/dist.log 2>&1
python bench.py --dist-single --rs-ag --steps 10 --warmup 3 > $OUT/dist_rsag.log 2>&1
python bench.py --frozen-geometry --features-only-grad --views 16 --no-cpu-baseline > $OUT/frozen.log 2>&1
python bench.py --config cfg5 --frozen-geometry --features-only-grad --views 8 --no-cpu-baseline > $OUT/frozen_cfg5.log 2>&1
[ -n "$WITH_FASTEXP" ] && python bench.py --fast-exp --no-cpu-baseline > $OUT/fastexp.log 2>&1
