set -e
export TMPDIR=/tmp
O=$PWD/gpurun_out/final
rm -rf $O; mkdir -p $O
python tools/prof_transform.py 1024 5 both 0 > $O/idle0.log 2>&1
python tools/prof_transform.py 1024 5 both 0.5 > $O/idle05.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/ktrace -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline > $O/bench_under_rocprof.json 2> $O/bench_under_rocprof.err
rocprofv3 --pmc FETCH_SIZE -d $O/pmc_fetch --output-format csv -- python3 tools/prof_transform.py 1024 2 both > $O/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE -d $O/pmc_write --output-format csv -- python3 tools/prof_transform.py 1024 2 both > $O/pmc_write.log 2>&1
python tools/pmc_traffic.py 1024 $O/pmc_fetch $O/pmc_write > $O/traffic_1024.json
python bench.py > $O/bench_default.json 2> $O/bench_default.err
find $O -name "*kernel_stats.csv" -exec cp {} $O/kernel_stats.csv \;
find $O -name "*.db" -delete; find $O -name "*kernel_trace.csv" -delete
