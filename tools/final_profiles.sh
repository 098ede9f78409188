#!/bin/bash
# The round's rocprofv3 evidence, taken on the GPU box in one go (gpurun -- 'bash tools/final_profiles.sh rNN'):
#   gpurun_out/<round>_final/kernel: --kernel-trace --memory-copy-trace --stats of a short bench.py run (shipped defaults)
#   gpurun_out/<round>_final/pmc_*:  FETCH_SIZE / WRITE_SIZE passes (separate, as the guide prescribes) over the plain transforms
#   profiles/<round>/final_kernel_stats.csv, final_memory_copy_stats.csv, final_kernel_stats.meta.json, traffic_1024.json
# Nothing here combines --pmc with a trace domain; the profiled program comes right after `--` (no wrapper hop).
set -e -o pipefail
round=${1:-r05}
R=$PWD
out=$R/gpurun_out/${round}_final
mkdir -p $out $R/profiles/$round
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --memory-copy-trace --stats -d $out/kernel --output-format csv -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --secondary-steps 0 --no-warm-transforms > $out/bench_k3_under_rocprofv3.json 2> $out/bench_k3_under_rocprofv3.err
rocprofv3 --pmc FETCH_SIZE -d $out/pmc_fetch --output-format csv -- python3 $R/tools/prof_transform.py 1024 2 both > $out/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE -d $out/pmc_write --output-format csv -- python3 $R/tools/prof_transform.py 1024 2 both > $out/pmc_write.log 2>&1
rocprofv3 --pmc SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE -d $out/pmc_lds --output-format csv -- python3 $R/tools/prof_transform.py 1024 2 both > $out/pmc_lds.log 2>&1
cd $R
python3 tools/kernel_stats_meta.py $out/kernel gpurun_out/${round}_final/profiles "bench.py --steps 3 --warmup 1 --no-cpu-baseline --secondary-steps 0 --no-warm-transforms (under rocprofv3 --kernel-trace --memory-copy-trace --stats)"
python3 tools/pmc_traffic.py 1024 $out/pmc_fetch $out/pmc_write > gpurun_out/${round}_final/profiles/traffic_1024.json
python3 tools/pmc_sq.py $out/pmc_lds > gpurun_out/${round}_final/profiles/sq_lds_counters_1024.txt
cp $out/bench_k3_under_rocprofv3.json gpurun_out/${round}_final/profiles/
# keep the merge small: the raw traces stay on the box
rm -rf $out/kernel $out/pmc_fetch $out/pmc_write $out/pmc_lds
ls -la gpurun_out/${round}_final/profiles
