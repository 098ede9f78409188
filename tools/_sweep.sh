set -e
python tools/rc_speed.py waverange_amd/libwaverange_amd.so 64 > gpurun_out/rc_speed_multi.log 2>&1
for cfg in "1 8" "3 1" "4 1" "5 1"; do
  set -- $cfg
  echo "== jobs $1 threads $2" >> gpurun_out/sweep.log
  timeout -k 10 400 python bench.py --no-cpu-baseline --jobs $1 --threads $2 --steps 2 --warmup 1 >> gpurun_out/sweep.log 2>&1
done
