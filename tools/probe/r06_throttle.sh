# which limiter holds the shader clock below 2.4 GHz during the SSOD step?  amd-smi throttle-violation accumulators around a 400-step run
set -u
OUT=gpurun_out/r06t3; mkdir -p $OUT
amd-smi metric -g 0 --violation > $OUT/viol_before.txt 2>&1
python bench.py --steps 400 --warmup 10 --no-cpu-baseline --no-teacher-alone > $OUT/bench_long.json 2>/dev/null &
B=$!
sleep 14
amd-smi metric -g 0 --violation > $OUT/viol_during.txt 2>&1
amd-smi metric -g 0 --power --temperature > $OUT/power_temp_during.txt 2>&1
wait $B
amd-smi metric -g 0 --violation > $OUT/viol_after.txt 2>&1
python -c "import json; d=json.load(open('$OUT/bench_long.json')); print('400 steps ms_per_step', round(d['ms_per_step'],2))"
echo "--- before"; cat $OUT/viol_before.txt | head -60
echo "--- after"; cat $OUT/viol_after.txt | head -60
echo "--- power / temperature during"; cat $OUT/power_temp_during.txt | head -40
