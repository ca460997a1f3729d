#!/bin/bash
# sample GPU clock / power while the step bench runs: is the step power- or clock-limited?
OUT=${1:-gpurun_out/power}; mkdir -p $OUT
python bench.py --steps 60 --warmup 5 --no-cpu-baseline > $OUT/bench.json 2> $OUT/bench.err &
BP=$!
sleep 20      # model build + warm-up
for i in $(seq 1 40); do
  rocm-smi --showclocks --showpower --showuse 2>/dev/null | grep -E "sclk|mclk|Power|GPU use|fclk" | tr '\n' ' ' >> $OUT/samples.txt; echo >> $OUT/samples.txt
  sleep 0.1
  kill -0 $BP 2>/dev/null || break
done
wait $BP
echo idle >> $OUT/samples.txt
rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|Power" | tr '\n' ' ' >> $OUT/samples.txt
cut -c1-200 $OUT/bench.json
head -5 $OUT/samples.txt; echo ...; tail -4 $OUT/samples.txt
