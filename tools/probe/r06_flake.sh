set -u
OUT=gpurun_out/r06flake; mkdir -p $OUT
for K in 0 1; do
  n=0; f=0
  for i in $(seq 1 12); do
    ET_CONV_BUF_DMA=$K timeout 300 python -m pytest "tests/test_fullsize.py::test_conv_adjoint_and_linearity_full_size" -x -q -m gpu > $OUT/run_${K}_$i.log 2>&1 && n=$((n+1)) || { f=$((f+1)); grep -E "^E  +Assert|^tests.*FAILED|full_size\[" $OUT/run_${K}_$i.log | head -3; }
  done
  echo "ET_CONV_BUF_DMA=$K: passed $n failed $f"
done | tee $OUT/summary.txt
