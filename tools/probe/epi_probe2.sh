#!/bin/bash
# r03: the epilogue's share of a tile AFTER the hardware bf16 conversion and the guard-free store pass (tools/probe/epi_probe.sh of r02
# was measured before them).  forward with statistics (the training forward) and forward with scale/bias/SiLU (the teacher)
for L in S S60; do
  echo "== lib $L"
  for shape in "256 256 3 1 40 64" "128 128 3 1 80 64" "128 128 1 1 80 64" "256 256 1 1 40 64"; do
    ET_HIP_LIB=$PWD/tools/probe/libet_$L.so timeout 120 python tools/probe/ts_conv.py $shape 2>&1 | grep TS | cut -c1-420
  done
done
