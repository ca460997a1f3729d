#!/bin/bash
mkdir -p gpurun_out
for L in default tools/probe/libet_keep2.so tools/probe/libet_one.so; do
  echo "== $L"
  if [ $L = default ]; then timeout 600 python tools/probe/buf_pattern.py 60 2>&1 | grep -v amdgpu.ids; else ET_HIP_LIB=$PWD/$L timeout 600 python tools/probe/buf_pattern.py 60 2>&1 | grep -v amdgpu.ids; fi
done > gpurun_out/r06_pattern.txt 2>&1
cat gpurun_out/r06_pattern.txt
