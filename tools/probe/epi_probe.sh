#!/bin/bash
# what the shared conv epilogue costs: s_memtime stamps per workgroup (build -DET_STAMPS) with and without its global stores (-DET_ABLATE=60)
# build first (in the container):
#   for v in "S:-DET_STAMPS" "S60:-DET_STAMPS -DET_ABLATE=60"; do n=${v%%:*}; f=${v#*:}; hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -I include $f -c tools/probe/conv_probe.hip -o /tmp/conv_$n.o;
#     hipcc --offload-arch=gfx950 -shared -fPIC -o tools/probe/libet_$n.so /tmp/conv_$n.o $(ls efficientteacher_amd/csrc/_obj/*.o | grep -v conv.o); done
for L in S S60; do
  echo "== lib $L"
  for shape in "256 256 3 1 40 64" "128 128 3 1 80 64" "128 128 1 1 80 64" "256 256 1 1 40 64"; do
    ET_HIP_LIB=$PWD/tools/probe/libet_$L.so timeout 120 python tools/probe/ts_conv.py $shape 2>&1 | grep TS | cut -c1-420
  done
  ET_HIP_LIB=$PWD/tools/probe/libet_$L.so MB_REF=0 MB_ONLY=pp timeout 300 python tools/microbench.py conv 2>&1 | tail -1 | cut -c1-160
done
