#!/bin/bash
# experiment builds of libet_hip.so with -DSWP_ABL=<bits> (1: the lean register epilogue stores 64-byte-contiguous lane quads -- wrong pixels, timing only)
set -e
cd "$(dirname "$0")/../.."
python -m efficientteacher_amd.csrc.build >/dev/null
for n in "$@"; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -I include -DSWP_ABL=$n -c efficientteacher_amd/csrc/conv.hip -o /tmp/conv_swpabl$n.o &
done
wait
for n in "$@"; do
  objs=$(ls efficientteacher_amd/csrc/_obj/*.o | grep -v conv.o)
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o tools/probe/libet_swpabl$n.so /tmp/conv_swpabl$n.o $objs
done
