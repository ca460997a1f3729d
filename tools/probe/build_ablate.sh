#!/bin/bash
# experiment builds of libet_hip.so with -DET_ABLATE=<n> on conv.hip (see the ET_ABLATE hooks there)
set -e
cd "$(dirname "$0")/../.."
python -m efficientteacher_amd.csrc.build >/dev/null
for n in "$@"; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -I include -DET_ABLATE=$n -c efficientteacher_amd/csrc/conv.hip -o /tmp/conv_abl$n.o &
done
wait
for n in "$@"; do
  objs=$(ls efficientteacher_amd/csrc/_obj/*.o | grep -v conv.o)
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o tools/probe/libet_abl$n.so /tmp/conv_abl$n.o $objs
done
ls -la tools/probe/*.so
