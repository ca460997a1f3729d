#!/bin/bash
# experiment builds of libet_hip.so with -DET_ABLATE=<n> on tools/probe/conv_probe.hip (the r02 copy of csrc/conv.hip that still carries the ET_ABLATE / ET_STAMPS hooks; the product source has none)
set -e
cd "$(dirname "$0")/../.."
python -m efficientteacher_amd.csrc.build >/dev/null
for n in "$@"; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -I include -DET_ABLATE=$n -c tools/probe/conv_probe.hip -o /tmp/conv_abl$n.o &
done
wait
for n in "$@"; do
  objs=$(ls efficientteacher_amd/csrc/_obj/*.o | grep -v conv.o)
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o tools/probe/libet_abl$n.so /tmp/conv_abl$n.o $objs
done
ls -la tools/probe/*.so
