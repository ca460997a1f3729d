#!/bin/bash
# experiment builds of libet_hip.so with -DRS_ABL=<bits> (1 no fragment masks, 2 no row shift, 4 no A staging after the first unit): timing only
set -e
cd "$(dirname "$0")/../.."
python -m efficientteacher_amd.csrc.build >/dev/null
for n in "$@"; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -I include -DRS_ABL=$n -c efficientteacher_amd/csrc/conv.hip -o /tmp/conv_rs$n.o &
done
wait
for n in "$@"; do
  objs=$(ls efficientteacher_amd/csrc/_obj/*.o | grep -v conv.o)
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o tools/probe/libet_rs$n.so /tmp/conv_rs$n.o $objs
done
