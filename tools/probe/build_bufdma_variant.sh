#!/bin/bash
# usage: build_bufdma_variant.sh <name> <-D flags...>   ->  tools/probe/libet_<name>.so  (conv.hip rebuilt with the flags, other objects reused)
set -e
cd "$(dirname "$0")/../.."
N=$1; shift
O=efficientteacher_amd/csrc/_obj
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function -I include "$@" -c efficientteacher_amd/csrc/conv.hip -o /tmp/conv_$N.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o tools/probe/libet_$N.so $O/abi.o $O/augment.o /tmp/conv_$N.o $O/detect.o $O/loss.o $O/nms.o $O/norm.o $O/optim.o $O/pseudo_label.o $O/spatial.o $O/tal.o
ls -la tools/probe/libet_$N.so
