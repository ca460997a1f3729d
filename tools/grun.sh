#!/bin/bash
# build the gfx950 library here (hipcc cross-compiles), then run a command on an MI355X box through gpurun
set -e
cd "$(dirname "$0")/.."
python -m efficientteacher_amd.csrc.build > /dev/null
T=${GRUN_TIMEOUT:-1500}
/usr/local/graft/bin/gpurun --timeout $T -- "$@"
