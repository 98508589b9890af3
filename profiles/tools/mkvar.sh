#!/bin/bash
# Builds a diagnostic variant of libsedhip.so: profiles/tools/mkvar.sh NAME [extra hipcc flags, e.g. -DSED_X_MASK=64]
# -> profiles/tools/build/libsedhip_NAME.so (front-end restricted to the headline kernel unless FULL=1).
set -e
here=$(cd "$(dirname "$0")" && pwd)
src=$here/../../sound-event-detection_amd/csrc
name=$1; shift
only=${FULL:+}
[ -z "${FULL:-}" ] && only=-DSED_ONLY_BENCH
mkdir -p $here/build
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -I$here/../../include -Wall -Wno-unused-function $only "$@" \
  -c ${SRC:-$src/sed_frontend.hip} -o $here/build/front_$name.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $here/build/front_$name.o $src/sed_decode.o $src/sed_resample.o -o $here/build/libsedhip_$name.so
echo built $here/build/libsedhip_$name.so
