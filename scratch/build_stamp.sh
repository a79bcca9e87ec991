#!/bin/bash
# second build of the library with -DSM3_STAMP (diagnostic; never shipped): scratch/_stamp/libsm3hip_stamp.so
set -e
cd "$(dirname "$0")/.."
mkdir -p scratch/_stamp
for f in skin-sm3_amd/csrc/*.hip; do
  o=scratch/_stamp/$(basename $f .hip).o
  if [ ! -f $o ] || [ $f -nt $o ] || [ skin-sm3_amd/csrc/conv_common.h -nt $o ] || [ skin-sm3_amd/csrc/common.h -nt $o ]; then
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -munsafe-fp-atomics -DSM3_STAMP -c $f -o $o &
  fi
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC scratch/_stamp/*.o -o scratch/_stamp/libsm3hip_stamp.so
ls -la scratch/_stamp/libsm3hip_stamp.so
