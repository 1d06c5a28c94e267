#!/bin/bash
# tools/build_variant.sh <name> [hipcc flags...]  -> build/abl/libfemhip_<name>.so
set -e
cd /root/repo
name=$1; shift
mkdir -p build/abl
/opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -fPIC -Wall -Wno-unused-result "$@" -c fem_amd/csrc/fem_hip.hip -o build/abl/fem_hip_$name.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -o build/abl/libfemhip_$name.so build/abl/fem_hip_$name.o fem_amd/csrc/fem_index_build.o fem_amd/csrc/fem_tail.o -L/opt/rocm/lib -lrccl -Wl,-rpath,/opt/rocm/lib
rm -f build/abl/fem_hip_$name.o
echo built $name
