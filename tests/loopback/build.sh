#!/bin/bash
# Builds tests/loopback/libloopback_nccl.so (test infrastructure, see loopback_nccl.cpp): host code + one small gfx950 kernel (the rank-order sum).
set -e
cd "$(dirname "$0")"
if [ ! -f libloopback_nccl.so ] || [ loopback_nccl.cpp -nt libloopback_nccl.so ]; then
  /opt/rocm/bin/hipcc -x hip --offload-arch=gfx950 -O2 -fPIC -shared -std=c++17 loopback_nccl.cpp -o libloopback_nccl.so -lpthread
fi
echo "built tests/loopback/libloopback_nccl.so"
