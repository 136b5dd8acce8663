#!/bin/bash
# Builds tests/loopback/libloopback_nccl.so (test infrastructure, see loopback_nccl.cpp).  Host code only.
set -e
cd "$(dirname "$0")"
if [ ! -f libloopback_nccl.so ] || [ loopback_nccl.cpp -nt libloopback_nccl.so ]; then
  /opt/rocm/bin/hipcc -O2 -fPIC -shared -std=c++17 loopback_nccl.cpp -o libloopback_nccl.so -lpthread
fi
echo "built tests/loopback/libloopback_nccl.so"
