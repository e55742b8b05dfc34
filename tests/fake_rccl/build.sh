#!/bin/bash
# test-only stand-in for librccl (see fake_rccl.cpp); the GPU tests build it on first use if it is missing
set -euo pipefail
HERE="$(cd "$(dirname "${BASH_SOURCE[0]}")" && pwd)"
"${HIPCC:-/opt/rocm/bin/hipcc}" -O2 -std=c++17 -fPIC -shared -Wall -x c++ -D__HIP_PLATFORM_AMD__ -I/opt/rocm/include \
    "$HERE/fake_rccl.cpp" -o "$HERE/libfake_rccl.so" -L/opt/rocm/lib -lamdhip64 -lrt -lpthread
echo "built $HERE/libfake_rccl.so"
