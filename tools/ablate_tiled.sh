#!/bin/bash
# builds timing-only variants of the library into tools/_bin/ (they travel to the GPU box with gpurun)
set -e
cd "$(dirname "$0")/.."
mkdir -p tools/_bin
for v in FULL NOGATHER STAMPS; do
  flags=""; [ "$v" = NOGATHER ] && flags="-DPDLP_ABL_NOGATHER"; [ "$v" = STAMPS ] && flags="-DPDLP_STAMPS"
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -ffp-contract=off -Iinclude $flags \
     torchpdlp_amd/csrc/pdlp_hip.hip -o tools/_bin/libpdlp_$v.so 2>/dev/null &
done
wait; ls tools/_bin
