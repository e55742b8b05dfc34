#!/bin/bash
# Builds timing variants of the library into tools/_bin/ (they travel to the GPU box with gpurun); each switches ONE of the
# round-2 scheduling changes of the tiled / CSR kernels back to its earlier form ("What moved the tiled
# kernel").  Time them in one process:  python tools/ab_kernels.py torchpdlp_amd/libpdlp_hip.so tools/_bin/libpdlp_<v>.so ...
#   noprio        no wave priority for pass 1                    selectsum  compare/select row sums instead of clamp weights
#   noscan2       compiler-scheduled count scans                 burst      all products of a group after its last gathers
#   round2 / round1  gather rounds of 2 / of 1 per lane instead of 4
#   cnttop        count words loaded at the top of their tile    csrbranchy conditional loads in the CSR kernel
#   nogather / stamps   ablation (wrong results) / cycle stamps per phase, both diagnostic only
set -e
cd "$(dirname "$0")/.."
mkdir -p tools/_bin
build() { name=$1; shift; /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -ffp-contract=off -Iinclude "$@" \
     torchpdlp_amd/csrc/pdlp_hip.hip -o tools/_bin/libpdlp_$name.so 2>/dev/null; }
build noprio -DPDLP_PRIO_P1=0 -DPDLP_PRIO_P2=0 &
build selectsum -DPDLP_SELECT_SUM &
build noscan2 -DPDLP_NO_SCAN2 &
build burst -DPDLP_BURST_PRODUCTS &
wait
build round2 -DPDLP_ROUND=2 &
build round1 -DPDLP_ROUND=1 &
build cnttop -DPDLP_COUNTS_AT_TOP &
build csrbranchy -DPDLP_CSR_BRANCHY &
wait
build nogather -DPDLP_ABL_NOGATHER &
build stamps -DPDLP_STAMPS &
wait
rm -f tools/_bin/*.hipfb
ls tools/_bin
