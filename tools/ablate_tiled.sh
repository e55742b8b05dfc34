#!/bin/bash
# Builds timing variants of the library into tools/_bin/ (they travel to the GPU box with gpurun); each switches ONE of the
# round-2 scheduling changes of the tiled / CSR kernels back to its earlier form ("What moved the tiled
# kernel").  Time them in one process:  python tools/ab_kernels.py torchpdlp_amd/libpdlp_hip.so tools/_bin/libpdlp_<v>.so ...
#   noprio        no wave priority for pass 1                    selectsum  compare/select row sums instead of clamp weights
#   noscan2       compiler-scheduled count scans                 burst      all products of a group after its last gathers
#   round2 / round1  gather rounds of 2 / of 1 per lane instead of 4
#   cnttop        count words loaded at the top of their tile    csrbranchy conditional loads in the CSR kernel
#   nogather / stamps   ablation (wrong results) / cycle stamps per phase, both diagnostic only
# round 3 (profiles/r03_ab*.log):
#   abl_noval / abl_nocounts   no value stream / count words loaded once per workgroup (wrong results, timing only)
#   cnt_thread_major           the count-word layout of rounds 1-2 (time it with "lib.so:PDLP_CNT_LAYOUT=thread" in ab_kernels.py)
#   slide1 / slide2            8 / 12 gathers per lane in flight (sliding window)
#   nt_gather, stagger_200     gathers bypassing L1; the second workgroup of a CU starts 200 x 64 clocks late
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
build abl_noval -DPDLP_ABL_NOVAL &
build abl_nocounts -DPDLP_ABL_NOCOUNTS &
build cnt_thread_major -DPDLP_COUNTS_THREAD_MAJOR &
build slide1 -DPDLP_SLIDE=1 &
wait
build slide2 -DPDLP_SLIDE=2 &
build nt_gather -DPDLP_NT_GATHER &
build stagger_200 -DPDLP_STAGGER=200 &
wait
rm -f tools/_bin/*.hipfb
ls tools/_bin
