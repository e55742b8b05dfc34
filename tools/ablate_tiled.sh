#!/bin/bash
# Timing variants of the library for A/B runs (tools/ab_kernels.py).  The product sources carry no experiment switches: the
# historical / wrong-result variants live in tools/experiments/kernel_lab.patch, which this script applies to a scratch copy of
# torchpdlp_amd/csrc before building with the variant's -D flag.  Outputs: tools/_bin/libpdlp_<name>.so (they travel to the GPU
# box with gpurun; delete them when the experiment is over).
#   tools/ablate_tiled.sh <name>[:-DFLAG[=v] ...] ...      e.g.  tools/ablate_tiled.sh noprio:-DPDLP_PRIO_P1=0:-DPDLP_PRIO_P2=0 abl_noval:-DPDLP_ABL_NOVAL
# Switches the patch restores (see the comments next to each #if in the patched sources):
#   PDLP_SELECT_SUM  PDLP_NO_SCAN2  PDLP_BURST_PRODUCTS  PDLP_COUNTS_AT_TOP  PDLP_COUNTS_THREAD_MAJOR  PDLP_CSR_BRANCHY
#   PDLP_SLIDE=n  PDLP_NT_GATHER  PDLP_STAGGER=n  PDLP_STAMPS        (timing only, WRONG RESULTS:) PDLP_ABL_NOVAL  PDLP_ABL_NOCOUNTS  PDLP_ABL_NOGATHER
# Tunables that need no patch (product sources): PDLP_PRIO_P1/_P2, PDLP_ROUND, PDLP_EPI_GROUP, PDLP_F32_RPT/_CAP/_TU, PDLP_TNT (threads per workgroup).
# Round 5 (tools/experiments/r05_gather_lab.patch, applied by hand to a scratch copy): PDLP_GATHER_X4=1|2 (16-byte gathers), PDLP_GATHER_LDSDMA=1, k_csr_fused<XLDS>.
set -e
cd "$(dirname "$0")/.."
mkdir -p tools/_bin
scratch=$(mktemp -d)
trap 'rm -rf "$scratch"' EXIT
cp -r torchpdlp_amd/csrc "$scratch/csrc_clean"
(cd "$scratch" && patch -s -p0 < "$OLDPWD/tools/experiments/kernel_lab.patch")
for spec in "$@"; do
  name=${spec%%:*}
  flags=$(echo "${spec#"$name"}" | tr ':' ' ')
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -ffp-contract=off -Iinclude $flags \
      "$scratch/csrc_clean/pdlp_hip.hip" -o "tools/_bin/libpdlp_$name.so" 2>/dev/null &
done
wait
rm -f tools/_bin/*.hipfb
ls tools/_bin
