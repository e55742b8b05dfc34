#!/bin/bash
# Build libpdlp_hip.so for gfx950 in-tree (cross-compiles without a GPU).
set -euo pipefail
HERE="$(cd "$(dirname "${BASH_SOURCE[0]}")" && pwd)"
ROOT="$(cd "$HERE/../.." && pwd)"
HIPCC="${HIPCC:-/opt/rocm/bin/hipcc}"
OUT="$ROOT/torchpdlp_amd/libpdlp_hip.so"
# -ffp-contract=off: the reference's eager torch ops never fuse a multiply into an add; keep the same
# roundings (the path is bandwidth-bound, FMA contraction buys nothing).
"$HIPCC" --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -ffp-contract=off \
    -Wall -Wextra -Wno-unused-parameter \
    -I"$ROOT/include" "$HERE/pdlp_hip.hip" -o "$OUT" "$@"
echo "built $OUT"
