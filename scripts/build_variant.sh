#!/bin/bash
# Development tool: a variant of libmmk_hip.so whose mmk_unet.hip (or another source: $SRC) is compiled with extra flags, for A/B runs
# through MMK_LIB.   bash scripts/build_variant.sh <tag> <extra flags...>   ->  build_exp/lib_<tag>.so
set -e
tag=$1; shift
SRC=${SRC:-mmk_unet}
R=$(cd "$(dirname "$0")/.." && pwd)
O=$R/mm_masking_amd/csrc/_obj
BASE="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -I $R/include"
EXTRA=""
[ "$SRC" = "mmk_unet" ] && EXTRA="-mllvm -amdgpu-mfma-vgpr-form"
/opt/rocm/bin/hipcc $BASE $EXTRA "$@" -c $R/mm_masking_amd/csrc/$SRC.hip -o $R/build_exp/flags/${SRC}_$tag.o
objs=""
for s in mmk_api mmk_icp mmk_radar mmk_unet mmk_unet_driver mmk_loader mmk_loss; do
  if [ "$s" = "$SRC" ]; then objs="$objs $R/build_exp/flags/${SRC}_$tag.o"; else objs="$objs $O/$s.o"; fi
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $R/build_exp/lib_$tag.so $objs
echo built build_exp/lib_$tag.so
