#!/bin/bash
# Build an experimental variant of libenf_hip.so:  scripts/build_variant.sh NAME "-DFLAG=.. -f.." ["src1 src2 .."]
# -> variants/libenf_NAME.so (git-ignored, travels to the GPU box); select with ENF_HIP_LIB=variants/libenf_NAME.so
# With a third argument only those sources are compiled with the flags; the rest are the in-tree objects (run `make` first).
set -e
cd "$(dirname "$0")/.."
NAME=$1; FLAGS=$2; ONLY=$3
mkdir -p variants build_variants/$NAME
SRC=enf-pde_amd/csrc
ALL="enf_api enf_pack enf_prologue enf_loss enf_wz enf_pair_fwd enf_pair_bwd enf_xtd enf_tail enf_train enf_debug enf_ode enf_ode_basis enf_ode_block"
OBJS=""
for f in $ALL; do
  if [ -z "$ONLY" ] || [[ " $ONLY " == *" $f "* ]] || [ ! -f $SRC/$f.o ]; then
    ( /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-function -Wno-unknown-pragmas $FLAGS -c $SRC/$f.hip -o build_variants/$NAME/$f.o ) &
    OBJS="$OBJS build_variants/$NAME/$f.o"
  else
    OBJS="$OBJS $SRC/$f.o"
  fi
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $OBJS -o variants/libenf_$NAME.so
echo built variants/libenf_$NAME.so
