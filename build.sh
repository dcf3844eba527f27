#!/bin/bash
# Build libgpz_hip.so (gfx950) in-tree.  Usage: ./build.sh [--dev]
#   --dev: the variant with the developer A/B switches of gpz_options.h compiled in (-DGPZ_DEV_SWITCHES) -> gpz_amd/lib/libgpz_hip_dev.so
set -e
cd "$(dirname "$0")"
SRC=gpz_amd/csrc
OUT=gpz_amd/lib
OBJ=build
LIB=libgpz_hip.so
EXTRA=""
if [ "$1" = "--dev" ]; then OBJ=build/dev; LIB=libgpz_hip_dev.so; EXTRA="-DGPZ_DEV_SWITCHES"; fi
mkdir -p $OUT $OBJ
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Iinclude -I$SRC $EXTRA"
UNITS="k_phi k_gemm k_syrk_small k_small k_chol k_pinv k_rows k_gen k_psi k_psi32 k_psi32m k_pmiss k_pmiss_cov k_pmiss_cov64 k_pmiss_covg k_lbfgs k_wide k_cpsi k_cpsi4 k_cpsi4w k_cpsi4wp k_pmc4 gpz_options gpz_arena gpz_ctx gpz_eval gpz_graph gpz_predict gpz_mgpu"
[ "$1" = "--dev" ] && UNITS="$UNITS k_oz"   # the int8-sliced T-GEMM: a measured route that is not the default
HDRS="$SRC/gpz_kernels.h $SRC/gpz_dev.h $SRC/gpz_options.h $SRC/gpz_ctx.h $SRC/k_cpsi4_impl.h $SRC/gpz_mgpu_sync.h include/gpz_hip.h"
pids=()
objs=""
for f in $UNITS; do
  [ -f $SRC/$f.hip ] || continue
  objs="$objs $OBJ/$f.o"
  stale=0
  [ -f $OBJ/$f.o ] || stale=1
  [ $stale = 0 ] && [ $SRC/$f.hip -nt $OBJ/$f.o ] && stale=1
  for h in $HDRS; do [ $stale = 0 ] && [ -f $h ] && [ $h -nt $OBJ/$f.o ] && stale=1; done
  [ $f = k_pmiss_cov64 ] && [ $stale = 0 ] && [ $SRC/k_pmiss_cov.hip -nt $OBJ/$f.o ] && stale=1
  if [ $stale = 1 ]; then
    hipcc $FLAGS -c $SRC/$f.hip -o $OBJ/$f.o &
    pids+=($!)
  fi
done
for p in "${pids[@]}"; do wait $p; done
hipcc --offload-arch=gfx950 -shared -fPIC $objs -ldl -lpthread -o $OUT/$LIB
echo "built $OUT/$LIB"
