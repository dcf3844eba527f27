#!/bin/bash
# Build libgpz_hip.so (gfx950) in-tree.  Usage: ./build.sh [-j N]
set -e
cd "$(dirname "$0")"
SRC=gpz_amd/csrc
OUT=gpz_amd/lib
mkdir -p $OUT build
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Iinclude -I$SRC"
pids=()
for f in k_phi k_gemm k_chol k_pinv k_rows k_gen k_psi k_psi32 k_psi32m k_pmiss k_pmiss_cov k_pmiss_cov64 k_pmiss_covg k_lbfgs k_wide k_cpsi k_cpsi4 k_cpsi4w k_cpsi4wp k_pmc4 gpz_ctx gpz_mgpu; do
  if [ ! -f build/$f.o ] || [ $SRC/$f.hip -nt build/$f.o ] || [ $SRC/gpz_kernels.h -nt build/$f.o ] || [ $SRC/gpz_dev.h -nt build/$f.o ] || [ $SRC/k_cpsi4_impl.h -nt build/$f.o ] || [ $SRC/gpz_mgpu_sync.h -nt build/$f.o ] || { [ $f = k_pmiss_cov64 ] && [ $SRC/k_pmiss_cov.hip -nt build/$f.o ]; } || [ include/gpz_hip.h -nt build/$f.o ]; then
    hipcc $FLAGS -c $SRC/$f.hip -o build/$f.o &
    pids+=($!)
  fi
done
for p in "${pids[@]}"; do wait $p; done
hipcc --offload-arch=gfx950 -shared -fPIC build/k_phi.o build/k_gemm.o build/k_chol.o build/k_pinv.o build/k_rows.o build/k_gen.o build/k_psi.o build/k_psi32.o build/k_psi32m.o build/k_pmiss.o build/k_pmiss_cov.o build/k_pmiss_cov64.o build/k_pmiss_covg.o build/k_lbfgs.o build/k_wide.o build/k_cpsi.o build/k_cpsi4.o build/k_cpsi4w.o build/k_cpsi4wp.o build/k_pmc4.o build/gpz_ctx.o build/gpz_mgpu.o -ldl -lpthread -o $OUT/libgpz_hip.so
echo "built $OUT/libgpz_hip.so"
