#!/bin/bash
# Developer tool: build a variant of libgpz_hip.so with extra -D flags into gpz_amd/lib/libgpz_hip_<tag>.so
#   tools/build_variant.sh <tag> <file-to-rebuild> [-DNAME=VALUE ...]      e.g.  tools/build_variant.sh ur12 k_rows -DGPZ_MOM_UR=12
set -e
cd "$(dirname "$0")/.."
tag=$1; f=$2; shift 2
mkdir -p build/var_$tag
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Iinclude -Igpz_amd/csrc "$@" -c gpz_amd/csrc/$f.hip -o build/var_$tag/$f.o
objs=""
for g in k_phi k_gemm k_chol k_pinv k_rows k_gen k_psi k_psi32 k_pmiss k_pmiss_cov k_lbfgs gpz_ctx; do
  if [ "$g" == "$f" ]; then objs="$objs build/var_$tag/$g.o"; else objs="$objs build/$g.o"; fi
done
hipcc --offload-arch=gfx950 -shared -fPIC $objs -o gpz_amd/lib/libgpz_hip_$tag.so
echo "built gpz_amd/lib/libgpz_hip_$tag.so"
