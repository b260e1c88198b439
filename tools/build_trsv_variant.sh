#!/bin/bash
# Developer helper: like build_variant.sh, but only the three TRSV translation
# units are compiled with the extra flags; the other objects are taken from the
# default build (accessor-blas_amd/build, `make` first). For A/B runs of TRSV
# kernel variants (tools/trsv_lib_ab2.py).
# usage: tools/build_trsv_variant.sh name "-DFLAG=1 ..."
set -e
name=$1; shift
flags="$*"
root=$(cd "$(dirname "$0")/../accessor-blas_amd" && pwd)
out=$root/build_x/$name
mkdir -p "$out" "$root/lib_$name"
common="-std=c++17 -O3 -fPIC --offload-arch=gfx950 -ffp-contract=fast -I$root/include -I$root/csrc -Wall -Wno-unused-function -Wno-pass-failed -Wno-unused-lambda-capture $flags"
pids=()
for src in accblas_trsv accblas_trsv_f32 accblas_trsv_f64; do
  /opt/rocm/bin/hipcc $common -c "$root/csrc/$src.hip" -o "$out/$src.o" &
  pids+=($!)
done
for p in "${pids[@]}"; do wait $p; done
for src in accblas_runtime accblas_gemv accblas_dot accblas_scaled; do cp "$root/build/$src.o" "$out/$src.o"; done
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 "$out"/*.o -o "$root/lib_$name/libaccblas_hip.so"
echo "built $root/lib_$name/libaccblas_hip.so with: $flags"
