#!/bin/bash
# Developer helper: builds libaccblas_hip.so with extra compiler flags into
# accessor-blas_amd/lib_<name>/ (git-ignored) for in-one-call A/B runs
# (ACCBLAS_LIB_PATH selects the build in the tools/*.py probes).
# usage: tools/build_variant.sh name "-DFLAG=1 ..."
set -e
name=$1; shift
flags="$*"
root=$(cd "$(dirname "$0")/../accessor-blas_amd" && pwd)
out=$root/build_x/$name
mkdir -p "$out" "$root/lib_$name"
common="-std=c++17 -O3 -fPIC --offload-arch=gfx950 -ffp-contract=fast -I$root/include -I$root/csrc -Wall -Wno-unused-function -Wno-pass-failed $flags"
pids=()
for src in accblas_runtime accblas_gemv accblas_dot accblas_trsv accblas_trsv_f32 accblas_trsv_f64 accblas_scaled; do
  /opt/rocm/bin/hipcc $common -c "$root/csrc/$src.hip" -o "$out/$src.o" &
  pids+=($!)
done
for p in "${pids[@]}"; do wait $p; done
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 "$out"/*.o -o "$root/lib_$name/libaccblas_hip.so"
echo "built $root/lib_$name/libaccblas_hip.so with: $flags"
