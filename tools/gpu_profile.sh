#!/bin/bash
# Runs on the GPU box (via gpurun): bench line + rocprofv3 kernel trace + PMC
# passes for the GEMV headline workload. Outputs land in gpurun_out/.
set -o pipefail
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
python3 $REPO/bench.py --steps 50 --warmup 5 > $OUT/bench.json 2> $OUT/bench.err || exit 1
tail -1 $OUT/bench.json
# the SAME command as the bench line above, under the kernel trace
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_stats -o gemv -- \
    python3 $REPO/bench.py --steps 50 --warmup 5 > $OUT/prof_stats.log 2>&1 || exit 2
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/prof_fetch -o gemv -- \
    python3 $REPO/bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-other-configs > $OUT/prof_fetch.log 2>&1 || exit 3
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/prof_write -o gemv -- \
    python3 $REPO/bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-other-configs > $OUT/prof_write.log 2>&1 || exit 4
find $OUT -name '*.csv' | head -20
# dot and TRSV kernels at the BASELINE sizes (kernel-trace stats only)
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_dot -o dot -- \
    python3 $REPO/tools/probe_perf.py dot > $OUT/prof_dot.log 2>&1 || exit 5
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_trsv -o trsv -- \
    python3 $REPO/tools/probe_perf.py trsv > $OUT/prof_trsv.log 2>&1 || exit 6
# HBM traffic of the other entries of the bench line (`other_configs`): separate
# --pmc passes per counter and case. Condensed afterwards, on the CPU box, by
#   python3 tools/summarize_profiles.py rNN                       (headline; writes profiles/traffic.json's first entry)
#   python3 tools/summarize_pmc.py rNN trsv trsv
#   python3 tools/summarize_pmc.py rNN dot '::dot|3dotI'
#   python3 tools/summarize_pmc.py rNN gemv_other gemv_vec
#   python3 tools/update_traffic.py entry=profiles/rNN_<tag>_pmc.json:<needle> ...   (ties each ratio to the kernel sources' hash)
if [ "${ACCBLAS_PROFILE_OTHER:-1}" = "1" ]; then
    bash $REPO/tools/gpu_profile_trsv.sh trsv '65536 f64 f32 lower 5 1 -1' '24000 f64 f32 upper 7 1 -1' \
        '24000 f32 f32 upper 7 1 -1' '24000 f64 f64 upper 7 1 -1' > $OUT/pmc_trsv.log 2>&1 || exit 7
    CASE=dot bash $REPO/tools/gpu_profile_trsv.sh dot '268435456 f32 f16 5' '33554432 f32 f16 8' > $OUT/pmc_dot.log 2>&1 || exit 8
    CASE=gemv bash $REPO/tools/gpu_profile_trsv.sh gemv_other '16384 16384 f64 f64 5' '4096 32768 f64 f32 8 32768' \
        > $OUT/pmc_gemv_other.log 2>&1 || exit 9
fi
