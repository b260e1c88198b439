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
