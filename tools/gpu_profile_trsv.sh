#!/bin/bash
# Runs on the GPU box (via gpurun): HBM traffic counters of the TRSV kernel
# (separate --pmc passes, the program directly after `--`), plus the dot kernel.
# usage: gpu_profile_trsv.sh TAG "N ar st uplo reps relay wide" ...
#        CASE=dot gpu_profile_trsv.sh TAG "N ar st reps" ...   (tools/dot_case.py)
set -o pipefail
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
TAG=$1; shift
i=0
for cfg in "$@"; do
    i=$((i+1))
    for ctr in FETCH_SIZE WRITE_SIZE; do
        rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d $OUT/pmc_${TAG}_${i}_$ctr -o trsv -- \
            python3 $REPO/tools/${CASE:-trsv}_case.py $cfg > $OUT/pmc_${TAG}_${i}_$ctr.log 2>&1 || exit $((10+i))
        tail -1 $OUT/pmc_${TAG}_${i}_$ctr.log
    done
done
