set -e
O=gpurun_out/r3i; mkdir -p $O
for v in base nopoll base nopoll; do
  timeout -k 10 300 accessor-blas_amd/build/tune_trsv_wide_$v 24000 5 50 $O/stamps_24000_$v.csv > $O/wide_24000_$v.log 2>&1
  sed -n 3p $O/wide_24000_$v.log
  timeout -k 10 300 accessor-blas_amd/build/tune_trsv_wide_$v 65536 3 50 $O/stamps_65536_$v.csv > $O/wide_65536_$v.log 2>&1
  sed -n 3p $O/wide_65536_$v.log
done
echo ALL DONE
