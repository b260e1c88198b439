// acc_trsv kernels with f64 arithmetic (see accblas_trsv_common.hpp).
#include "accblas_trsv_common.hpp"

namespace accblas {

int acc_trsv_f64(accblas_dtype st, accblas_uplo uplo, accblas_diag diag,
                 int64_t n, const void *A, int64_t lda, void *x, int64_t incx,
                 uint32_t *helper, void *stream,
                 const dev::trsv_session *external)
{
    return acc_trsv_typed<double>(st, uplo, diag, n, A, lda, x, incx, helper,
                               stream, external);
}

}  // namespace accblas
