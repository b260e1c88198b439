// C-ABI DOT entry points (include/accblas.h). Views are built as the
// reference launchers do (`/root/reference/cuda/dot_kernels.cuh:192-206` plain,
// `:224-263` accessor); the handle supplies what `myBlasHandle` supplied
// (`:29-65`): the CU count that sizes the persistent grid and device scratch.
#include <array>

#include "accblas_internal.hpp"
#include "dot_kernels.hpp"


namespace {
/** status[1] <- status[0], status[0] <- 0, as one atomic exchange (the 16-byte
 *  status slot of a handle's workspace: word 0 = the sticky mark the dot
 *  kernels write, word 1 = what the last `accblas_dot_status` took). */
__global__ void take_dot_status(std::uint32_t *status)
{
    status[1] = atomicExch(&status[0], 0u);
}
}  // namespace


extern "C" {


int accblas_dot(accblas_handle *handle, accblas_dtype vt, int64_t n,
                const void *x, int64_t incx, const void *y, int64_t incy,
                void *res, void *stream)
{
    const int status = accblas::validate_dot(handle, n, x, incx, y, incy, res);
    if (status != ACCBLAS_OK) {
        return status;
    }
    return accblas::dispatch_arithmetic(vt, [&](auto tag) {
        using value_type = typename decltype(tag)::type;
        using namespace accblas::dev;
        const plain_view<const value_type> x_view{
            static_cast<const value_type *>(x), {n, 1}, incx};
        const plain_view<const value_type> y_view{
            static_cast<const value_type *>(y), {n, 1}, incy};
        return accblas::check_hip(
            launch_dot<value_type>(handle->dot_ws,
                                   handle->dot_mode == ACCBLAS_DOT_ATOMIC,
                                   x_view, y_view,
                                   static_cast<value_type *>(res),
                                   accblas::as_stream(stream)),
            "dot launch");
    });
}


int accblas_acc_dot(accblas_handle *handle, accblas_dtype ar, accblas_dtype st,
                    accblas_dtype res_t, int64_t n, const void *x,
                    int64_t incx, const void *y, int64_t incy, void *res,
                    void *stream)
{
    const int status = accblas::validate_dot(handle, n, x, incx, y, incy, res);
    if (status != ACCBLAS_OK) {
        return status;
    }
    return accblas::dispatch_arithmetic(ar, [&](auto ar_tag) {
        using ar_type = typename decltype(ar_tag)::type;
        return accblas::dispatch_type(st, [&](auto st_tag) {
            using st_type = typename decltype(st_tag)::type;
            return accblas::dispatch_type(res_t, [&](auto res_tag) {
                using res_type = typename decltype(res_tag)::type;
                // Accessor setup, as cuda/dot_kernels.cuh:233-243
                constexpr std::size_t dimensionality{2};
                using accessor =
                    gko::acc::reduced_row_major<dimensionality, ar_type,
                                                st_type>;
                using c_range =
                    gko::acc::range<typename accessor::const_accessor>;
                using size_array = std::array<gko::acc::size_type, 2>;
                using stride_array = std::array<gko::acc::size_type, 1>;
                const auto x_acc = c_range(size_array{{n, 1}},
                                           static_cast<const st_type *>(x),
                                           stride_array{{incx}});
                const auto y_acc = c_range(size_array{{n, 1}},
                                           static_cast<const st_type *>(y),
                                           stride_array{{incy}});
                return accblas::check_hip(
                    accblas::dev::launch_dot<ar_type>(
                        handle->dot_ws,
                        handle->dot_mode == ACCBLAS_DOT_ATOMIC, x_acc, y_acc,
                        static_cast<res_type *>(res),
                        accblas::as_stream(stream)),
                    "acc_dot launch");
            });
        });
    });
}


int accblas_dot_status(accblas_handle *handle, void *stream, int *expired)
{
    if (handle == nullptr || expired == nullptr) {
        return accblas::set_error(ACCBLAS_ERR_NULL_POINTER,
                                  "dot_status: handle or expired is NULL");
    }
    // Everything on the caller's stream (nothing on the legacy null stream,
    // which would synchronise with every blocking stream): a one-thread kernel
    // takes the mark with an atomic exchange - read and clear in one step, so a
    // dot of this handle that expires on another stream meanwhile is reported
    // by the next call instead of being wiped -, the copy follows it in stream
    // order, then the stream is synchronised.
    hipStream_t queue = accblas::as_stream(stream);
    std::uint32_t *status_words = handle->dot_ws.status;
    take_dot_status<<<dim3(1), dim3(1), 0, queue>>>(status_words);
    int status = accblas::check_hip(hipGetLastError(), "dot status launch");
    if (status != ACCBLAS_OK) {
        return status;
    }
    std::uint32_t word = 0;
    status = accblas::check_hip(
        hipMemcpyAsync(&word, status_words + 1, sizeof(word),
                       hipMemcpyDeviceToHost, queue),
        "hipMemcpyAsync(dot status)");
    if (status != ACCBLAS_OK) {
        return status;
    }
    status = accblas::check_hip(hipStreamSynchronize(queue),
                                "hipStreamSynchronize");
    if (status != ACCBLAS_OK) {
        return status;
    }
    *expired = word == accblas::dev::dot_expired_word ? 1 : 0;
    return status;
}


}  // extern "C"
