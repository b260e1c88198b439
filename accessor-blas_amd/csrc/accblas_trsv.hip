// C-ABI TRSV entry points (include/accblas.h). Views are built as the
// reference launchers do (`/root/reference/cuda/trsv_kernels.cuh:455-488`
// plain, `:918-961` accessor).
#include <array>
#include <atomic>
#include <map>
#include <mutex>
#include <utility>

#include "accblas_trsv_common.hpp"


namespace {
std::atomic<int> trsv_split_option{0};
std::atomic<int> trsv_wide_option{0};
std::atomic<int> trsv_relay_option{1};
std::atomic<int> gemv_skew_option_store{1};
}

int accblas::gemv_skew_option_value() { return gemv_skew_option_store.load(); }

bool accblas::trsv_split_enabled() { return trsv_split_option.load() != 0; }
bool accblas::trsv_wide_enabled() { return trsv_wide_option.load() != 0; }
int accblas::trsv_relay_option_value() { return trsv_relay_option.load(); }

extern "C" int accblas_set_option(const char *name, int value)
{
    if (name != nullptr && std::string(name) == "trsv_split") {
        trsv_split_option.store(value);
        return ACCBLAS_OK;
    }
    if (name != nullptr && std::string(name) == "trsv_wide") {
        trsv_wide_option.store(value);
        return ACCBLAS_OK;
    }
    if (name != nullptr && std::string(name) == "trsv_relay") {
        if (value < 0 || value > 2) {
            return accblas::set_error(ACCBLAS_ERR_INVALID_ARGUMENT,
                                      "trsv_relay takes 0, 1 or 2");
        }
        trsv_relay_option.store(value);
        return ACCBLAS_OK;
    }
    if (name != nullptr && std::string(name) == "gemv_skew") {
        if (value < 0 || value > 2) {
            return accblas::set_error(ACCBLAS_ERR_INVALID_ARGUMENT,
                                      "gemv_skew takes 0, 1 or 2");
        }
        gemv_skew_option_store.store(value);
        return ACCBLAS_OK;
    }
    return accblas::set_error(ACCBLAS_ERR_INVALID_ARGUMENT,
                              "accblas_set_option: unknown option");
}

int accblas::trsv_num_cus()
{
    static int cached = 0;
    if (cached == 0) {
        int device = 0;
        hipDeviceProp_t prop;
        if (hipGetDevice(&device) == hipSuccess &&
            hipGetDeviceProperties(&prop, device) == hipSuccess) {
            cached = prop.multiProcessorCount;
        } else {
            cached = 1;
        }
    }
    return cached;
}


namespace {


// Hand-off workspace of the solve kernels: a 4 KiB header (the
// `trsv_state` of `kernel::trsv`) followed by 8-byte {tag, value} granules, one or
// two per element of x. The reference's launcher signature has no handle to
// keep it in, so the library owns one buffer per (device, stream) and grows it
// on demand; growth is the only time a TRSV call allocates (and synchronises
// that stream first, so a solve still reading the old buffer finishes). A new
// buffer is zeroed once; `kernel::trsv` re-arms it itself (epoch tags).
struct workspace_entry {
    void *ptr{nullptr};
    std::size_t bytes{0};
};

constexpr std::size_t workspace_header_bytes = 4096;

std::mutex workspace_guard;
std::map<std::pair<int, hipStream_t>, workspace_entry> workspace_cache;

/** Caller holds `workspace_guard`. */
int ensure_workspace(hipStream_t stream, std::size_t granule_bytes,
                     workspace_entry **out)
{
    int device = 0;
    int status = accblas::check_hip(hipGetDevice(&device), "hipGetDevice");
    if (status != ACCBLAS_OK) {
        return status;
    }
    auto &entry = workspace_cache[std::make_pair(device, stream)];
    const std::size_t bytes = granule_bytes + workspace_header_bytes;
    if (entry.bytes < bytes) {
        if (entry.ptr != nullptr) {
            status = accblas::check_hip(hipStreamSynchronize(stream),
                                        "hipStreamSynchronize");
            if (status != ACCBLAS_OK) {
                return status;
            }
            (void)hipFree(entry.ptr);
            entry = workspace_entry{};
        }
        const std::size_t rounded =
            std::max<std::size_t>(bytes + bytes / 2, std::size_t{1} << 16);
        status = accblas::check_hip(hipMalloc(&entry.ptr, rounded),
                                    "hipMalloc(trsv workspace)");
        if (status != ACCBLAS_OK) {
            entry = workspace_entry{};
            return status;
        }
        entry.bytes = rounded;
        status = accblas::check_hip(
            hipMemsetAsync(entry.ptr, 0, rounded, stream),
            "hipMemsetAsync(trsv workspace)");
        if (status != ACCBLAS_OK) {
            return status;
        }
        // header = accblas::dev::trsv_state{0, 0, first epoch, 0}
        static const accblas::dev::trsv_state initial{
            0u, 0u, accblas::dev::trsv_first_epoch, 0u};
        status = accblas::check_hip(
            hipMemcpyAsync(entry.ptr, &initial, sizeof(initial),
                           hipMemcpyHostToDevice, stream),
            "hipMemcpyAsync(trsv state)");
        if (status != ACCBLAS_OK) {
            return status;
        }
    }
    *out = &entry;
    return ACCBLAS_OK;
}

accblas::dev::granule_word *granules_of(const workspace_entry &entry)
{
    return reinterpret_cast<accblas::dev::granule_word *>(
        static_cast<char *>(entry.ptr) + workspace_header_bytes);
}

}  // namespace


int accblas::trsv_get_workspace(hipStream_t stream, std::size_t bytes,
                                accblas::dev::granule_word **out)
{
    std::lock_guard<std::mutex> lock(workspace_guard);
    workspace_entry *entry = nullptr;
    const int status = ensure_workspace(stream, bytes, &entry);
    if (status == ACCBLAS_OK) {
        *out = granules_of(*entry);
    }
    return status;
}

int accblas::trsv_get_session(hipStream_t stream, std::size_t bytes,
                              accblas::dev::trsv_session *out)
{
    std::lock_guard<std::mutex> lock(workspace_guard);
    workspace_entry *entry = nullptr;
    const int status = ensure_workspace(stream, bytes, &entry);
    if (status == ACCBLAS_OK) {
        out->state = static_cast<accblas::dev::trsv_state *>(entry->ptr);
        out->granules = granules_of(*entry);
        out->capacity_words = static_cast<accblas::dev::index_type>(
            (entry->bytes - workspace_header_bytes) /
            sizeof(accblas::dev::granule_word));
    }
    return status;
}


namespace {



int validate_trsv(accblas_uplo uplo, accblas_diag diag, int64_t n,
                  const void *A, int64_t lda, const void *x, int64_t incx,
                  const uint32_t *helper)
{
    if (uplo != ACCBLAS_UPPER && uplo != ACCBLAS_LOWER) {
        return accblas::set_error(ACCBLAS_ERR_INVALID_ARGUMENT,
                                  "trsv: unknown uplo");
    }
    if (diag != ACCBLAS_UNIT && diag != ACCBLAS_NON_UNIT) {
        return accblas::set_error(ACCBLAS_ERR_INVALID_ARGUMENT,
                                  "trsv: unknown diag");
    }
    if (n < 0) {
        return accblas::set_error(ACCBLAS_ERR_INVALID_ARGUMENT, "trsv: n < 0");
    }
    if (lda < n) {
        return accblas::set_error(ACCBLAS_ERR_INVALID_ARGUMENT,
                                  "trsv: row stride smaller than n");
    }
    if (incx < 1) {
        return accblas::set_error(ACCBLAS_ERR_INVALID_ARGUMENT,
                                  "trsv: increment must be >= 1");
    }
    if (helper == nullptr) {
        return accblas::set_error(ACCBLAS_ERR_NULL_POINTER,
                                  "trsv: helper is NULL");
    }
    if (n > 0 && (A == nullptr || x == nullptr)) {
        return accblas::set_error(ACCBLAS_ERR_NULL_POINTER,
                                  "trsv: A or x is NULL");
    }
    return ACCBLAS_OK;
}


}  // namespace


extern "C" {


int accblas_trsv(accblas_dtype vt, accblas_uplo uplo, accblas_diag diag,
                 int64_t n, const void *A, int64_t lda, void *x, int64_t incx,
                 uint32_t *helper, void *stream)
{
    const int status = validate_trsv(uplo, diag, n, A, lda, x, incx, helper);
    if (status != ACCBLAS_OK) {
        return status;
    }
    return accblas::dispatch_arithmetic(vt, [&](auto tag) {
        using value_type = typename decltype(tag)::type;
        using namespace accblas::dev;
        const plain_view<const value_type> m_view{
            static_cast<const value_type *>(A), {n, n}, lda};
        const plain_view<value_type> x_view{static_cast<value_type *>(x),
                                            {n, 1}, incx};
        return accblas::trsv_dispatch<value_type, value_type>(
            uplo == ACCBLAS_UPPER, diag == ACCBLAS_UNIT, n, m_view, x_view,
            helper, stream);
    });
}


int accblas_acc_trsv(accblas_dtype ar, accblas_dtype st, accblas_uplo uplo,
                     accblas_diag diag, int64_t n, const void *A, int64_t lda,
                     void *x, int64_t incx, uint32_t *helper, void *stream)
{
    const int status = validate_trsv(uplo, diag, n, A, lda, x, incx, helper);
    if (status != ACCBLAS_OK) {
        return status;
    }
    if (ar == ACCBLAS_F32) {
        return accblas::acc_trsv_f32(st, uplo, diag, n, A, lda, x, incx,
                                     helper, stream);
    }
    if (ar == ACCBLAS_F64) {
        return accblas::acc_trsv_f64(st, uplo, diag, n, A, lda, x, incx,
                                     helper, stream);
    }
    return accblas::set_error(ACCBLAS_ERR_UNSUPPORTED_TYPES,
                              "arithmetic type must be f32 or f64");
}


}  // extern "C"
