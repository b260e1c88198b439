// C-ABI TRSV entry points (include/accblas.h). Views are built as the
// reference launchers do (`/root/reference/cuda/trsv_kernels.cuh:455-488`
// plain, `:918-961` accessor).
#include <array>
#include <atomic>
#include <map>
#include <mutex>
#include <utility>
#include <vector>

#include "accblas_trsv_common.hpp"


namespace {
std::atomic<int> trsv_wide_option{1};
std::atomic<int> trsv_relay_option{1};
std::atomic<int> trsv_split_from_option{0};
std::atomic<int> trsv_prefetch_option{1};
std::atomic<int> trsv_wide_persistent_option{0};
std::atomic<int> trsv_small_option{1};
std::atomic<int> trsv_vector_rows_option{1};
std::atomic<int> gemv_skew_option_store{1};
std::atomic<int> gemv_vector_rows_option{1};
}

int accblas::gemv_skew_option_value() { return gemv_skew_option_store.load(); }
int accblas::gemv_vector_rows_option_value()
{
    return gemv_vector_rows_option.load();
}
int accblas::trsv_wide_option_value() { return trsv_wide_option.load(); }
int accblas::trsv_relay_option_value() { return trsv_relay_option.load(); }
int accblas::trsv_split_from_option_value()
{
    return trsv_split_from_option.load();
}

int accblas::trsv_prefetch_option_value() { return trsv_prefetch_option.load(); }
int accblas::trsv_small_option_value() { return trsv_small_option.load(); }
int accblas::trsv_wide_persistent_option_value()
{
    return trsv_wide_persistent_option.load();
}
int accblas::trsv_vector_rows_option_value()
{
    return trsv_vector_rows_option.load();
}

int accblas::trsv_num_cus()
{
    // per device: a process may drive several
    static std::mutex guard;
    static std::map<int, int> cached;
    int device = 0;
    if (hipGetDevice(&device) != hipSuccess) {
        return 1;
    }
    std::lock_guard<std::mutex> lock(guard);
    auto found = cached.find(device);
    if (found == cached.end()) {
        int count = 0;
        if (hipDeviceGetAttribute(&count, hipDeviceAttributeMultiprocessorCount,
                                  device) != hipSuccess ||
            count <= 0) {
            count = 1;
        }
        found = cached.emplace(device, count).first;
    }
    return found->second;
}

extern "C" int accblas_set_option(const char *name, int value)
{
    const std::string key = name != nullptr ? name : "";
    if (key == "trsv_wide_split_from") {
        if (value < 0) {
            return accblas::set_error(ACCBLAS_ERR_INVALID_ARGUMENT,
                                      "trsv_wide_split_from takes a count >= 0");
        }
        trsv_split_from_option.store(value);
        return ACCBLAS_OK;
    }
    std::atomic<int> *target = nullptr;
    if (key == "trsv_wide") {
        target = &trsv_wide_option;
    } else if (key == "trsv_relay") {
        target = &trsv_relay_option;
    } else if (key == "trsv_prefetch") {
        target = &trsv_prefetch_option;
    } else if (key == "trsv_wide_persistent") {
        target = &trsv_wide_persistent_option;
    } else if (key == "trsv_small") {
        target = &trsv_small_option;
    } else if (key == "trsv_vector_rows") {
        target = &trsv_vector_rows_option;
    } else if (key == "gemv_vector_rows") {
        target = &gemv_vector_rows_option;
    } else if (key == "gemv_skew") {
        target = &gemv_skew_option_store;
    } else {
        return accblas::set_error(ACCBLAS_ERR_INVALID_ARGUMENT,
                                  "accblas_set_option: unknown option");
    }
    if (value < 0 || value > 2) {
        return accblas::set_error(ACCBLAS_ERR_INVALID_ARGUMENT,
                                  key + " takes 0, 1 or 2");
    }
    target->store(value);
    return ACCBLAS_OK;
}


namespace {


// Hand-off workspace of the solve kernels: a 4 KiB header (the `trsv_state`
// of the kernels) followed by 8-byte {tag, value} granules, one or two per
// element of x. The reference's launcher signature has no handle to keep it
// in (its `helper` is two words), so for the drop-in entry points the library
// owns one buffer per (device, stream) and grows it on demand. Lifetime rules:
//   * a buffer that was handed to a launch is never freed or re-used for
//     anything else until accblas_trsv_release_workspace(stream): growth
//     allocates a NEW buffer and retires the old one, so a hipGraph that
//     captured a solve keeps replaying on memory that is still there (its
//     own state word and granules; solves on different buffers of one stream
//     do not interact);
//   * the lock is held from "pointer taken" to "kernel enqueued"
//     (`trsv_workspace_lease`);
//   * growth needs hipMalloc + a memset: it is refused while the stream is
//     being captured (run one eager solve of that size first, or pass your own
//     workspace: accblas_acc_trsv_ws).
// A new buffer is zeroed once; the kernels re-arm it themselves (epoch tags).
struct workspace_entry {
    void *ptr{nullptr};
    std::size_t bytes{0};
    std::vector<void *> retired;
};

std::mutex workspace_guard;
std::map<std::pair<int, hipStream_t>, workspace_entry> workspace_cache;

int init_workspace(void *ptr, std::size_t bytes, hipStream_t stream)
{
    int status = accblas::check_hip(hipMemsetAsync(ptr, 0, bytes, stream),
                                    "hipMemsetAsync(trsv workspace)");
    if (status != ACCBLAS_OK) {
        return status;
    }
    // header = a clean `trsv_state`; the source of the copy must outlive the
    // asynchronous copy: static storage
    static const accblas::dev::trsv_state initial{
        accblas::dev::trsv_clean_tickets, 0u, accblas::dev::trsv_first_epoch,
        0u, 0u};
    return accblas::check_hip(
        hipMemcpyAsync(ptr, &initial, sizeof(initial), hipMemcpyHostToDevice,
                       stream),
        "hipMemcpyAsync(trsv state)");
}

accblas::dev::trsv_session session_of(void *ptr, std::size_t bytes)
{
    accblas::dev::trsv_session session{};
    session.state = static_cast<accblas::dev::trsv_state *>(ptr);
    session.granules = reinterpret_cast<accblas::dev::granule_word *>(
        static_cast<char *>(ptr) + accblas::trsv_workspace_header_bytes);
    session.capacity_words = static_cast<accblas::dev::index_type>(
        (bytes - accblas::trsv_workspace_header_bytes) /
        sizeof(accblas::dev::granule_word));
    return session;
}

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
    const std::size_t bytes =
        granule_bytes + accblas::trsv_workspace_header_bytes;
    if (entry.bytes < bytes) {
        hipStreamCaptureStatus capture = hipStreamCaptureStatusNone;
        if (hipStreamIsCapturing(stream, &capture) == hipSuccess &&
            capture != hipStreamCaptureStatusNone) {
            return accblas::set_error(
                ACCBLAS_ERR_INVALID_ARGUMENT,
                "trsv: the library-owned workspace of this stream would have "
                "to grow during stream capture; run one eager solve of this "
                "size first or use accblas_acc_trsv_ws");
        }
        const std::size_t rounded =
            std::max<std::size_t>(bytes + bytes / 2, std::size_t{1} << 16);
        void *fresh = nullptr;
        status = accblas::check_hip(hipMalloc(&fresh, rounded),
                                    "hipMalloc(trsv workspace)");
        if (status != ACCBLAS_OK) {
            return status;
        }
        status = init_workspace(fresh, rounded, stream);
        if (status != ACCBLAS_OK) {
            (void)hipFree(fresh);
            return status;
        }
        if (entry.ptr != nullptr) {
            entry.retired.push_back(entry.ptr);
        }
        entry.ptr = fresh;
        entry.bytes = rounded;
    }
    *out = &entry;
    return ACCBLAS_OK;
}

}  // namespace


int accblas::trsv_workspace_lease::acquire(hipStream_t stream,
                                           std::size_t granule_bytes)
{
    lock_ = std::unique_lock<std::mutex>(workspace_guard);
    workspace_entry *entry = nullptr;
    const int status = ensure_workspace(stream, granule_bytes, &entry);
    if (status == ACCBLAS_OK) {
        session_ = session_of(entry->ptr, entry->bytes);
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


namespace {


template <typename St>
std::size_t workspace_bytes_for(int64_t n)
{
    return accblas::trsv_workspace_total_bytes<St>(n);
}

int workspace_session(void *workspace, std::size_t workspace_bytes,
                      accblas::dev::trsv_session *out)
{
    if (workspace == nullptr) {
        return accblas::set_error(ACCBLAS_ERR_NULL_POINTER,
                                  "trsv: workspace is NULL");
    }
    if (workspace_bytes <= accblas::trsv_workspace_header_bytes ||
        !accblas::dev::is_aligned(workspace, 512)) {
        return accblas::set_error(
            ACCBLAS_ERR_INVALID_ARGUMENT,
            "trsv: workspace must be 512-byte aligned and larger than its "
            "4 KiB header");
    }
    *out = session_of(workspace, workspace_bytes);
    return ACCBLAS_OK;
}

int plain_trsv(accblas_dtype vt, accblas_uplo uplo, accblas_diag diag,
               int64_t n, const void *A, int64_t lda, void *x, int64_t incx,
               uint32_t *helper, void *stream,
               const accblas::dev::trsv_session *external)
{
    return accblas::dispatch_arithmetic(vt, [&](auto tag) {
        using value_type = typename decltype(tag)::type;
        using namespace accblas::dev;
        const plain_view<const value_type> m_view{
            static_cast<const value_type *>(A), {n, n}, lda};
        const plain_view<value_type> x_view{static_cast<value_type *>(x),
                                            {n, 1}, incx};
        return accblas::trsv_dispatch<value_type, value_type>(
            uplo == ACCBLAS_UPPER, diag == ACCBLAS_UNIT, n, m_view, x_view,
            helper, stream, external);
    });
}

int acc_trsv_any(accblas_dtype ar, accblas_dtype st, accblas_uplo uplo,
                 accblas_diag diag, int64_t n, const void *A, int64_t lda,
                 void *x, int64_t incx, uint32_t *helper, void *stream,
                 const accblas::dev::trsv_session *external)
{
    if (ar == ACCBLAS_F32) {
        return accblas::acc_trsv_f32(st, uplo, diag, n, A, lda, x, incx,
                                     helper, stream, external);
    }
    if (ar == ACCBLAS_F64) {
        return accblas::acc_trsv_f64(st, uplo, diag, n, A, lda, x, incx,
                                     helper, stream, external);
    }
    return accblas::set_error(ACCBLAS_ERR_UNSUPPORTED_TYPES,
                              "arithmetic type must be f32 or f64");
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
    return plain_trsv(vt, uplo, diag, n, A, lda, x, incx, helper, stream,
                      nullptr);
}


int accblas_acc_trsv(accblas_dtype ar, accblas_dtype st, accblas_uplo uplo,
                     accblas_diag diag, int64_t n, const void *A, int64_t lda,
                     void *x, int64_t incx, uint32_t *helper, void *stream)
{
    const int status = validate_trsv(uplo, diag, n, A, lda, x, incx, helper);
    if (status != ACCBLAS_OK) {
        return status;
    }
    return acc_trsv_any(ar, st, uplo, diag, n, A, lda, x, incx, helper, stream,
                        nullptr);
}


int accblas_trsv_status(const uint32_t *helper, void *stream, int *expired)
{
    if (helper == nullptr || expired == nullptr) {
        return accblas::set_error(ACCBLAS_ERR_NULL_POINTER,
                                  "trsv_status: helper or expired is NULL");
    }
    int status = accblas::check_hip(
        hipStreamSynchronize(accblas::as_stream(stream)),
        "hipStreamSynchronize");
    if (status != ACCBLAS_OK) {
        return status;
    }
    std::uint32_t word = 0;
    status = accblas::check_hip(
        hipMemcpy(&word, helper, sizeof(word), hipMemcpyDeviceToHost),
        "hipMemcpy(trsv helper)");
    if (status != ACCBLAS_OK) {
        return status;
    }
    *expired = word == accblas::dev::kernel::detail::trsv_error_word ? 1 : 0;
    return ACCBLAS_OK;
}


int accblas_trsv_workspace_bytes(accblas_dtype st, int64_t n, size_t *bytes)
{
    if (bytes == nullptr) {
        return accblas::set_error(ACCBLAS_ERR_NULL_POINTER,
                                  "trsv_workspace_bytes: bytes is NULL");
    }
    if (n < 0) {
        return accblas::set_error(ACCBLAS_ERR_INVALID_ARGUMENT,
                                  "trsv_workspace_bytes: n < 0");
    }
    return accblas::dispatch_type(st, [&](auto tag) {
        using st_type = typename decltype(tag)::type;
        *bytes = workspace_bytes_for<st_type>(n);
        return ACCBLAS_OK;
    });
}


int accblas_trsv_workspace_init(void *workspace, size_t workspace_bytes,
                                void *stream)
{
    accblas::dev::trsv_session session{};
    const int status = workspace_session(workspace, workspace_bytes, &session);
    if (status != ACCBLAS_OK) {
        return status;
    }
    return init_workspace(workspace, workspace_bytes,
                          accblas::as_stream(stream));
}


int accblas_trsv_ws(accblas_dtype vt, accblas_uplo uplo, accblas_diag diag,
                    int64_t n, const void *A, int64_t lda, void *x,
                    int64_t incx, uint32_t *helper, void *workspace,
                    size_t workspace_bytes, void *stream)
{
    int status = validate_trsv(uplo, diag, n, A, lda, x, incx, helper);
    if (status != ACCBLAS_OK) {
        return status;
    }
    accblas::dev::trsv_session session{};
    status = workspace_session(workspace, workspace_bytes, &session);
    if (status != ACCBLAS_OK) {
        return status;
    }
    return plain_trsv(vt, uplo, diag, n, A, lda, x, incx, helper, stream,
                      &session);
}


int accblas_acc_trsv_ws(accblas_dtype ar, accblas_dtype st, accblas_uplo uplo,
                        accblas_diag diag, int64_t n, const void *A,
                        int64_t lda, void *x, int64_t incx, uint32_t *helper,
                        void *workspace, size_t workspace_bytes, void *stream)
{
    int status = validate_trsv(uplo, diag, n, A, lda, x, incx, helper);
    if (status != ACCBLAS_OK) {
        return status;
    }
    accblas::dev::trsv_session session{};
    status = workspace_session(workspace, workspace_bytes, &session);
    if (status != ACCBLAS_OK) {
        return status;
    }
    return acc_trsv_any(ar, st, uplo, diag, n, A, lda, x, incx, helper, stream,
                        &session);
}


int accblas_trsv_release_workspace(void *stream)
{
    int device = 0;
    int status = accblas::check_hip(hipGetDevice(&device), "hipGetDevice");
    if (status != ACCBLAS_OK) {
        return status;
    }
    // detach the entry under the lock, wait and free outside it: launches on
    // other streams and devices take the same lock (their leases) and must
    // not stall for the length of this stream's work
    workspace_entry detached;
    {
        std::lock_guard<std::mutex> lock(workspace_guard);
        const auto found = workspace_cache.find(
            std::make_pair(device, accblas::as_stream(stream)));
        if (found == workspace_cache.end()) {
            return ACCBLAS_OK;
        }
        detached = std::move(found->second);
        workspace_cache.erase(found);
    }
    // solves still running on these buffers finish first
    status = accblas::check_hip(
        hipStreamSynchronize(accblas::as_stream(stream)),
        "hipStreamSynchronize");
    for (void *old : detached.retired) {
        (void)hipFree(old);
    }
    if (detached.ptr != nullptr) {
        (void)hipFree(detached.ptr);
    }
    return status;
}


}  // extern "C"
