// C-ABI runtime entry points: diagnostics, device, memory, streams, events,
// the handle and the device-side convert. HIP replacements for the CUDA
// runtime use in `/root/reference/cuda/utils.cuh:80-182`,
// `cuda/memory.cuh:28-121` and `myBlasHandle` (`cuda/dot_kernels.cuh:29-65`).
#include <cstring>

#include "accblas_internal.hpp"


namespace accblas {


namespace {
thread_local std::string last_error = "no error";
}


int set_error(int code, const std::string &message)
{
    last_error = message;
    return code;
}

int check_hip(hipError_t err, const char *what)
{
    if (err == hipSuccess) {
        return ACCBLAS_OK;
    }
    return set_error(static_cast<int>(err),
                     std::string(what) + ": " + hipGetErrorString(err));
}


namespace kernel {


template <typename Dst, typename Src>
__global__ __launch_bounds__(256) void convert(Dst *dst, const Src *src,
                                               std::int64_t n)
{
    const std::int64_t stride = std::int64_t{gridDim.x} * blockDim.x;
    for (std::int64_t i = std::int64_t{blockIdx.x} * blockDim.x + threadIdx.x;
         i < n; i += stride) {
        dst[i] = static_cast<Dst>(src[i]);
    }
}


}  // namespace kernel
}  // namespace accblas


using accblas::check_hip;
using accblas::set_error;


extern "C" {


const char *accblas_last_error_string(void)
{
    return accblas::last_error.c_str();
}

int accblas_version(void) { return 100; }

size_t accblas_dtype_size(accblas_dtype t)
{
    switch (t) {
    case ACCBLAS_F16:
        return 2;
    case ACCBLAS_F32:
        return 4;
    case ACCBLAS_F64:
        return 8;
    case ACCBLAS_I16:
        return 2;
    case ACCBLAS_I32:
        return 4;
    }
    return 0;
}

int accblas_device_count(int *count)
{
    if (count == nullptr) {
        return set_error(ACCBLAS_ERR_NULL_POINTER, "count is NULL");
    }
    return check_hip(hipGetDeviceCount(count), "hipGetDeviceCount");
}

int accblas_set_device(int device)
{
    return check_hip(hipSetDevice(device), "hipSetDevice");
}

int accblas_get_device(int *device)
{
    if (device == nullptr) {
        return set_error(ACCBLAS_ERR_NULL_POINTER, "device is NULL");
    }
    return check_hip(hipGetDevice(device), "hipGetDevice");
}

int accblas_device_info(int device, int *num_cus, int *wavefront_size,
                        size_t *total_mem_bytes, char *name, size_t name_len)
{
    hipDeviceProp_t prop;
    const int status =
        check_hip(hipGetDeviceProperties(&prop, device), "hipGetDeviceProperties");
    if (status != ACCBLAS_OK) {
        return status;
    }
    if (num_cus != nullptr) {
        *num_cus = prop.multiProcessorCount;
    }
    if (wavefront_size != nullptr) {
        *wavefront_size = prop.warpSize;
    }
    if (total_mem_bytes != nullptr) {
        *total_mem_bytes = prop.totalGlobalMem;
    }
    if (name != nullptr && name_len > 0) {
        std::strncpy(name, prop.name, name_len - 1);
        name[name_len - 1] = '\0';
    }
    return ACCBLAS_OK;
}

int accblas_device_synchronize(void)
{
    return check_hip(hipDeviceSynchronize(), "hipDeviceSynchronize");
}

int accblas_malloc(void **ptr, size_t bytes)
{
    if (ptr == nullptr) {
        return set_error(ACCBLAS_ERR_NULL_POINTER, "ptr is NULL");
    }
    *ptr = nullptr;
    if (bytes == 0) {
        return ACCBLAS_OK;
    }
    return check_hip(hipMalloc(ptr, bytes), "hipMalloc");
}

int accblas_free(void *ptr)
{
    if (ptr == nullptr) {
        return ACCBLAS_OK;
    }
    return check_hip(hipFree(ptr), "hipFree");
}

int accblas_memcpy_h2d(void *dst, const void *src, size_t bytes)
{
    if (bytes == 0) {
        return ACCBLAS_OK;
    }
    return check_hip(hipMemcpy(dst, src, bytes, hipMemcpyHostToDevice),
                     "hipMemcpy(H2D)");
}

int accblas_memcpy_d2h(void *dst, const void *src, size_t bytes)
{
    if (bytes == 0) {
        return ACCBLAS_OK;
    }
    return check_hip(hipMemcpy(dst, src, bytes, hipMemcpyDeviceToHost),
                     "hipMemcpy(D2H)");
}

int accblas_memcpy_d2d(void *dst, const void *src, size_t bytes)
{
    if (bytes == 0) {
        return ACCBLAS_OK;
    }
    return check_hip(hipMemcpy(dst, src, bytes, hipMemcpyDeviceToDevice),
                     "hipMemcpy(D2D)");
}

int accblas_memset(void *dst, int value, size_t bytes)
{
    if (bytes == 0) {
        return ACCBLAS_OK;
    }
    return check_hip(hipMemset(dst, value, bytes), "hipMemset");
}

int accblas_stream_create(void **stream)
{
    if (stream == nullptr) {
        return set_error(ACCBLAS_ERR_NULL_POINTER, "stream is NULL");
    }
    hipStream_t s;
    const int status = check_hip(hipStreamCreate(&s), "hipStreamCreate");
    *stream = status == ACCBLAS_OK ? s : nullptr;
    return status;
}

int accblas_stream_destroy(void *stream)
{
    return check_hip(hipStreamDestroy(accblas::as_stream(stream)),
                     "hipStreamDestroy");
}

int accblas_stream_synchronize(void *stream)
{
    return check_hip(hipStreamSynchronize(accblas::as_stream(stream)),
                     "hipStreamSynchronize");
}

int accblas_event_create(void **event)
{
    if (event == nullptr) {
        return set_error(ACCBLAS_ERR_NULL_POINTER, "event is NULL");
    }
    hipEvent_t e;
    const int status = check_hip(hipEventCreate(&e), "hipEventCreate");
    *event = status == ACCBLAS_OK ? e : nullptr;
    return status;
}

int accblas_event_destroy(void *event)
{
    return check_hip(hipEventDestroy(static_cast<hipEvent_t>(event)),
                     "hipEventDestroy");
}

int accblas_event_record(void *event, void *stream)
{
    return check_hip(hipEventRecord(static_cast<hipEvent_t>(event),
                                    accblas::as_stream(stream)),
                     "hipEventRecord");
}

int accblas_event_synchronize(void *event)
{
    return check_hip(hipEventSynchronize(static_cast<hipEvent_t>(event)),
                     "hipEventSynchronize");
}

int accblas_event_elapsed_ms(float *ms, void *start, void *stop)
{
    if (ms == nullptr) {
        return set_error(ACCBLAS_ERR_NULL_POINTER, "ms is NULL");
    }
    return check_hip(hipEventElapsedTime(ms, static_cast<hipEvent_t>(start),
                                         static_cast<hipEvent_t>(stop)),
                     "hipEventElapsedTime");
}

int accblas_handle_create(accblas_handle **handle)
{
    if (handle == nullptr) {
        return set_error(ACCBLAS_ERR_NULL_POINTER, "handle is NULL");
    }
    *handle = nullptr;
    int device = 0;
    int status = check_hip(hipGetDevice(&device), "hipGetDevice");
    if (status != ACCBLAS_OK) {
        return status;
    }
    hipDeviceProp_t prop;
    status = check_hip(hipGetDeviceProperties(&prop, device),
                       "hipGetDeviceProperties");
    if (status != ACCBLAS_OK) {
        return status;
    }
    constexpr size_t slot_bytes = accblas::dev::dot_slot_bytes;
    constexpr size_t total_bytes = slot_bytes + 3 * 16;
    void *workspace = nullptr;
    status = check_hip(hipMalloc(&workspace, total_bytes), "hipMalloc");
    if (status != ACCBLAS_OK) {
        return status;
    }
    status = check_hip(hipMemset(workspace, 0, total_bytes), "hipMemset");
    if (status != ACCBLAS_OK) {
        (void)hipFree(workspace);
        return status;
    }
    auto *h = new accblas_handle{};
    h->device = device;
    h->num_cus = prop.multiProcessorCount;
    h->dot_mode = ACCBLAS_DOT_DETERMINISTIC;
    h->workspace = workspace;
    auto *bytes = static_cast<char *>(workspace);
    h->dot_ws.slots = reinterpret_cast<accblas::dev::dot_slot_word *>(bytes);
    h->dot_ws.scalar = bytes + slot_bytes;
    h->dot_ws.num_cus = h->num_cus;
    h->dot_ws.serial = 1;
    h->trsv_helper = reinterpret_cast<std::uint32_t *>(bytes + slot_bytes + 16);
    h->dot_ws.status =
        reinterpret_cast<std::uint32_t *>(bytes + slot_bytes + 32);
    *handle = h;
    return ACCBLAS_OK;
}

int accblas_handle_destroy(accblas_handle *handle)
{
    if (handle == nullptr) {
        return ACCBLAS_OK;
    }
    // as Memory<T>'s destructor (cuda/memory.cuh:60-63): wait for the device
    // before releasing memory a kernel may still use
    (void)hipDeviceSynchronize();
    const int status = check_hip(hipFree(handle->workspace), "hipFree");
    delete handle;
    return status;
}

int accblas_handle_num_cus(const accblas_handle *handle, int *num_cus)
{
    if (handle == nullptr || num_cus == nullptr) {
        return set_error(ACCBLAS_ERR_NULL_POINTER, "handle or num_cus is NULL");
    }
    *num_cus = handle->num_cus;
    return ACCBLAS_OK;
}

int accblas_handle_set_dot_mode(accblas_handle *handle, accblas_dot_mode mode)
{
    if (handle == nullptr) {
        return set_error(ACCBLAS_ERR_NULL_POINTER, "handle is NULL");
    }
    if (mode != ACCBLAS_DOT_DETERMINISTIC && mode != ACCBLAS_DOT_ATOMIC) {
        return set_error(ACCBLAS_ERR_INVALID_ARGUMENT, "unknown dot mode");
    }
    handle->dot_mode = mode;
    return ACCBLAS_OK;
}

int accblas_handle_device_scratch(accblas_handle *handle, void **ptr)
{
    if (handle == nullptr || ptr == nullptr) {
        return set_error(ACCBLAS_ERR_NULL_POINTER, "handle or ptr is NULL");
    }
    *ptr = handle->dot_ws.scalar;
    return ACCBLAS_OK;
}

int accblas_handle_trsv_helper(accblas_handle *handle, uint32_t **helper)
{
    if (handle == nullptr || helper == nullptr) {
        return set_error(ACCBLAS_ERR_NULL_POINTER, "handle or helper is NULL");
    }
    *helper = handle->trsv_helper;
    return ACCBLAS_OK;
}

int accblas_convert(accblas_dtype dst_t, void *dst, accblas_dtype src_t,
                    const void *src, int64_t n, void *stream)
{
    if (n < 0) {
        return set_error(ACCBLAS_ERR_INVALID_ARGUMENT, "n < 0");
    }
    if (n == 0) {
        return ACCBLAS_OK;
    }
    if (dst == nullptr || src == nullptr) {
        return set_error(ACCBLAS_ERR_NULL_POINTER, "dst or src is NULL");
    }
    return accblas::dispatch_type(dst_t, [&](auto dst_tag) {
        using dst_type = typename decltype(dst_tag)::type;
        return accblas::dispatch_type(src_t, [&](auto src_tag) {
            using src_type = typename decltype(src_tag)::type;
            const std::int64_t blocks =
                std::min<std::int64_t>((n + 255) / 256, 256 * 16);
            accblas::kernel::convert<<<dim3(static_cast<unsigned>(blocks)),
                                       dim3(256), 0,
                                       accblas::as_stream(stream)>>>(
                static_cast<dst_type *>(dst),
                static_cast<const src_type *>(src), n);
            return check_hip(hipGetLastError(), "convert launch");
        });
    });
}


}  // extern "C"
