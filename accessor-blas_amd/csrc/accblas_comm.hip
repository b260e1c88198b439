// libaccblas_rccl.so: the multi-GPU context of include/accblas_comm.h.
// Built only on the public C ABI (accblas.h) plus RCCL: local compute is the
// single-GPU entry point, the exchange is one RCCL collective on the caller's
// stream. The reference has no multi-GPU code; the sharding follows SURVEY.md
// §8e (row blocks for GEMV, chunks + one scalar all-reduce for dot).
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/accblas_comm.h"


struct accblas_comm {
    ncclComm_t comm;
    int world_size;
    int rank;
    int device;
    void *scalar;  // ring of scalar_slots 16-byte device slots, one per dot in
                   // flight: [0, 8) this rank's partial dot in `ar`, [8, 16)
                   // the reduced one when a cast must follow
    unsigned next_slot;    // slot the next accblas_comm_acc_dot takes
    int group_pending;     // dots recorded in the open group (their
                           // all-reduces have not been enqueued yet)
    bool always_collective;  // ACCBLAS_COMM_FORCE_COLLECTIVE=1: a world of one
                             // rank still runs the collectives (tests)
};


namespace {


// Every accblas_comm_acc_dot takes the next slot of the ring, so several dots
// may be in flight on one communicator: inside a group the chunk dots run at
// once while their all-reduces (and casts) wait for the group end, and a
// second dot must not overwrite what the first all-reduce has yet to read.
constexpr unsigned scalar_slots = 64;
constexpr size_t scalar_slot_bytes = 16;

thread_local std::string last_error;

// Calls bracketed by accblas_comm_group_begin/_end (one host thread driving
// several communicators): RCCL only RECORDS a collective issued inside a group
// and enqueues it at ncclGroupEnd, so anything that must run after the
// collective on the same stream is recorded here and enqueued by
// accblas_comm_group_end, after ncclGroupEnd.
struct pending_cast {
    int device;
    accblas_dtype dst_t;
    void *dst;
    accblas_dtype src_t;
    const void *src;
    void *stream;
};
thread_local int group_depth = 0;
thread_local std::vector<pending_cast> pending_casts;
thread_local std::vector<accblas_comm *> group_members;

int fail(int code, const std::string &message)
{
    last_error = message;
    return code;
}

int check_hip(hipError_t err, const char *what)
{
    if (err == hipSuccess) {
        return ACCBLAS_OK;
    }
    return fail(static_cast<int>(err),
                std::string(what) + ": " + hipGetErrorString(err));
}

int check_rccl(ncclResult_t res, const char *what)
{
    if (res == ncclSuccess) {
        return ACCBLAS_OK;
    }
    return fail(ACCBLAS_ERR_COMM,
                std::string(what) + ": " + ncclGetErrorString(res));
}

/** Forwards a failure of a libaccblas_hip call with its message. */
int check_local(int status)
{
    if (status != ACCBLAS_OK) {
        last_error = accblas_last_error_string();
    }
    return status;
}

bool rccl_type(accblas_dtype t, ncclDataType_t *out)
{
    switch (t) {
    case ACCBLAS_F16:
        *out = ncclFloat16;
        return true;
    case ACCBLAS_F32:
        *out = ncclFloat32;
        return true;
    case ACCBLAS_F64:
        *out = ncclFloat64;
        return true;
    }
    return false;
}

/** Binds `device` for the lifetime of the object, restores the previous one. */
class device_guard {
public:
    explicit device_guard(int device) : previous_{-1}, status_{ACCBLAS_OK}
    {
        status_ = check_hip(hipGetDevice(&previous_), "hipGetDevice");
        if (status_ == ACCBLAS_OK && previous_ != device) {
            status_ = check_hip(hipSetDevice(device), "hipSetDevice");
        } else {
            previous_ = -1;
        }
    }
    ~device_guard()
    {
        if (previous_ >= 0) {
            (void)hipSetDevice(previous_);
        }
    }
    int status() const { return status_; }

private:
    int previous_;
    int status_;
};

int make_comm(accblas_comm **out, ncclComm_t raw, int world_size, int rank,
              int device)
{
    void *scalar = nullptr;
    const int status = check_hip(
        hipMalloc(&scalar, scalar_slots * scalar_slot_bytes), "hipMalloc");
    if (status != ACCBLAS_OK) {
        return status;
    }
    const char *force = std::getenv("ACCBLAS_COMM_FORCE_COLLECTIVE");
    *out = new accblas_comm{raw,    world_size, rank, device,
                            scalar, 0u,         0,    force != nullptr && force[0] == '1'};
    return ACCBLAS_OK;
}


}  // namespace


extern "C" {


const char *accblas_comm_last_error_string(void) { return last_error.c_str(); }

int accblas_comm_unique_id(void *id)
{
    if (id == nullptr) {
        return fail(ACCBLAS_ERR_NULL_POINTER, "id is NULL");
    }
    static_assert(sizeof(ncclUniqueId) == ACCBLAS_COMM_ID_BYTES,
                  "ACCBLAS_COMM_ID_BYTES must match RCCL's id");
    ncclUniqueId uid;
    const int status = check_rccl(ncclGetUniqueId(&uid), "ncclGetUniqueId");
    if (status == ACCBLAS_OK) {
        std::memcpy(id, &uid, sizeof(uid));
    }
    return status;
}

int accblas_comm_create(accblas_comm **comm, int world_size, int rank,
                        const void *id)
{
    if (comm == nullptr || id == nullptr) {
        return fail(ACCBLAS_ERR_NULL_POINTER, "comm or id is NULL");
    }
    if (world_size < 1 || rank < 0 || rank >= world_size) {
        return fail(ACCBLAS_ERR_INVALID_ARGUMENT,
                    "need world_size >= 1 and 0 <= rank < world_size");
    }
    int device = 0;
    int status = check_hip(hipGetDevice(&device), "hipGetDevice");
    if (status != ACCBLAS_OK) {
        return status;
    }
    ncclUniqueId uid;
    std::memcpy(&uid, id, sizeof(uid));
    ncclComm_t raw{};
    status = check_rccl(ncclCommInitRank(&raw, world_size, uid, rank),
                        "ncclCommInitRank");
    if (status != ACCBLAS_OK) {
        return status;
    }
    status = make_comm(comm, raw, world_size, rank, device);
    if (status != ACCBLAS_OK) {
        (void)ncclCommDestroy(raw);
    }
    return status;
}

int accblas_comm_create_all(accblas_comm **comms, int n_gpus,
                            const int *devices)
{
    if (comms == nullptr) {
        return fail(ACCBLAS_ERR_NULL_POINTER, "comms is NULL");
    }
    if (n_gpus < 1) {
        return fail(ACCBLAS_ERR_INVALID_ARGUMENT, "need n_gpus >= 1");
    }
    std::vector<int> devs(n_gpus);
    for (int g = 0; g < n_gpus; ++g) {
        devs[g] = devices != nullptr ? devices[g] : g;
    }
    int visible = 0;
    int status = check_hip(hipGetDeviceCount(&visible), "hipGetDeviceCount");
    if (status != ACCBLAS_OK) {
        return status;
    }
    for (int g = 0; g < n_gpus; ++g) {
        if (devs[g] < 0 || devs[g] >= visible) {
            return fail(ACCBLAS_ERR_INVALID_ARGUMENT,
                        "device index outside the visible devices");
        }
    }
    std::vector<ncclComm_t> raw(n_gpus);
    status = check_rccl(ncclCommInitAll(raw.data(), n_gpus, devs.data()),
                        "ncclCommInitAll");
    if (status != ACCBLAS_OK) {
        return status;
    }
    for (int g = 0; g < n_gpus; ++g) {
        comms[g] = nullptr;
    }
    for (int g = 0; g < n_gpus && status == ACCBLAS_OK; ++g) {
        device_guard guard(devs[g]);
        status = guard.status();
        if (status == ACCBLAS_OK) {
            status = make_comm(&comms[g], raw[g], n_gpus, g, devs[g]);
        }
    }
    if (status != ACCBLAS_OK) {
        const std::string keep = last_error;
        for (int g = 0; g < n_gpus; ++g) {
            if (comms[g] != nullptr) {
                (void)accblas_comm_destroy(comms[g]);
                comms[g] = nullptr;
            } else {
                (void)ncclCommDestroy(raw[g]);
            }
        }
        last_error = keep;
    }
    return status;
}

int accblas_comm_destroy(accblas_comm *comm)
{
    if (comm == nullptr) {
        return ACCBLAS_OK;
    }
    int status = ACCBLAS_OK;
    {
        device_guard guard(comm->device);
        status = guard.status();
        if (status == ACCBLAS_OK) {
            status = check_hip(hipDeviceSynchronize(), "hipDeviceSynchronize");
        }
        const int free_status = check_hip(hipFree(comm->scalar), "hipFree");
        if (status == ACCBLAS_OK) {
            status = free_status;
        }
    }
    const int comm_status =
        check_rccl(ncclCommDestroy(comm->comm), "ncclCommDestroy");
    delete comm;
    return status != ACCBLAS_OK ? status : comm_status;
}

int accblas_comm_rank(const accblas_comm *comm, int *rank)
{
    if (comm == nullptr || rank == nullptr) {
        return fail(ACCBLAS_ERR_NULL_POINTER, "comm or rank is NULL");
    }
    *rank = comm->rank;
    return ACCBLAS_OK;
}

int accblas_comm_world_size(const accblas_comm *comm, int *world_size)
{
    if (comm == nullptr || world_size == nullptr) {
        return fail(ACCBLAS_ERR_NULL_POINTER, "comm or world_size is NULL");
    }
    *world_size = comm->world_size;
    return ACCBLAS_OK;
}

int accblas_comm_device(const accblas_comm *comm, int *device)
{
    if (comm == nullptr || device == nullptr) {
        return fail(ACCBLAS_ERR_NULL_POINTER, "comm or device is NULL");
    }
    *device = comm->device;
    return ACCBLAS_OK;
}

int accblas_comm_group_begin(void)
{
    const int status = check_rccl(ncclGroupStart(), "ncclGroupStart");
    if (status == ACCBLAS_OK) {
        ++group_depth;
    }
    return status;
}

int accblas_comm_group_end(void)
{
    int status = check_rccl(ncclGroupEnd(), "ncclGroupEnd");
    if (group_depth > 0) {
        --group_depth;
    }
    if (group_depth > 0) {
        return status;
    }
    // the collectives of the group are enqueued now: what has to follow them
    // in stream order can be
    for (const pending_cast &cast : pending_casts) {
        if (status != ACCBLAS_OK) {
            break;
        }
        device_guard guard(cast.device);
        status = guard.status();
        if (status == ACCBLAS_OK) {
            status = check_local(accblas_convert(cast.dst_t, cast.dst,
                                                 cast.src_t, cast.src, 1,
                                                 cast.stream));
        }
    }
    pending_casts.clear();
    for (accblas_comm *member : group_members) {
        member->group_pending = 0;
    }
    group_members.clear();
    return status;
}


int accblas_shard_bounds(int64_t n, int world_size, int rank, int64_t align,
                         int64_t *begin, int64_t *end)
{
    if (begin == nullptr || end == nullptr) {
        return fail(ACCBLAS_ERR_NULL_POINTER, "begin or end is NULL");
    }
    if (n < 0 || align < 0 || world_size < 1 || rank < 0 ||
        rank >= world_size) {
        return fail(ACCBLAS_ERR_INVALID_ARGUMENT,
                    "need n, align >= 0 and 0 <= rank < world_size");
    }
    const int64_t world = world_size;
    const int64_t base = n / world;
    const int64_t extra = n % world;
    const auto boundary = [&](int64_t r) {
        if (r >= world) {
            return n;
        }
        const int64_t edge = r * base + (r < extra ? r : extra);
        return align > 1 ? edge - edge % align : edge;
    };
    *begin = boundary(rank);
    *end = boundary(rank + int64_t{1});
    return ACCBLAS_OK;
}

int accblas_comm_shard(const accblas_comm *comm, int64_t n, int64_t align,
                       int64_t *begin, int64_t *end)
{
    if (comm == nullptr) {
        return fail(ACCBLAS_ERR_NULL_POINTER, "comm is NULL");
    }
    return accblas_shard_bounds(n, comm->world_size, comm->rank, align, begin,
                                end);
}

int accblas_comm_acc_dot(accblas_comm *comm, accblas_handle *handle,
                         accblas_dtype ar, accblas_dtype st,
                         accblas_dtype res_t, int64_t n_local, const void *x,
                         int64_t incx, const void *y, int64_t incy, void *res,
                         void *stream)
{
    if (comm == nullptr || res == nullptr) {
        return fail(ACCBLAS_ERR_NULL_POINTER, "comm or res is NULL");
    }
    ncclDataType_t wire{};
    if ((ar != ACCBLAS_F32 && ar != ACCBLAS_F64) || !rccl_type(ar, &wire)) {
        return fail(ACCBLAS_ERR_UNSUPPORTED_TYPES,
                    "arithmetic type must be f32 or f64");
    }
    device_guard guard(comm->device);
    if (guard.status() != ACCBLAS_OK) {
        return guard.status();
    }
    const bool collective = comm->world_size > 1 || comm->always_collective;
    if (!collective) {
        // a world of one rank: the chunk dot writes the result directly
        return check_local(accblas_acc_dot(handle, ar, st, res_t, n_local, x,
                                           incx, y, incy, res, stream));
    }
    if (group_depth > 0) {
        if (comm->group_pending >= static_cast<int>(scalar_slots)) {
            return fail(ACCBLAS_ERR_INVALID_ARGUMENT,
                        "more than 64 dots on one communicator inside one "
                        "group: close the group first");
        }
        if (comm->group_pending++ == 0) {
            group_members.push_back(comm);
        }
    }
    char *const slot = static_cast<char *>(comm->scalar) +
                       (comm->next_slot++ % scalar_slots) * scalar_slot_bytes;
    void *const partial = slot;
    void *const reduced = slot + 8;
    // chunk -> device scalar in the ARITHMETIC type (an empty chunk yields 0)
    int status = check_local(accblas_acc_dot(handle, ar, st, ar, n_local, x,
                                             incx, y, incy, partial, stream));
    if (status != ACCBLAS_OK) {
        return status;
    }
    // Stream order: chunk dot -> all-reduce -> (cast). When the result type
    // IS the arithmetic type the all-reduce writes `res` itself and nothing
    // follows it. Otherwise the cast must come after the exchange; inside a
    // group (accblas_comm_group_begin) the all-reduce is only enqueued by
    // ncclGroupEnd, so the cast is deferred to accblas_comm_group_end.
    const bool direct = res_t == ar;
    status = check_rccl(
        ncclAllReduce(partial, direct ? res : reduced, 1, wire, ncclSum,
                      comm->comm, static_cast<hipStream_t>(stream)),
        "ncclAllReduce");
    if (status != ACCBLAS_OK || direct) {
        return status;
    }
    if (group_depth > 0) {
        pending_casts.push_back(
            pending_cast{comm->device, res_t, res, ar, reduced, stream});
        return ACCBLAS_OK;
    }
    return check_local(accblas_convert(res_t, res, ar, reduced, 1, stream));
}

int accblas_comm_acc_gemv(accblas_comm *comm, accblas_dtype ar,
                          accblas_dtype st, int64_t rows_local, int64_t cols,
                          double alpha, const void *A, int64_t lda,
                          const void *x, int64_t incx, double beta,
                          void *res_local, int64_t incres, void *res_full,
                          void *stream)
{
    if (comm == nullptr) {
        return fail(ACCBLAS_ERR_NULL_POINTER, "comm is NULL");
    }
    ncclDataType_t wire{};
    if (res_full != nullptr) {
        if (!rccl_type(st, &wire)) {
            return fail(ACCBLAS_ERR_INVALID_ARGUMENT, "unknown storage type");
        }
        if (incres != 1) {
            return fail(ACCBLAS_ERR_INVALID_ARGUMENT,
                        "the all-gather needs a contiguous result slice");
        }
    }
    device_guard guard(comm->device);
    if (guard.status() != ACCBLAS_OK) {
        return guard.status();
    }
    int status = check_local(accblas_acc_gemv(ar, st, rows_local, cols, alpha,
                                              A, lda, x, incx, beta, res_local,
                                              incres, stream));
    if (status != ACCBLAS_OK || res_full == nullptr) {
        return status;
    }
    if (comm->world_size == 1 && !comm->always_collective) {
        if (res_full == res_local) {
            return ACCBLAS_OK;
        }
        return check_hip(
            hipMemcpyAsync(res_full, res_local,
                           static_cast<size_t>(rows_local) *
                               accblas_dtype_size(st),
                           hipMemcpyDeviceToDevice,
                           static_cast<hipStream_t>(stream)),
            "hipMemcpyAsync");
    }
    return check_rccl(
        ncclAllGather(res_local, res_full, static_cast<size_t>(rows_local),
                      wire, comm->comm, static_cast<hipStream_t>(stream)),
        "ncclAllGather");
}


}  // extern "C"
