/*
 * accblas.h — C ABI of the MI355X (gfx950) accessor-BLAS hot path.
 *
 * This is the drop-in boundary: every entry point replaces one
 * host launcher template (or runtime helper) of the reference's `cuda/` tree,
 * cited per function as `file:line` relative to /root/reference. Host code
 * (C++14, g++) never sees a HIP header: it includes this file and links
 * libaccblas_hip.so. Conventions:
 *   - every function returns an int status: 0 = ok, >0 = hipError_t value,
 *     <0 = ACCBLAS_ERR_*; nothing throws across the boundary;
 *     accblas_last_error_string() describes the last failure of the calling
 *     thread;
 *   - sizes, strides and increments are int64_t, counted in ELEMENTS;
 *   - device buffers are opaque `void *` / `const void *` owned by the caller;
 *   - alpha/beta travel as double and are narrowed to the arithmetic type
 *     inside;
 *   - `stream` is a hipStream_t passed as `void *`; NULL = the default stream;
 *   - launches are asynchronous: the caller synchronises (as the reference's
 *     `benchmark_function` does, cuda/utils.cuh:236-262).
 * fp16 buffers hold IEEE binary16 bit patterns.
 */
#ifndef ACCBLAS_H_
#define ACCBLAS_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- status codes ------------------------------------------------------ */
#define ACCBLAS_OK 0
#define ACCBLAS_ERR_UNSUPPORTED_TYPES (-1) /* (arithmetic, storage) pair not built */
#define ACCBLAS_ERR_INVALID_ARGUMENT (-2)
#define ACCBLAS_ERR_NULL_POINTER (-3)
#define ACCBLAS_ERR_INTERNAL (-4)

/* Element types. Storage: f16/f32/f64. Arithmetic: f32/f64. */
typedef enum accblas_dtype {
    ACCBLAS_F16 = 0,
    ACCBLAS_F32 = 1,
    ACCBLAS_F64 = 2,
    /* integer STORAGE types: accepted by the accblas_scaled_* entry points
     * only, every other call returns ACCBLAS_ERR_UNSUPPORTED_TYPES for them */
    ACCBLAS_I16 = 3,
    ACCBLAS_I32 = 4
} accblas_dtype;

/* Mirrors `enum class tmtx_t { upper, lower }` (cuda/trsv_kernels.cuh:22). */
typedef enum accblas_uplo { ACCBLAS_UPPER = 0, ACCBLAS_LOWER = 1 } accblas_uplo;

/* Mirrors `enum class dmtx_t { non_unit, unit }` (cuda/trsv_kernels.cuh:29). */
typedef enum accblas_diag {
    ACCBLAS_NON_UNIT = 0,
    ACCBLAS_UNIT = 1
} accblas_diag;

/* How the per-workgroup dot partials are combined. */
typedef enum accblas_dot_mode {
    /* one launch; every workgroup hands its partial over as a slot tagged
     * with the launch's serial number, the last workgroup of the grid collects
     * the slots and sums them in a fixed order: bitwise reproducible
     * (default). There are no counters to re-arm: whatever an earlier call -
     * finished, aborted or scribbled over - left in the handle's scratch
     * cannot be taken for a partial of this one. Should the collector's wait
     * ever give up (seconds; it cannot happen on a healthy device) the result
     * is NaN. */
    ACCBLAS_DOT_DETERMINISTIC = 0,
    /* reference behaviour: zero the result, then one atomic_add per
     * workgroup (cuda/dot_kernels.cuh:114,159; cuda/atomics.cuh:80-103) */
    ACCBLAS_DOT_ATOMIC = 1
} accblas_dot_mode;

/* Replaces `myBlasHandle` (cuda/dot_kernels.cuh:29-65): cached device
 * properties of the current device plus device scratch (dot partials, a
 * 16-byte result slot, the 2-word TRSV helper). A handle serialises its dot
 * calls: use one handle per stream (calls on one stream may share one). */
typedef struct accblas_handle accblas_handle;

/* ---- diagnostics ------------------------------------------------------- */
const char *accblas_last_error_string(void);
/* ABI version: major*10000 + minor*100 + patch */
int accblas_version(void);
size_t accblas_dtype_size(accblas_dtype t);

/* Tuning knobs (no counterpart in the reference, whose choices are
 * compile-time constants). The knobs take 0, 1 or 2 unless said otherwise.
 * Known names:
 *   "trsv_wide"   1 (default) = systems of 2- or 4-byte storage with
 *                 n >= 3328 use
 *                 128-row blocks (`kernel::trsv_wide`: half as many dependent
 *                 hops through memory, the folded panel held in registers),
 *                 everything else - 8-byte storage below n = 61440 - 64-row blocks
 *                 (`kernel::trsv`); 0 = 64-row blocks only; 2 = 128-row blocks
 *                 whenever the operands allow it.
 *   "trsv_relay"  64-row kernel only: 1 (default) = the far sweep's x goes
 *                 through a loader wave and LDS where that was measured to pay
 *                 (fp16 storage, fp32 arithmetic; fp64 arithmetic from n = 960
 *                 with 8-byte and from n = 1152 with 4-byte storage);
 *                 0 = never; 2 = always.
 *   "trsv_small"  1 (default) = systems of at most 128 unknowns are solved by
 *                 one workgroup without a workspace (`kernel::trsv_small`:
 *                 forward substitution in the wave; any alignment, row stride
 *                 and increment); 0 = the blocked kernels at every size.
 *   "trsv_vector_rows"  1 (default) = the blocked TRSV kernels read the matrix
 *                 with 16-byte loads whatever its base address and row stride
 *                 (the hardware takes them at any alignment); 0 = only for
 *                 16-byte aligned bases and strides, other operands run the
 *                 element-wise 64-row kernel (the rule until round 3).
 *   "gemv_vector_rows"  the same for GEMV: 1 (default) = rows (and x) of any
 *                 alignment are read with 16-byte loads - long rows on their
 *                 own 128-byte aligned column grid -; 0 = 8-byte loads or the
 *                 element-wise kernel for operands that are not 16-byte aligned.
 *   "gemv_skew"   1 (default) = GEMV with fp64 arithmetic and a power-of-two
 *                 row pitch (>= 16 KiB) uses the time-skewed sweep (rows of
 *                 a group walk the columns one chunk apart, against channel
 *                 aliasing of power-of-two pitches); 0 = never; 2 = whenever
 *                 the shape allows it.
 *   "trsv_wide_split_from"  (any count >= 0) the 128-row kernel gives the row
 *                 blocks that cannot be resident from the start - those beyond
 *                 the number of CUs, except the first (up to 64) of them -
 *                 several workgroups each (helpers sum shares of the far
 *                 columns, the owner adds their sums in a fixed order);
 *                 0 (default) = that rule, k > 0 = from row block k on (at
 *                 least 98).
 *   "trsv_prefetch"  128-row kernel: 1 (default) = solves of 8-byte storage
 *                 with 32 to 144 row blocks (n = 4096 ... 18432; reached with
 *                 "trsv_wide" = 2 only) launch extra workgroups on the CUs the
 *                 row blocks leave idle, which pull the tiles of the next block
 *                 columns into the Infinity Cache ahead of the chain's front
 *                 (12-26 % faster; same bits). 4-byte storage no longer: under
 *                 the front loop it is 2-25 % slower with them;
 *                 0 = never; 2 = whenever CUs are idle.
 *   "trsv_wide_persistent"  128-row kernel, solves with more tickets (row
 *                 blocks + helpers' shares) than CUs: 1 = one workgroup per CU
 *                 draws tickets until none is left (the hardware dispatcher
 *                 binds workgroup k of a launch to XCD k mod 8: with a
 *                 workgroup per ticket the last tickets start when a CU of
 *                 THEIR XCD frees up); 0 = one workgroup per ticket. Same bits.
 * Unknown names and other values return ACCBLAS_ERR_INVALID_ARGUMENT. */
int accblas_set_option(const char *name, int value);

/* ---- device / runtime (replaces cuda/utils.cuh:80-107, memory.cuh) ------ */
int accblas_device_count(int *count);
int accblas_set_device(int device);
int accblas_get_device(int *device);
/* name_len bytes are written at most (NUL terminated); any out pointer may
 * be NULL */
int accblas_device_info(int device, int *num_cus, int *wavefront_size,
                        size_t *total_mem_bytes, char *name, size_t name_len);
/* cudaDeviceSynchronize() of cuda/utils.cuh:107 */
int accblas_device_synchronize(void);

/* cudaMalloc / cudaFree / cudaMemcpy of cuda/memory.cuh:28-41,56-66,99-121 */
int accblas_malloc(void **ptr, size_t bytes);
int accblas_free(void *ptr);
int accblas_memcpy_h2d(void *dst, const void *src, size_t bytes);
int accblas_memcpy_d2h(void *dst, const void *src, size_t bytes);
int accblas_memcpy_d2d(void *dst, const void *src, size_t bytes);
int accblas_memset(void *dst, int value, size_t bytes);

/* streams */
int accblas_stream_create(void **stream);
int accblas_stream_destroy(void *stream);
int accblas_stream_synchronize(void *stream);

/* events: `cuda_event` / `CudaTimer` of cuda/utils.cuh:113-182 */
int accblas_event_create(void **event);
int accblas_event_destroy(void *event);
int accblas_event_record(void *event, void *stream);
int accblas_event_synchronize(void *event);
int accblas_event_elapsed_ms(float *ms, void *start, void *stop);

/* ---- handle ------------------------------------------------------------ */
int accblas_handle_create(accblas_handle **handle);
int accblas_handle_destroy(accblas_handle *handle);
/* number of compute units cached in the handle (multiProcessorCount of
 * cuda/dot_kernels.cuh:199) */
int accblas_handle_num_cus(const accblas_handle *handle, int *num_cus);
int accblas_handle_set_dot_mode(accblas_handle *handle, accblas_dot_mode mode);
/* 16-byte device scratch, `get_device_value_ptr` of cuda/dot_kernels.cuh:49-55 */
int accblas_handle_device_scratch(accblas_handle *handle, void **ptr);
/* 2 x uint32 device words usable as the TRSV helper */
int accblas_handle_trsv_helper(accblas_handle *handle, uint32_t **helper);

/* ---- GEMV: res = alpha * A * x + beta * res ------------------------------
 * A row-major rows x cols with row stride lda (elements), x(i) = x[i*incx],
 * res(i) = res[i*incres]. If beta == 0, res is not read. */

/* `gemv<ValueType>` (cuda/gemv_kernels.cuh:136-147 -> kernel :30-64):
 * plain-pointer kernel, everything in `vt`. */
int accblas_gemv(accblas_dtype vt, int64_t rows, int64_t cols, double alpha,
                 const void *A, int64_t lda, const void *x, int64_t incx,
                 double beta, void *res, int64_t incres, void *stream);

/* `acc_gemv<ArType, StType>` (cuda/gemv_kernels.cuh:168-193 -> kernel
 * :79-113): kernel driven through range<reduced_row_major<2, ar, st>>:
 * loads `st`, accumulates `ar`, stores `st`. */
int accblas_acc_gemv(accblas_dtype ar, accblas_dtype st, int64_t rows,
                     int64_t cols, double alpha, const void *A, int64_t lda,
                     const void *x, int64_t incx, double beta, void *res,
                     int64_t incres, void *stream);

/* ---- DOT: *res = sum_i x[i*incx] * y[i*incy] -----------------------------
 * `res` is a DEVICE scalar. */

/* `dot<ValueType>` (cuda/dot_kernels.cuh:192-206 -> kernels :78-116). */
int accblas_dot(accblas_handle *handle, accblas_dtype vt, int64_t n,
                const void *x, int64_t incx, const void *y, int64_t incy,
                void *res, void *stream);

/* `acc_dot<ArType, StType, ResType>` (cuda/dot_kernels.cuh:224-263 ->
 * kernels :131-173): loads `st`, accumulates `ar`, writes the device scalar
 * as `res_t`. */
int accblas_acc_dot(accblas_handle *handle, accblas_dtype ar, accblas_dtype st,
                    accblas_dtype res_t, int64_t n, const void *x,
                    int64_t incx, const void *y, int64_t incy, void *res,
                    void *stream);

/* Did a deterministic dot on `handle` give up? In the default mode the last
 * workgroup of the grid collects the other workgroups' partial sums and waits
 * for them (bounded: seconds). They need nothing from it, so the wait ends as
 * soon as they have run - but if another stream's kernel keeps them off the
 * CUs for that long, the collector writes NaN to *res instead of a partial
 * result, and this call says so: it synchronises `stream`, sets *expired to 1
 * if any dot on this handle gave up since the last call (and clears the
 * mark), else to 0. The mark is read and cleared by one atomic exchange in a
 * kernel on `stream`, followed by a copy and a synchronisation of that stream;
 * nothing runs on the null stream. It synchronises, so it must not be called
 * while `stream` is being captured into a graph. No reference counterpart (its
 * atomics cannot stall, cuda/dot_kernels.cuh:158-160); ACCBLAS_DOT_ATOMIC mode
 * never waits. */
int accblas_dot_status(accblas_handle *handle, void *stream, int *expired);

/* ---- other accessor kinds: scaled_reduced_row_major --------------------
 * The reference's README lists the accessor family its kernels accept ("all
 * others work accordingly", README.md:19); its launchers only build
 * reduced_row_major (cuda/gemv_kernels.cuh:178-189, dot_kernels.cuh:233-243).
 * These two build `range<scaled_reduced_row_major<2, ar, st, 0b10>>` - value =
 * (ar)storage x scalar, one arithmetic-type scalar per COLUMN index - for the
 * matrix of a GEMV / the first vector of a dot, and reduced_row_major for the
 * operands beside it. Scaled storage: I16, I32, F16 or F32; the other
 * operands: F32 or F64 storage; arithmetic F32 or F64 (anything else:
 * ACCBLAS_ERR_UNSUPPORTED_TYPES).
 *   gemv: res = alpha * A x + beta * res, A(i, j) = (ar)A[i * lda + j] *
 *         scalars[j]; `scalars` = cols device values of the arithmetic type.
 *   dot:  *res = sum_i ((ar)x[i * incx] * x_scalar[0]) * y[i * incy]; x is an
 *         (n, 1) range, its one column has ONE scalar (a device value of the
 *         arithmetic type); `res` = a device scalar of the arithmetic type. */
int accblas_scaled_acc_gemv(accblas_dtype ar, accblas_dtype a_st,
                            accblas_dtype v_st, int64_t rows, int64_t cols,
                            double alpha, const void *A, int64_t lda,
                            const void *scalars, const void *x, int64_t incx,
                            double beta, void *res, int64_t incres,
                            void *stream);
int accblas_scaled_acc_dot(accblas_handle *handle, accblas_dtype ar,
                           accblas_dtype x_st, accblas_dtype y_st, int64_t n,
                           const void *x, int64_t incx, const void *x_scalar,
                           const void *y, int64_t incy, void *res,
                           void *stream);

/* ---- TRSV: x <- T^{-1} x in place ----------------------------------------
 * T = upper/lower triangle of the row-major n x n matrix A (row stride lda),
 * unit or non-unit diagonal. `helper` = 2 device uint32 words owned by the
 * caller (cuda/trsv_benchmark.cu:98); NULL is not accepted. A solve is ONE
 * kernel launch (the reference's `trsv_init` launch, trsv_kernels.cuh:38-42, is
 * folded into it; nothing is kept on the host, so a captured call can be
 * replayed from a hipGraph). After the solve helper[0] is 0xffffffff, or
 * 0xdeadbeef if an internal wait gave up (the result is then invalid; it
 * cannot happen on a healthy device), and helper[1] is the number of 64-row
 * blocks.
 *
 * Hand-off workspace between row blocks (a 4 KiB state header + one or two
 * 8-byte words per element of x). The reference's launchers have no argument
 * for it, so the drop-in entry points (accblas_trsv / accblas_acc_trsv) use a
 * workspace owned by the library, one per (device, stream): it grows on
 * demand, a buffer that was handed to a launch is only freed by
 * accblas_trsv_release_workspace(stream) (so graphs that captured a solve stay
 * valid when a larger solve makes the stream's workspace grow), and growth is
 * refused during stream capture. Callers that want to own it use the *_ws
 * entry points: allocate accblas_trsv_workspace_bytes(st, n_max) bytes
 * (512-byte aligned; accblas_malloc is), initialise them ONCE with
 * accblas_trsv_workspace_init, and pass the buffer to any number of solves
 * with n <= n_max and any storage type no wider than `st`. A workspace serves one
 * solve at a time: solves sharing one must be ordered (same stream or events).
 * Every launch carries a serial number; a launch that finds the workspace
 * header left by another launch that never finished (aborted, or its bounded
 * waits gave up), or overwritten, repairs it - counters reset, a fresh epoch
 * so that stale hand-off words are not taken for new ones - before any row
 * block proceeds. Re-initialising by hand is never needed. (A hipGraph replays
 * the serial it was captured with; that is fine after a solve that completed,
 * which always leaves the header clean.) */

/* `trsv<ValueType>` (cuda/trsv_kernels.cuh:455-488 -> kernels :38-42,
 * :69-235, :263-432). */
int accblas_trsv(accblas_dtype vt, accblas_uplo uplo, accblas_diag diag,
                 int64_t n, const void *A, int64_t lda, void *x, int64_t incx,
                 uint32_t *helper, void *stream);

/* `acc_trsv<ArType, StType>` (cuda/trsv_kernels.cuh:918-961 -> kernels
 * :527-692, :725-893). */
int accblas_acc_trsv(accblas_dtype ar, accblas_dtype st, accblas_uplo uplo,
                     accblas_diag diag, int64_t n, const void *A, int64_t lda,
                     void *x, int64_t incx, uint32_t *helper, void *stream);

/* Did the last solve that used `helper` complete? Synchronises `stream`,
 * reads helper[0] back and sets *expired to 1 if an internal wait of that
 * solve gave up (the solution is then invalid), else to 0. Callers need not
 * know the encoding of helper[0]. A solve that gave up does not poison later
 * ones: the next launch on the same workspace recognises the unfinished
 * state and repairs it before it starts (see the workspace notes above). */
int accblas_trsv_status(const uint32_t *helper, void *stream, int *expired);

/* Bytes of hand-off workspace for solves of up to n unknowns stored as `st`. */
int accblas_trsv_workspace_bytes(accblas_dtype st, int64_t n, size_t *bytes);
/* Zeroes the workspace and arms its state (asynchronous on `stream`); once
 * per buffer, not per solve. */
int accblas_trsv_workspace_init(void *workspace, size_t workspace_bytes,
                                void *stream);
/* accblas_trsv / accblas_acc_trsv on a caller-owned workspace. */
int accblas_trsv_ws(accblas_dtype vt, accblas_uplo uplo, accblas_diag diag,
                    int64_t n, const void *A, int64_t lda, void *x,
                    int64_t incx, uint32_t *helper, void *workspace,
                    size_t workspace_bytes, void *stream);
int accblas_acc_trsv_ws(accblas_dtype ar, accblas_dtype st, accblas_uplo uplo,
                        accblas_diag diag, int64_t n, const void *A,
                        int64_t lda, void *x, int64_t incx, uint32_t *helper,
                        void *workspace, size_t workspace_bytes, void *stream);
/* Synchronises `stream` and frees every library-owned workspace buffer of
 * (current device, stream), including retired ones. Graphs that captured a
 * solve through accblas_trsv / accblas_acc_trsv on that stream must not be
 * replayed afterwards. Call it before hipStreamDestroy: the buffers are keyed
 * by the stream handle and otherwise stay allocated (a later stream that gets
 * the same handle value would inherit them, which is harmless but not a
 * release). */
int accblas_trsv_release_workspace(void *stream);

/* ---- helpers used by the benchmark setup -------------------------------- */

/* Element-wise conversion on the device: dst[i] = static_cast<dst_t>(src[i]),
 * i < n. The reference converts on the host (`convert_mtx`,
 * cuda/matrix_helper.cuh:91-103); this is the same cast done in HBM so that
 * large operands need not cross PCIe twice. */
int accblas_convert(accblas_dtype dst_t, void *dst, accblas_dtype src_t,
                    const void *src, int64_t n, void *stream);

#ifdef __cplusplus
} /* extern "C" */
#endif

#endif /* ACCBLAS_H_ */
