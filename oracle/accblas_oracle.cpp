// ORACLE — TEST INFRASTRUCTURE ONLY. Not part of the product.
//
// Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
// build, load or call this file; the product (accessor-blas_amd/) never does
// and fails loudly without its HIP library.
//
// What it is: a CPU restatement (C++14, g++) of the reference's hot path,
// every function citing the file:line of /root/reference it follows:
//   * the input generator and the St conversions (matrix_helper.cuh),
//   * the CPU error-check path `reduce` / `compare` (utils.cuh:281-332) — the
//     only CPU numerics the reference itself has,
//   * ORDER-FAITHFUL emulations of the CUDA kernels gemv / acc_gemv, dot /
//     acc_dot, lower/upper (acc_)trsv: same thread-to-element mapping, same
//     shared-memory tree and shuffle butterflies, FMA contraction as nvcc does
//     by default — i.e. what the reference's GPU kernels compute (the dot's
//     atomics are applied in block-index order, one of the orders the hardware
//     may take),
//   * a high-precision semantic oracle (long double accumulation of the same
//     storage-rounded inputs) used as the truth that tolerances are stated
//     against.
// All kernels run through the HOST side of the build's own accessor headers,
// as the reference's kernels run through Ginkgo's.
//
// PINNING STATUS: the reference ships no golden vectors, tests or fixtures,
// and cannot be compiled here (CUDA toolkit, cuBLAS/cuSOLVER and the Ginkgo
// headers are absent), so there are no reference OUTPUTS to
// pin against: kernel-level PARITY IS UNPINNED. What is pinned
// (tests/test_oracle.py, tests/golden/): the generator stream and the C1 dot
// value derived in SURVEY.md §8c C-4, fp16 conversion bit patterns against
// numpy, the structural known answers (plain == Acc<T,T> bit for bit,
// manufactured TRSV solutions, reduce on n in {1,2,3,odd,2^k}) and the error
// magnitudes the reference publishes in its plots.
#include <algorithm>
#include <array>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <random>
#include <type_traits>
#include <vector>

#include <accessor/range.hpp>
#include <accessor/reduced_row_major.hpp>

#include "../accessor-blas_amd/hip/float16.hpp"


namespace {


using size_type = std::int64_t;

enum dtype : int { F16 = 0, F32 = 1, F64 = 2 };


// ---------------------------------------------------------------------------
// ranges, exactly as the reference launchers build them
// (cuda/gemv_kernels.cuh:178-189)
// ---------------------------------------------------------------------------
template <typename Ar, typename St>
using acc_range = gko::acc::range<gko::acc::reduced_row_major<2, Ar, St>>;

template <typename Ar, typename St>
acc_range<Ar, St> make_range(size_type rows, size_type cols, St *data,
                             size_type stride)
{
    return acc_range<Ar, St>(std::array<size_type, 2>{{rows, cols}}, data,
                             std::array<size_type, 1>{{stride}});
}


template <typename T>
T fused(T a, T b, T c, bool use_fma)
{
    return use_fma ? std::fma(a, b, c) : a * b + c;
}


// kernel_utils.cuh:69-89 (block tree down to 32 values) followed by
// kernel_utils.cuh:30-53 (xor butterfly 1,2,4,8,16). `shared` has block_size
// entries; returns shared[0] as thread 0 sees it.
template <typename T>
T block_reduce_emulated(std::vector<T> &shared)
{
    const std::size_t block_size = shared.size();
    for (std::size_t i = block_size / 2; i >= 32; i /= 2) {
        for (std::size_t t = 0; t < i; ++t) {
            shared[t] = shared[t] + shared[t + i];
        }
    }
    std::array<T, 32> lanes;
    for (int l = 0; l < 32; ++l) {
        lanes[l] = shared[l];
    }
    for (int mask = 1; mask < 32; mask <<= 1) {
        std::array<T, 32> next;
        for (int l = 0; l < 32; ++l) {
            next[l] = lanes[l] + lanes[l ^ mask];
        }
        lanes = next;
    }
    return lanes[0];
}

// sub-warp butterfly alone (kernel_utils.cuh:30-53) over 32 lanes
template <typename T>
T butterfly32(const std::array<T, 32> &in)
{
    std::array<T, 32> lanes = in;
    for (int mask = 1; mask < 32; mask <<= 1) {
        std::array<T, 32> next;
        for (int l = 0; l < 32; ++l) {
            next[l] = lanes[l] + lanes[l ^ mask];
        }
        lanes = next;
    }
    return lanes[0];
}


// ---------------------------------------------------------------------------
// GEMV — order-faithful emulation of kernel::acc_gemv<512>
// (cuda/gemv_kernels.cuh:79-113; kernel::gemv :30-64 is the same with
// Ar == St). One block per row, thread t takes columns t, t+512, ...
// ---------------------------------------------------------------------------
template <typename Ar, typename St>
void ref_gemv(size_type rows, size_type cols, Ar alpha, const St *A,
              size_type lda, const St *x, size_type incx, Ar beta, St *res,
              size_type incres, bool use_fma)
{
    constexpr std::size_t block_size = 512;
    auto mtx = make_range<Ar, const St>(rows, cols, A, lda);
    auto xr = make_range<Ar, const St>(cols, 1, x, incx);
    auto rr = make_range<Ar, St>(rows, 1, res, incres);
    std::vector<Ar> shared(block_size);
    for (size_type row = 0; row < rows; ++row) {
        for (std::size_t t = 0; t < block_size; ++t) {
            Ar local{};
            for (size_type col = t; col < cols; col += block_size) {
                local = fused<Ar>(mtx(row, col), xr(col, 0), local, use_fma);
            }
            shared[t] = local;
        }
        const Ar sum = block_reduce_emulated(shared);
        if (beta == Ar{0}) {
            rr(row, 0) = alpha * sum;
        } else {
            const Ar old = rr(row, 0);
            rr(row, 0) = fused<Ar>(alpha, sum, beta * old, use_fma);
        }
    }
}

// Semantic oracle: same storage-rounded inputs, accumulation in long double,
// one rounding to Ar, then the alpha/beta update in long double and one
// rounding to Ar before the narrowing store.
template <typename Ar, typename St>
void exact_gemv(size_type rows, size_type cols, Ar alpha, const St *A,
                size_type lda, const St *x, size_type incx, Ar beta, St *res,
                size_type incres)
{
    auto mtx = make_range<Ar, const St>(rows, cols, A, lda);
    auto xr = make_range<Ar, const St>(cols, 1, x, incx);
    auto rr = make_range<Ar, St>(rows, 1, res, incres);
    for (size_type row = 0; row < rows; ++row) {
        long double sum = 0;
        for (size_type col = 0; col < cols; ++col) {
            sum += static_cast<long double>(mtx(row, col)) *
                   static_cast<long double>(xr(col, 0));
        }
        long double value = static_cast<long double>(alpha) * sum;
        if (beta != Ar{0}) {
            value += static_cast<long double>(beta) *
                     static_cast<long double>(Ar(rr(row, 0)));
        }
        rr(row, 0) = static_cast<Ar>(value);
    }
}


// ---------------------------------------------------------------------------
// DOT — order-faithful emulation of kernel::acc_dot<1024>
// (cuda/dot_kernels.cuh:131-161; kernel::dot :87-116 is the same with
// Ar == St). Grid of `num_blocks` blocks (reference: SMs * 32,
// dot_kernels.cuh:20-21,196-199), global thread g takes g, g + T, ...;
// atomic_add of the block results applied in block order after init_res
// (:78-82); optional cast_result (:168-173).
// ---------------------------------------------------------------------------
template <typename Ar, typename St, typename Res>
void ref_dot(size_type n, const St *x, size_type incx, const St *y,
             size_type incy, Res *res, size_type num_blocks, bool use_fma)
{
    constexpr std::size_t block_size = 1024;
    auto xr = make_range<Ar, const St>(n, 1, x, incx);
    auto yr = make_range<Ar, const St>(n, 1, y, incy);
    const size_type increment = num_blocks * static_cast<size_type>(block_size);
    std::vector<Ar> shared(block_size);
    Ar result{0};
    for (size_type b = 0; b < num_blocks; ++b) {
        for (std::size_t t = 0; t < block_size; ++t) {
            Ar local{};
            for (size_type idx = b * block_size + t; idx < n;
                 idx += increment) {
                local = fused<Ar>(xr(idx, 0), yr(idx, 0), local, use_fma);
            }
            shared[t] = local;
        }
        result = result + block_reduce_emulated(shared);
    }
    *res = static_cast<Res>(result);
}

template <typename Ar, typename St>
void exact_dot(size_type n, const St *x, size_type incx, const St *y,
               size_type incy, long double *sum_out, long double *scale_out)
{
    auto xr = make_range<Ar, const St>(n, 1, x, incx);
    auto yr = make_range<Ar, const St>(n, 1, y, incy);
    // pairwise-blocked long double sum: error << 1 ulp(double) of the result
    long double total = 0, total_sq = 0;
    const size_type chunk = 4096;
    for (size_type start = 0; start < n; start += chunk) {
        long double part = 0, part_sq = 0;
        const size_type end = std::min(n, start + chunk);
        for (size_type i = start; i < end; ++i) {
            const long double p = static_cast<long double>(xr(i, 0)) *
                                  static_cast<long double>(yr(i, 0));
            part += p;
            part_sq += p * p;
        }
        total += part;
        total_sq += part_sq;
    }
    *sum_out = total;
    *scale_out = std::sqrt(total_sq);  // Q of SURVEY.md §8d D4
}


// ---------------------------------------------------------------------------
// TRSV — order-faithful emulation of kernel::acc_lower_trsv / acc_upper_trsv
// <32, 4> (cuda/trsv_kernels.cuh:527-692 / :725-893; the plain kernels
// :69-235 / :263-432 are the same with Ar == St). Blocks are processed in
// ticket order 0, 1, 2, ... which is a schedule the hardware may take.
// ---------------------------------------------------------------------------
template <typename Ar, typename St>
void ref_trsv(bool upper, bool unit, size_type n, const St *A, size_type lda,
              St *x, size_type incx, bool use_fma)
{
    constexpr int ss = 32;        // swarp_size
    constexpr int spb = 4;        // swarps_per_block
    constexpr int ts = ss + 1;    // triang_stride
    constexpr int nlr = ss / spb; // num_local_rows
    auto mtx = make_range<Ar, const St>(n, n, A, lda);
    auto xr = make_range<Ar, St>(n, 1, x, incx);
    const size_type num_blocks = (n + ss - 1) / ss;
    std::vector<Ar> triang(ss * ts);
    std::array<Ar, ss> x_correction;

    for (size_type rb = 0; rb < num_blocks; ++rb) {
        // --- load the diagonal tile, column-major (:553-567 / :751-768)
        for (int row = 0; row < ss; ++row) {
            for (int col = 0; col < ss; ++col) {
                Ar value{0};
                if (!upper) {
                    const size_type grow = rb * ss + row;
                    const size_type gcol = rb * ss + col;
                    if (unit && col == row) {
                        value = Ar{1};
                    } else if (col <= row && grow < n && gcol < n) {
                        value = mtx(grow, gcol);
                    }
                } else {
                    const size_type grow = n - (rb + 1) * ss + row;
                    const size_type gcol = n - (rb + 1) * ss + col;
                    if (unit && col == row) {
                        value = Ar{1};
                    } else if (row <= col && 0 <= grow && 0 <= gcol) {
                        value = mtx(grow, gcol);
                    }
                }
                triang[col * ts + row] = value;
            }
        }
        // --- non-unit: scale rows by 1/diag (:571-586 / :772-788)
        if (!unit) {
            if (!upper) {
                for (int row = 0; row < ss; ++row) {
                    const Ar diag_el = Ar{1} / triang[row * ts + row];
                    for (int col = 0; col <= row; ++col) {
                        const int idx = col * ts + row;
                        triang[idx] =
                            (row == col) ? diag_el : triang[idx] * diag_el;
                    }
                }
            } else {
                // the reference starts at row = swarp_size (a no-op step)
                for (int row = ss - 1; row >= 0; --row) {
                    const Ar diag_el = Ar{1} / triang[row * ts + row];
                    for (int col = row; col < ss; ++col) {
                        const int idx = col * ts + row;
                        triang[idx] =
                            (row == col) ? diag_el : triang[idx] * diag_el;
                    }
                }
            }
        }
        // --- Gauss-Jordan inversion in place (:589-611 / :791-815)
        if (!upper) {
            for (int diag_id = 0; diag_id < ss; ++diag_id) {
                const Ar diag_el = triang[diag_id * ts + diag_id];
                for (int row = diag_id + 1; row < ss; ++row) {
                    const Ar factor = -triang[diag_id * ts + row];
                    for (int col = 0; col < row; ++col) {
                        const int idx = col * ts + row;
                        triang[idx] =
                            (col == diag_id)
                                ? factor * diag_el
                                : fused<Ar>(factor,
                                            triang[col * ts + diag_id],
                                            triang[idx], use_fma);
                    }
                }
            }
        } else {
            for (int diag_id = ss - 1; diag_id >= 0; --diag_id) {
                const Ar diag_el = triang[diag_id * ts + diag_id];
                for (int row = diag_id - 1; row >= 0; --row) {
                    const Ar factor = -triang[diag_id * ts + row];
                    for (int col = row + 1; col < ss; ++col) {
                        const int idx = col * ts + row;
                        triang[idx] =
                            (col == diag_id)
                                ? factor * diag_el
                                : fused<Ar>(factor,
                                            triang[col * ts + diag_id],
                                            triang[idx], use_fma);
                    }
                }
            }
        }
        // --- off-diagonal sweep: thread (tx, ty) accumulates nlr rows
        // (:614-654 / :818-860), then butterfly per row (:657-662 / :863-868)
        std::vector<Ar> local(static_cast<std::size_t>(ss) * spb * nlr, Ar{0});
        auto local_at = [&](int tx, int ty, int lri) -> Ar & {
            return local[(static_cast<std::size_t>(ty) * nlr + lri) * ss + tx];
        };
        if (!upper) {
            for (size_type cb = 0; cb < rb; ++cb) {
                for (int ty = 0; ty < spb; ++ty) {
                    const size_type end_lri = (n - 1 - rb * ss - ty) / spb;
                    for (int tx = 0; tx < ss; ++tx) {
                        const size_type gcol = cb * ss + tx;
                        const Ar x_cached = xr(gcol, 0);
                        for (int lri = 0; lri < nlr; ++lri) {
                            if (lri <= end_lri) {
                                const size_type grow =
                                    rb * ss + ty + lri * spb;
                                local_at(tx, ty, lri) = fused<Ar>(
                                    x_cached, mtx(grow, gcol),
                                    local_at(tx, ty, lri), use_fma);
                            }
                        }
                    }
                }
            }
        } else {
            for (size_type cb = 1; cb <= rb; ++cb) {
                for (int ty = 0; ty < spb; ++ty) {
                    // ceildiv(-n + (rb+1)*ss - ty, spb) (:841-844), with the
                    // reference's ceildiv semantics (utils.cuh:71-75)
                    const size_type a = -n + (rb + 1) * ss - ty;
                    const size_type start_lri =
                        (a <= 0) ? a / spb : (a - 1) / spb + 1;
                    for (int tx = 0; tx < ss; ++tx) {
                        const size_type gcol = n - cb * ss + tx;
                        const Ar x_cached = xr(gcol, 0);
                        for (int lri = 0; lri < nlr; ++lri) {
                            if (start_lri <= lri) {
                                const size_type grow =
                                    n - (rb + 1) * ss + ty + lri * spb;
                                local_at(tx, ty, lri) = fused<Ar>(
                                    x_cached, mtx(grow, gcol),
                                    local_at(tx, ty, lri), use_fma);
                            }
                        }
                    }
                }
            }
        }
        for (int lri = 0; lri < nlr; ++lri) {
            for (int ty = 0; ty < spb; ++ty) {
                std::array<Ar, 32> lanes;
                for (int tx = 0; tx < ss; ++tx) {
                    lanes[tx] = local_at(tx, ty, lri);
                }
                x_correction[lri * spb + ty] = butterfly32(lanes);
            }
        }
        // --- solve with the inverted tile (:665-691 / :871-892)
        std::array<Ar, ss> local_x;
        std::array<bool, ss> valid;
        for (int row = 0; row < ss; ++row) {
            const size_type x_idx =
                upper ? n - (rb + 1) * ss + row : rb * ss + row;
            valid[row] = upper ? (0 <= x_idx) : (x_idx < n);
            local_x[row] = valid[row] ? Ar(xr(x_idx, 0)) - x_correction[row]
                                      : Ar{0};
        }
        for (int row = 0; row < ss; ++row) {
            Ar solution{0};
            for (int col = 0; col < ss; ++col) {
                solution = fused<Ar>(triang[col * ts + row], local_x[col],
                                     solution, use_fma);
            }
            if (valid[row]) {
                const size_type x_idx =
                    upper ? n - (rb + 1) * ss + row : rb * ss + row;
                xr(x_idx, 0) = solution;
            }
        }
    }
}

// Semantic TRSV: plain substitution in long double on the storage-rounded
// matrix, result rounded once to Ar and stored as St. Later rows use the
// St-ROUNDED earlier solutions, as every kernel does (x is re-read through
// the accessor, trsv_kernels.cuh:641).
template <typename Ar, typename St>
void exact_trsv(bool upper, bool unit, size_type n, const St *A, size_type lda,
                St *x, size_type incx)
{
    auto mtx = make_range<Ar, const St>(n, n, A, lda);
    auto xr = make_range<Ar, St>(n, 1, x, incx);
    for (size_type k = 0; k < n; ++k) {
        const size_type i = upper ? n - 1 - k : k;
        long double value = static_cast<long double>(Ar(xr(i, 0)));
        if (!upper) {
            for (size_type j = 0; j < i; ++j) {
                value -= static_cast<long double>(mtx(i, j)) *
                         static_cast<long double>(Ar(xr(j, 0)));
            }
        } else {
            for (size_type j = i + 1; j < n; ++j) {
                value -= static_cast<long double>(mtx(i, j)) *
                         static_cast<long double>(Ar(xr(j, 0)));
            }
        }
        if (!unit) {
            value /= static_cast<long double>(mtx(i, i));
        }
        xr(i, 0) = static_cast<Ar>(value);
    }
}


// ---------------------------------------------------------------------------
// dispatch helpers
// ---------------------------------------------------------------------------
template <typename F>
int with_storage(int st, F &&f)
{
    switch (st) {
    case F16:
        return f(float16{});
    case F32:
        return f(float{});
    case F64:
        return f(double{});
    }
    return -1;
}

template <typename F>
int with_arithmetic(int ar, F &&f)
{
    switch (ar) {
    case F32:
        return f(float{});
    case F64:
        return f(double{});
    }
    return -1;
}


struct engine_box {
    std::default_random_engine engine;
    std::uniform_real_distribution<double> dist{-1.0, 1.0};
};


}  // namespace


extern "C" {


// ---- generator (cuda/gemv_benchmark.cu:81-83: default_random_engine(42),
// uniform_real_distribution<double>(-1, 1)) -------------------------------
void *oracle_engine_create(std::uint32_t seed)
{
    auto *box = new engine_box;
    box->engine.seed(seed);
    return box;
}

void oracle_engine_destroy(void *engine) { delete static_cast<engine_box *>(engine); }

// gen_mtx (cuda/matrix_helper.cuh:28-51): row-major fill, stride gaps
// untouched, redraw until std::isnormal
int oracle_gen_mtx(void *engine, size_type rows, size_type cols,
                   size_type stride, double *out)
{
    if (stride < cols) {
        return -2;  // "Wrong use of stride"
    }
    auto *box = static_cast<engine_box *>(engine);
    for (size_type row = 0; row < rows; ++row) {
        for (size_type col = 0; col < cols; ++col) {
            double val{};
            do {
                val = box->dist(box->engine);
            } while (!std::isnormal(val));
            out[row * stride + col] = val;
        }
    }
    return 0;
}

// write_random (cuda/matrix_helper.cuh:66-76): no isnormal filter
void oracle_write_random(void *engine, size_type rows, size_type cols,
                         size_type stride, double *out)
{
    auto *box = static_cast<engine_box *>(engine);
    for (size_type row = 0; row < rows; ++row) {
        for (size_type col = 0; col < cols; ++col) {
            out[row * stride + col] = box->dist(box->engine);
        }
    }
}

// convert_mtx with static_cast (cuda/matrix_helper.cuh:91-103,
// gemv_memory.cuh:92-97); element-wise over a flat buffer
int oracle_convert(int dst_t, void *dst, int src_t, const void *src,
                   size_type n)
{
    return with_storage(dst_t, [&](auto d) {
        using D = decltype(d);
        return with_storage(src_t, [&](auto s) {
            using S = decltype(s);
            auto *out = static_cast<D *>(dst);
            const auto *in = static_cast<const S *>(src);
            for (size_type i = 0; i < n; ++i) {
                out[i] = static_cast<D>(static_cast<double>(in[i]));
            }
            return 0;
        });
    });
}

// ---- CPU error-check path of the reference ------------------------------
// reduce (cuda/utils.cuh:281-300) with op = a + b (mode 0) or |a| + |b|
// (mode 1, gemv_benchmark.cu:224-227). Overwrites tmp.
double oracle_reduce(size_type n, size_type stride, double *tmp, int mode)
{
    auto op = [mode](double a, double b) {
        return mode == 1 ? std::abs(a) + std::abs(b) : a + b;
    };
    auto ceildiv = [](size_type a, size_type b) {
        return (a <= 0) ? a / b : (a - 1) / b + 1;
    };
    size_type end = n;
    for (size_type halfway = ceildiv(n, 2); halfway > 1;
         halfway = ceildiv(halfway, 2)) {
        for (size_type row = 0; row < halfway; ++row) {
            if (row + halfway < end) {
                tmp[row * stride] =
                    op(tmp[row * stride], tmp[(row + halfway) * stride]);
            }
        }
        end = halfway;
    }
    return n == 1 ? op(tmp[0], 0.0) : op(tmp[0], tmp[stride]);
}

// compare (cuda/utils.cuh:315-332): tmp = |ref - other| in double, then
// reduce with +. `other` is of type other_t (the St result is widened).
double oracle_compare(size_type n, size_type stride, const double *ref,
                      int other_t, const void *other, double *tmp)
{
    with_storage(other_t, [&](auto s) {
        using S = decltype(s);
        const auto *o = static_cast<const S *>(other);
        for (size_type row = 0; row < n; ++row) {
            const double v1 = ref[row * stride];
            const double v2 = static_cast<double>(o[row * stride]);
            tmp[row * stride] = std::abs(v1 - v2);
        }
        return 0;
    });
    return oracle_reduce(n, stride, tmp, 0);
}

// ---- kernels ---------------------------------------------------------------
int oracle_ref_gemv(int ar, int st, size_type rows, size_type cols,
                    double alpha, const void *A, size_type lda, const void *x,
                    size_type incx, double beta, void *res, size_type incres,
                    int use_fma)
{
    return with_arithmetic(ar, [&](auto a) {
        using Ar = decltype(a);
        return with_storage(st, [&](auto s) {
            using St = decltype(s);
            ref_gemv<Ar, St>(rows, cols, static_cast<Ar>(alpha),
                             static_cast<const St *>(A), lda,
                             static_cast<const St *>(x), incx,
                             static_cast<Ar>(beta), static_cast<St *>(res),
                             incres, use_fma != 0);
            return 0;
        });
    });
}

int oracle_exact_gemv(int ar, int st, size_type rows, size_type cols,
                      double alpha, const void *A, size_type lda,
                      const void *x, size_type incx, double beta, void *res,
                      size_type incres)
{
    return with_arithmetic(ar, [&](auto a) {
        using Ar = decltype(a);
        return with_storage(st, [&](auto s) {
            using St = decltype(s);
            exact_gemv<Ar, St>(rows, cols, static_cast<Ar>(alpha),
                               static_cast<const St *>(A), lda,
                               static_cast<const St *>(x), incx,
                               static_cast<Ar>(beta), static_cast<St *>(res),
                               incres);
            return 0;
        });
    });
}

int oracle_ref_dot(int ar, int st, int res_t, size_type n, const void *x,
                   size_type incx, const void *y, size_type incy, void *res,
                   size_type num_blocks, int use_fma)
{
    return with_arithmetic(ar, [&](auto a) {
        using Ar = decltype(a);
        return with_storage(st, [&](auto s) {
            using St = decltype(s);
            return with_storage(res_t, [&](auto r) {
                using Res = decltype(r);
                Ar value{};
                ref_dot<Ar, St, Ar>(n, static_cast<const St *>(x), incx,
                                    static_cast<const St *>(y), incy, &value,
                                    num_blocks, use_fma != 0);
                *static_cast<Res *>(res) =
                    static_cast<Res>(static_cast<double>(value));
                return 0;
            });
        });
    });
}

// sum_out = long double sum rounded to double; scale_out = sqrt(sum (x*y)^2)
int oracle_exact_dot(int ar, int st, size_type n, const void *x,
                     size_type incx, const void *y, size_type incy,
                     double *sum_out, double *scale_out)
{
    return with_arithmetic(ar, [&](auto a) {
        using Ar = decltype(a);
        return with_storage(st, [&](auto s) {
            using St = decltype(s);
            long double sum = 0, scale = 0;
            exact_dot<Ar, St>(n, static_cast<const St *>(x), incx,
                              static_cast<const St *>(y), incy, &sum, &scale);
            *sum_out = static_cast<double>(sum);
            *scale_out = static_cast<double>(scale);
            return 0;
        });
    });
}

int oracle_ref_trsv(int ar, int st, int upper, int unit, size_type n,
                    const void *A, size_type lda, void *x, size_type incx,
                    int use_fma)
{
    return with_arithmetic(ar, [&](auto a) {
        using Ar = decltype(a);
        return with_storage(st, [&](auto s) {
            using St = decltype(s);
            ref_trsv<Ar, St>(upper != 0, unit != 0, n,
                             static_cast<const St *>(A), lda,
                             static_cast<St *>(x), incx, use_fma != 0);
            return 0;
        });
    });
}

int oracle_exact_trsv(int ar, int st, int upper, int unit, size_type n,
                      const void *A, size_type lda, void *x, size_type incx)
{
    return with_arithmetic(ar, [&](auto a) {
        using Ar = decltype(a);
        return with_storage(st, [&](auto s) {
            using St = decltype(s);
            exact_trsv<Ar, St>(upper != 0, unit != 0, n,
                               static_cast<const St *>(A), lda,
                               static_cast<St *>(x), incx);
            return 0;
        });
    });
}

// CPU GEMV over a row block / dot over a chunk through the host accessor: the
// "host-CPU baseline" of BASELINE.md §3 item 2 (the reference has no CPU BLAS
// kernels). Single-threaded bodies; bench.py runs one per host thread on
// disjoint row blocks / chunks (ctypes releases the GIL).
int oracle_accessor_gemv_rows(int ar, int st, size_type row_begin,
                              size_type row_end, size_type cols, double alpha,
                              const void *A, size_type lda, const void *x,
                              size_type incx, double beta, void *res,
                              size_type incres)
{
    return with_arithmetic(ar, [&](auto a) {
        using Ar = decltype(a);
        return with_storage(st, [&](auto s) {
            using St = decltype(s);
            auto mtx = make_range<Ar, const St>(row_end, cols,
                                                static_cast<const St *>(A), lda);
            auto xr = make_range<Ar, const St>(
                cols, 1, static_cast<const St *>(x), incx);
            auto rr = make_range<Ar, St>(row_end, 1, static_cast<St *>(res),
                                         incres);
            const Ar al = static_cast<Ar>(alpha), be = static_cast<Ar>(beta);
            for (size_type row = row_begin; row < row_end; ++row) {
                Ar partial[4] = {};
                size_type col = 0;
                for (; col + 3 < cols; col += 4) {
                    for (int u = 0; u < 4; ++u) {
                        partial[u] += mtx(row, col + u) * xr(col + u, 0);
                    }
                }
                for (; col < cols; ++col) {
                    partial[0] += mtx(row, col) * xr(col, 0);
                }
                const Ar sum =
                    (partial[0] + partial[1]) + (partial[2] + partial[3]);
                if (be == Ar{0}) {
                    rr(row, 0) = al * sum;
                } else {
                    rr(row, 0) = al * sum + be * rr(row, 0);
                }
            }
            return 0;
        });
    });
}

int oracle_accessor_dot_chunk(int ar, int st, size_type begin, size_type end,
                              const void *x, size_type incx, const void *y,
                              size_type incy, double *partial_out)
{
    return with_arithmetic(ar, [&](auto a) {
        using Ar = decltype(a);
        return with_storage(st, [&](auto s) {
            using St = decltype(s);
            auto xr = make_range<Ar, const St>(
                end, 1, static_cast<const St *>(x), incx);
            auto yr = make_range<Ar, const St>(
                end, 1, static_cast<const St *>(y), incy);
            Ar partial[4] = {};
            size_type i = begin;
            for (; i + 3 < end; i += 4) {
                for (int u = 0; u < 4; ++u) {
                    partial[u] += xr(i + u, 0) * yr(i + u, 0);
                }
            }
            for (; i < end; ++i) {
                partial[0] += xr(i, 0) * yr(i, 0);
            }
            *partial_out = static_cast<double>((partial[0] + partial[1]) +
                                               (partial[2] + partial[3]));
            return 0;
        });
    });
}


}  // extern "C"


// Advances the generator exactly as gen_mtx would for `count` values without
// storing them (used to reach the draw positions of x / res behind a large A,
// cuda/gemv_memory.cuh:45-47, without allocating A).
extern "C" void oracle_engine_discard(void *engine, std::int64_t count)
{
    auto *box = static_cast<engine_box *>(engine);
    for (std::int64_t i = 0; i < count; ++i) {
        double val{};
        do {
            val = box->dist(box->engine);
        } while (!std::isnormal(val));
    }
}
