#include <hip/hip_runtime.h>
#include <type_traits>
using f4 = float __attribute__((ext_vector_type(4)));
constexpr int NI = 8;
struct stage { f4 a[NI]; };
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

template <int Variant>
__global__ __launch_bounds__(512) void k(const float *A, long lda, const unsigned long long *flags,
                                         float *out, int steps)
{
    __shared__ float xs[256];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const float *row_ptr[NI];
    for (int i = 0; i < NI; ++i) row_ptr[i] = A + (long)(wave * 16 + i * 2 + lane / 32) * lda + (lane % 32) * 4;
    auto issue = [&](int cb, stage &s) {
#pragma unroll
        for (int i = 0; i < NI; ++i) s.a[i] = __builtin_nontemporal_load((const f4 *)(row_ptr[i] + cb * 128));
    };
    double acc[NI] = {};
    int chunk_end = 0;
    auto next_chunk = [&](int cb) {
        lds_barrier();
        if (wave == 0) {
            unsigned long long v = __hip_atomic_load(flags + cb * 64 + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            while (!__all((v >> 32) == 7)) v = __hip_atomic_load(flags + cb * 64 + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            xs[lane] = __uint_as_float((unsigned)v);
        }
        lds_barrier();
        chunk_end = cb + 1;
    };
    auto consume = [&](int cb, stage &s) {
        if (cb == chunk_end) next_chunk(cb);
        const float x0 = xs[(lane % 32) * 4], x1 = xs[(lane % 32) * 4 + 1], x2 = xs[(lane % 32) * 4 + 2], x3 = xs[(lane % 32) * 4 + 3];
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            acc[i] += (double)s.a[i].x * x0; acc[i] += (double)s.a[i].y * x1;
            acc[i] += (double)s.a[i].z * x2; acc[i] += (double)s.a[i].w * x3;
        }
    };
    stage ring[2];
    issue(0, ring[0]);
    asm volatile("" ::: "memory");  // slot 0's loads strictly before slot 1's
    issue(1, ring[1]);
    asm volatile("" ::: "memory");
    if (Variant == 0) {  // guarded, as the kernel's loop
        for (int cb0 = 0; cb0 < steps; cb0 += 2) {
#pragma unroll
            for (int d = 0; d < 2; ++d) {
                const int cb = cb0 + d;
                if (cb < steps) {
                    consume(cb, ring[d]);
                    if (cb + 2 < steps) issue(cb + 2, ring[d]);
                }
            }
        }
    } else if (Variant == 2) {  // guarded + compiler barrier between use and refill
        for (int cb0 = 0; cb0 < steps; cb0 += 2) {
#pragma unroll
            for (int d = 0; d < 2; ++d) {
                const int cb = cb0 + d;
                if (cb < steps) {
                    consume(cb, ring[d]);
                    asm volatile("" ::: "memory");
                    if (cb + 2 < steps) issue(cb + 2, ring[d]);
                }
            }
        }
    } else if (Variant == 3) {  // guarded + sched_barrier
        for (int cb0 = 0; cb0 < steps; cb0 += 2) {
#pragma unroll
            for (int d = 0; d < 2; ++d) {
                const int cb = cb0 + d;
                if (cb < steps) {
                    consume(cb, ring[d]);
                    __builtin_amdgcn_sched_barrier(0);
                    if (cb + 2 < steps) issue(cb + 2, ring[d]);
                }
            }
        }
    } else if (Variant == 5) {  // refill address made to depend on the accumulators
        for (int cb0 = 0; cb0 < steps; cb0 += 2) {
#pragma unroll
            for (int d = 0; d < 2; ++d) {
                const int cb = cb0 + d;
                if (cb < steps) {
                    consume(cb, ring[d]);
                    int after = 0;
                    asm volatile("" : "+v"(after) : "v"(acc[0]), "v"(acc[1]), "v"(acc[2]), "v"(acc[3]),
                                 "v"(acc[4]), "v"(acc[5]), "v"(acc[6]), "v"(acc[7]));
                    if (cb + 2 < steps) {
#pragma unroll
                        for (int i = 0; i < NI; ++i)
                            ring[d].a[i] = __builtin_nontemporal_load(
                                (const f4 *)(row_ptr[i] + (cb + 2) * 128 + after));
                    }
                }
            }
        }
    } else if (Variant == 6) {  // whole groups + refill after the FMAs (false dependency)
        auto refill = [&](int cb, stage &st) {
            int after = 0;
            asm volatile("" : "+v"(after) : "v"(acc[0]), "v"(acc[1]), "v"(acc[2]), "v"(acc[3]),
                         "v"(acc[4]), "v"(acc[5]), "v"(acc[6]), "v"(acc[7]));
#pragma unroll
            for (int i = 0; i < NI; ++i)
                st.a[i] = __builtin_nontemporal_load((const f4 *)(row_ptr[i] + cb * 128 + after));
        };
        int cb0 = 0;
        for (; cb0 + 4 <= steps; cb0 += 2) {
            consume(cb0, ring[0]); refill(cb0 + 2, ring[0]);
            consume(cb0 + 1, ring[1]); refill(cb0 + 3, ring[1]);
        }
        for (; cb0 < steps; cb0 += 2) {
            consume(cb0, ring[0]);
            if (cb0 + 1 < steps) consume(cb0 + 1, ring[1]);
        }
    } else if (Variant == 13 || Variant == 14) {
        const int wave_u = __builtin_amdgcn_readfirstlane(wave);
        unsigned spins = 0;
        auto expired = [&]() {
            ++spins;
            if ((spins & 1023u) != 0) return false;
            const unsigned st = __hip_atomic_load((const unsigned *)out, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (st == 0xdeadbeefu) return true;
            if (spins >= (1u << 21)) { __hip_atomic_store((unsigned *)out, 0xdeadbeefu, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); return true; }
            return false;
        };
        int mode = steps & 1;
        auto next_chunk3 = [&](int cb) {
            lds_barrier();
            if (wave_u == 0 && mode) {
                unsigned long long p0 = __hip_atomic_load(flags + cb * 128 + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                unsigned long long p1 = __hip_atomic_load(flags + cb * 128 + 64 + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                while (!__all((p0 >> 32) == 7 && (p1 >> 32) == 7) && !(Variant == 13 && expired())) {
                    p0 = __hip_atomic_load(flags + cb * 128 + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    p1 = __hip_atomic_load(flags + cb * 128 + 64 + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
                xs[lane] = __uint_as_float((unsigned)p0);
                xs[64 + lane] = __uint_as_float((unsigned)p1);
            }
            lds_barrier();
            if (!mode) {
                const int elements = (steps & 7) * 128;
                for (int e0 = 0; e0 < elements; e0 += 8 * 512) {
                    unsigned long long raw[8];
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        const int e = e0 + j * 512 + threadIdx.x;
                        if (Variant == 13 || e < elements) raw[j] = __hip_atomic_load(flags + e, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    }
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        const int e = e0 + j * 512 + threadIdx.x;
                        if (Variant == 13 || e < elements) {
                            while ((raw[j] >> 32) != 7 && !(Variant == 13 && expired())) raw[j] = __hip_atomic_load(flags + e, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                            xs[e & 255] = __uint_as_float((unsigned)raw[j]);
                        }
                    }
                }
                lds_barrier();
            }
            chunk_end = cb + 1;
        };
        auto consume3 = [&](int cb, stage &st) {
            if (cb == chunk_end) next_chunk3(cb);
            const float x0 = xs[(lane % 32) * 4], x1 = xs[(lane % 32) * 4 + 1], x2 = xs[(lane % 32) * 4 + 2], x3 = xs[(lane % 32) * 4 + 3];
#pragma unroll
            for (int i = 0; i < NI; ++i) {
                acc[i] += (double)st.a[i].x * x0; acc[i] += (double)st.a[i].y * x1;
                acc[i] += (double)st.a[i].z * x2; acc[i] += (double)st.a[i].w * x3;
            }
        };
        auto refill = [&](int cb, stage &st) {
            int after = 0;
            asm volatile("" : "+v"(after) : "v"(acc[0]), "v"(acc[1]), "v"(acc[2]), "v"(acc[3]),
                         "v"(acc[4]), "v"(acc[5]), "v"(acc[6]), "v"(acc[7]));
#pragma unroll
            for (int i = 0; i < NI; ++i)
                st.a[i] = __builtin_nontemporal_load((const f4 *)(row_ptr[i] + cb * 128 + after));
        };
        int cb0 = 0;
        for (; cb0 + 4 <= steps; cb0 += 2) {
            consume3(cb0, ring[0]); refill(cb0 + 2, ring[0]);
            consume3(cb0 + 1, ring[1]); refill(cb0 + 3, ring[1]);
        }
        for (; cb0 < steps; cb0 += 2) {
            consume3(cb0, ring[0]);
            if (cb0 + 1 < steps) consume3(cb0 + 1, ring[1]);
        }
    } else if (Variant >= 7 && Variant <= 12) {
        const int wave_u = __builtin_amdgcn_readfirstlane(wave);
        unsigned long long probe = 0;
        bool pending = false;
        int mode = steps & 1;  // runtime: bulk or front
        auto next_chunk2 = [&](int cb) {
            lds_barrier();
            if ((Variant >= 11 ? wave_u == 0 : wave == 0) && mode) {
                if (!pending) probe = __hip_atomic_load(flags + cb * 64 + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                while (!__all((probe >> 32) == 7)) probe = __hip_atomic_load(flags + cb * 64 + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                pending = false;
                xs[lane] = __uint_as_float((unsigned)probe);
            }
            lds_barrier();
            if (!mode) {  // bulk: all threads fetch
                unsigned long long v = __hip_atomic_load(flags + cb * 512 + threadIdx.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                while ((v >> 32) != 7) v = __hip_atomic_load(flags + cb * 512 + threadIdx.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (threadIdx.x < 256) xs[threadIdx.x] = __uint_as_float((unsigned)v);
                lds_barrier();
            }
            chunk_end = cb + 1;
        };
        auto consume2 = [&](int cb, stage &st) {
            if (cb == chunk_end) next_chunk2(cb);
            const float x0 = xs[(lane % 32) * 4], x1 = xs[(lane % 32) * 4 + 1], x2 = xs[(lane % 32) * 4 + 2], x3 = xs[(lane % 32) * 4 + 3];
#pragma unroll
            for (int i = 0; i < NI; ++i) {
                acc[i] += (double)st.a[i].x * x0; acc[i] += (double)st.a[i].y * x1;
                acc[i] += (double)st.a[i].z * x2; acc[i] += (double)st.a[i].w * x3;
            }
            if (Variant == 9) {  // every wave, every step: unconditional early probe
                probe = __hip_atomic_load(flags + (cb + 1) * 64 + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                pending = true;
            }
            if (Variant == 11 && wave_u == 0) {  // scalar condition, every step
                probe = __hip_atomic_load(flags + (cb + 1) * 64 + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                pending = true;
            }
            if (Variant == 12 && wave_u == 0 && mode && !pending && cb + 1 == chunk_end) {  // scalar, as the kernel
                probe = __hip_atomic_load(flags + (cb + 1) * 64 + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                pending = true;
            }
            if (Variant == 10 && wave == 0) {  // wave 0 only, but every step
                probe = __hip_atomic_load(flags + (cb + 1) * 64 + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                pending = true;
            }
            if (Variant == 7 && wave == 0 && mode && !pending && cb + 1 == chunk_end) {  // early probe before the refill
                probe = __hip_atomic_load(flags + (cb + 1) * 64 + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                pending = true;
            }
        };
        auto refill = [&](int cb, stage &st) {
            int after = 0;
            asm volatile("" : "+v"(after) : "v"(acc[0]), "v"(acc[1]), "v"(acc[2]), "v"(acc[3]),
                         "v"(acc[4]), "v"(acc[5]), "v"(acc[6]), "v"(acc[7]));
#pragma unroll
            for (int i = 0; i < NI; ++i)
                st.a[i] = __builtin_nontemporal_load((const f4 *)(row_ptr[i] + cb * 128 + after));
        };
        int cb0 = 0;
        for (; cb0 + 4 <= steps; cb0 += 2) {
            consume2(cb0, ring[0]); refill(cb0 + 2, ring[0]);
            consume2(cb0 + 1, ring[1]); refill(cb0 + 3, ring[1]);
        }
        for (; cb0 < steps; cb0 += 2) {
            consume2(cb0, ring[0]);
            if (cb0 + 1 < steps) consume2(cb0 + 1, ring[1]);
        }
    } else if (Variant == 4) {  // asm loads refilling IN PLACE + manual waits, whole groups
        auto issue_asm = [&](int cb, stage &st) {
#pragma unroll
            for (int i = 0; i < NI; ++i)
                asm volatile("global_load_dwordx4 %0, %1, off nt" : "+v"(st.a[i]) : "v"(row_ptr[i] + cb * 128) : "memory");
        };
        auto arrived = [&](stage &st, auto later) {
            asm volatile("s_waitcnt vmcnt(%8)"
                         : "+v"(st.a[0]), "+v"(st.a[1]), "+v"(st.a[2]), "+v"(st.a[3]), "+v"(st.a[4]),
                           "+v"(st.a[5]), "+v"(st.a[6]), "+v"(st.a[7])
                         : "n"(decltype(later)::value) : "memory");
        };
        stage r0{}, r1{};
        issue_asm(0, r0);
        issue_asm(1, r1);
        int cb0 = 0;
        for (; cb0 + 4 <= steps; cb0 += 2) {
            if (cb0 == chunk_end) next_chunk(cb0);
            arrived(r0, std::integral_constant<int, 8>{});
            consume(-1, r0);
            issue_asm(cb0 + 2, r0);
            if (cb0 + 1 == chunk_end) next_chunk(cb0 + 1);
            arrived(r1, std::integral_constant<int, 8>{});
            consume(-1, r1);
            issue_asm(cb0 + 3, r1);
        }
        for (; cb0 < steps; cb0 += 2) {
            if (cb0 == chunk_end) next_chunk(cb0);
            arrived(r0, std::integral_constant<int, 0>{});
            consume(-1, r0);
            if (cb0 + 1 < steps) {
                if (cb0 + 1 == chunk_end) next_chunk(cb0 + 1);
                arrived(r1, std::integral_constant<int, 0>{});
                consume(-1, r1);
            }
        }
    } else {  // whole groups, unconditional
        int cb0 = 0;
        for (; cb0 + 4 <= steps; cb0 += 2) {
            consume(cb0, ring[0]); issue(cb0 + 2, ring[0]);
            consume(cb0 + 1, ring[1]); issue(cb0 + 3, ring[1]);
        }
        for (; cb0 < steps; cb0 += 2) {
            consume(cb0, ring[0]);
            if (cb0 + 1 < steps) consume(cb0 + 1, ring[1]);
        }
    }
    double t = 0;
    for (int i = 0; i < NI; ++i) t += acc[i];
    out[threadIdx.x + blockIdx.x * 512] = (float)t;
}
template __global__ void k<0>(const float *, long, const unsigned long long *, float *, int);
template __global__ void k<1>(const float *, long, const unsigned long long *, float *, int);
template __global__ void k<2>(const float *, long, const unsigned long long *, float *, int);
template __global__ void k<3>(const float *, long, const unsigned long long *, float *, int);
template __global__ void k<4>(const float *, long, const unsigned long long *, float *, int);
template __global__ void k<5>(const float *, long, const unsigned long long *, float *, int);
template __global__ void k<6>(const float *, long, const unsigned long long *, float *, int);
template __global__ void k<7>(const float *, long, const unsigned long long *, float *, int);
template __global__ void k<8>(const float *, long, const unsigned long long *, float *, int);
template __global__ void k<9>(const float *, long, const unsigned long long *, float *, int);
template __global__ void k<10>(const float *, long, const unsigned long long *, float *, int);
template __global__ void k<11>(const float *, long, const unsigned long long *, float *, int);
template __global__ void k<12>(const float *, long, const unsigned long long *, float *, int);
template __global__ void k<13>(const float *, long, const unsigned long long *, float *, int);
template __global__ void k<14>(const float *, long, const unsigned long long *, float *, int);
