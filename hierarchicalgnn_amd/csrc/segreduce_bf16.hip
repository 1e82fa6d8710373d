// bf16 feature rows for the HBM-bound kernels (BASELINE config 4 dtype: latent=512 bf16, a 1-KiB
// row again).  Same plan, same one-wave-per-list structure as segreduce.hip; a row of F bf16 is read
// as F/8 16-byte columns (8 bf16 per lane), sums are accumulated in fp32 and rounded to bf16 once
// (v_cvt_pk_bf16_f32), so the result does not depend on the list length the way bf16 atomics would
// (MI355X_MICROARCH.md, global float atomics: pk_add_bf16 rounds every add).
// Partial sums of split lists stay fp32; the combine pass rounds once.
#include "common.h"
#include <type_traits>

namespace hgnn {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned short u16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

struct Acc8 {
    f32x4 lo, hi;
};

__device__ __forceinline__ Acc8 widen(u16x8 v) {
    Acc8 a;
    a.lo = f32x4{__builtin_bit_cast(float, (unsigned)v[0] << 16), __builtin_bit_cast(float, (unsigned)v[1] << 16),
                 __builtin_bit_cast(float, (unsigned)v[2] << 16), __builtin_bit_cast(float, (unsigned)v[3] << 16)};
    a.hi = f32x4{__builtin_bit_cast(float, (unsigned)v[4] << 16), __builtin_bit_cast(float, (unsigned)v[5] << 16),
                 __builtin_bit_cast(float, (unsigned)v[6] << 16), __builtin_bit_cast(float, (unsigned)v[7] << 16)};
    return a;
}

__device__ __forceinline__ u16x8 narrow(const Acc8& a) {
    bf16x8 r;
    r[0] = (__bf16)a.lo.x; r[1] = (__bf16)a.lo.y; r[2] = (__bf16)a.lo.z; r[3] = (__bf16)a.lo.w;
    r[4] = (__bf16)a.hi.x; r[5] = (__bf16)a.hi.y; r[6] = (__bf16)a.hi.z; r[7] = (__bf16)a.hi.w;
    return __builtin_bit_cast(u16x8, r);
}

// 8 consecutive elements starting at element offset `off` of a row
template <bool IN_F32>
__device__ __forceinline__ Acc8 load8(const void* row, int off) {
    if (IN_F32) {
        const float* p = (const float*)row + off;
        Acc8 a;
        a.lo = *(const f32x4*)p;
        a.hi = *(const f32x4*)(p + 4);
        return a;
    }
    return widen(*(const u16x8*)((const unsigned short*)row + off));
}

template <bool OUT_F32>
__device__ __forceinline__ void store8(void* row, int off, const Acc8& a) {
    if (OUT_F32) {
        float* p = (float*)row + off;
        *(f32x4*)p = a.lo;
        *(f32x4*)(p + 4) = a.hi;
    } else {
        *(u16x8*)((unsigned short*)row + off) = narrow(a);
    }
}

// IN_F32: rows of `src` are fp32 (the combine pass over the partial buffer); otherwise bf16.
// Items whose target is >= 0 write bf16 rows of `out`; negative targets write fp32 partial rows.
template <int RL, int U, bool HAS_W, bool HAS_RS, bool IN_F32, int WPB>
__global__ __launch_bounds__(WPB * 64) void k_seg_reduce_bf16(
    const void* __restrict__ src, int F, int ncol, const int32_t* __restrict__ src_row,
    const int32_t* __restrict__ perm, const float* __restrict__ weight, const float* __restrict__ row_scale,
    const int32_t* __restrict__ wi_begin, const int32_t* __restrict__ wi_end,
    const int32_t* __restrict__ wi_target, const int32_t* __restrict__ n_items_ptr, int64_t max_items,
    unsigned short* __restrict__ out, float* __restrict__ partial) {
    constexpr int G = 64 / RL;
    const int lane = threadIdx.x & 63;
    const int64_t item = (int64_t)blockIdx.x * WPB + (threadIdx.x >> 6);
    const int n_items = *n_items_ptr;
    if (item >= n_items || item >= max_items) return;
    const int begin = __builtin_amdgcn_readfirstlane(wi_begin[item]);
    const int end = __builtin_amdgcn_readfirstlane(wi_end[item]);
    const int target = __builtin_amdgcn_readfirstlane(wi_target[item]);
    const int g = lane / RL;
    const int c = lane % RL;
    const size_t row_bytes = (size_t)F * (IN_F32 ? 4 : 2);
    Acc8 acc;
    acc.lo = acc.hi = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int base = begin; base < end; base += 64) {
        const int n = (end - base) < 64 ? (end - base) : 64;
        int my_row = 0;
        float my_w = 1.f;
        if (lane < n) {
            const int p = base + lane;
            my_row = src_row != nullptr ? src_row[p] : p;
            if (HAS_W) my_w = weight[perm != nullptr ? perm[p] : p];
            if (HAS_RS) my_w *= row_scale[my_row];
        }
        for (int j = 0; j < n; j += G * U) {
            // keep the rows in flight in their RAW form (bf16: 4 VGPRs per row, not 8) and widen
            // one at a time while accumulating
            typedef typename std::conditional<IN_F32, Acc8, u16x8>::type Raw;
            Raw val[U];
            float w[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int e = j + u * G + g;
                int r;
                if (G == 1) {
                    r = __builtin_amdgcn_readlane(my_row, (j + u) & 63);
                    w[u] = (HAS_W || HAS_RS)
                               ? __builtin_bit_cast(float, __builtin_amdgcn_readlane(
                                                               __builtin_bit_cast(int, my_w), (j + u) & 63))
                               : 1.f;
                } else {
                    r = __shfl(my_row, e & 63);
                    w[u] = (HAS_W || HAS_RS) ? __shfl(my_w, e & 63) : 1.f;
                }
                const char* rp = (const char*)src + (size_t)r * row_bytes;
                if (e < n && c < ncol) {
                    if constexpr (IN_F32) val[u] = load8<true>(rp, c * 8);
                    else val[u] = *(const u16x8*)((const unsigned short*)rp + c * 8);
                } else {
                    if constexpr (IN_F32) val[u].lo = val[u].hi = f32x4{0.f, 0.f, 0.f, 0.f};
                    else val[u] = u16x8{0, 0, 0, 0, 0, 0, 0, 0};
                }
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                Acc8 x;
                if constexpr (IN_F32) x = val[u];
                else x = widen(val[u]);
                if (HAS_W || HAS_RS) {
                    acc.lo += x.lo * w[u];
                    acc.hi += x.hi * w[u];
                } else {
                    acc.lo += x.lo;
                    acc.hi += x.hi;
                }
            }
        }
    }
    if (G > 1) {
#pragma unroll
        for (int off = RL; off < 64; off <<= 1) {
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                acc.lo[k] += __shfl_xor(acc.lo[k], off);
                acc.hi[k] += __shfl_xor(acc.hi[k], off);
            }
        }
    }
    if (g == 0 && c < ncol) {
        if (target >= 0)
            store8<false>(out + (size_t)target * (size_t)F, c * 8, acc);
        else
            store8<true>(partial + (size_t)(~target) * (size_t)F, c * 8, acc);
    }
}

// out[perm[p],:] = w[perm[p]] * table[dst,:]  (destination order; bf16 in, bf16 out)
template <int RL, bool HAS_W, int WPB>
__global__ __launch_bounds__(WPB * 64) void k_spread_rows_bf16(
    const unsigned short* __restrict__ table, int F, int ncol, const int32_t* __restrict__ perm,
    const float* __restrict__ weight, const int32_t* __restrict__ wi_begin, const int32_t* __restrict__ wi_end,
    const int32_t* __restrict__ wi_dst, const int32_t* __restrict__ n_items_ptr, int64_t max_items,
    unsigned short* __restrict__ out) {
    constexpr int G = 64 / RL;
    const int lane = threadIdx.x & 63;
    const int64_t item = (int64_t)blockIdx.x * WPB + (threadIdx.x >> 6);
    const int n_items = *n_items_ptr;
    if (item >= n_items || item >= max_items) return;
    const int begin = __builtin_amdgcn_readfirstlane(wi_begin[item]);
    const int end = __builtin_amdgcn_readfirstlane(wi_end[item]);
    if (begin >= end) return;
    const int dst = __builtin_amdgcn_readfirstlane(wi_dst[item]);
    const int g = lane / RL;
    const int c = lane % RL;
    u16x8 raw = {0, 0, 0, 0, 0, 0, 0, 0};
    if (c < ncol) raw = *(const u16x8*)(table + (size_t)dst * (size_t)F + c * 8);
    const Acc8 row = widen(raw);
    for (int base = begin; base < end; base += 64) {
        const int n = (end - base) < 64 ? (end - base) : 64;
        int my_e = 0;
        float my_w = 1.f;
        if (lane < n) {
            my_e = perm[base + lane];
            if (HAS_W) my_w = weight[my_e];
        }
        for (int j = 0; j < n; j += G) {
            const int k = j + g;
            const int e = __shfl(my_e, k & 63);
            const float w = HAS_W ? __shfl(my_w, k & 63) : 1.f;
            if (k < n && c < ncol) {
                unsigned short* op = out + (size_t)e * (size_t)F + c * 8;
                if (HAS_W) {
                    Acc8 x;
                    x.lo = row.lo * w;
                    x.hi = row.hi * w;
                    *(u16x8*)op = narrow(x);
                } else {
                    *(u16x8*)op = raw;
                }
            }
        }
    }
}

// out[e,:] = w[e] * table[idx[e],:]  (edge order; idx < 0 -> zeros)
template <int RL, bool HAS_W>
__global__ __launch_bounds__(256) void k_gather_rows_bf16(const unsigned short* __restrict__ table, int F, int ncol,
                                                          const int32_t* __restrict__ idx, int64_t M,
                                                          const float* __restrict__ weight,
                                                          unsigned short* __restrict__ out) {
    constexpr int G = 64 / RL;
    const int lane = threadIdx.x & 63;
    const int g = lane / RL;
    const int c = lane % RL;
    const int64_t n_tiles = (M + 63) / 64;
    for (int64_t tile = (int64_t)blockIdx.x * kWavesPerBlock + (threadIdx.x >> 6); tile < n_tiles;
         tile += (int64_t)gridDim.x * kWavesPerBlock) {
        const int64_t base = tile * 64;
        const int n = (M - base) < 64 ? (int)(M - base) : 64;
        int my_idx = -1;
        float my_w = 1.f;
        if (lane < n) {
            my_idx = idx[base + lane];
            if (HAS_W) my_w = weight[base + lane];
        }
        for (int j = 0; j < n; j += G) {
            const int k = j + g;
            const int r = __shfl(my_idx, k & 63);
            const float w = HAS_W ? __shfl(my_w, k & 63) : 1.f;
            if (k < n && c < ncol) {
                u16x8 raw = {0, 0, 0, 0, 0, 0, 0, 0};
                if (r >= 0) raw = *(const u16x8*)(table + (size_t)r * (size_t)F + c * 8);
                if (HAS_W) {
                    Acc8 x = widen(raw);
                    x.lo *= w;
                    x.hi *= w;
                    raw = narrow(x);
                }
                *(u16x8*)(out + (size_t)(base + k) * (size_t)F + c * 8) = raw;
            }
        }
    }
}

static bool bf16_shape_ok(int F) { return F > 0 && F % 8 == 0 && F <= 512; }

template <int RL, int U, int WPB>
static void launch_seg_bf16(const hgnn_plan* plan, const void* src, int F, const float* weight,
                            const float* row_scale, unsigned short* out, float* partial, hipStream_t s) {
    const int ncol = F / 8;
    const unsigned grid = (unsigned)ceil_div(plan->max_work, WPB);
    const int32_t* n_items = plan->counts + HGNN_CNT_WORK;
#define HGNN_B(W_, RS_)                                                                                   \
    k_seg_reduce_bf16<RL, U, W_, RS_, false, WPB><<<grid, WPB * 64, 0, s>>>(                              \
        src, F, ncol, plan->src_row, plan->perm, weight, row_scale, plan->wi_begin, plan->wi_end,         \
        plan->wi_target, n_items, plan->max_work, out, partial)
    if (grid) {
        if (weight && row_scale) HGNN_B(true, true);
        else if (weight) HGNN_B(true, false);
        else HGNN_B(false, false);
    }
#undef HGNN_B
    // combine pass: fp32 partial rows -> bf16 output rows of the split destinations
    const unsigned grid2 = (unsigned)ceil_div(plan->max_split, 4);
    if (grid2)
        k_seg_reduce_bf16<RL, 4, false, false, true, 4><<<grid2, 256, 0, s>>>(
            partial, F, ncol, nullptr, nullptr, nullptr, nullptr, plan->split_pbegin, plan->split_pbegin + 1,
            plan->split_dst, plan->counts + HGNN_CNT_SPLIT, plan->max_split, out, partial);
}

}  // namespace hgnn

using namespace hgnn;

extern "C" int hgnn_segment_reduce_bf16(const hgnn_plan* plan, const void* src, int32_t F, const float* weight,
                                        const float* row_scale, void* out, float* partial,
                                        hgnn_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    HGNN_REQUIRE(plan != nullptr, "hgnn_segment_reduce_bf16: plan is NULL");
    if (!bf16_shape_ok(F)) {
        set_error("hgnn_segment_reduce_bf16: F must be a multiple of 8 and <= 512 (got %d)", F);
        return HGNN_ERR_UNSUPPORTED;
    }
    if (plan->n_dst == 0) return HGNN_OK;
    HGNN_REQUIRE(out != nullptr && (plan->n_rows == 0 || src != nullptr) && partial != nullptr,
                 "hgnn_segment_reduce_bf16: NULL pointer");
    HGNN_REQUIRE(!(row_scale && !weight), "hgnn_segment_reduce_bf16: row_scale requires weight");
    HGNN_REQUIRE(plan->src_row != nullptr || !plan->has_gather,
                 "hgnn_segment_reduce_bf16: src_row may only be NULL for a sorted plan without gather");
    HGNN_REQUIRE((uintptr_t)src % 16 == 0 && (uintptr_t)out % 16 == 0 && (uintptr_t)partial % 16 == 0,
                 "hgnn_segment_reduce_bf16: src/out/partial must be 16-byte aligned");
    const int ncol = F / 8;
    unsigned short* o = (unsigned short*)out;
    if (ncol <= 4) launch_seg_bf16<4, 4, 4>(plan, src, F, weight, row_scale, o, partial, stream);
    else if (ncol <= 8) launch_seg_bf16<8, 4, 4>(plan, src, F, weight, row_scale, o, partial, stream);
    else if (ncol <= 16) launch_seg_bf16<16, 4, 4>(plan, src, F, weight, row_scale, o, partial, stream);
    else if (ncol <= 32) launch_seg_bf16<32, 4, 8>(plan, src, F, weight, row_scale, o, partial, stream);
    else launch_seg_bf16<64, 16, 16>(plan, src, F, weight, row_scale, o, partial, stream);
    HGNN_CHECK_HIP(hipGetLastError());
    return HGNN_OK;
}

extern "C" int hgnn_spread_rows_bf16(const hgnn_plan* plan, const void* table, int32_t F, const float* weight,
                                     void* out, hgnn_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    HGNN_REQUIRE(plan != nullptr, "hgnn_spread_rows_bf16: plan is NULL");
    if (!bf16_shape_ok(F)) {
        set_error("hgnn_spread_rows_bf16: F must be a multiple of 8 and <= 512 (got %d)", F);
        return HGNN_ERR_UNSUPPORTED;
    }
    if (plan->n_rows == 0 || plan->n_dst == 0) return HGNN_OK;
    HGNN_REQUIRE(!plan->has_gather, "hgnn_spread_rows_bf16: plan must be a plain destination plan");
    HGNN_REQUIRE(table != nullptr && out != nullptr, "hgnn_spread_rows_bf16: NULL pointer");
    HGNN_REQUIRE((uintptr_t)table % 16 == 0 && (uintptr_t)out % 16 == 0,
                 "hgnn_spread_rows_bf16: table/out must be 16-byte aligned");
    const int ncol = F / 8;
    const int32_t* n_items = plan->counts + HGNN_CNT_WORK;
    const unsigned short* t = (const unsigned short*)table;
    unsigned short* o = (unsigned short*)out;
#define HGNN_S(RL)                                                                                          \
    do {                                                                                                    \
        const unsigned grid = (unsigned)ceil_div(plan->max_work, 8);                                        \
        if (weight)                                                                                         \
            k_spread_rows_bf16<RL, true, 8><<<grid, 512, 0, stream>>>(t, F, ncol, plan->perm, weight,       \
                plan->wi_begin, plan->wi_end, plan->wi_dst, n_items, plan->max_work, o);                    \
        else                                                                                                \
            k_spread_rows_bf16<RL, false, 8><<<grid, 512, 0, stream>>>(t, F, ncol, plan->perm, weight,      \
                plan->wi_begin, plan->wi_end, plan->wi_dst, n_items, plan->max_work, o);                    \
    } while (0)
    if (ncol <= 4) HGNN_S(4);
    else if (ncol <= 8) HGNN_S(8);
    else if (ncol <= 16) HGNN_S(16);
    else if (ncol <= 32) HGNN_S(32);
    else HGNN_S(64);
#undef HGNN_S
    HGNN_CHECK_HIP(hipGetLastError());
    return HGNN_OK;
}

extern "C" int hgnn_gather_rows_bf16(const void* table, int64_t table_rows, int32_t F, const int32_t* idx,
                                     int64_t M, const float* weight, void* out, hgnn_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    HGNN_REQUIRE(M >= 0 && table_rows >= 0, "hgnn_gather_rows_bf16: bad sizes");
    if (!bf16_shape_ok(F)) {
        set_error("hgnn_gather_rows_bf16: F must be a multiple of 8 and <= 512 (got %d)", F);
        return HGNN_ERR_UNSUPPORTED;
    }
    if (M == 0) return HGNN_OK;
    HGNN_REQUIRE(table != nullptr && idx != nullptr && out != nullptr, "hgnn_gather_rows_bf16: NULL pointer");
    HGNN_REQUIRE((uintptr_t)table % 16 == 0 && (uintptr_t)out % 16 == 0,
                 "hgnn_gather_rows_bf16: table/out must be 16-byte aligned");
    const int ncol = F / 8;
    int64_t blocks = ceil_div(ceil_div(M, 64), kWavesPerBlock);
    if (blocks > 8192) blocks = 8192;
    const unsigned grid = (unsigned)blocks;
    const unsigned short* t = (const unsigned short*)table;
    unsigned short* o = (unsigned short*)out;
#define HGNN_G(RL)                                                                                   \
    do {                                                                                             \
        if (weight) k_gather_rows_bf16<RL, true><<<grid, 256, 0, stream>>>(t, F, ncol, idx, M, weight, o); \
        else k_gather_rows_bf16<RL, false><<<grid, 256, 0, stream>>>(t, F, ncol, idx, M, weight, o);  \
    } while (0)
    if (ncol <= 4) HGNN_G(4);
    else if (ncol <= 8) HGNN_G(8);
    else if (ncol <= 16) HGNN_G(16);
    else if (ncol <= 32) HGNN_G(32);
    else HGNN_G(64);
#undef HGNN_G
    HGNN_CHECK_HIP(hipGetLastError());
    return HGNN_OK;
}
