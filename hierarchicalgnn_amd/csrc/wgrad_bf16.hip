// Weight gradient of one Linear layer of the bf16 training path, hand-written for bf16 MFMA:
//
//     dW[ho][hi] = sum_m dz[m][ho] * a[m][hi]            dz: [M, Ho] bf16,  a: [M, Hi] bf16,  dW: fp32
//
// (backward of Modules/utils.py:169-196's Linear layers as used by the edge / node networks,
// Modules/gnn_utils.py:22-41, over the M = 2,000,000 directed edges of an event).  This is a "TN" GEMM whose
// REDUCTION runs over the 2M rows and whose result is tiny (512 x 256): the library picks an output-tile
// decomposition for it and reaches 140-370 TFLOP/s (3.3-4.3 ms at latent 256, tools/bench_gemm_bf16_lib.py);
// the operation is really an HBM stream (3 GB of rows per 0.5 TFLOP), so the design here is:
//   * split-K: the rows are cut into S slices, one workgroup per (output tile, slice); every row of dz / a
//     is read Hi/TI resp. Ho/TO times in total, at 16 B per lane, coalesced;
//   * both MFMA operands need 8 CONSECUTIVE m per lane for a fixed column -- the transpose of the row-major
//     tiles.  The tiles are stored row-major in LDS exactly as they arrive ([32 m][T columns], ds_write_b128)
//     and read with gfx950's transposing LDS load ds_read_b64_tr_b16: per 16-lane group a 4-row x 16-column
//     block comes back column-major, two such reads = one 16x16x32 operand (k = 8*(lane/16) .. +7).
//     Row stride = T*2 + 32 bytes puts the 8 rows a 32-lane half touches on disjoint banks;
//   * D[ho][hi] accumulates in registers over the whole slice (64 fp32 per lane for a 64 x 64 wave tile),
//     written once as a partial [S][Ho][Hi]; a second tiny kernel adds the S partials in slice order.
//     No atomics: the result is bitwise reproducible.
#include "common.h"

namespace hgnn {
namespace wg {

typedef unsigned short u16x8 __attribute__((ext_vector_type(8)));
typedef short v4s __attribute__((ext_vector_type(4)));
typedef short v8s __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int KT = 32;  // rows of m per step = the k depth of one v_mfma_f32_16x16x32_bf16

// transposed 16x16x32 operand: columns c0 .. c0+15 of rows 0 .. 31 of an LDS tile with `stride` bytes per row
__device__ __forceinline__ bf16x8 tr_operand(const char* tile, int stride, int c0, int lane) {
    const int kg = lane >> 4, q = (lane & 15) >> 2, p = lane & 3;
    const char* a0 = tile + (8 * kg + q) * stride + (c0 + 4 * p) * 2;
    const v4s lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) v4s*)a0);
    const v4s hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) v4s*)(a0 + 4 * stride));
    const v8s v = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
    return __builtin_bit_cast(bf16x8, v);
}

// (hi, mid) bf16 parts of 8 fp32 values: x ~= hi + mid to 16 significand bits (split-bf16 products, mlp_split3_f32.hip)
__device__ __forceinline__ void split8(const f32x4 lo4, const f32x4 hi4, u16x8& h, u16x8& m) {
#pragma unroll
    for (int c = 0; c < 8; ++c) {
        const float x = c < 4 ? lo4[c] : hi4[c - 4];
        const __bf16 hb = (__bf16)x;
        h[c] = __builtin_bit_cast(unsigned short, hb);
        m[c] = __builtin_bit_cast(unsigned short, (__bf16)(x - (float)hb));
    }
}

// TO x TI output tile per workgroup, WO x WI waves, each wave (TO/WO) x (TI/WI).
// S3: the operands are FP32 rows (A, B point at floats, lda / ldb count floats); every tile is split into a hi and a
// mid bf16 plane while it is written to LDS and  dz^T a ~= mid.mid + mid.hi + hi.mid + hi.hi  runs as four MFMAs -- the weight
// gradient of the fp32 training backward at 3/16 of the fp32 matrix instruction's time (hgnn_wgrad_f32_split3).
template <int TO, int TI, int WO, int WI, bool S3 = false>
__global__ __launch_bounds__(WO * WI * 64) void k_wgrad_bf16(const unsigned short* __restrict__ A, long long lda,
                                                             const unsigned short* __restrict__ B, long long ldb,
                                                             long long M, int Ho, int Hi, float* __restrict__ partial,
                                                             long long rows_per_slice, float* __restrict__ partial_cs) {
    constexpr int NTHR = WO * WI * 64;
    constexpr int SA = TO * 2 + 32, SB = TI * 2 + 32;  // LDS row strides in bytes
    constexpr int PA = TO / 8, PB = TI / 8;            // 16-byte pieces per tile row
    constexpr int NPA = KT * PA / NTHR, NPB = KT * PB / NTHR;
    static_assert(KT * PA % NTHR == 0 && KT * PB % NTHR == 0, "tile pieces must divide over the threads");
    constexpr int FO = TO / WO / 16, FI = TI / WI / 16;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* As = smem;
    char* Bs = smem + 2 * KT * SA;
    constexpr int MID = 2 * KT * (SA + SB);   // S3: the mid planes follow the hi planes
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wo = wave % WO, wi = wave / WO;
    const int n_to = (Ho + TO - 1) / TO;
    const int ho0 = (int)(blockIdx.x % n_to) * TO;
    const int hi0 = (int)(blockIdx.x / n_to) * TI;
    const long long m_begin = (long long)blockIdx.y * rows_per_slice;
    long long m_end = m_begin + rows_per_slice;
    if (m_end > M) m_end = M;
    const int steps = m_end > m_begin ? (int)((m_end - m_begin + KT - 1) / KT) : 0;

    u16x8 ra[NPA], rb[NPB];
    f32x4 fa[S3 ? NPA : 1][2], fb[S3 ? NPB : 1][2];
    auto load = [&](int s) {
        if constexpr (S3) {
            const long long m0 = m_begin + (long long)s * KT;
            const float* Af = (const float*)A;
            const float* Bf = (const float*)B;
            const f32x4 z4 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int i = 0; i < NPA; ++i) {
                const int idx = tid + i * NTHR;
                const int row = idx / PA, col = ho0 + (idx % PA) * 8;
                const bool ok = m0 + row < m_end && col < Ho;
                const float* p = Af + (size_t)(m0 + row) * (size_t)lda + col;
                fa[i][0] = ok ? *(const f32x4*)p : z4;
                fa[i][1] = ok ? *(const f32x4*)(p + 4) : z4;
            }
#pragma unroll
            for (int i = 0; i < NPB; ++i) {
                const int idx = tid + i * NTHR;
                const int row = idx / PB, col = hi0 + (idx % PB) * 8;
                const bool ok = m0 + row < m_end && col < Hi;
                const float* p = Bf + (size_t)(m0 + row) * (size_t)ldb + col;
                fb[i][0] = ok ? *(const f32x4*)p : z4;
                fb[i][1] = ok ? *(const f32x4*)(p + 4) : z4;
            }
            return;
        }
        const long long m0 = m_begin + (long long)s * KT;
#pragma unroll
        for (int i = 0; i < NPA; ++i) {
            const int idx = tid + i * NTHR;
            const int row = idx / PA, col = ho0 + (idx % PA) * 8;
            const u16x8 zero = {0, 0, 0, 0, 0, 0, 0, 0};
            ra[i] = (m0 + row < m_end && col < Ho) ? *(const u16x8*)(A + (size_t)(m0 + row) * (size_t)lda + col) : zero;
        }
#pragma unroll
        for (int i = 0; i < NPB; ++i) {
            const int idx = tid + i * NTHR;
            const int row = idx / PB, col = hi0 + (idx % PB) * 8;
            const u16x8 zero = {0, 0, 0, 0, 0, 0, 0, 0};
            rb[i] = (m0 + row < m_end && col < Hi) ? *(const u16x8*)(B + (size_t)(m0 + row) * (size_t)ldb + col) : zero;
        }
    };
    auto store = [&](int buf) {
        if constexpr (S3) {
#pragma unroll
            for (int i = 0; i < NPA; ++i) {
                const int idx = tid + i * NTHR;
                u16x8 h, m;
                split8(fa[i][0], fa[i][1], h, m);
                char* d = As + buf * KT * SA + (idx / PA) * SA + (idx % PA) * 16;
                *(u16x8*)d = h;
                *(u16x8*)(d + MID) = m;
            }
#pragma unroll
            for (int i = 0; i < NPB; ++i) {
                const int idx = tid + i * NTHR;
                u16x8 h, m;
                split8(fb[i][0], fb[i][1], h, m);
                char* d = Bs + buf * KT * SB + (idx / PB) * SB + (idx % PB) * 16;
                *(u16x8*)d = h;
                *(u16x8*)(d + MID) = m;
            }
            return;
        }
#pragma unroll
        for (int i = 0; i < NPA; ++i) {
            const int idx = tid + i * NTHR;
            *(u16x8*)(As + buf * KT * SA + (idx / PA) * SA + (idx % PA) * 16) = ra[i];
        }
#pragma unroll
        for (int i = 0; i < NPB; ++i) {
            const int idx = tid + i * NTHR;
            *(u16x8*)(Bs + buf * KT * SB + (idx / PB) * SB + (idx % PB) * 16) = rb[i];
        }
    };

    f32x4 acc[FO][FI];
#pragma unroll
    for (int fo = 0; fo < FO; ++fo)
#pragma unroll
        for (int fi = 0; fi < FI; ++fi) acc[fo][fi] = f32x4{0.f, 0.f, 0.f, 0.f};
    // optional column sums of A (= the Linear's bias gradient, sum_m dz[m][ho]): one more MFMA per A fragment against
    // an all-ones operand, by the waves of the first hi-tile column only (wave-uniform condition)
    const bool do_cs = partial_cs != nullptr && hi0 == 0 && wi == 0;
    f32x4 acs[FO];
#pragma unroll
    for (int fo = 0; fo < FO; ++fo) acs[fo] = f32x4{0.f, 0.f, 0.f, 0.f};
    const bf16x8 ones = {(__bf16)1.0f, (__bf16)1.0f, (__bf16)1.0f, (__bf16)1.0f,
                         (__bf16)1.0f, (__bf16)1.0f, (__bf16)1.0f, (__bf16)1.0f};

    if (steps > 0) {
        load(0);
        store(0);
    }
    __syncthreads();
    for (int s = 0; s < steps; ++s) {   // `steps` is uniform over the workgroup: every wave takes every barrier
        const bool more = s + 1 < steps;
        if (more) load(s + 1);
        const char* at = As + (s & 1) * KT * SA;
        const char* bt = Bs + (s & 1) * KT * SB;
        bf16x8 b[FI];
#pragma unroll
        for (int fi = 0; fi < FI; ++fi) b[fi] = tr_operand(bt, SB, wi * (TI / WI) + fi * 16, lane);
        if constexpr (S3) {
            bf16x8 bm[FI];
#pragma unroll
            for (int fi = 0; fi < FI; ++fi) bm[fi] = tr_operand(bt + MID, SB, wi * (TI / WI) + fi * 16, lane);
#pragma unroll
            for (int fo = 0; fo < FO; ++fo) {
                const bf16x8 a = tr_operand(at, SA, wo * (TO / WO) + fo * 16, lane);
                const bf16x8 am = tr_operand(at + MID, SA, wo * (TO / WO) + fo * 16, lane);
#pragma unroll
                for (int fi = 0; fi < FI; ++fi) {
                    // four products: the kernel is an HBM stream, the mid.mid term (2^-16 of a product) is free here
                    // and halves the truncation error of a reduction over millions of rows
                    acc[fo][fi] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(am, bm[fi], acc[fo][fi], 0, 0, 0);
                    acc[fo][fi] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(am, b[fi], acc[fo][fi], 0, 0, 0);
                    acc[fo][fi] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, bm[fi], acc[fo][fi], 0, 0, 0);
                    acc[fo][fi] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b[fi], acc[fo][fi], 0, 0, 0);
                }
                if (do_cs) {
                    acs[fo] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(am, ones, acs[fo], 0, 0, 0);
                    acs[fo] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, ones, acs[fo], 0, 0, 0);
                }
            }
        } else {
#pragma unroll
            for (int fo = 0; fo < FO; ++fo) {
                const bf16x8 a = tr_operand(at, SA, wo * (TO / WO) + fo * 16, lane);
#pragma unroll
                for (int fi = 0; fi < FI; ++fi)
                    acc[fo][fi] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b[fi], acc[fo][fi], 0, 0, 0);
                if (do_cs) acs[fo] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, ones, acs[fo], 0, 0, 0);
            }
        }
        if (more) store((s + 1) & 1);
        __syncthreads();
    }
    // D layout: column (hi) = lane & 15, rows (ho) = 4 * (lane >> 4) + r
    float* out = partial + (size_t)blockIdx.y * (size_t)Ho * (size_t)Hi;
#pragma unroll
    for (int fo = 0; fo < FO; ++fo) {
#pragma unroll
        for (int fi = 0; fi < FI; ++fi) {
            const int hi = hi0 + wi * (TI / WI) + fi * 16 + (lane & 15);
            const int ho = ho0 + wo * (TO / WO) + fo * 16 + 4 * (lane >> 4);
            if (hi < Hi) {
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    if (ho + r < Ho) out[(size_t)(ho + r) * (size_t)Hi + hi] = acc[fo][fi][r];
            }
        }
    }
    if (do_cs && (lane & 15) == 0) {   // every column of the all-ones product holds the same sums: take column 0
#pragma unroll
        for (int fo = 0; fo < FO; ++fo) {
            const int ho = ho0 + wo * (TO / WO) + fo * 16 + 4 * (lane >> 4);
#pragma unroll
            for (int r = 0; r < 4; ++r)
                if (ho + r < Ho) partial_cs[(size_t)blockIdx.y * (size_t)Ho + ho + r] = acs[fo][r];
        }
    }
}

__global__ __launch_bounds__(256) void k_wgrad_reduce(const float* __restrict__ partial, int slices, int Ho, int Hi,
                                                      float* __restrict__ out, long long ldo,
                                                      const float* __restrict__ partial_cs, float* __restrict__ colsum) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    const long long n = (long long)Ho * Hi;
    if (colsum != nullptr && i < Ho) {
        float s = 0.f;
        for (int k = 0; k < slices; ++k) s += partial_cs[(size_t)k * (size_t)Ho + i];   // slice order
        colsum[i] = s;
    }
    if (i >= n) return;
    // four independent chains (slices k, k+1, k+2, k+3 mod 4) keep four loads in flight; the chains are then
    // added in a fixed order: deterministic
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    int k = 0;
    for (; k + 3 < slices; k += 4) {
        s0 += partial[(size_t)k * (size_t)n + i];
        s1 += partial[(size_t)(k + 1) * (size_t)n + i];
        s2 += partial[(size_t)(k + 2) * (size_t)n + i];
        s3 += partial[(size_t)(k + 3) * (size_t)n + i];
    }
    for (; k < slices; ++k) s0 += partial[(size_t)k * (size_t)n + i];
    out[(size_t)(i / Hi) * (size_t)ldo + (i % Hi)] = (s0 + s1) + (s2 + s3);
}

struct Shape {
    int to, ti, tiles, slices;
    long long rows_per_slice;
};

static Shape shape_for(long long M, int Ho, int Hi) {
    Shape s;
    s.to = Ho > 128 ? 256 : 128;
    s.ti = (Ho > 128 && Hi > 128) ? 256 : 128;   // larger tiles = fewer re-reads of the 2M rows
    s.tiles = (int)(ceil_div(Ho, s.to) * ceil_div(Hi, s.ti));
    // ~2 workgroups per CU in total: every slice costs a [Ho, Hi] fp32 partial that the reduce pass reads back
    // (512 slices of a 512 x 256 gradient = 268 MB: the reduce took 184 us per call, 2/3 of the GEMM itself)
    long long want = ceil_div((long long)512, s.tiles);
    long long max_slices = ceil_div(M, (long long)KT * 8);   // at least 8 steps per slice
    if (want > max_slices) want = max_slices;
    if (want < 1) want = 1;
    s.rows_per_slice = ceil_div(ceil_div(M, want), (long long)KT) * KT;
    if (s.rows_per_slice < KT) s.rows_per_slice = KT;
    s.slices = (int)ceil_div(M > 0 ? M : 1, s.rows_per_slice);
    return s;
}

}  // namespace wg
}  // namespace hgnn

using namespace hgnn;

extern "C" int hgnn_wgrad_workspace_bytes(int64_t M, int32_t Ho, int32_t Hi, size_t* bytes) {
    HGNN_REQUIRE(bytes != nullptr && M >= 0 && Ho > 0 && Hi > 0, "hgnn_wgrad_workspace_bytes: bad argument");
    const wg::Shape s = wg::shape_for(M, Ho, Hi);
    *bytes = (size_t)s.slices * ((size_t)Ho * (size_t)Hi + (size_t)Ho) * sizeof(float);
    return HGNN_OK;
}

extern "C" int hgnn_wgrad_bf16(const void* A, int64_t lda, const void* B, int64_t ldb, int64_t M, int32_t Ho,
                               int32_t Hi, float* out, int64_t ldo, float* colsum, void* workspace,
                               size_t workspace_bytes, hgnn_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    HGNN_REQUIRE(M >= 0 && Ho > 0 && Hi > 0 && Ho % 8 == 0 && Hi % 8 == 0,
                 "hgnn_wgrad_bf16: Ho and Hi must be positive multiples of 8 (got %d, %d)", Ho, Hi);
    HGNN_REQUIRE(out != nullptr && ldo >= Hi, "hgnn_wgrad_bf16: out is NULL or ldo < Hi");
    HGNN_REQUIRE(lda >= Ho && ldb >= Hi && lda % 8 == 0 && ldb % 8 == 0,
                 "hgnn_wgrad_bf16: row strides must be multiples of 8 elements and cover the columns");
    HGNN_REQUIRE(M == 0 || (A != nullptr && B != nullptr && (uintptr_t)A % 16 == 0 && (uintptr_t)B % 16 == 0),
                 "hgnn_wgrad_bf16: operands are NULL or not 16-byte aligned");
    const wg::Shape s = wg::shape_for(M, Ho, Hi);
    const size_t need = (size_t)s.slices * ((size_t)Ho * (size_t)Hi + (size_t)Ho) * sizeof(float);
    HGNN_REQUIRE(workspace != nullptr && workspace_bytes >= need && (uintptr_t)workspace % 16 == 0,
                 "hgnn_wgrad_bf16: workspace too small (%zu < %zu) or unaligned", workspace_bytes, need);
    float* partial = (float*)workspace;
    float* partial_cs = colsum != nullptr ? partial + (size_t)s.slices * (size_t)Ho * (size_t)Hi : nullptr;
    const dim3 grid((unsigned)s.tiles, (unsigned)s.slices);
    if (s.to == 256 && s.ti == 256) {
        constexpr int TO = 256, TI = 256;
        const size_t lds = 2 * wg::KT * (size_t)((TO * 2 + 32) + (TI * 2 + 32));
        auto kern = wg::k_wgrad_bf16<TO, TI, 2, 4>;
        HGNN_CHECK_HIP(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        kern<<<grid, 512, lds, stream>>>((const unsigned short*)A, lda, (const unsigned short*)B, ldb, M, Ho, Hi, partial,
                                         s.rows_per_slice, partial_cs);
    } else if (s.to == 256) {
        constexpr int TO = 256, TI = 128;
        const size_t lds = 2 * wg::KT * (size_t)((TO * 2 + 32) + (TI * 2 + 32));
        wg::k_wgrad_bf16<TO, TI, 4, 2><<<grid, 512, lds, stream>>>((const unsigned short*)A, lda, (const unsigned short*)B,
                                                                   ldb, M, Ho, Hi, partial, s.rows_per_slice, partial_cs);
    } else {
        constexpr int TO = 128, TI = 128;
        const size_t lds = 2 * wg::KT * (size_t)((TO * 2 + 32) + (TI * 2 + 32));
        wg::k_wgrad_bf16<TO, TI, 2, 2><<<grid, 256, lds, stream>>>((const unsigned short*)A, lda, (const unsigned short*)B,
                                                                   ldb, M, Ho, Hi, partial, s.rows_per_slice, partial_cs);
    }
    wg::k_wgrad_reduce<<<(unsigned)ceil_div((int64_t)Ho * Hi, 256), 256, 0, stream>>>(partial, s.slices, Ho, Hi, out, ldo,
                                                                                      partial_cs, colsum);
    HGNN_CHECK_HIP(hipGetLastError());
    return HGNN_OK;
}

/* the same weight gradient for FP32 rows, products as split-bf16 (hi.hi + mid.hi + hi.mid): see k_wgrad_bf16<.., S3> */
extern "C" int hgnn_wgrad_f32_split3(const float* A, int64_t lda, const float* B, int64_t ldb, int64_t M, int32_t Ho,
                                     int32_t Hi, float* out, int64_t ldo, float* colsum, void* workspace,
                                     size_t workspace_bytes, hgnn_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    HGNN_REQUIRE(M >= 0 && Ho > 0 && Hi > 0 && Ho % 8 == 0 && Hi % 8 == 0,
                 "hgnn_wgrad_f32_split3: Ho and Hi must be positive multiples of 8 (got %d, %d)", Ho, Hi);
    HGNN_REQUIRE(out != nullptr && ldo >= Hi, "hgnn_wgrad_f32_split3: out is NULL or ldo < Hi");
    HGNN_REQUIRE(lda >= Ho && ldb >= Hi && lda % 4 == 0 && ldb % 4 == 0,
                 "hgnn_wgrad_f32_split3: row strides must be multiples of 4 floats and cover the columns");
    HGNN_REQUIRE(M == 0 || (A != nullptr && B != nullptr && (uintptr_t)A % 16 == 0 && (uintptr_t)B % 16 == 0),
                 "hgnn_wgrad_f32_split3: operands are NULL or not 16-byte aligned");
    const wg::Shape s = wg::shape_for(M, Ho, Hi);
    const size_t need = (size_t)s.slices * ((size_t)Ho * (size_t)Hi + (size_t)Ho) * sizeof(float);
    HGNN_REQUIRE(workspace != nullptr && workspace_bytes >= need && (uintptr_t)workspace % 16 == 0,
                 "hgnn_wgrad_f32_split3: workspace too small (%zu < %zu) or unaligned", workspace_bytes, need);
    float* partial = (float*)workspace;
    float* partial_cs = colsum != nullptr ? partial + (size_t)s.slices * (size_t)Ho * (size_t)Hi : nullptr;
    const dim3 grid((unsigned)s.tiles, (unsigned)s.slices);
    const unsigned short* Au = (const unsigned short*)A;
    const unsigned short* Bu = (const unsigned short*)B;
#define HGNN_WG3(TO_, TI_, WO_, WI_)                                                                              \
    do {                                                                                                          \
        const size_t lds = 2 * 2 * wg::KT * (size_t)((TO_ * 2 + 32) + (TI_ * 2 + 32));                            \
        auto kern = wg::k_wgrad_bf16<TO_, TI_, WO_, WI_, true>;                                                   \
        HGNN_CHECK_HIP(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); \
        kern<<<grid, WO_ * WI_ * 64, lds, stream>>>(Au, lda, Bu, ldb, M, Ho, Hi, partial, s.rows_per_slice, partial_cs); \
    } while (0)
    if (s.to == 256 && s.ti == 256) HGNN_WG3(256, 256, 2, 4);
    else if (s.to == 256) HGNN_WG3(256, 128, 4, 2);
    else HGNN_WG3(128, 128, 2, 2);
#undef HGNN_WG3
    wg::k_wgrad_reduce<<<(unsigned)ceil_div((int64_t)Ho * Hi, 256), 256, 0, stream>>>(partial, s.slices, Ho, Hi, out, ldo,
                                                                                      partial_cs, colsum);
    HGNN_CHECK_HIP(hipGetLastError());
    return HGNN_OK;
}
