// Shared device code of the feature-split bf16 MLP kernels: the forward (mlp_split_bf16.hip) and the fused
// backward layer (mlp_bwd_bf16.hip).  See mlp_split_bf16.hip for the decomposition.
#pragma once
#include "mlp_common.h"

namespace hgnn {
namespace fs {

typedef unsigned short u16x8 __attribute__((ext_vector_type(8)));
typedef unsigned short u16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x2 __attribute__((ext_vector_type(2)));


__device__ __forceinline__ bf16x8 as_bf16(u16x8 v) { return __builtin_bit_cast(bf16x8, v); }
__device__ __forceinline__ unsigned short bf16_bits(float x) { return __builtin_bit_cast(unsigned short, (__bf16)x); }
__device__ __forceinline__ float bf16_float(unsigned short b) { return __builtin_bit_cast(float, (unsigned)b << 16); }

// GELU in its tanh form folded to x * sigmoid(2u) (same as mlp_fused_bf16.hip: |error| vs the erf
// form <= 5e-4 absolute, below one bf16 ulp of the result)
__device__ __forceinline__ float gelu_t(float x) {
    const float t = x * fmaf(-0.10294324f, x * x, -2.30220820f);
    return x * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(t));
}

template <int ACT>
__device__ __forceinline__ float act_t(float x, int act) {
    const int code = ACT >= 0 ? ACT : act;
    if (code == HGNN_ACT_GELU) return gelu_t(x);
    return act_apply(x, code);
}

template <int NT>
struct Ring {
    static constexpr int R = NT > 8 ? NT : 8;  // fragments in flight per wave (16 measured: L=256 2.89 vs 2.79 ms (12 spilled registers), L=128 1.29 vs 1.08 ms (occupancy 3 -> 2), L=512 8.44 vs 8.52 ms: the ring depth is not what parks the waves)
    static constexpr int CPI = R / NT;         // = chunks the ring runs ahead
};

// the ring starts with chunks 0 .. CPI-1;  wp = this lane's pointer to fragment (chunk 0, tile 0 of the wave)
template <int NT, int NW>
__device__ __forceinline__ void ring_fill(u16x8 (&w)[Ring<NT>::R], const u16x8* __restrict__ wp, int total) {
    constexpr int CPI = Ring<NT>::CPI;
#pragma unroll
    for (int cc = 0; cc < CPI; ++cc) {
        const int c = cc < total ? cc : total - 1;
        const u16x8* p = wp + (size_t)c * (NW * NT * 64);
#pragma unroll
        for (int t = 0; t < NT; ++t) w[cc * NT + t] = p[t * 64];
    }
}

// acc[t][j] += W(chunks gc0 .. gc0+n) * B,  B = n k-chunks read from LDS rows of stride RS bytes
// (`bsrc` = this lane's base: row e, k-group g).  Chunk (c, t) sits in ring slot (c % CPI)*NT + t and
// is replaced, right after its 4 MFMAs, by chunk c+CPI of the weight stream (clamped at the end: a
// few unused loads instead of a branch inside the unrolled body).  n % max(2, CPI) == 0.
template <int NT, int RS, int VAR, int NW, int NJ>
__device__ __forceinline__ void gemm_lds(f32x4 (&acc)[NT][NJ], u16x8 (&w)[Ring<NT>::R],
                                         const u16x8* __restrict__ wp, int gc0, int total,
                                         const char* bsrc, int kx, int n, int ablate) {
    constexpr int CPI = Ring<NT>::CPI;
    constexpr int U = CPI < 2 ? 2 : CPI;
    // NJ = 4: the next chunk's B fragments are read into a second register set at the start of the
    // chunk; NJ = 8 (32 registers per set): each fragment is re-read IN PLACE right after its last
    // MFMA of the chunk (the last weight tile), 8 MFMAs = 128 cycles before its next use
    constexpr bool INPLACE = NJ > 4;
    u16x8 b[INPLACE ? 1 : 2][NJ];
#pragma unroll
    // LDS rows are XOR-swizzled in 16-byte pieces: logical piece q of row r sits at piece q ^ (r & 15).  A lane
    // reads piece 4c + g of row 16j + e, i.e. physical piece (4 (c ^ kx)) + (g ^ (e & 3)) with kx = e >> 2: `bsrc`
    // already carries the (g ^ (e & 3)) part, the chunk index is XORed here.  Every 16-lane group of ds_read_b128
    // ({0-3,12-15,20-27}, ...) then touches 16 distinct 16-byte slots of the 64 banks: conflict-free (with the
    // former +16-byte row padding the groups' lanes (e=11,g=1) and (e=12,g=0) shared a slot: 2-way).
    for (int j = 0; j < NJ; ++j) b[0][j] = *(const u16x8*)(bsrc + j * 16 * RS + (kx << 6));
    for (int c = 0; c < n; c += U) {
#pragma unroll
        for (int u = 0; u < U; ++u) {
            // VAR 2 (two waves per SIMD, 32 MFMAs per chunk): a chunk = one uninterrupted burst.  Park
            // ONCE until the whole chunk's operands have landed, then issue its MFMAs (and the next
            // chunk's loads) back to back; the partner wave's loads fly meanwhile.  With per-fragment
            // counted waits both waves stall in small steps all the time (A/B in one process at
            // L=256: 3.3 -> 2.7 ms; s_setprio around the burst: no change; a counted wait every 4
            // fragments instead: 4.5 ms).  Shorter chunks (L=128)
            // and one wave per SIMD (L=512) are faster with the counted waits (VAR 0).
            if (VAR == 2) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
            int cn = c + u + 1 < n ? c + u + 1 : n - 1;
            if (ablate & 16) cn = 0;
            constexpr int cur = INPLACE ? 0 : 0;
            const int bi = INPLACE ? 0 : (u & 1);
            if constexpr (!INPLACE) {
                const int cx = (cn ^ kx) << 6;
#pragma unroll
                for (int j = 0; j < NJ; ++j) b[(u + 1) & 1][j] = *(const u16x8*)(bsrc + j * 16 * RS + cx);
                // pin the issue order (hipcc otherwise sinks every load to just before its first use,
                // i.e. an L2 round trip behind 4 MFMAs): the next chunk's B reads first, then per tile
                // NJ MFMAs followed by the ring refill that runs 8 fragments ahead
                __builtin_amdgcn_sched_group_barrier(0x100, NJ, 0);
            }
            (void)cur;
            int gn = gc0 + c + u + CPI < total ? gc0 + c + u + CPI : total - 1;
            if (ablate & 1) gn = 0;
            const u16x8* wn = wp + (size_t)gn * (NW * NT * 64);
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                const int slot = (u % CPI) * NT + t;
#pragma unroll
                for (int j = 0; j < NJ; ++j) {
                    acc[t][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(as_bf16(w[slot]), as_bf16(b[bi][j]),
                                                                        acc[t][j], 0, 0, 0);
                    if (INPLACE && t == NT - 1) {
                        b[0][j] = *(const u16x8*)(bsrc + j * 16 * RS + ((cn ^ kx) << 6));
                        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                    }
                }
                w[slot] = wn[t * 64];
                if (!(INPLACE && t == NT - 1)) __builtin_amdgcn_sched_group_barrier(0x008, NJ, 0);
                __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
            }
        }
    }
}

constexpr int cmax(int a, int b) { return a > b ? a : b; }

}  // namespace fs
}  // namespace hgnn
