// Shared device helpers of the fused MLP kernels (fp32: mlp_fused.hip, bf16: mlp_fused_bf16.hip).
#pragma once
#include "common.h"

namespace hgnn {

typedef float f32x4 __attribute__((ext_vector_type(4)));

// Activations.  v_mfma_f32_16x16x4_f32 runs on the SIMD's fp32 vector ALUs (its rate IS the VALU FMA
// rate), so epilogue VALU work does not hide under a co-resident wave's MFMAs: every VALU
// instruction here is paid in full (ablation: LayerNorm+act were 15.7 % of the kernel with libm
// erff/tanhf).  Hence branch-free forms built on v_exp_f32 / v_rcp_f32:
//   erf : Abramowitz-Stegun 7.1.26, |abs error| <= 1.5e-7  (exact-GELU parity bar is 1e-4 rel)
//   tanh: 1 - 2/(exp(2|x|)+1), abs error ~1e-7
__device__ __forceinline__ float fast_erf(float x) {
    const float ax = __builtin_fabsf(x);
    const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, ax, 1.0f));
    float p = fmaf(1.061405429f, t, -1.453152027f);
    p = fmaf(p, t, 1.421413741f);
    p = fmaf(p, t, -0.284496736f);
    p = fmaf(p, t, 0.254829592f);
    p *= t;
    const float e = __builtin_amdgcn_exp2f(-1.44269504088896341f * ax * ax);
    const float r = fmaf(-p, e, 1.0f);
    return __builtin_copysignf(r, x);
}

__device__ __forceinline__ float fast_tanh(float x) {
    const float ax = __builtin_fabsf(x);
    const float e = __builtin_amdgcn_exp2f(2.88539008177792681f * ax);  // exp(2|x|)
    const float r = fmaf(-2.0f, __builtin_amdgcn_rcpf(e + 1.0f), 1.0f);
    return __builtin_copysignf(r, x);
}

__device__ __forceinline__ float act_apply(float x, int act) {
    switch (act) {
        case HGNN_ACT_GELU: return 0.5f * x * (1.0f + fast_erf(x * 0.70710678118654752440f));
        case HGNN_ACT_TANH: return fast_tanh(x);
        case HGNN_ACT_RELU: return x > 0.f ? x : 0.f;
        default: return x;
    }
}

// exact-GELU derivative: Phi(y) + y * phi(y)
__device__ __forceinline__ float act_grad(float y, float a, int act) {
    switch (act) {
        case HGNN_ACT_GELU: {
            const float cdf = 0.5f * (1.0f + fast_erf(y * 0.70710678118654752440f));
            const float pdf = 0.3989422804014327f * __builtin_amdgcn_exp2f(-0.72134752044448170368f * y * y);
            return fmaf(y, pdf, cdf);
        }
        case HGNN_ACT_TANH: return fmaf(-a, a, 1.0f);
        case HGNN_ACT_RELU: return y > 0.f ? 1.0f : 0.f;
        default: return 1.0f;
    }
}

// Weight staging.  W[NF][Kdim] row-major (torch Linear.weight); chunk = columns [k0, k0+16).
// One 1-KiB LDS-DMA piece covers 16 rows: lane (i = lane&15, g = lane>>4) fetches
// W[16p+i][k0+4g .. +3] and the DMA lands it at lane*16 bytes, i.e. the piece is stored in exactly
// the order the MFMA A-fragment read (ds_read_b128 at lane*16) wants: conflict-free, no swizzle.
// Address = wave-uniform piece base (SALU) + a fixed 32-bit per-lane byte offset.
// A tiny state machine so that only ONE per-lane 64-bit source pointer and one scalar LDS address
// stay live across the MFMA loop (eight precomputed piece addresses cost 16 VGPRs and pushed the
// L=256 kernel into scratch).
struct WStage {
    const char* src;        // per-lane source of the NEXT piece to issue
    unsigned lds;           // LDS byte address of the NEXT piece's destination (wave-uniform)
    unsigned piece_stride;  // bytes between this wave's consecutive pieces in W (64 rows)
};

__device__ __forceinline__ void wait_dma() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }

// Issued as inline asm on purpose: hipcc orders every later ds_read behind a builtin LDS-DMA with
// s_waitcnt vmcnt(0) (it cannot see that the DMA fills the OTHER buffer), which would serialise
// each piece's memory latency into the MFMA loop.  The hand-off is done by hand instead: every wave
// drains its DMAs (wait_dma) right before the barrier that publishes them.
__device__ __forceinline__ void dma_piece(const char* src, unsigned lds_addr) {
    asm volatile(
        "s_mov_b32 m0, %1\n\t"
        "s_nop 0\n\t"
        "global_load_lds_dwordx4 %0, off"
        :
        : "v"(src), "s"(lds_addr)
        : "memory");  // m0 is a reserved register: hipcc re-materialises it before each of its own uses
}

__device__ __forceinline__ unsigned lds_addr_of(float* p) {
    return (unsigned)(size_t)(__attribute__((address_space(3))) float*)p;
}

// start staging one chunk (16 bytes per lane and row: 4 fp32 or 8 bf16 k-values per lane group) of a
// row-major weight matrix into `lds`: points at this wave's first piece
// (NW = waves per workgroup: piece p of a chunk is staged by wave p % NW)
template <int NW = 4>
__device__ __forceinline__ WStage begin_stage_bytes(const void* W, size_t row_bytes, size_t chunk_byte_off,
                                                    float* lds, int wave, int lane) {
    WStage st;
    st.src = (const char*)W + (size_t)(wave * 16 + (lane & 15)) * row_bytes + chunk_byte_off + (lane >> 4) * 16;
    st.lds = lds_addr_of(lds) + (unsigned)wave * 1024u;
    st.piece_stride = (unsigned)(NW * 16 * row_bytes);
    return st;
}

template <int NW = 4>
__device__ __forceinline__ WStage begin_stage(const float* W, int Kdim, int k0, float* lds, int wave, int lane) {
    return begin_stage_bytes<NW>(W, (size_t)Kdim * sizeof(float), (size_t)k0 * sizeof(float), lds, wave, lane);
}

template <int NW = 4>
__device__ __forceinline__ void stage_next(WStage& st) {
    dma_piece(st.src, st.lds);
    st.src += st.piece_stride;
    st.lds += NW * 1024u;  // NW pieces further
}

// number of pieces wave `wave` owns of an NF-row chunk (pieces p = wave, wave+NW, ...)
template <int NF, int NW = 4>
__device__ __forceinline__ int pieces_of(int wave) {
    constexpr int PIECES = NF / 16;
    return PIECES % NW == 0 ? PIECES / NW : (PIECES / NW + (wave < PIECES % NW ? 1 : 0));
}

}  // namespace hgnn
