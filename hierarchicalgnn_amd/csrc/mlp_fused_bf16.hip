// bf16 variant of the fused  gather -> concat -> [Linear -> LayerNorm -> act] x {2,3} -> (+skip)  MLP
// (BASELINE config 4 dtype) on v_mfma_f32_16x16x32_bf16: 16x the fp32 matrix rate.
//
// Same mapping as mlp_fused.hip -- transposed GEMMs D[f][e] = sum_k W[f][k] X[e][k], one wave owns
// 16 edges and every feature, LayerNorm statistics in registers (fp32), weights through LDS by
// LDS-DMA, activations straight from global memory -- with these differences:
//   * a k-chunk is 32 values: lane (e = lane&15, g = lane>>4) holds X[e][k0+8g .. +7] (one 16-byte
//     load of the bf16 row), the A fragment is W[f][k0+8g .. +7] (one ds_read_b128);
//   * the fp32 accumulator tile of layer i is converted to bf16 and packed into the B operand of
//     layer i+1: a 32-wide k-block is built from TWO 16-feature tiles, so k-slot (g, j) carries
//     feature 32kb + (j<4 ? 4g+j : 16+4g+j-4).  The host stores W_{i+1} with its columns in that
//     slot order (fused.py), so the A fragment is still one contiguous 16-byte read;
//   * inputs / skip / output are bf16, weights bf16 (converted from the fp32 master copy per call),
//     bias and LayerNorm parameters fp32, all accumulation and LayerNorm arithmetic fp32;
//   * bf16 MFMA does NOT share the fp32 vector ALUs, so LayerNorm/GELU epilogues of one wave overlap
//     the co-resident waves' MFMAs.
#include "mlp_common.h"

namespace hgnn {

typedef unsigned short u16x8 __attribute__((ext_vector_type(8)));
typedef unsigned short u16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

struct MlpArgsBf16 {
    const unsigned short* seg_table[3];
    const int32_t* seg_index[3];
    int seg_width[3];
    int n_seg;
    int K1;
    const unsigned short* W[3];  // bf16, row-major [out][in]; layers >= 1 column-permuted (see header)
    const float* b[3];
    const float* lnw[3];
    const float* lnb[3];
    int act[3];
    float eps;
    const unsigned short* skip;
    unsigned short* out;
    long long M;
};

__device__ __forceinline__ bf16x8 as_bf16(u16x8 v) { return __builtin_bit_cast(bf16x8, v); }

__device__ __forceinline__ unsigned short to_bf16_bits(float x) {
    return __builtin_bit_cast(unsigned short, (__bf16)x);
}

// pack two activated fp32 tiles (features 32kb+4g+r and 32kb+16+4g+r) into one B fragment
__device__ __forceinline__ u16x8 pack_kblock(const f32x4& t0, const f32x4& t1) {
    u16x8 r;
    r[0] = to_bf16_bits(t0.x); r[1] = to_bf16_bits(t0.y); r[2] = to_bf16_bits(t0.z); r[3] = to_bf16_bits(t0.w);
    r[4] = to_bf16_bits(t1.x); r[5] = to_bf16_bits(t1.y); r[6] = to_bf16_bits(t1.z); r[7] = to_bf16_bits(t1.w);
    return r;
}

// bf16 epilogue.  bf16 MFMA leaves the fp32 vector ALUs to the epilogue, and at 16x the matrix rate
// the kernel is VALU-bound (LayerNorm + GELU per hidden element), so the epilogue is written for
// instruction count at bf16-level accuracy (outputs are rounded to 8 significant bits anyway):
//   * one-pass LayerNorm statistics (sum and sum of squares in fp32),
//   * LayerNorm applied as two FMAs,
//   * GELU in its tanh form folded to x * sigmoid(2u): 7 instructions instead of 17
//     (|error| vs the erf form <= 5e-4 absolute, below one bf16 ulp of the result).
__device__ __forceinline__ float gelu_bf16(float x) {
    // -2*log2(e)*sqrt(2/pi) * (1 + 0.044715 x^2) * x
    const float t = x * fmaf(-0.10294324f, x * x, -2.30220820f);
    return x * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(t));
}

template <int ACT>
__device__ __forceinline__ float act_b(float x, int act) {
    const int code = ACT >= 0 ? ACT : act;
    if (code == HGNN_ACT_GELU) return gelu_bf16(x);
    return act_apply(x, code);
}

template <int NT, int ACT>
__device__ __forceinline__ void layernorm_act_b(f32x4 (&acc)[NT], const float* __restrict__ lnw,
                                                const float* __restrict__ lnb, int act, float eps, int g) {
    constexpr float inv_n = 1.0f / (float)(NT * 16);
    float s = 0.f, q = 0.f;
#pragma unroll
    for (int T = 0; T < NT; ++T) {
        s += (acc[T].x + acc[T].y) + (acc[T].z + acc[T].w);
        q = fmaf(acc[T].x, acc[T].x, q);
        q = fmaf(acc[T].y, acc[T].y, q);
        q = fmaf(acc[T].z, acc[T].z, q);
        q = fmaf(acc[T].w, acc[T].w, q);
    }
    s += __shfl_xor(s, 16);
    q += __shfl_xor(q, 16);
    s += __shfl_xor(s, 32);
    q += __shfl_xor(q, 32);
    const float mean = s * inv_n;
    const float var = fmaxf(fmaf(-mean, mean, q * inv_n), 0.f);
    const float rstd = 1.0f / sqrtf(var + eps);
    const float shift = -mean * rstd;
#pragma unroll
    for (int T = 0; T < NT; ++T) {
        const f32x4 w4 = *(const f32x4*)(lnw + T * 16 + g * 4);
        const f32x4 b4 = *(const f32x4*)(lnb + T * 16 + g * 4);
        f32x4 v;
        v.x = act_b<ACT>(fmaf(fmaf(acc[T].x, rstd, shift), w4.x, b4.x), act);
        v.y = act_b<ACT>(fmaf(fmaf(acc[T].y, rstd, shift), w4.y, b4.y), act);
        v.z = act_b<ACT>(fmaf(fmaf(acc[T].z, rstd, shift), w4.z, b4.z), act);
        v.w = act_b<ACT>(fmaf(fmaf(acc[T].w, rstd, shift), w4.w, b4.w), act);
        acc[T] = v;
    }
}

// one 32-wide k-chunk:  acc[eg][T] += Wchunk[T] * b[eg]  for every 16-feature tile and each of the
// wave's EG edge groups (one A fragment feeds EG MFMAs), A fragments pipelined one pair ahead, the
// LDS-DMA pieces of a later chunk issued between MFMA groups
template <int EG, int NT, int NF_NEXT>
__device__ __forceinline__ void mma_chunk_b(f32x4 (&acc)[EG][NT], const float* __restrict__ wb,
                                            const u16x8 (&b)[EG], WStage& st, int n_pieces) {
    static_assert(NT % 2 == 0, "tiles are processed in pairs");
    constexpr int PAIRS = NT / 2;
    constexpr int PER_WAVE = (NF_NEXT / 16 + 3) / 4;
    constexpr int EVERY = PAIRS >= PER_WAVE ? PAIRS / PER_WAVE : 1;
    bf16x8 bb[EG];
#pragma unroll
    for (int eg = 0; eg < EG; ++eg) bb[eg] = as_bf16(b[eg]);
    u16x8 w0 = *(const u16x8*)(wb);
    u16x8 w1 = *(const u16x8*)(wb + 256);
    __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
    int issued = 0;
#pragma unroll
    for (int T = 0; T < NT; T += 2) {
        u16x8 n0 = w0, n1 = w1;
        if (T + 2 < NT) {
            n0 = *(const u16x8*)(wb + (T + 2) * 256);
            n1 = *(const u16x8*)(wb + (T + 3) * 256);
        }
        if (((T / 2) % EVERY == 0) && issued < PER_WAVE) {
            if (__builtin_amdgcn_readfirstlane(issued < n_pieces ? 1 : 0)) stage_next(st);
            ++issued;
        }
#pragma unroll
        for (int eg = 0; eg < EG; ++eg) {
            acc[eg][T] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(as_bf16(w0), bb[eg], acc[eg][T], 0, 0, 0);
            acc[eg][T + 1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(as_bf16(w1), bb[eg], acc[eg][T + 1], 0, 0, 0);
        }
        w0 = n0;
        w1 = n1;
        if (T + 2 < NT) __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
        __builtin_amdgcn_sched_group_barrier(0x008, 2 * EG, 0);
    }
    for (int i = issued; i < n_pieces; ++i) stage_next(st);
}

template <int NT>
__device__ __forceinline__ void init_bias_b(f32x4 (&acc)[NT], const float* __restrict__ b, int g) {
#pragma unroll
    for (int T = 0; T < NT; ++T) acc[T] = *(const f32x4*)(b + T * 16 + g * 4);
}

// s_waitcnt vmcnt(N) with a compile-time N (the asm immediate)
template <int N>
__device__ __forceinline__ void wait_vm() {
    static_assert(N >= 0 && N < 64, "vmcnt is a 6-bit field");
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

// register-resident layer: `in[eg]` = packed k-blocks of the previous layer's activations.
// NBUF-deep LDS ring: the pieces of chunk c+NBUF-1 are issued during chunk c, so a chunk's DMA has
// NBUF-1 chunk times to land (one chunk of bf16 MFMAs is shorter than an L2 round trip).
template <int EG, int NBUF, int NKB, int NTO>
__device__ __forceinline__ void dense_from_regs_b(const u16x8 (&in)[EG][NKB], f32x4 (&out)[EG][NTO],
                                                  const unsigned short* __restrict__ W, float* lds, int wave,
                                                  int lane) {
    constexpr int KD = NKB * 32;          // input features
    constexpr int BUF = NTO * 256;        // floats (= 1 KiB pieces) per chunk buffer
    constexpr int PIECES = NTO;           // 16-row pieces per chunk
    constexpr bool EVEN = PIECES % 4 == 0;
    constexpr int PW = PIECES / 4;        // pieces per wave when EVEN
    const int n_pieces = pieces_of<NTO * 16>(wave);
    const size_t row_bytes = (size_t)KD * 2;
    __syncthreads();
#pragma unroll
    for (int p = 0; p < NBUF - 1; ++p) {
        if (p < NKB) {
            WStage st = begin_stage_bytes(W, row_bytes, (size_t)p * 64, lds + p * BUF, wave, lane);
            for (int i = 0; i < n_pieces; ++i) stage_next(st);
        }
    }
#pragma unroll
    for (int c = 0; c < NKB; ++c) {
        // chunk c must have landed; chunks c+1 .. c+NBUF-2 (younger) may stay in flight
        constexpr int AHEAD = NBUF - 2;
        if (EVEN && c + AHEAD < NKB) wait_vm<(EVEN ? AHEAD * PW : 0)>();
        else wait_vm<0>();
        __syncthreads();
        const float* wb = lds + (c % NBUF) * BUF + lane * 4;
        const int cn = c + NBUF - 1;
        WStage st = begin_stage_bytes(W, row_bytes, (size_t)cn * 64, lds + (cn % NBUF) * BUF, wave, lane);
        u16x8 b[EG];
#pragma unroll
        for (int eg = 0; eg < EG; ++eg) b[eg] = in[eg][c];
        mma_chunk_b<EG, NTO, NTO * 16>(out, wb, b, st, cn < NKB ? n_pieces : 0);
    }
}

template <int NT>
__device__ __forceinline__ void pack_all(const f32x4 (&acc)[NT], u16x8 (&out)[NT / 2]) {
#pragma unroll
    for (int kb = 0; kb < NT / 2; ++kb) out[kb] = pack_kblock(acc[2 * kb], acc[2 * kb + 1]);
}

template <int NT>
__device__ __forceinline__ void store_out_b(const f32x4 (&acc)[NT], const MlpArgsBf16& a, long long e, bool valid,
                                            int g) {
    if (!valid) return;
    constexpr int NOUT = NT * 16;
    unsigned short* op = a.out + (size_t)e * NOUT + g * 4;
    const unsigned short* sp = a.skip != nullptr ? a.skip + (size_t)e * NOUT + g * 4 : nullptr;
#pragma unroll
    for (int T = 0; T < NT; ++T) {
        f32x4 v = acc[T];
        if (sp != nullptr) {
            const u16x4 s = *(const u16x4*)(sp + T * 16);
            v.x += __builtin_bit_cast(float, (unsigned)s[0] << 16);
            v.y += __builtin_bit_cast(float, (unsigned)s[1] << 16);
            v.z += __builtin_bit_cast(float, (unsigned)s[2] << 16);
            v.w += __builtin_bit_cast(float, (unsigned)s[3] << 16);
        }
        u16x4 o;
        o[0] = to_bf16_bits(v.x);
        o[1] = to_bf16_bits(v.y);
        o[2] = to_bf16_bits(v.z);
        o[3] = to_bf16_bits(v.w);
        *(u16x4*)(op + T * 16) = o;
    }
}

// per-lane X stream of one edge: a running pointer that hops to the next segment's row
struct XStream {
    const unsigned short* q1;
    const unsigned short* q2;
    const unsigned short* px;
};

// EG: 16-edge groups per wave (1 or 2); NBUF: LDS weight-ring depth (2 or 3)
template <int NT1, int NT2, int NT3, int MINW, int ACT_H, int ACT_O, int EG, int NBUF>
__global__ __launch_bounds__(256, MINW) void k_fused_mlp_bf16(const MlpArgsBf16 a) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int ei = lane & 15;
    const int g = lane >> 4;
    long long e[EG];
    bool valid[EG];
    XStream xs[EG];
#pragma unroll
    for (int eg = 0; eg < EG; ++eg) {
        e[eg] = ((long long)blockIdx.x * 4 + wave) * (16 * EG) + eg * 16 + ei;
        valid[eg] = e[eg] < a.M;
        const long long er = valid[eg] ? e[eg] : 0;
        long long r = a.seg_index[0] != nullptr ? (long long)a.seg_index[0][er] : er;
        xs[eg].px = a.seg_table[0] + (size_t)(r < 0 ? 0 : r) * (size_t)a.seg_width[0] + g * 8;
        xs[eg].q1 = xs[eg].px;
        xs[eg].q2 = xs[eg].px;
        if (a.n_seg > 1) {
            r = a.seg_index[1] != nullptr ? (long long)a.seg_index[1][er] : er;
            xs[eg].q1 = a.seg_table[1] + (size_t)(r < 0 ? 0 : r) * (size_t)a.seg_width[1] + g * 8;
        }
        if (a.n_seg > 2) {
            r = a.seg_index[2] != nullptr ? (long long)a.seg_index[2][er] : er;
            xs[eg].q2 = a.seg_table[2] + (size_t)(r < 0 ? 0 : r) * (size_t)a.seg_width[2] + g * 8;
        }
    }
    const int nc = a.K1 / 32;
    const int c1 = a.seg_width[0] / 32;
    const int c2 = c1 + (a.n_seg > 1 ? a.seg_width[1] / 32 : nc);
    // The X operand also travels by LDS-DMA: every wave gathers its own B fragments (per-lane source
    // row, lane*16-byte destination = B-fragment order) into a private 4-slot ring and reads them back
    // with ds_read_b128.  Reason: with X in ordinary global loads hipcc guards their first use with
    // `s_waitcnt vmcnt(k)`, k counting only ITS loads; the hand-issued weight DMAs sit in the same
    // in-order queue, so that wait drained the weight ring every chunk (59 % of wave time parked).
    // With every VMEM op of the loop issued by hand, the counted drains below are the only waits.
    constexpr int MAXNT = NT1 > NT2 ? (NT1 > NT3 ? NT1 : NT3) : (NT2 > NT3 ? NT2 : NT3);
    float* xring = lds + NBUF * MAXNT * 256 + wave * (4 * EG * 256);
    int cl = 0;  // next chunk the X streams will fetch (the same for every edge group)
    auto issue_x = [&]() {
#pragma unroll
        for (int eg = 0; eg < EG; ++eg) {
            if (cl < nc) dma_piece((const char*)xs[eg].px, lds_addr_of(xring + ((cl & 3) * EG + eg) * 256));
            xs[eg].px += 32;
            if (cl + 1 == c1) xs[eg].px = xs[eg].q1;
            if (cl + 1 == c2) xs[eg].px = xs[eg].q2;
        }
        ++cl;
    };

    // ---------------- layer 1
    f32x4 acc1[EG][NT1];
#pragma unroll
    for (int eg = 0; eg < EG; ++eg) init_bias_b<NT1>(acc1[eg], a.b[0], g);
    __builtin_amdgcn_s_setprio(2);
    {
        constexpr int BUF = NT1 * 256;
        constexpr bool EVEN = NT1 % 4 == 0;
        constexpr int PW = NT1 / 4;
        const int n_pieces = pieces_of<NT1 * 16>(wave);
        const size_t row_bytes = (size_t)a.K1 * 2;
        // prologue = the "virtual iterations" -3, -2, -1 in the SAME issue order as the steady state
        // (weights of chunk i+NBUF-1, then X of chunk i+3), so that the counted drain holds from c = 0
#pragma unroll
        for (int i = -3; i < 0; ++i) {
            const int cw = i + NBUF - 1;
            if (cw >= 0 && cw < nc) {
                WStage st = begin_stage_bytes(a.W[0], row_bytes, (size_t)cw * 64, lds + (cw % NBUF) * BUF, wave, lane);
                for (int k = 0; k < n_pieces; ++k) stage_next(st);
            }
            issue_x();
        }
        // Queue order per iteration: [weight pieces of chunk c+NBUF-1] then [EG X pieces of chunk c+3].
        // When iteration c begins, the ops YOUNGER than the data it needs (weights of chunk c, issued in
        // iteration c-NBUF+1; X of chunk c, older still) are (NBUF-2) later chunks of weight pieces and
        // (NBUF-1) iterations' X pieces: that many may stay in flight.  The last iterations issue fewer ops and drain completely.
        for (int c = 0; c < nc; ++c) {
            if (EVEN && c + NBUF + 2 < nc) wait_vm<(EVEN ? (NBUF - 2) * PW + (NBUF - 1) * EG : 0)>();
            else wait_vm<0>();
            __syncthreads();
            const float* wb = lds + (c % NBUF) * BUF + lane * 4;
            const int cn = c + NBUF - 1;
            WStage st = begin_stage_bytes(a.W[0], row_bytes, (size_t)cn * 64, lds + (cn % NBUF) * BUF, wave, lane);
            u16x8 x0[EG];
#pragma unroll
            for (int eg = 0; eg < EG; ++eg) x0[eg] = *(const u16x8*)(xring + ((c & 3) * EG + eg) * 256 + lane * 4);
            mma_chunk_b<EG, NT1, NT1 * 16>(acc1, wb, x0, st, cn < nc ? n_pieces : 0);
            issue_x();  // chunk c+3 into the slot chunk c-1 was read from
        }
    }
    __builtin_amdgcn_s_setprio(0);
    u16x8 h1[EG][NT1 / 2];
#pragma unroll
    for (int eg = 0; eg < EG; ++eg) {
        layernorm_act_b<NT1, ACT_H>(acc1[eg], a.lnw[0], a.lnb[0], a.act[0], a.eps, g);
        pack_all<NT1>(acc1[eg], h1[eg]);
    }

    // ---------------- layer 2 (and 3)
    f32x4 acc2[EG][NT2];
#pragma unroll
    for (int eg = 0; eg < EG; ++eg) init_bias_b<NT2>(acc2[eg], a.b[1], g);
    __builtin_amdgcn_s_setprio(2);
    dense_from_regs_b<EG, NBUF, NT1 / 2, NT2>(h1, acc2, a.W[1], lds, wave, lane);
    __builtin_amdgcn_s_setprio(0);
    if constexpr (NT3 == 0) {
#pragma unroll
        for (int eg = 0; eg < EG; ++eg) {
            layernorm_act_b<NT2, ACT_O>(acc2[eg], a.lnw[1], a.lnb[1], a.act[1], a.eps, g);
            store_out_b<NT2>(acc2[eg], a, e[eg], valid[eg], g);
        }
    } else {
        u16x8 h2[EG][NT2 / 2];
#pragma unroll
        for (int eg = 0; eg < EG; ++eg) {
            layernorm_act_b<NT2, ACT_H>(acc2[eg], a.lnw[1], a.lnb[1], a.act[1], a.eps, g);
            pack_all<NT2>(acc2[eg], h2[eg]);
        }
        f32x4 acc3[EG][NT3];
#pragma unroll
        for (int eg = 0; eg < EG; ++eg) init_bias_b<NT3>(acc3[eg], a.b[2], g);
        __builtin_amdgcn_s_setprio(2);
        dense_from_regs_b<EG, NBUF, NT2 / 2, NT3>(h2, acc3, a.W[2], lds, wave, lane);
        __builtin_amdgcn_s_setprio(0);
#pragma unroll
        for (int eg = 0; eg < EG; ++eg) {
            layernorm_act_b<NT3, ACT_O>(acc3[eg], a.lnw[2], a.lnb[2], a.act[2], a.eps, g);
            store_out_b<NT3>(acc3[eg], a, e[eg], valid[eg], g);
        }
    }
}

template <int NT1, int NT2, int NT3, int MINW, int ACT_H, int ACT_O, int EG, int NBUF>
static int launch_b_act(const MlpArgsBf16& a, hipStream_t s) {
    constexpr int maxnt = NT1 > NT2 ? (NT1 > NT3 ? NT1 : NT3) : (NT2 > NT3 ? NT2 : NT3);
    const size_t lds_bytes = ((size_t)NBUF * maxnt + 4 * 4 * EG) * 256 * sizeof(float);  // weight ring + X rings
    const unsigned grid = (unsigned)ceil_div(a.M, 64 * EG);
    auto kern = k_fused_mlp_bf16<NT1, NT2, NT3, MINW, ACT_H, ACT_O, EG, NBUF>;
    if (lds_bytes > 64 * 1024) {
        HGNN_CHECK_HIP(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize,
                                           (int)lds_bytes));
    }
    kern<<<grid, 256, lds_bytes, s>>>(a);
    HGNN_CHECK_HIP(hipGetLastError());
    return HGNN_OK;
}


template <int NT1, int NT2, int NT3, int MINW, int EG, int NBUF>
static int launch_b_shape(const MlpArgsBf16& a, hipStream_t s) {
    const int n = NT3 == 0 ? 2 : 3;
    bool hidden_gelu = true;
    for (int l = 0; l + 1 < n; ++l) hidden_gelu = hidden_gelu && a.act[l] == HGNN_ACT_GELU;
    const int out = a.act[n - 1];
    if (hidden_gelu && out == HGNN_ACT_TANH) return launch_b_act<NT1, NT2, NT3, MINW, HGNN_ACT_GELU, HGNN_ACT_TANH, EG, NBUF>(a, s);
    if (hidden_gelu && out == HGNN_ACT_GELU) return launch_b_act<NT1, NT2, NT3, MINW, HGNN_ACT_GELU, HGNN_ACT_GELU, EG, NBUF>(a, s);
    return launch_b_act<NT1, NT2, NT3, MINW, -1, -1, EG, NBUF>(a, s);
}

template <int NT1, int NT2, int NT3, int MINW>
static int launch_b(const MlpArgsBf16& a, hipStream_t s) {
    // wide layers: 32 edges per wave (one weight fragment feeds two MFMAs), one workgroup per CU,
    // 3-deep weight ring; narrow layers keep 16 edges per wave at higher occupancy
    if constexpr (NT1 >= 16) {
        return launch_b_shape<NT1, NT2, NT3, 1, 2, 3>(a, s);
    }
    return launch_b_shape<NT1, NT2, NT3, MINW, 1, 2>(a, s);
}

}  // namespace hgnn

using namespace hgnn;

extern "C" int hgnn_mlp_supported_bf16(const hgnn_mlp_desc* d) {
    if (d == nullptr) return 0;
    if (d->n_seg < 1 || d->n_seg > 3 || (d->n_layers != 2 && d->n_layers != 3)) return 0;
    int k = 0;
    for (int s = 0; s < d->n_seg; ++s) {
        if (d->seg_width[s] <= 0 || d->seg_width[s] % 32 != 0) return 0;
        k += d->seg_width[s];
    }
    if (k != d->width[0] || d->w0_cols != 0 || d->w_last_rows != 0) return 0;
    const int n = d->n_layers;
    for (int l = 0; l < n; ++l)
        if (d->W[l] == nullptr || d->b[l] == nullptr || d->ln_w[l] == nullptr || d->ln_b[l] == nullptr) return 0;
    if (d->save_pre[0] || d->save_pre[1] || d->save_pre[2] || d->n_pre != 0) return 0;
    const int h = d->width[1];
    const int o = d->width[n];
    if (n == 3 && d->width[2] != h) return 0;
    if (h != 2 * o) return 0;
    return (o == 32 || o == 64 || o == 128 || o == 256) ? 1 : 0;
}

extern "C" int hgnn_mlp_forward_bf16(const hgnn_mlp_desc* d, void* out, hgnn_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    HGNN_REQUIRE(d != nullptr && out != nullptr, "hgnn_mlp_forward_bf16: NULL argument");
    if (!hgnn_mlp_supported_bf16(d)) {
        set_error("hgnn_mlp_forward_bf16: unsupported shape (K -> 2L (-> 2L) -> L, LayerNorm on every layer, "
                  "L in {32,64,128,256}, every segment a multiple of 32 wide)");
        return HGNN_ERR_UNSUPPORTED;
    }
    if (d->M == 0) return HGNN_OK;
    HGNN_REQUIRE(d->M > 0, "hgnn_mlp_forward_bf16: bad M");
    MlpArgsBf16 a;
    for (int s = 0; s < 3; ++s) {
        a.seg_table[s] = s < d->n_seg ? (const unsigned short*)d->seg_table[s] : nullptr;
        a.seg_index[s] = s < d->n_seg ? d->seg_index[s] : nullptr;
        a.seg_width[s] = s < d->n_seg ? d->seg_width[s] : 0;
        if (s < d->n_seg) {
            HGNN_REQUIRE(a.seg_table[s] != nullptr && (uintptr_t)a.seg_table[s] % 16 == 0,
                         "hgnn_mlp_forward_bf16: segment table %d is NULL or not 16-byte aligned", s);
        }
    }
    a.n_seg = d->n_seg;
    a.K1 = d->width[0];
    for (int l = 0; l < 3; ++l) {
        const bool on = l < d->n_layers;
        a.W[l] = on ? (const unsigned short*)d->W[l] : nullptr;
        a.b[l] = on ? d->b[l] : nullptr;
        a.lnw[l] = on ? d->ln_w[l] : nullptr;
        a.lnb[l] = on ? d->ln_b[l] : nullptr;
        a.act[l] = on ? d->act[l] : 0;
        if (on) {
            HGNN_REQUIRE((uintptr_t)a.W[l] % 16 == 0 && (uintptr_t)a.b[l] % 16 == 0 &&
                             (uintptr_t)a.lnw[l] % 16 == 0 && (uintptr_t)a.lnb[l] % 16 == 0,
                         "hgnn_mlp_forward_bf16: layer %d parameters must be 16-byte aligned", l);
        }
    }
    a.eps = d->ln_eps;
    a.skip = (const unsigned short*)d->skip;
    a.out = (unsigned short*)out;
    a.M = d->M;
    HGNN_REQUIRE((uintptr_t)out % 8 == 0 && (uintptr_t)a.skip % 8 == 0,
                 "hgnn_mlp_forward_bf16: out/skip must be 8-byte aligned");
    const int o = d->width[d->n_layers];
    if (d->n_layers == 2) {
        switch (o) {
            case 32: return launch_b<4, 2, 0, 2>(a, stream);
            case 64: return launch_b<8, 4, 0, 2>(a, stream);
            case 128: return launch_b<16, 8, 0, 2>(a, stream);
            case 256: return launch_b<32, 16, 0, 2>(a, stream);
        }
    } else {
        switch (o) {
            case 32: return launch_b<4, 4, 2, 2>(a, stream);
            case 64: return launch_b<8, 8, 4, 2>(a, stream);
            case 128: return launch_b<16, 16, 8, 2>(a, stream);
            case 256: return launch_b<32, 32, 16, 2>(a, stream);
        }
    }
    set_error("hgnn_mlp_forward_bf16: no instantiation");
    return HGNN_ERR_UNSUPPORTED;
}
