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

// one 32-wide k-chunk:  acc[T] += Wchunk[T] * b  for every 16-feature tile, A fragments pipelined one
// pair ahead, the next chunk's LDS-DMA pieces issued between MFMA groups
template <int NT, int NF_NEXT>
__device__ __forceinline__ void mma_chunk_b(f32x4 (&acc)[NT], const float* __restrict__ wb, const u16x8 b,
                                            WStage& st, int n_pieces) {
    static_assert(NT % 2 == 0, "tiles are processed in pairs");
    constexpr int PAIRS = NT / 2;
    constexpr int PER_WAVE = (NF_NEXT / 16 + 3) / 4;
    constexpr int EVERY = PAIRS >= PER_WAVE ? PAIRS / PER_WAVE : 1;
    const bf16x8 bb = as_bf16(b);
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
        acc[T] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(as_bf16(w0), bb, acc[T], 0, 0, 0);
        acc[T + 1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(as_bf16(w1), bb, acc[T + 1], 0, 0, 0);
        w0 = n0;
        w1 = n1;
        if (T + 2 < NT) __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
        __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
    }
    for (int i = issued; i < n_pieces; ++i) stage_next(st);
}

template <int NT>
__device__ __forceinline__ void init_bias_b(f32x4 (&acc)[NT], const float* __restrict__ b, int g) {
#pragma unroll
    for (int T = 0; T < NT; ++T) acc[T] = *(const f32x4*)(b + T * 16 + g * 4);
}

// register-resident layer: `in` = packed k-blocks of the previous layer's activations
template <int NKB, int NTO>
__device__ __forceinline__ void dense_from_regs_b(const u16x8 (&in)[NKB], f32x4 (&out)[NTO],
                                                  const unsigned short* __restrict__ W, float* lds, int wave,
                                                  int lane) {
    constexpr int KD = NKB * 32;          // input features
    constexpr int BUF = NTO * 256;        // floats (= 1 KiB pieces) per chunk buffer
    const int n_pieces = pieces_of<NTO * 16>(wave);
    const size_t row_bytes = (size_t)KD * 2;
    __syncthreads();
    {
        WStage st = begin_stage_bytes(W, row_bytes, 0, lds, wave, lane);
        for (int i = 0; i < n_pieces; ++i) stage_next(st);
    }
#pragma unroll
    for (int c = 0; c < NKB; ++c) {
        wait_dma();
        __syncthreads();
        const float* wb = lds + (c & 1) * BUF + lane * 4;
        WStage st = begin_stage_bytes(W, row_bytes, (size_t)(c + 1) * 64, lds + ((c + 1) & 1) * BUF, wave, lane);
        mma_chunk_b<NTO, NTO * 16>(out, wb, in[c], st, c + 1 < NKB ? n_pieces : 0);
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

template <int NT1, int NT2, int NT3, int MINW, int ACT_H, int ACT_O>
__global__ __launch_bounds__(256, MINW) void k_fused_mlp_bf16(const MlpArgsBf16 a) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int ei = lane & 15;
    const int g = lane >> 4;
    const long long e = (long long)blockIdx.x * 64 + wave * 16 + ei;
    const bool valid = e < a.M;
    const long long er = valid ? e : 0;

    // per-lane row start (+ 8g elements) of every input segment; one running pointer hops segments
    const unsigned short* q0;
    const unsigned short* q1;
    const unsigned short* q2;
    {
        long long r = a.seg_index[0] != nullptr ? (long long)a.seg_index[0][er] : er;
        q0 = a.seg_table[0] + (size_t)(r < 0 ? 0 : r) * (size_t)a.seg_width[0] + g * 8;
        q1 = q0;
        q2 = q0;
        if (a.n_seg > 1) {
            r = a.seg_index[1] != nullptr ? (long long)a.seg_index[1][er] : er;
            q1 = a.seg_table[1] + (size_t)(r < 0 ? 0 : r) * (size_t)a.seg_width[1] + g * 8;
        }
        if (a.n_seg > 2) {
            r = a.seg_index[2] != nullptr ? (long long)a.seg_index[2][er] : er;
            q2 = a.seg_table[2] + (size_t)(r < 0 ? 0 : r) * (size_t)a.seg_width[2] + g * 8;
        }
    }
    const int nc = a.K1 / 32;
    const int c1 = a.seg_width[0] / 32;
    const int c2 = c1 + (a.n_seg > 1 ? a.seg_width[1] / 32 : nc);
    const unsigned short* px = q0;
    int cl = 0;
    auto next_x = [&]() -> u16x8 {
        u16x8 v = {0, 0, 0, 0, 0, 0, 0, 0};
        if (cl < nc) v = *(const u16x8*)px;
        ++cl;
        px += 32;
        if (cl == c1) px = q1;
        if (cl == c2) px = q2;
        return v;
    };

    // ---------------- layer 1
    f32x4 acc1[NT1];
    init_bias_b<NT1>(acc1, a.b[0], g);
    __builtin_amdgcn_s_setprio(2);
    {
        constexpr int BUF = NT1 * 256;
        const int n_pieces = pieces_of<NT1 * 16>(wave);
        const size_t row_bytes = (size_t)a.K1 * 2;
        {
            WStage st = begin_stage_bytes(a.W[0], row_bytes, 0, lds, wave, lane);
            for (int i = 0; i < n_pieces; ++i) stage_next(st);
        }
        // X stream: four chunks in flight.  A chunk's MFMAs take only ~512 cycles here (16x the fp32
        // rate), far less than a global-load latency, so (i) the prefetch distance is 4 chunks and
        // (ii) the new X load is issued AFTER this iteration's DMA pieces, which lets the drain before
        // the barrier be `vmcnt(1)`: every DMA piece (older) has landed, the youngest X load stays in
        // flight.  With `vmcnt(0)` each chunk waited out a full memory latency (5.2 ms -> see DESIGN).
        u16x8 x0 = next_x();
        u16x8 x1 = next_x();
        u16x8 x2 = next_x();
        u16x8 x3 = next_x();
        bool x_in_flight = nc > 3;  // was an X load issued after the previous iteration's DMA pieces?
        for (int c = 0; c < nc; ++c) {
            if (x_in_flight)
                asm volatile("s_waitcnt vmcnt(1)" ::: "memory");
            else
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // tail: the youngest op IS a DMA piece
            __syncthreads();
            const float* wb = lds + (c & 1) * BUF + lane * 4;
            WStage st = begin_stage_bytes(a.W[0], row_bytes, (size_t)(c + 1) * 64, lds + ((c + 1) & 1) * BUF, wave, lane);
            mma_chunk_b<NT1, NT1 * 16>(acc1, wb, x0, st, c + 1 < nc ? n_pieces : 0);
            x0 = x1;
            x1 = x2;
            x2 = x3;
            x_in_flight = cl < nc;  // next_x() below really loads (wave-uniform)
            x3 = next_x();
        }
    }
    __builtin_amdgcn_s_setprio(0);
    layernorm_act_b<NT1, ACT_H>(acc1, a.lnw[0], a.lnb[0], a.act[0], a.eps, g);
    u16x8 h1[NT1 / 2];
    pack_all<NT1>(acc1, h1);

    // ---------------- layer 2 (and 3)
    f32x4 acc2[NT2];
    init_bias_b<NT2>(acc2, a.b[1], g);
    __builtin_amdgcn_s_setprio(2);
    dense_from_regs_b<NT1 / 2, NT2>(h1, acc2, a.W[1], lds, wave, lane);
    __builtin_amdgcn_s_setprio(0);
    layernorm_act_b<NT2, (NT3 == 0 ? ACT_O : ACT_H)>(acc2, a.lnw[1], a.lnb[1], a.act[1], a.eps, g);
    if constexpr (NT3 == 0) {
        store_out_b<NT2>(acc2, a, e, valid, g);
    } else {
        u16x8 h2[NT2 / 2];
        pack_all<NT2>(acc2, h2);
        f32x4 acc3[NT3];
        init_bias_b<NT3>(acc3, a.b[2], g);
        __builtin_amdgcn_s_setprio(2);
        dense_from_regs_b<NT2 / 2, NT3>(h2, acc3, a.W[2], lds, wave, lane);
        __builtin_amdgcn_s_setprio(0);
        layernorm_act_b<NT3, ACT_O>(acc3, a.lnw[2], a.lnb[2], a.act[2], a.eps, g);
        store_out_b<NT3>(acc3, a, e, valid, g);
    }
}

template <int NT1, int NT2, int NT3, int MINW, int ACT_H, int ACT_O>
static int launch_b_act(const MlpArgsBf16& a, hipStream_t s) {
    constexpr int maxnt = NT1 > NT2 ? (NT1 > NT3 ? NT1 : NT3) : (NT2 > NT3 ? NT2 : NT3);
    const size_t lds_bytes = (size_t)2 * maxnt * 256 * sizeof(float);
    const unsigned grid = (unsigned)ceil_div(a.M, 64);
    auto kern = k_fused_mlp_bf16<NT1, NT2, NT3, MINW, ACT_H, ACT_O>;
    if (lds_bytes > 64 * 1024) {
        HGNN_CHECK_HIP(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize,
                                           (int)lds_bytes));
    }
    kern<<<grid, 256, lds_bytes, s>>>(a);
    HGNN_CHECK_HIP(hipGetLastError());
    return HGNN_OK;
}

template <int NT1, int NT2, int NT3, int MINW>
static int launch_b(const MlpArgsBf16& a, hipStream_t s) {
    const int n = NT3 == 0 ? 2 : 3;
    bool hidden_gelu = true;
    for (int l = 0; l + 1 < n; ++l) hidden_gelu = hidden_gelu && a.act[l] == HGNN_ACT_GELU;
    const int out = a.act[n - 1];
    if (hidden_gelu && out == HGNN_ACT_TANH) return launch_b_act<NT1, NT2, NT3, MINW, HGNN_ACT_GELU, HGNN_ACT_TANH>(a, s);
    if (hidden_gelu && out == HGNN_ACT_GELU) return launch_b_act<NT1, NT2, NT3, MINW, HGNN_ACT_GELU, HGNN_ACT_GELU>(a, s);
    return launch_b_act<NT1, NT2, NT3, MINW, -1, -1>(a, s);
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
    if (d->save_pre[0] || d->save_pre[1] || d->save_pre[2]) return 0;
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
