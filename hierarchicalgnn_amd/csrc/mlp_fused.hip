// Fused  gather -> concat -> [Linear -> LayerNorm -> act] x {2,3} -> (+skip)  on fp32 MFMA.
//
// Replaces, for one make_mlp Sequential (reference Modules/utils.py:169-196) and the
// concat/gather feeding it (Modules/gnn_utils.py:52-53, :61-62, :126, :134, :144, :152):
//     cat([nodes[g0], nodes[g1], edges]) -> Linear -> LN -> GELU -> Linear -> LN -> Tanh -> + edges
// without ever writing the [M,3L] concat, the gathered copies or the [M,H] hidden
// activations to HBM (>= 16 GB of avoidable traffic per cell at L=256, M=2M).
//
// Mapping (wave64, v_mfma_f32_16x16x4_f32, exact fp32 = fmaf chain):
//   * the GEMMs are evaluated TRANSPOSED:  D[f][e] = sum_k W[f][k] * X[e][k]
//     A operand = weight fragment, B operand = activations.  The D layout then
//     puts the edge on the lane (col = lane&15) and the features in registers
//     (row = 4*(lane>>4) + reg), so
//       - LayerNorm statistics over features are in-register sums plus two
//         cross-lane adds (no LDS, no cross-wave traffic);
//       - the activated accumulator tile of layer i IS the B operand of layer
//         i+1 (regs r=0..3 of tile c hold k = 16c + 4*(lane>>4) + r, exactly the
//         four k-steps of that tile) -- hidden activations never leave registers.
//   * one wave owns 16 edges and ALL features of every layer; a workgroup is 4
//     waves = 64 edges.  Weights stream through LDS in k-chunks of 16
//     (A-fragment order, filled by LDS-DMA global_load_lds_dwordx4, double
//     buffered, one barrier per chunk); the activations X are read straight from
//     global memory into the B-operand layout (16 rows x 64 B per wave load),
//     with the row gather folded into the per-lane address.
//   * 2 workgroups per CU (<=256 VGPRs); the 3-layer L=256 node network needs 320
//     accumulator registers and runs at 1 workgroup per CU.  NOTE: the fp32 MFMA
//     executes on the SIMD's fp32 vector ALUs, so epilogue VALU work does NOT hide
//     under a co-resident wave's MFMAs (measured): the epilogue is kept
//     short (branch-free erf/tanh) rather than overlapped.
//   * small-K mode (encoders, K = 3 / 6) and width-1 plain last layers (heads) are
//     handled by zero padding (see hgnn_mlp_desc in include/hgnn_hip.h); `save_pre`
//     dumps the pre-LayerNorm outputs for the opt-in differentiable variant.
#include "mlp_common.h"

namespace hgnn {

int g_opt_mlp_ablate = 0;   // set through hgnn_set_option("mlp_ablate", bits): DIAGNOSTIC, wrong results

struct MlpArgs {
    const float* seg_table[3];
    const int32_t* seg_index[3];
    int seg_width[3];
    int n_seg;
    int K1;       // columns stored in W[0] (multiple of 16)
    int K1_real;  // concatenated input width; < K1 only in small-K mode (K1_real <= 16, W[0] zero padded)
    int n_out_real;  // real output columns; < (tiles of the last layer)*16 only for PARTIAL kernels (heads with a
                     // plain 1- or emb_dim-wide last layer, supernode encoder L-8): W/b/ln of the last layer are
                     // zero padded to the tile width, statistics and stores use the real width
    const float* W[3];
    const float* b[3];
    const float* lnw[3];
    const float* lnb[3];
    int act[3];
    float eps;
    const float* skip;
    float* out;
    float* save_pre[3];  // optional [M, width_l] dumps of each layer's pre-LayerNorm output (training)
    const float* pre_table[2];    // pre-projected gathered segments [rows, NT1*16] (see hgnn_mlp_desc.n_pre)
    const int32_t* pre_index[2];
    int n_pre;
    long long M;
    int ablate;        // DIAGNOSTIC ONLY (wrong results): 1 = skip LN/act, 2 = skip weight DMA, 4 = skip barriers
};

template <int NF, int NW>
__device__ __forceinline__ void stage_w(const float* __restrict__ W, int Kdim, int k0, float* lds,
                                        int wave, int lane) {
    WStage st = begin_stage<NW>(W, Kdim, k0, lds, wave, lane);
    const int n = pieces_of<NF, NW>(wave);
    for (int i = 0; i < n; ++i) stage_next<NW>(st);
}

// acc[T][r] (edge = lane&15, feature = 16T + 4*(lane>>4) + r): LayerNorm over features, then act
// PARTIAL: only n_real < NT*16 features are real; the padded ones have zero weights and bias, so their
// accumulators are exactly 0: they add nothing to the sum and mean^2 each to the squared deviations
template <int NT, int ACT, bool LN = true, bool PARTIAL = false>
__device__ __forceinline__ void layernorm_act(f32x4 (&acc)[NT], const float* __restrict__ lnw,
                                              const float* __restrict__ lnb, int act, float eps, int g,
                                              int n_real = NT * 16) {
    if (!LN) return;  // plain last layer of a head: bias only
    const float inv_n = PARTIAL ? 1.0f / (float)n_real : 1.0f / (float)(NT * 16);
    float s = 0.f;
#pragma unroll
    for (int T = 0; T < NT; ++T) s += (acc[T].x + acc[T].y) + (acc[T].z + acc[T].w);
    s += __shfl_xor(s, 16);
    s += __shfl_xor(s, 32);
    const float mean = s * inv_n;
    float q = 0.f;
#pragma unroll
    for (int T = 0; T < NT; ++T) {
        f32x4 d = acc[T] - mean;
        q += (d.x * d.x + d.y * d.y) + (d.z * d.z + d.w * d.w);
    }
    q += __shfl_xor(q, 16);
    q += __shfl_xor(q, 32);
    if (PARTIAL) q -= (float)(NT * 16 - n_real) * mean * mean;
    const float rstd = 1.0f / sqrtf(q * inv_n + eps);
#pragma unroll
    for (int T = 0; T < NT; ++T) {
        const f32x4 w4 = *(const f32x4*)(lnw + T * 16 + g * 4);
        const f32x4 b4 = *(const f32x4*)(lnb + T * 16 + g * 4);
        f32x4 v = (acc[T] - mean) * rstd * w4 + b4;
        const int code = ACT >= 0 ? ACT : act;  // compile-time for the common networks
        v.x = act_apply(v.x, code);
        v.y = act_apply(v.y, code);
        v.z = act_apply(v.z, code);
        v.w = act_apply(v.w, code);
        acc[T] = v;
    }
}

template <int NT>
__device__ __forceinline__ void init_bias(f32x4 (&acc)[NT], const float* __restrict__ b, int g) {
#pragma unroll
    for (int T = 0; T < NT; ++T) acc[T] = *(const f32x4*)(b + T * 16 + g * 4);
}

// One k-chunk (16 k-values) of one layer:  acc[T] += Wchunk[T] * b  for every 16-feature tile T.
// Tiles are processed in pairs (two independent accumulators cover the 40-cycle dependent-issue
// latency of v_mfma_f32_16x16x4_f32) and the NEXT pair's weight fragments are fetched from LDS
// before the current pair's 8 MFMAs, pinned with sched_group_barrier so that hipcc does not
// sink the reads back to their first use.  The LDS-DMA pieces that stage the NEXT chunk's weights
// are issued one at a time BETWEEN MFMA groups (each costs the issuing wave ~60 cycles of VMEM
// issue; in a burst ahead of the loop they were 10 % of the kernel, in the MFMA shadow they hide).
template <int NT, int NF_NEXT, int NW>
__device__ __forceinline__ void mma_chunk(f32x4 (&acc)[NT], const float* __restrict__ wb, const f32x4 b,
                                          WStage& st, int n_pieces) {
    static_assert(NT % 2 == 0, "tiles are processed in pairs");
    constexpr int PAIRS = NT / 2;
    constexpr int PER_WAVE = (NF_NEXT / 16 + NW - 1) / NW;
    constexpr int EVERY = PAIRS >= PER_WAVE ? PAIRS / PER_WAVE : 1;
    f32x4 w0 = *(const f32x4*)(wb);
    f32x4 w1 = *(const f32x4*)(wb + 256);
    __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);  // pipeline prologue: the first pair's reads
    int issued = 0;
#pragma unroll
    for (int T = 0; T < NT; T += 2) {
        f32x4 n0 = w0, n1 = w1;
        if (T + 2 < NT) {
            n0 = *(const f32x4*)(wb + (T + 2) * 256);
            n1 = *(const f32x4*)(wb + (T + 3) * 256);
        }
        if (((T / 2) % EVERY == 0) && issued < PER_WAVE) {
            if (__builtin_amdgcn_readfirstlane(issued < n_pieces ? 1 : 0)) stage_next<NW>(st);  // scalar branch
            ++issued;
        }
        acc[T] = __builtin_amdgcn_mfma_f32_16x16x4f32(w0.x, b.x, acc[T], 0, 0, 0);
        acc[T + 1] = __builtin_amdgcn_mfma_f32_16x16x4f32(w1.x, b.x, acc[T + 1], 0, 0, 0);
        acc[T] = __builtin_amdgcn_mfma_f32_16x16x4f32(w0.y, b.y, acc[T], 0, 0, 0);
        acc[T + 1] = __builtin_amdgcn_mfma_f32_16x16x4f32(w1.y, b.y, acc[T + 1], 0, 0, 0);
        acc[T] = __builtin_amdgcn_mfma_f32_16x16x4f32(w0.z, b.z, acc[T], 0, 0, 0);
        acc[T + 1] = __builtin_amdgcn_mfma_f32_16x16x4f32(w1.z, b.z, acc[T + 1], 0, 0, 0);
        acc[T] = __builtin_amdgcn_mfma_f32_16x16x4f32(w0.w, b.w, acc[T], 0, 0, 0);
        acc[T + 1] = __builtin_amdgcn_mfma_f32_16x16x4f32(w1.w, b.w, acc[T + 1], 0, 0, 0);
        w0 = n0;
        w1 = n1;
        if (T + 2 < NT) __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);  // 2 DS reads (next pair)
        __builtin_amdgcn_sched_group_barrier(0x008, 8, 0);                  // 8 MFMAs (this pair)
    }
    for (int i = issued; i < n_pieces; ++i) stage_next<NW>(st);  // more pieces than tile pairs
}

// one register-resident layer: out[NTO tiles] = W[NTO*16][NTI*16] * in  (in = previous accumulators)
template <int NTI, int NTO, int NW>
__device__ __forceinline__ void dense_from_regs(const f32x4 (&in)[NTI], f32x4 (&out)[NTO],
                                                const float* __restrict__ W, float* lds, int wave, int lane,
                                                int ablate) {
    constexpr int KD = NTI * 16;
    constexpr int BUF = NTO * 256;  // floats per chunk buffer
    const int n_pieces = pieces_of<NTO * 16, NW>(wave);
    __syncthreads();                // everyone is done with both buffers of the previous layer
    stage_w<NTO * 16, NW>(W, KD, 0, lds, wave, lane);
#pragma unroll
    for (int c = 0; c < NTI; ++c) {
        wait_dma();                           // this wave's pieces of chunk c have landed
        if (!(ablate & 4)) __syncthreads();   // everyone's have; buffer (c+1)&1 is free again
        const float* wb = lds + (c & 1) * BUF + lane * 4;
        WStage st = begin_stage<NW>(W, KD, (c + 1) * 16, lds + ((c + 1) & 1) * BUF, wave, lane);
        mma_chunk<NTO, NTO * 16, NW>(out, wb, in[c], st, (c + 1 < NTI && !(ablate & 2)) ? n_pieces : 0);
    }
}

template <int NT, bool PARTIAL = false>
__device__ __forceinline__ void store_out(const f32x4 (&acc)[NT], const MlpArgs& a, long long e, bool valid,
                                          int g) {
    if (!valid) return;
    if constexpr (PARTIAL) {  // out[e][0 .. n_out_real): the columns past the real width are padding
        const int nr = a.n_out_real;
        float* op = a.out + (size_t)e * (size_t)nr;
        if ((nr & 3) == 0) {
#pragma unroll
            for (int T = 0; T < NT; ++T) {
                const int col = T * 16 + g * 4;
                if (col < nr) *(f32x4*)(op + col) = acc[T];
            }
        } else {
#pragma unroll
            for (int T = 0; T < NT; ++T) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int col = T * 16 + g * 4 + r;
                    if (col < nr) op[col] = acc[T][r];
                }
            }
        }
        return;
    }
    constexpr int NOUT = NT * 16;
    float* op = a.out + (size_t)e * NOUT + g * 4;
    if (a.skip != nullptr) {
        const float* sp = a.skip + (size_t)e * NOUT + g * 4;
#pragma unroll
        for (int T = 0; T < NT; ++T) *(f32x4*)(op + T * 16) = acc[T] + *(const f32x4*)(sp + T * 16);
    } else {
#pragma unroll
        for (int T = 0; T < NT; ++T) *(f32x4*)(op + T * 16) = acc[T];
    }
}

// training: dump a layer's pre-LayerNorm activations z[e][f] (what the hand-written backward in
// fused.py needs; the hidden activations themselves are recomputed from it, never stored)
template <int NT>
__device__ __forceinline__ void dump_pre(const f32x4 (&acc)[NT], float* base, long long e, bool valid, int g) {
    if (base == nullptr || !valid) return;  // base is wave-uniform
    float* op = base + (size_t)e * (NT * 16) + g * 4;
#pragma unroll
    for (int T = 0; T < NT; ++T) *(f32x4*)(op + T * 16) = acc[T];
}

// NT1/NT2/NT3: 16-feature tiles of layer 1 / 2 / 3 outputs (NT3 == 0: two-layer MLP)
// ACT_H / ACT_O: activation of the hidden layers / of the last layer (HGNN_ACT_*), or -1 = read
// it from the descriptor per element (keeps rare combinations working without an instantiation)
// PLAIN_LAST: the last layer has no LayerNorm / activation (classifier heads, width-1 output)
// PARTIAL: the last layer's real width is a.n_out_real < its tile width (see MlpArgs)
template <int NT1, int NT2, int NT3, int MINW, int ACT_H, int ACT_O, bool PLAIN_LAST = false, int NW = 4,
          bool PARTIAL = false>
__global__ __launch_bounds__(NW * 64, MINW) void k_fused_mlp(const MlpArgs a) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int ei = lane & 15;
    const int g = lane >> 4;
    const long long e = (long long)blockIdx.x * (NW * 16) + wave * 16 + ei;
    const bool valid = e < a.M;
    const long long er = valid ? e : 0;

    // per-lane row start of every input segment (the gather is folded in here).  The X stream is
    // read through ONE running pointer that hops to the next segment's row at a segment boundary,
    // so nothing is runtime-indexed (cdna_hip_programming.md 5.4 rule 20).
    const float* q0;
    const float* q1;
    const float* q2;
    {
        long long r = a.seg_index[0] != nullptr ? (long long)a.seg_index[0][er] : er;
        q0 = a.seg_table[0] + (size_t)(r < 0 ? 0 : r) * (size_t)a.seg_width[0] + g * 4;
        q1 = q0;
        q2 = q0;
        if (a.n_seg > 1) {
            r = a.seg_index[1] != nullptr ? (long long)a.seg_index[1][er] : er;
            q1 = a.seg_table[1] + (size_t)(r < 0 ? 0 : r) * (size_t)a.seg_width[1] + g * 4;
        }
        if (a.n_seg > 2) {
            r = a.seg_index[2] != nullptr ? (long long)a.seg_index[2][er] : er;
            q2 = a.seg_table[2] + (size_t)(r < 0 ? 0 : r) * (size_t)a.seg_width[2] + g * 4;
        }
    }
    const int nc = a.K1 / 16;
    const int c1 = a.seg_width[0] / 16;                       // first chunk of segment 1
    const int c2 = c1 + (a.n_seg > 1 ? a.seg_width[1] / 16 : nc);  // first chunk of segment 2
    const float* px = q0;
    int cl = 0;  // next chunk the X stream will load
    // small-K mode (encoders: K = 3 or 6 spatial coordinates): one zero-padded chunk, assembled
    // from scalar loads because a 12-byte row is not 16-byte addressable
    const bool smallk = a.K1_real < a.K1;  // wave-uniform
    auto small_x = [&]() -> f32x4 {
        const float* r0 = q0 - g * 4;
        const float* r1 = q1 - g * 4;
        const float* r2 = q2 - g * 4;
        const int w0 = a.seg_width[0], w1 = a.n_seg > 1 ? a.seg_width[1] : 0, w2 = a.n_seg > 2 ? a.seg_width[2] : 0;
        f32x4 v;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int k = g * 4 + r;
            float t = 0.f;
            if (k < w0) t = r0[k];
            else if (k < w0 + w1) t = r1[k - w0];
            else if (k < w0 + w1 + w2) t = r2[k - w0 - w1];
            v[r] = t;
        }
        return v;
    };
    auto next_x = [&](f32x4 fallback) -> f32x4 {
        f32x4 v = fallback;
        if (cl < nc && !smallk) v = *(const f32x4*)px;
        ++cl;
        px += 16;
        if (cl == c1) px = q1;
        if (cl == c2) px = q2;
        return v;
    };

    // ---------------- layer 1: K1 (runtime) -> NT1*16, weights through LDS, X from global
    f32x4 acc1[NT1];
    init_bias<NT1>(acc1, a.b[0], g);
    // pre-projected gathered segments: the row's accumulators start at b + sum_s P_s[idx_s[e]]
    // (this lane's 4 features of every 16-feature tile: 64-byte pieces of the 16 P rows per load)
    if (a.n_pre > 0) {
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            if (s < a.n_pre) {
                const long long r = (long long)a.pre_index[s][er];
                const float* p = a.pre_table[s] + (size_t)(r < 0 ? 0 : r) * (size_t)(NT1 * 16) + g * 4;
#pragma unroll
                for (int T = 0; T < NT1; ++T) {
                    const f32x4 v = *(const f32x4*)(p + T * 16);
                    acc1[T].x += v.x;
                    acc1[T].y += v.y;
                    acc1[T].z += v.z;
                    acc1[T].w += v.w;
                }
            }
        }
    }
    __builtin_amdgcn_s_setprio(2);
    {
        constexpr int BUF = NT1 * 256;
        const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
        stage_w<NT1 * 16, NW>(a.W[0], a.K1, 0, lds, wave, lane);
        const int n_pieces = pieces_of<NT1 * 16, NW>(wave);
        f32x4 x0 = smallk ? small_x() : next_x(zero);
        f32x4 x1 = next_x(zero);
        for (int c = 0; c < nc; ++c) {
            wait_dma();
            if (!(a.ablate & 4)) __syncthreads();
            const f32x4 x2 = next_x(zero);
            const float* wb = lds + (c & 1) * BUF + lane * 4;
            WStage st = begin_stage<NW>(a.W[0], a.K1, (c + 1) * 16, lds + ((c + 1) & 1) * BUF, wave, lane);
            mma_chunk<NT1, NT1 * 16, NW>(acc1, wb, x0, st, (c + 1 < nc && !(a.ablate & 2)) ? n_pieces : 0);
            x0 = x1;
            x1 = x2;
        }
    }
    // Priority outranks age in the SIMD's issue arbitration: a wave in its (dense, wait-free) VALU
    // epilogue would otherwise starve the co-resident workgroup's MFMA stream whenever it is the
    // older of the two.  Epilogues run at priority 0, MFMA loops at priority 2.
    __builtin_amdgcn_s_setprio(0);
    dump_pre<NT1>(acc1, a.save_pre[0], e, valid, g);
    if constexpr (NT2 == 0) {
        // single-layer launch (fp32 at latent 512: a 1024-wide hidden layer is 256 accumulators per lane, so the
        // layers of an MLP run as separate launches and the hidden rows make one trip through HBM)
        if (!(a.ablate & 1)) layernorm_act<NT1, ACT_O>(acc1, a.lnw[0], a.lnb[0], a.act[0], a.eps, g);
        store_out<NT1>(acc1, a, e, valid, g);
        return;
    }
    if (!(a.ablate & 1)) layernorm_act<NT1, ACT_H>(acc1, a.lnw[0], a.lnb[0], a.act[0], a.eps, g);
    __builtin_amdgcn_s_setprio(2);

    // ---------------- layer 2 (and 3): activations stay in registers
    f32x4 acc2[NT2 > 0 ? NT2 : 2];
    constexpr int N2 = NT2 > 0 ? NT2 : 2;   // (NT2 == 0 returned above; N2 only keeps the dead code well-formed)
    init_bias<N2>(acc2, a.b[1], g);
    dense_from_regs<NT1, N2, NW>(acc1, acc2, a.W[1], lds, wave, lane, a.ablate);
    __builtin_amdgcn_s_setprio(0);
    dump_pre<N2>(acc2, a.save_pre[1], e, valid, g);
    if (!(a.ablate & 1))
        layernorm_act<N2, (NT3 == 0 ? ACT_O : ACT_H), !(PLAIN_LAST && NT3 == 0), (PARTIAL && NT3 == 0)>(
            acc2, a.lnw[1], a.lnb[1], a.act[1], a.eps, g, NT3 == 0 ? a.n_out_real : N2 * 16);
    if constexpr (NT3 == 0) {
        store_out<N2, PARTIAL>(acc2, a, e, valid, g);
    } else {
        f32x4 acc3[NT3];
        init_bias<NT3>(acc3, a.b[2], g);
        __builtin_amdgcn_s_setprio(2);
        dense_from_regs<N2, NT3, NW>(acc2, acc3, a.W[2], lds, wave, lane, a.ablate);
        __builtin_amdgcn_s_setprio(0);
        dump_pre<NT3>(acc3, a.save_pre[2], e, valid, g);
        if (!(a.ablate & 1))
            layernorm_act<NT3, ACT_O, !PLAIN_LAST, PARTIAL>(acc3, a.lnw[2], a.lnb[2], a.act[2], a.eps, g, a.n_out_real);
        store_out<NT3, PARTIAL>(acc3, a, e, valid, g);
    }
}

template <int NT1, int NT2, int NT3, int MINW, int ACT_H, int ACT_O, bool PLAIN_LAST = false, int NW = 4,
          bool PARTIAL = false>
static int launch_mlp_act(const MlpArgs& a, hipStream_t s) {
    constexpr int maxnt = NT1 > NT2 ? (NT1 > NT3 ? NT1 : NT3) : (NT2 > NT3 ? NT2 : NT3);
    const size_t lds_bytes = (size_t)2 * maxnt * 256 * sizeof(float);
    const unsigned grid = (unsigned)ceil_div(a.M, NW * 16);
    auto kern = k_fused_mlp<NT1, NT2, NT3, MINW, ACT_H, ACT_O, PLAIN_LAST, NW, PARTIAL>;
    if (lds_bytes > 64 * 1024) {
        HGNN_CHECK_HIP(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize,
                                           (int)lds_bytes));
    }
    kern<<<grid, NW * 64, lds_bytes, s>>>(a);
    HGNN_CHECK_HIP(hipGetLastError());
    return HGNN_OK;
}

// encoders whose output is narrower than their last tile row (supernode encoder: L - emb_dim,
// BipartiteClassification/Models/HGNN_GMM.py:117): LayerNorm + activation over the real width
template <int NT1, int NT2, int NT3, int MINW>
static int launch_mlp_partial(const MlpArgs& a, hipStream_t s) {
    const int n = NT3 == 0 ? 2 : 3;
    bool gelu = true;
    for (int l = 0; l < n; ++l) gelu = gelu && a.act[l] == HGNN_ACT_GELU;
    if (gelu) return launch_mlp_act<NT1, NT2, NT3, MINW, HGNN_ACT_GELU, HGNN_ACT_GELU, false, 4, true>(a, s);
    return launch_mlp_act<NT1, NT2, NT3, MINW, -1, -1, false, 4, true>(a, s);
}

template <int NT1, int NT2, int NT3, int MINW>
static int launch_mlp(const MlpArgs& a, hipStream_t s) {
    const int n = NT3 == 0 ? 2 : 3;
    bool hidden_gelu = true;
    for (int l = 0; l + 1 < n; ++l) hidden_gelu = hidden_gelu && a.act[l] == HGNN_ACT_GELU;
    const int out = a.act[n - 1];
    // (round-2 null result, removed: 8 waves = 128 rows per workgroup, one workgroup per CU -- bitwise equal,
    // 3.6 % slower: one phase-locked workgroup does not overlap its epilogue with another's MFMAs)
    if (hidden_gelu && out == HGNN_ACT_TANH) return launch_mlp_act<NT1, NT2, NT3, MINW, HGNN_ACT_GELU, HGNN_ACT_TANH>(a, s);
    if (hidden_gelu && out == HGNN_ACT_GELU) return launch_mlp_act<NT1, NT2, NT3, MINW, HGNN_ACT_GELU, HGNN_ACT_GELU>(a, s);
    return launch_mlp_act<NT1, NT2, NT3, MINW, -1, -1>(a, s);
}

// one Linear -> LayerNorm -> act (+ skip) layer: the pieces of an fp32 MLP at latent 512
template <int NT1, int MINW>
static int launch_single(const MlpArgs& a, hipStream_t s) {
    switch (a.act[0]) {
        case HGNN_ACT_GELU: return launch_mlp_act<NT1, 0, 0, MINW, HGNN_ACT_GELU, HGNN_ACT_GELU>(a, s);
        case HGNN_ACT_TANH: return launch_mlp_act<NT1, 0, 0, MINW, HGNN_ACT_TANH, HGNN_ACT_TANH>(a, s);
    }
    return launch_mlp_act<NT1, 0, 0, MINW, -1, -1>(a, s);
}

// heads: K -> H -> H -> w with a PLAIN last layer of w <= 32 real outputs (padded to 32 rows): the width-1
// classifiers (IN.py:107-115, HGNN_GMM.py:313-321) and the emb_dim-wide embedding head (HGNN_GMM.py:74-82);
// LayerNorm + act on the two hidden layers only
template <int NTH, int MINW>
static int launch_head(const MlpArgs& a, hipStream_t s) {
    if (a.act[0] == HGNN_ACT_GELU && a.act[1] == HGNN_ACT_GELU)
        return launch_mlp_act<NTH, NTH, 2, MINW, HGNN_ACT_GELU, HGNN_ACT_NONE, true, 4, true>(a, s);
    if (a.act[0] == HGNN_ACT_TANH && a.act[1] == HGNN_ACT_TANH)
        return launch_mlp_act<NTH, NTH, 2, MINW, HGNN_ACT_TANH, HGNN_ACT_NONE, true, 4, true>(a, s);
    return launch_mlp_act<NTH, NTH, 2, MINW, -1, HGNN_ACT_NONE, true, 4, true>(a, s);
}

// plain (no LayerNorm / activation) last layer of a 3-layer network = a head
static bool is_head(const hgnn_mlp_desc* d) { return d->n_layers == 3 && d->ln_w[2] == nullptr; }
// LayerNorm'ed last layer narrower than its zero-padded storage (supernode encoder)
static bool is_partial(const hgnn_mlp_desc* d) {
    return !is_head(d) && d->w_last_rows != 0 && d->w_last_rows != d->width[d->n_layers];
}

}  // namespace hgnn

using namespace hgnn;

extern "C" int hgnn_mlp_supported(const hgnn_mlp_desc* d) {
    if (d == nullptr) return 0;
    if (d->n_seg < 1 || d->n_seg > 3 || d->n_layers < 1 || d->n_layers > 3) return 0;
    int k = 0;
    bool aligned16 = true;
    for (int s = 0; s < d->n_seg; ++s) {
        if (d->seg_width[s] <= 0) return 0;
        aligned16 = aligned16 && d->seg_width[s] % 16 == 0;
        k += d->seg_width[s];
    }
    if (k != d->width[0]) return 0;
    if (!aligned16) {  // small-K mode: one zero-padded 16-column chunk
        if (k > 16 || d->w0_cols != 16) return 0;
    } else if (d->w0_cols != 0 && d->w0_cols != k) {
        return 0;
    }
    const int n = d->n_layers;
    for (int l = 0; l < n; ++l)
        if (d->W[l] == nullptr || d->b[l] == nullptr) return 0;
    if (d->n_pre < 0 || d->n_pre > 2) return 0;
    for (int s = 0; s < d->n_pre; ++s)
        if (d->pre_table[s] == nullptr || d->pre_index[s] == nullptr) return 0;
    const int h = d->width[1];
    const int o = d->width[n];
    if (n == 1) {
        // a single Linear -> LayerNorm -> act (+ skip) layer, 512 or 1024 wide: the building block of the fp32
        // MLPs at latent 512 (hidden 1024), which do not fit one launch
        if (d->ln_w[0] == nullptr || d->ln_b[0] == nullptr || d->w_last_rows != 0 || d->save_pre[0] != nullptr) return 0;
        return (o == 512 || o == 1024) ? 1 : 0;
    }
    if (n == 3 && d->width[2] != h) return 0;
    if (is_head(d)) {
        // K -> H -> H -> w (w <= 32): LayerNorm on the hidden layers only, plain last layer stored as 32 rows
        if (o < 1 || o > 32 || d->w_last_rows != 32 || d->act[2] != HGNN_ACT_NONE) return 0;
        if (d->ln_w[0] == nullptr || d->ln_b[0] == nullptr || d->ln_w[1] == nullptr || d->ln_b[1] == nullptr) return 0;
        if (d->skip != nullptr) return 0;
        return (h == 64 || h == 128 || h == 256 || h == 512) ? 1 : 0;
    }
    for (int l = 0; l < n; ++l)
        if (d->ln_w[l] == nullptr || d->ln_b[l] == nullptr) return 0;
    if (is_partial(d)) {
        // K -> 2P (-> 2P) -> o with o < P = w_last_rows real outputs, zero padded to P rows (W, b, ln_w, ln_b)
        const int P = d->w_last_rows;
        if (o < 4 || o >= P || (o & 3) != 0 || h != 2 * P || d->skip != nullptr) return 0;
        for (int l = 0; l < n; ++l)
            if (d->save_pre[l] != nullptr) return 0;
        return (n == 3 && (P == 32 || P == 64 || P == 128 || P == 256)) ? 1 : 0;
    }
    if (h != 2 * o) return 0;
    return (o == 32 || o == 64 || o == 128 || o == 256) ? 1 : 0;
}

extern "C" int hgnn_mlp_forward_f32(const hgnn_mlp_desc* d, float* out, hgnn_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    HGNN_REQUIRE(d != nullptr && out != nullptr, "hgnn_mlp_forward_f32: NULL argument");
    if (!hgnn_mlp_supported(d)) {
        set_error("hgnn_mlp_forward_f32: unsupported shape (see hgnn_mlp_supported in include/hgnn_hip.h: "
                  "K -> 2L (-> 2L) -> L with LayerNorm everywhere, L in {32,64,128,256}, segments multiples "
                  "of 16 or K <= 16 zero-padded; or a K -> H -> H -> 1 head)");
        return HGNN_ERR_UNSUPPORTED;
    }
    if (d->M == 0) return HGNN_OK;
    HGNN_REQUIRE(d->M > 0 && d->M < ((int64_t)1 << 31) * 64, "hgnn_mlp_forward_f32: bad M");
    MlpArgs a;
    for (int s = 0; s < 3; ++s) {
        a.seg_table[s] = s < d->n_seg ? d->seg_table[s] : nullptr;
        a.seg_index[s] = s < d->n_seg ? d->seg_index[s] : nullptr;
        a.seg_width[s] = s < d->n_seg ? d->seg_width[s] : 0;
        if (s < d->n_seg) {
            HGNN_REQUIRE(a.seg_table[s] != nullptr && (uintptr_t)a.seg_table[s] % 16 == 0,
                         "hgnn_mlp_forward_f32: segment table %d is NULL or not 16-byte aligned", s);
        }
    }
    a.n_seg = d->n_seg;
    a.K1_real = d->width[0];
    a.K1 = d->w0_cols != 0 ? d->w0_cols : d->width[0];
    a.n_out_real = d->width[d->n_layers];
    for (int l = 0; l < 3; ++l) {
        const bool on = l < d->n_layers;
        a.W[l] = on ? d->W[l] : nullptr;
        a.b[l] = on ? d->b[l] : nullptr;
        a.lnw[l] = on ? d->ln_w[l] : nullptr;
        a.lnb[l] = on ? d->ln_b[l] : nullptr;
        a.act[l] = on ? d->act[l] : 0;
        if (on) {
            HGNN_REQUIRE((uintptr_t)a.W[l] % 16 == 0 && (uintptr_t)a.b[l] % 16 == 0 &&
                             (uintptr_t)a.lnw[l] % 16 == 0 && (uintptr_t)a.lnb[l] % 16 == 0,
                         "hgnn_mlp_forward_f32: layer %d parameters must be 16-byte aligned", l);
        }
        if (on && a.lnw[l] == nullptr) a.lnw[l] = a.lnb[l] = a.b[l];  // never dereferenced (plain layer)
    }
    a.ablate = g_opt_mlp_ablate;
    for (int l = 0; l < 3; ++l) {
        a.save_pre[l] = l < d->n_layers ? d->save_pre[l] : nullptr;
        HGNN_REQUIRE((uintptr_t)a.save_pre[l] % 16 == 0, "hgnn_mlp_forward_f32: save_pre[%d] must be 16-byte aligned", l);
    }
    // heads: the two LayerNorm'ed hidden layers can dump (training forward of the score heads); the plain last layer's
    // "pre-LayerNorm output" is the result itself
    HGNN_REQUIRE(!(is_head(d) && d->save_pre[2]), "hgnn_mlp_forward_f32: save_pre[2] is not available for heads");
    a.eps = d->ln_eps;
    a.skip = d->skip;
    a.out = out;
    a.n_pre = d->n_pre;
    for (int s = 0; s < 2; ++s) {
        a.pre_table[s] = s < d->n_pre ? d->pre_table[s] : nullptr;
        a.pre_index[s] = s < d->n_pre ? d->pre_index[s] : nullptr;
        HGNN_REQUIRE((uintptr_t)a.pre_table[s] % 16 == 0, "hgnn_mlp_forward_f32: pre_table[%d] must be 16-byte aligned", s);
    }
    a.M = d->M;
    HGNN_REQUIRE((uintptr_t)out % 16 == 0 && (uintptr_t)a.skip % 16 == 0,
                 "hgnn_mlp_forward_f32: out/skip must be 16-byte aligned");
    if (is_head(d)) {
        switch (d->width[1]) {
            case 64: return launch_head<4, 2>(a, stream);
            case 128: return launch_head<8, 2>(a, stream);
            case 256: return launch_head<16, 2>(a, stream);
            case 512: return launch_head<32, 1>(a, stream);
        }
    }
    if (is_partial(d)) {
        switch (d->w_last_rows) {
            case 32: return launch_mlp_partial<4, 4, 2, 2>(a, stream);
            case 64: return launch_mlp_partial<8, 8, 4, 2>(a, stream);
            case 128: return launch_mlp_partial<16, 16, 8, 2>(a, stream);
            case 256: return launch_mlp_partial<32, 32, 16, 1>(a, stream);
        }
    }
    const int o = d->width[d->n_layers];
    if (d->n_layers == 1) {
        switch (o) {
            case 512: return launch_single<32, 2>(a, stream);
            case 1024: return launch_single<64, 1>(a, stream);
        }
    }
    if (d->n_layers == 2) {
        switch (o) {
            case 32: return launch_mlp<4, 2, 0, 2>(a, stream);
            case 64: return launch_mlp<8, 4, 0, 2>(a, stream);
            case 128: return launch_mlp<16, 8, 0, 2>(a, stream);
            case 256: return launch_mlp<32, 16, 0, 2>(a, stream);
        }
    } else {
        switch (o) {
            case 32: return launch_mlp<4, 4, 2, 2>(a, stream);
            case 64: return launch_mlp<8, 8, 4, 2>(a, stream);
            case 128: return launch_mlp<16, 16, 8, 2>(a, stream);
            case 256: return launch_mlp<32, 32, 16, 1>(a, stream);
        }
    }
    set_error("hgnn_mlp_forward_f32: no instantiation");
    return HGNN_ERR_UNSUPPORTED;
}
