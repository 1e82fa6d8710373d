// Fused backward of ONE layer boundary of the bf16 training path (hand-written data gradient):
//
//   EPI_LN  :  da   = dz . W                       [M, K] x [K, N]   (W = the Linear's weight [K out, N in])
//              dz'  = dLayerNorm(act'(LN(z')) * da)                   z' = the previous layer's pre-LayerNorm rows
//              a'   = act(LN(z'))                                    (what the weight gradient of W needs)
//              + per-workgroup column partials of dgamma' / dbeta'
//   EPI_SKIP:  dx   = dz . W (+ skip)                                first layer: gradient of a direct input segment,
//                                                                    the skip connection's gradient added in place
//
// i.e. the data-gradient GEMM of Modules/utils.py:169-196's Linear fused with the LayerNorm / activation backward of
// the layer below it (unfused: one library GEMM writing da, one row pass reading z', da and writing dz', one more
// recomputing a': 13 B/element of HBM traffic instead of 7).  The decomposition is the forward feature-split
// kernel's (mlp_split_bf16.hip): a workgroup owns 64 rows, wave w a quarter of the N output features; the rows of dz
// reach the MFMAs through LDS panels, W^T streams from L2 in A-fragment order through the register ring
// (gemm_lds); the epilogue works in the accumulator layout (lane = row, registers = features):
//   * z' is loaded in that layout as raw bf16 (2 registers per 4 values) and converted each time it is needed,
//     so that da (128 registers) and z' (64) fit beside each other;
//   * row statistics / the two LayerNorm-backward row sums cross the 4 waves through LDS, as in the forward;
//   * dgamma' / dbeta' are reduced over a tile's 16 row lanes by shuffles and accumulated in LDS by the ONE lane
//     that owns the column (deterministic), written once per workgroup at the end (persistent workgroups);
//     the bias gradient (column sums of dz') comes out of hgnn_wgrad_bf16 (colsum) instead.
#include "mlp_split_common.h"

namespace hgnn {
extern int g_opt_mlp_ablate;
namespace bw {
using namespace fs;

enum { EPI_LN = 0, EPI_SKIP = 1 };

struct Args {
    const unsigned short* dz;      // [M, K] bf16
    int K;
    const unsigned short* Wt;      // bf16, A-fragment order of the [N][K] matrix W^T (see hgnn_mlp_forward_bf16_split)
    const unsigned short* z_prev;  // [M, N] bf16 (EPI_LN)
    const float* lnw;
    const float* lnb;
    int act;
    float eps;
    const unsigned short* skip;    // [M, N] bf16 or NULL (EPI_SKIP)
    unsigned short* out;           // [M, N] bf16: dz' (EPI_LN) / dx (EPI_SKIP)
    unsigned short* a_prev;        // [M, N] bf16 or NULL (EPI_LN)
    float* partials;               // [gridDim.x][2][N] (EPI_LN)
    long long M;
};

__device__ __forceinline__ f32x4 cvt4(u16x4 v) {
    f32x4 o;
    o.x = bf16_float(v[0]);
    o.y = bf16_float(v[1]);
    o.z = bf16_float(v[2]);
    o.w = bf16_float(v[3]);
    return o;
}
__device__ __forceinline__ u16x4 pack4(f32x4 v) {
    u16x4 o;
    o[0] = bf16_bits(v.x);
    o[1] = bf16_bits(v.y);
    o[2] = bf16_bits(v.z);
    o[3] = bf16_bits(v.w);
    return o;
}
// activation value and derivative from ONE evaluation of the transcendental part (act_apply + act_grad would
// evaluate the erf polynomial / exp twice)
__device__ __forceinline__ void act_val_grad(float y, int act, float& val, float& grad) {
    switch (act) {
        case HGNN_ACT_GELU: {
            // Phi(y) = 0.5 (1 + erf(y / sqrt 2)) by Abramowitz-Stegun 7.1.26 (as fast_erf), whose exp(-(y/sqrt 2)^2)
            // IS exp(-y^2 / 2), the Gaussian of the derivative's phi(y): one exp for both
            const float ax = __builtin_fabsf(y) * 0.70710678118654752440f;
            const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, ax, 1.0f));
            float p = fmaf(1.061405429f, t, -1.453152027f);
            p = fmaf(p, t, 1.421413741f);
            p = fmaf(p, t, -0.284496736f);
            p = fmaf(p, t, 0.254829592f);
            p *= t;
            const float e = __builtin_amdgcn_exp2f(-1.44269504088896341f * ax * ax);
            const float erf_abs = fmaf(-p, e, 1.0f);
            const float cdf = fmaf(0.5f, __builtin_copysignf(erf_abs, y), 0.5f);
            val = y * cdf;
            grad = fmaf(y, 0.3989422804014327f * e, cdf);
            return;
        }
        case HGNN_ACT_TANH: {
            const float t = fast_tanh(y);
            val = t;
            grad = fmaf(-t, t, 1.0f);
            return;
        }
        case HGNN_ACT_RELU:
            val = y > 0.f ? y : 0.f;
            grad = y > 0.f ? 1.0f : 0.f;
            return;
        default:
            val = y;
            grad = 1.0f;
    }
}

__device__ __forceinline__ float lanes16_sum(float v) {  // over the 16 row lanes (ei) of a k-group
    v += __shfl_xor(v, 1);
    v += __shfl_xor(v, 2);
    v += __shfl_xor(v, 4);
    v += __shfl_xor(v, 8);
    return v;
}

// KEEPZ: the raw z' tile stays in registers between the three epilogue phases (2 registers per 4 values); without
// it (N = 512 on 4 waves: 128 accumulators per lane) each phase reloads its pieces (L2 hits), which keeps the
// kernel at 2 workgroups per CU
template <int NT, int EPI, int VAR, int MINB, int NW = 4, int NJ = 4, bool KEEPZ = true>
__global__ __launch_bounds__(NW * 64, MINB) void k_mlp_bwd_layer(const Args a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int TE = 16 * NJ;
    constexpr int NTHR = NW * 64;
    constexpr int N = NT * NW * 16;
    constexpr int PK = 128;
    constexpr int PRS = PK * 2 + 16;
    constexpr int PANEL = TE * PRS;
    constexpr int CPP = PK / 32;
    constexpr float inv_n = 1.0f / (float)N;
    float* red = (float*)(smem + 2 * PANEL);              // [NW][TE][2]
    float* colacc = red + NW * TE * 2;                     // [2][N]  (EPI_LN)
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int ei = lane & 15;
    const int g = lane >> 4;
    const int fcol = wave * NT * 16 + 4 * g;               // this lane's first feature of tile 0
    if (EPI == EPI_LN) {
        for (int i = tid; i < 2 * N; i += NTHR) colacc[i] = 0.f;
    }
    __syncthreads();

    constexpr int LPR = PK * 2 / 16;
    constexpr int RPP = NTHR / LPR;
    constexpr int NP = TE / RPP;
    const int prow = tid / LPR;
    const int pcol = tid % LPR;
    const int np = a.K / PK;
    const int total = a.K / 32;
    const long long n_tiles = (a.M + TE - 1) / TE;

    for (long long tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {   // uniform trip count per workgroup
        const long long e0 = tile * TE;
        const unsigned short* px[NP];
#pragma unroll
        for (int i = 0; i < NP; ++i) {
            long long e = e0 + i * RPP + prow;
            if (e >= a.M) e = a.M - 1;
            px[i] = a.dz + (size_t)e * (size_t)a.K + pcol * 8;
        }
        u16x8 st[NP];
        auto load_panel = [&]() {
#pragma unroll
            for (int i = 0; i < NP; ++i) {
                st[i] = *(const u16x8*)px[i];
                px[i] += PK;
            }
        };
        auto store_panel = [&](int buf) {
#pragma unroll
            for (int i = 0; i < NP; ++i) *(u16x8*)(smem + buf * PANEL + (i * RPP + prow) * PRS + pcol * 16) = st[i];
        };

        f32x4 acc[NT][NJ];
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int j = 0; j < NJ; ++j) acc[t][j] = f32x4{0.f, 0.f, 0.f, 0.f};
        {
            const u16x8* wp = (const u16x8*)a.Wt + (size_t)(wave * NT) * 64 + lane;
            u16x8 w[Ring<NT>::R];
            load_panel();
            ring_fill<NT, NW>(w, wp, total);
            store_panel(0);
            __syncthreads();
            const char* blane = smem + ei * PRS + g * 16;
            for (int p = 0; p < np; ++p) {
                const bool more = p + 1 < np;
                if (more) load_panel();
                gemm_lds<NT, PRS, VAR, NW, NJ>(acc, w, wp, p * CPP, total, blane + (p & 1) * PANEL, 0, CPP, 0);
                if (more) store_panel((p + 1) & 1);
                __syncthreads();
            }
        }

        bool valid[NJ];
        size_t off[NJ];
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            const long long e = e0 + j * 16 + ei;
            valid[j] = e < a.M;
            off[j] = (size_t)(valid[j] ? e : a.M - 1) * N + fcol;
        }

        if constexpr (EPI == EPI_SKIP) {
#pragma unroll
            for (int j = 0; j < NJ; ++j) {
                if (!valid[j]) continue;
#pragma unroll
                for (int t = 0; t < NT; ++t) {
                    f32x4 v = acc[t][j];
                    if (a.skip != nullptr) v += cvt4(*(const u16x4*)(a.skip + off[j] + t * 16));
                    *(u16x4*)(a.out + off[j] + t * 16) = pack4(v);
                }
            }
        } else {
            // ---- E1: the previous layer's pre-LayerNorm rows in the accumulator layout, row statistics
            u16x4 zr[KEEPZ ? NT : 1][KEEPZ ? NJ : 1];
            auto zload = [&](int t, int j) -> u16x4 { return *(const u16x4*)(a.z_prev + off[j] + t * 16); };
            if constexpr (KEEPZ) {
#pragma unroll
                for (int j = 0; j < NJ; ++j)
#pragma unroll
                    for (int t = 0; t < NT; ++t) zr[t][j] = zload(t, j);
            }
#pragma unroll
            for (int j = 0; j < NJ; ++j) {
                float s = 0.f, q = 0.f;
#pragma unroll
                for (int t = 0; t < NT; ++t) {
                    const f32x4 x = cvt4(KEEPZ ? zr[KEEPZ ? t : 0][KEEPZ ? j : 0] : zload(t, j));
                    s += (x.x + x.y) + (x.z + x.w);
                    q = fmaf(x.x, x.x, q);
                    q = fmaf(x.y, x.y, q);
                    q = fmaf(x.z, x.z, q);
                    q = fmaf(x.w, x.w, q);
                }
                s += __shfl_xor(s, 16);
                q += __shfl_xor(q, 16);
                s += __shfl_xor(s, 32);
                q += __shfl_xor(q, 32);
                if (g == 0) {
                    f32x2 sq;
                    sq.x = s;
                    sq.y = q;
                    *(f32x2*)(red + (wave * TE + j * 16 + ei) * 2) = sq;
                }
                // reload variant: keep hipcc from hoisting the NEXT row tile's loads up here (it would hold all
                // 32 pieces = the 64 registers this variant exists to avoid)
                if (!KEEPZ) asm volatile("" ::: "memory");
            }
            __syncthreads();
            float mean[NJ], rstd[NJ];
#pragma unroll
            for (int j = 0; j < NJ; ++j) {
                float s = 0.f, q = 0.f;
#pragma unroll
                for (int w = 0; w < NW; ++w) {
                    const f32x2 sq = *(const f32x2*)(red + (w * TE + j * 16 + ei) * 2);
                    s += sq.x;
                    q += sq.y;
                }
                mean[j] = s * inv_n;
                const float var = fmaxf(fmaf(-mean[j], mean[j], q * inv_n), 0.f);
                rstd[j] = 1.0f / sqrtf(var + a.eps);
            }
            __syncthreads();  // everyone has read the statistics: `red` is free for the second exchange
            // ---- E2: activation', gamma; column partials; row sums of the LayerNorm backward
            float sg[NJ], sgx[NJ];
#pragma unroll
            for (int j = 0; j < NJ; ++j) sg[j] = sgx[j] = 0.f;
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                const f32x4 lw = *(const f32x4*)(a.lnw + fcol + t * 16);
                const f32x4 lb = *(const f32x4*)(a.lnb + fcol + t * 16);
                f32x4 dgs = f32x4{0.f, 0.f, 0.f, 0.f}, dbs = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int j = 0; j < NJ; ++j) {
                    const f32x4 x = cvt4(KEEPZ ? zr[KEEPZ ? t : 0][KEEPZ ? j : 0] : zload(t, j));
                    f32x4 av, gg;
#pragma unroll
                    for (int c = 0; c < 4; ++c) {
                        const float xh = (x[c] - mean[j]) * rstd[j];
                        const float y = fmaf(xh, lw[c], lb[c]);
                        float ac, gr;
                        act_val_grad(y, a.act, ac, gr);
                        const float dy = valid[j] ? acc[t][j][c] * gr : 0.f;
                        av[c] = ac;
                        dgs[c] = fmaf(dy, xh, dgs[c]);
                        dbs[c] += dy;
                        const float gv = dy * lw[c];
                        gg[c] = gv;
                        sg[j] += gv;
                        sgx[j] = fmaf(gv, xh, sgx[j]);
                    }
                    acc[t][j] = gg;
                    if (a.a_prev != nullptr && valid[j]) *(u16x4*)(a.a_prev + off[j] + t * 16) = pack4(av);
                }
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    dgs[c] = lanes16_sum(dgs[c]);
                    dbs[c] = lanes16_sum(dbs[c]);
                }
                if (ei == 0) {  // the one owner of these 4 columns in this workgroup
                    float* cg = colacc + fcol + t * 16;
                    *(f32x4*)cg = *(const f32x4*)cg + dgs;
                    *(f32x4*)(cg + N) = *(const f32x4*)(cg + N) + dbs;
                }
                if (!KEEPZ) asm volatile("" ::: "memory");
            }
#pragma unroll
            for (int j = 0; j < NJ; ++j) {
                float s = sg[j], q = sgx[j];
                s += __shfl_xor(s, 16);
                q += __shfl_xor(q, 16);
                s += __shfl_xor(s, 32);
                q += __shfl_xor(q, 32);
                if (g == 0) {
                    f32x2 sq;
                    sq.x = s;
                    sq.y = q;
                    *(f32x2*)(red + (wave * TE + j * 16 + ei) * 2) = sq;
                }
            }
            __syncthreads();
            // ---- E3: dz' = rstd * (g - mean(g) - xhat * mean(g * xhat))
#pragma unroll
            for (int j = 0; j < NJ; ++j) {
                float s = 0.f, q = 0.f;
#pragma unroll
                for (int w = 0; w < NW; ++w) {
                    const f32x2 sq = *(const f32x2*)(red + (w * TE + j * 16 + ei) * 2);
                    s += sq.x;
                    q += sq.y;
                }
                const float mg = s * inv_n, mgx = q * inv_n;
                if (!valid[j]) continue;
#pragma unroll
                for (int t = 0; t < NT; ++t) {
                    const f32x4 x = cvt4(KEEPZ ? zr[KEEPZ ? t : 0][KEEPZ ? j : 0] : zload(t, j));
                    f32x4 o;
#pragma unroll
                    for (int c = 0; c < 4; ++c) {
                        const float xh = (x[c] - mean[j]) * rstd[j];
                        o[c] = rstd[j] * (acc[t][j][c] - mg - xh * mgx);
                    }
                    *(u16x4*)(a.out + off[j] + t * 16) = pack4(o);
                }
                if (!KEEPZ) asm volatile("" ::: "memory");
            }
            __syncthreads();  // `red` is rewritten by the next tile
        }
    }
    if (EPI == EPI_LN) {
        __syncthreads();
        for (int i = tid; i < 2 * N; i += NTHR) a.partials[(size_t)blockIdx.x * 2 * N + i] = colacc[i];
    }
}

constexpr int kBwdGrid = HGNN_MLP_BWD_BLOCKS;

template <int NT, int EPI, int VAR, int MINB, int NW = 4, bool KEEPZ = true>
static int launch(const Args& a, hipStream_t s) {
    constexpr int NJ = 4, TE = 64, N = NT * NW * 16;
    const size_t lds = (size_t)2 * TE * (128 * 2 + 16) + NW * TE * 2 * sizeof(float) + (EPI == EPI_LN ? 2 * N * sizeof(float) : 0);
    const long long n_tiles = (a.M + TE - 1) / TE;
    // EPI_LN: a fixed grid (the partials are [HGNN_MLP_BWD_BLOCKS][2][N]; idle workgroups write zeros)
    const unsigned grid = EPI == EPI_LN ? (unsigned)kBwdGrid : (unsigned)(n_tiles < kBwdGrid ? n_tiles : kBwdGrid);
    k_mlp_bwd_layer<NT, EPI, VAR, MINB, NW, NJ, KEEPZ><<<grid, NW * 64, lds, s>>>(a);
    HGNN_CHECK_HIP(hipGetLastError());
    return HGNN_OK;
}

template <int EPI>
static int dispatch(const Args& a, int N, hipStream_t s) {
    switch (N) {
        case 128: return launch<2, EPI, 0, 2>(a, s);
        case 256: return launch<4, EPI, 0, 2>(a, s);
        // N = 512: 8 waves share the 64 rows (an eighth of the features each, one workgroup per CU): with 4 waves the
        // LayerNorm form needs the 128 accumulators AND the 64 registers of raw z' per lane and spills 171 registers
        case 512:
            // 8 waves share the 64 rows (an eighth of the features each, z' kept in registers, one workgroup per CU).
            // (round-2 null result, removed: 4 waves, 2 workgroups per CU, z' reloaded per phase -- the LayerNorm form
            // still spills 125 registers and measured 6.2 vs 3.2 ms at K = 256, M = 2M)
            return launch<4, EPI, 0, 1, 8>(a, s);
    }
    set_error("hgnn_mlp_backward_layer_bf16: N = %d has no instantiation (128, 256, 512)", N);
    return HGNN_ERR_UNSUPPORTED;
}

}  // namespace bw
}  // namespace hgnn

using namespace hgnn;

extern "C" int hgnn_mlp_backward_layer_supported_bf16(int32_t K, int32_t N) {
    return (K > 0 && K % 128 == 0 && (N == 128 || N == 256 || N == 512)) ? 1 : 0;
}

extern "C" int hgnn_mlp_backward_layer_bf16(const void* dz, int64_t M, int32_t K, int32_t N, const void* Wt_frag,
                                            const void* z_prev, const float* ln_w, const float* ln_b, int32_t act,
                                            float eps, const void* skip, void* out, void* a_prev, float* partials,
                                            hgnn_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    HGNN_REQUIRE(hgnn_mlp_backward_layer_supported_bf16(K, N), "hgnn_mlp_backward_layer_bf16: unsupported shape K=%d N=%d "
                 "(K a multiple of 128, N in {128, 256, 512})", K, N);
    HGNN_REQUIRE(M >= 0 && M <= 0x7fffffffLL, "hgnn_mlp_backward_layer_bf16: bad M");
    const bool ln = z_prev != nullptr;
    HGNN_REQUIRE(!ln || (ln_w != nullptr && ln_b != nullptr && partials != nullptr && skip == nullptr),
                 "hgnn_mlp_backward_layer_bf16: the LayerNorm form needs ln_w, ln_b, partials and no skip");
    HGNN_REQUIRE(ln || (a_prev == nullptr && partials == nullptr), "hgnn_mlp_backward_layer_bf16: a_prev / partials belong to the LayerNorm form");
    HGNN_REQUIRE(act >= HGNN_ACT_NONE && act <= HGNN_ACT_RELU, "hgnn_mlp_backward_layer_bf16: unknown activation code %d", act);
    if (M == 0 && !ln) return HGNN_OK;
    HGNN_REQUIRE(M == 0 || (dz != nullptr && Wt_frag != nullptr && out != nullptr), "hgnn_mlp_backward_layer_bf16: NULL pointer");
    HGNN_REQUIRE((uintptr_t)dz % 16 == 0 && (uintptr_t)Wt_frag % 16 == 0 && (uintptr_t)z_prev % 8 == 0 &&
                     (uintptr_t)skip % 8 == 0 && (uintptr_t)out % 8 == 0 && (uintptr_t)a_prev % 8 == 0 &&
                     (uintptr_t)ln_w % 16 == 0 && (uintptr_t)ln_b % 16 == 0 && (uintptr_t)partials % 16 == 0,
                 "hgnn_mlp_backward_layer_bf16: misaligned pointer");
    bw::Args a;
    a.dz = (const unsigned short*)dz;
    a.K = K;
    a.Wt = (const unsigned short*)Wt_frag;
    a.z_prev = (const unsigned short*)z_prev;
    a.lnw = ln_w;
    a.lnb = ln_b;
    a.act = act;
    a.eps = eps;
    a.skip = (const unsigned short*)skip;
    a.out = (unsigned short*)out;
    a.a_prev = (unsigned short*)a_prev;
    a.partials = partials;
    a.M = M;
    return ln ? bw::dispatch<bw::EPI_LN>(a, N, stream) : bw::dispatch<bw::EPI_SKIP>(a, N, stream);
}
