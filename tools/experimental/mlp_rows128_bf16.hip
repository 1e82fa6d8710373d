// bf16 fused  gather -> concat -> [Linear -> LayerNorm -> act] x {2,3} -> (+skip)  MLP at latent 256
// (K -> 512 (-> 512) -> 256), 128 ROWS PER WEIGHT FETCH.
//
// Why a second wide-layer kernel.  mlp_split_bf16.hip gives a workgroup 64 rows and streams each wave's weight
// slice from L2 into registers: 1 MiB of weights per 64 rows = 32.7 GB of L2 -> CU traffic per 2M-row launch.  An
// XCD's L2 delivers ~2 TB/s (16 channels x 64 B/clk; 16.8-18.8 TB/s chip-wide measured for L2-resident row gathers,
// MI355X_MICROARCH.md "Indexed rows"), and the bare MFMA + weight-load loop of that kernel runs at 15.8 TB/s of
// weight reads (round-2 ablation 31: 2.07 of its 2.75 ms): it is L2-BANDWIDTH bound, at any ring depth and with
// conflict-free LDS.  The only lever is rows per weight byte, and that is capped by accumulator space: 128 rows x 512
// hidden features of fp32 = half of a CU's register file.  So here ONE workgroup of 8 waves owns 128 rows per CU:
//   * wave w owns 1/8 of every layer's features for ALL 128 rows: 64 features x 128 rows in the 512-wide layers
//     (2 x 4 tiles of v_mfma_f32_32x32x16_bf16 = 128 accumulator registers), 32 features x 128 rows in the 256-wide
//     output layer; one 1-KiB weight fragment feeds 4 MFMAs of 32 cycles -- half the weight bytes per FLOP from L2
//     (16.4 GB per launch), a quarter of the vector-memory instructions per MFMA cycle;
//   * the 32x32x16 MFMA holds the SIMD's vector issue for 8 of its 32 cycles (8 of 16 for the 16x16x32 form), so
//     address arithmetic, LDS reads and the ring refills of both waves of a SIMD fit between MFMAs;
//   * weights are private to a wave (no LDS, no DMA, no per-chunk barrier): host stores W in the 32x32x16
//     A-fragment order (fragment (k16-step s, 32-feature tile T) = 1 KiB, lane l = W[32T + l%32][16s + 8(l/32) ..+7]),
//     streamed through a register ring D steps ahead;
//   * activations cross waves through LDS: gathered input rows in 128-wide k-panels (double-buffered, aliasing
//     the hidden rows), hidden rows as bf16 [128][512]; 16-byte pieces XOR-swizzled with the row so that every
//     16-lane group of a ds_read_b128 touches 16 distinct slots;
//   * persistent workgroups (grid = CUs): the next tile's indices / first panel are requested before the last
//     epilogue, the next layer's weight ring before each epilogue.
// The price: one workgroup per CU, all 8 waves phase-locked by the per-panel barriers, so the LayerNorm /
// activation epilogues (vector ALU only) no longer hide under another workgroup's MFMAs.
#include "mlp_split_common.h"

namespace hgnn {
extern int g_opt_mlp_ablate;
int g_opt_mlp_rows128 = 0;  // hgnn_set_option("mlp_rows128"): 1 = the latent-256 bf16 MLPs run here, 0 (default) = feature-split kernel (A/B: equal time)
namespace fr {

using fs::as_bf16;
using fs::act_t;
using fs::bf16_bits;
using fs::bf16_float;
using fs::cmax;
using fs::f32x2;
using fs::u16x4;
using fs::u16x8;
typedef float f32x16 __attribute__((ext_vector_type(16)));

struct Args {
    const unsigned short* seg_table[3];
    const int32_t* seg_index[3];
    int seg_width[3];
    int n_seg;
    int K1;
    const unsigned short* W[3];  // bf16, 32x32x16 A-fragment order
    const float* b[3];
    const float* lnw[3];
    const float* lnb[3];
    int act[3];
    float eps;
    const unsigned short* skip;
    unsigned short* out;
    long long M;
    unsigned short* save_pre[3];
    int ablate;  // DIAGNOSTIC (wrong results): 1 = weights from step 0 only, 2 = no LayerNorm / activation arithmetic,
                 // 4 = input panels after the first are not loaded (stale registers stored), 8 = no per-panel barriers,
                 // 16 = no static priority for waves 4-7
};

// Diagnostic build only (tools/rows128_stamps.hip): shader-clock stamps at the phase boundaries of every tile
#ifdef HGNN_ROWS128_STAMPS
__device__ unsigned long long g_stamps[256 * 64 * 8 * 8];   // [workgroup][tile][stamp][wave]
#define HGNN_STAMP(k)                                                                                      \
    do {                                                                                                   \
        if ((threadIdx.x & 63) == 0 && it < 64)                                                            \
            g_stamps[((blockIdx.x * 64 + it) * 8 + (k)) * 8 + (threadIdx.x >> 6)] = __builtin_amdgcn_s_memtime(); \
    } while (0)
#else
#define HGNN_STAMP(k)
#endif

constexpr int NW = 8;     // waves per workgroup
constexpr int NR = 4;     // 32-row tiles per workgroup
constexpr int TE = 128;   // rows per workgroup
constexpr int PK = 128;   // k-panel width of the input rows
constexpr int PRS = PK * 2;
constexpr int PANEL = TE * PRS;
constexpr int SPP = PK / 16;  // k16-steps per panel

template <int NF>
__device__ __forceinline__ void init_bias(f32x16 (&acc)[NF][NR], const float* __restrict__ b) {
#pragma unroll
    for (int f = 0; f < NF; ++f) {
        f32x16 v;
#pragma unroll
        for (int bb = 0; bb < 4; ++bb) {
            const f32x4 t = *(const f32x4*)(b + f * 32 + bb * 8);
            v[bb * 4 + 0] = t.x;
            v[bb * 4 + 1] = t.y;
            v[bb * 4 + 2] = t.z;
            v[bb * 4 + 3] = t.w;
        }
#pragma unroll
        for (int j = 0; j < NR; ++j) acc[f][j] = v;
    }
}

template <int NF, int D>
__device__ __forceinline__ void ring_fill(u16x8 (&w)[D * NF], const u16x8* __restrict__ wp, int wstride, int total) {
#pragma unroll
    for (int d = 0; d < D; ++d) {
        const int s = d < total ? d : total - 1;
        const u16x8* p = wp + (size_t)s * wstride;
#pragma unroll
        for (int f = 0; f < NF; ++f) w[d * NF + f] = p[f * 64];
    }
}

// acc[f][j] += W(steps gs0 .. gs0+n) * B, B = n k16-steps read from LDS rows of stride RS bytes.  `brow` = this
// lane's row (l%32) base + its k-half piece ((h ^ (x & 1)) << 4); step s of a row sits at 256-byte block s >> 3,
// 32-byte slot ((s & 7) ^ kx) (XOR swizzle, kx = (l & 15) >> 1).  Ring slot (step % D) * NF + f is refilled with
// step + D right after its last MFMA.  n % D == 0, D even.
template <int NF, int D, int RS>
__device__ __forceinline__ void gemm_lds(f32x16 (&acc)[NF][NR], u16x8 (&w)[D * NF], const u16x8* __restrict__ wp,
                                         int wstride, int gs0, int total, const char* brow, int kx, int n, int ablate) {
    u16x8 b[2][NR];
    {
        const char* p = brow + ((0 ^ kx) << 5);
#pragma unroll
        for (int j = 0; j < NR; ++j) b[0][j] = *(const u16x8*)(p + j * 32 * RS);
        // its own group: with a compile-time n the loop is straight-line code and the FIRST group of the loop
        // would otherwise swallow these reads (every step then read its own operands: no prefetch)
        __builtin_amdgcn_sched_group_barrier(0x100, NR, 0);
    }
    for (int c = 0; c < n; c += D) {
#pragma unroll
        for (int u = 0; u < D; ++u) {
            const int s = c + u;
            const int sn = s + 1 < n ? s + 1 : n - 1;
            const char* p = brow + ((sn >> 3) << 8) + (((sn & 7) ^ kx) << 5);
            int gn = gs0 + s + D < total ? gs0 + s + D : total - 1;
            if (ablate & 1) gn = 0;
            const u16x8* wn = wp + (size_t)gn * wstride;
#pragma unroll
            for (int f = 0; f < NF; ++f) {
                const int slot = u * NF + f;
#pragma unroll
                for (int j = 0; j < NR; ++j) {
                    acc[f][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_bf16(w[slot]), as_bf16(b[u & 1][j]),
                                                                        acc[f][j], 0, 0, 0);
                    // one LDS read of the NEXT step's operand behind each MFMA of the first tile row (a block of
                    // reads and address arithmetic at the head of the step drains the matrix pipe: stamps showed a
                    // wave alone issuing MFMAs only 55-60 % of its time)
                    if (f == 0) {
                        b[(u + 1) & 1][j] = *(const u16x8*)(p + j * 32 * RS);
                        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                    }
                }
                w[slot] = wn[f * 64];
                if (f != 0) __builtin_amdgcn_sched_group_barrier(0x008, NR, 0);
                __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
            }
        }
    }
}

// LayerNorm statistics over ALL features of the layer: this wave's partial (sum, sum of squares) per row -> LDS
template <int NF>
__device__ __forceinline__ void ln_partials(const f32x16 (&acc)[NF][NR], float* red, int wave, int n, int h) {
#pragma unroll
    for (int j = 0; j < NR; ++j) {
        float s = 0.f, q = 0.f;
#pragma unroll
        for (int f = 0; f < NF; ++f) {
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const float v = acc[f][j][i];
                s += v;
                q = fmaf(v, v, q);
            }
        }
        s += __shfl_xor(s, 32);
        q += __shfl_xor(q, 32);
        if (h == 0) {
            f32x2 sq;
            sq.x = s;
            sq.y = q;
            *(f32x2*)(red + (wave * TE + j * 32 + n) * 2) = sq;
        }
    }
}

__device__ __forceinline__ void ln_finish(const float* red, int n, float inv_n, float eps, float (&rstd)[NR],
                                          float (&shift)[NR]) {
#pragma unroll
    for (int j = 0; j < NR; ++j) {
        float s = 0.f, q = 0.f;
#pragma unroll
        for (int w = 0; w < NW; ++w) {
            const f32x2 sq = *(const f32x2*)(red + (w * TE + j * 32 + n) * 2);
            s += sq.x;
            q += sq.y;
        }
        const float mean = s * inv_n;
        const float var = fmaxf(fmaf(-mean, mean, q * inv_n), 0.f);
        rstd[j] = 1.0f / sqrtf(var + eps);
        shift[j] = -mean * rstd[j];
    }
}

// LayerNorm + activation of this wave's slice, written as bf16 hidden rows [128][HRS bytes] (swizzled pieces).
// lnw / lnb point at this lane's first feature (32 * first tile + 4h).
template <int NF, int ACT, int HRS>
__device__ __forceinline__ void apply_write_hidden(const f32x16 (&acc)[NF][NR], const float* __restrict__ lnw,
                                                   const float* __restrict__ lnb, int act, const float (&rstd)[NR],
                                                   const float (&shift)[NR], char* smem, int tile0, int n, int h, int x,
                                                   int ablate) {
#pragma unroll
    for (int f = 0; f < NF; ++f) {
        const int Tg = tile0 + f;  // wave-uniform
#pragma unroll
        for (int bb = 0; bb < 4; ++bb) {
            const f32x4 w4 = *(const f32x4*)(lnw + f * 32 + bb * 8);
            const f32x4 b4 = *(const f32x4*)(lnb + f * 32 + bb * 8);
            char* dst = smem + n * HRS + (Tg >> 2) * 256 + ((((Tg & 3) * 4 + bb) ^ x) << 4) + (h << 3);
#pragma unroll
            for (int j = 0; j < NR; ++j) {
                float v0 = acc[f][j][bb * 4 + 0], v1 = acc[f][j][bb * 4 + 1], v2 = acc[f][j][bb * 4 + 2],
                      v3 = acc[f][j][bb * 4 + 3];
                if (!(ablate & 2)) {
                    v0 = act_t<ACT>(fmaf(fmaf(v0, rstd[j], shift[j]), w4.x, b4.x), act);
                    v1 = act_t<ACT>(fmaf(fmaf(v1, rstd[j], shift[j]), w4.y, b4.y), act);
                    v2 = act_t<ACT>(fmaf(fmaf(v2, rstd[j], shift[j]), w4.z, b4.z), act);
                    v3 = act_t<ACT>(fmaf(fmaf(v3, rstd[j], shift[j]), w4.w, b4.w), act);
                }
                u16x4 o;
                o[0] = bf16_bits(v0);
                o[1] = bf16_bits(v1);
                o[2] = bf16_bits(v2);
                o[3] = bf16_bits(v3);
                *(u16x4*)(dst + j * 32 * HRS) = o;
            }
        }
    }
}

// LayerNorm + activation (+ skip) of the last layer -> bf16 rows staged in LDS (`stage`: [128][NOUT] bf16, 16-byte
// pieces XOR-swizzled with the row), from where whole rows go out with 16-byte stores.  The accumulator layout
// gives a lane 4 consecutive features of 32 different rows per instruction: as global stores (and skip loads) those
// 8-byte pieces are issue-bound (first build: 29.7k of a tile's 93k cycles in this epilogue).  `stage` already
// holds the skip rows when `has_skip` (added in fp32 before the one rounding).
template <int NF, int ACT, int NOUT>
__device__ __forceinline__ void apply_stage_out(const f32x16 (&acc)[NF][NR], const float* lnw, const float* lnb, int act,
                                                const float (&rstd)[NR], const float (&shift)[NR], char* stage,
                                                bool has_skip, int tile0, int n, int h, int x, int ablate) {
#pragma unroll
    for (int f = 0; f < NF; ++f) {
#pragma unroll
        for (int bb = 0; bb < 4; ++bb) {
            const f32x4 w4 = *(const f32x4*)(lnw + f * 32 + bb * 8);
            const f32x4 b4 = *(const f32x4*)(lnb + f * 32 + bb * 8);
            char* dst = stage + n * (NOUT * 2) + ((((tile0 + f) * 4 + bb) ^ x) << 4) + (h << 3);
#pragma unroll
            for (int j = 0; j < NR; ++j) {
                float v0 = acc[f][j][bb * 4 + 0], v1 = acc[f][j][bb * 4 + 1], v2 = acc[f][j][bb * 4 + 2],
                      v3 = acc[f][j][bb * 4 + 3];
                if (!(ablate & 2)) {
                    v0 = act_t<ACT>(fmaf(fmaf(v0, rstd[j], shift[j]), w4.x, b4.x), act);
                    v1 = act_t<ACT>(fmaf(fmaf(v1, rstd[j], shift[j]), w4.y, b4.y), act);
                    v2 = act_t<ACT>(fmaf(fmaf(v2, rstd[j], shift[j]), w4.z, b4.z), act);
                    v3 = act_t<ACT>(fmaf(fmaf(v3, rstd[j], shift[j]), w4.w, b4.w), act);
                }
                if (has_skip) {
                    const u16x4 sk = *(const u16x4*)(dst + j * 32 * (NOUT * 2));
                    v0 += bf16_float(sk[0]);
                    v1 += bf16_float(sk[1]);
                    v2 += bf16_float(sk[2]);
                    v3 += bf16_float(sk[3]);
                }
                u16x4 o;
                o[0] = bf16_bits(v0);
                o[1] = bf16_bits(v1);
                o[2] = bf16_bits(v2);
                o[3] = bf16_bits(v3);
                *(u16x4*)(dst + j * 32 * (NOUT * 2)) = o;
            }
        }
    }
}

// training: this wave's slice of the layer's pre-LayerNorm rows -> bf16 [M, NOUT]
template <int NF, int NOUT>
__device__ __forceinline__ void dump_pre(const f32x16 (&acc)[NF][NR], unsigned short* base, long long M, long long e0,
                                         int tile0, int n, int h) {
    if (base == nullptr) return;
#pragma unroll
    for (int j = 0; j < NR; ++j) {
        const long long e = e0 + j * 32 + n;
        if (e >= M) continue;
        const size_t off = (size_t)e * NOUT + (size_t)(tile0 * 32 + 4 * h);
#pragma unroll
        for (int f = 0; f < NF; ++f) {
#pragma unroll
            for (int bb = 0; bb < 4; ++bb) {
                u16x4 o;
                o[0] = bf16_bits(acc[f][j][bb * 4 + 0]);
                o[1] = bf16_bits(acc[f][j][bb * 4 + 1]);
                o[2] = bf16_bits(acc[f][j][bb * 4 + 2]);
                o[3] = bf16_bits(acc[f][j][bb * 4 + 3]);
                *(u16x4*)(base + off + f * 32 + bb * 8) = o;
            }
        }
    }
}

// NFH: 32-feature tiles per wave of the hidden layers (hidden width = 256 NFH), NFO: of the output layer;
// NL = 2 or 3 layers; D1 / DO: ring depth (k16-steps) of the hidden / output layers' weight streams
template <int NFH, int NFO, int NL, int ACT_H, int ACT_O, int D1, int DO>
__global__ __launch_bounds__(NW * 64, 1) void k_mlp_bf16_rows128(const Args a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int H = NFH * NW * 32;
    constexpr int O = NFO * NW * 32;
    constexpr int HRS = H * 2;  // hidden row stride (bytes): whole 256-byte bank rows, swizzled
    constexpr int REGION = cmax(TE * HRS, 2 * PANEL);
    float* red = (float*)(smem + REGION);                     // [NW][TE][sum, sumsq]
    int32_t* idx = (int32_t*)(smem + REGION + NW * TE * 8);   // [3][TE] gather rows of the current tile
    // bias / LayerNorm weight / LayerNorm bias of every layer, copied once: [layer][3][width] (an L2 round trip per
    // parameter vector and epilogue otherwise, queued behind whatever the wave has in flight)
    float* par = (float*)(smem + REGION + NW * TE * 8 + 3 * TE * 4);
    constexpr int PAR_O = (NL - 1) * 3 * H;   // the output layer's block
    for (int i = threadIdx.x; i < (NL - 1) * 3 * H + 3 * O; i += NW * 64) {
        const int l = i < PAR_O ? i / (3 * H) : NL - 1;
        const int r = i - l * 3 * H;
        const int wd = l < NL - 1 ? H : O;
        const float* src = r < wd ? a.b[l] : (r < 2 * wd ? a.lnw[l] : a.lnb[l]);
        par[i] = src[r % wd];
    }
    int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // the second-dispatched half loses every issue arbitration against its older SIMD partner (stamps: waves 0-3
    // finish a barrier-free GEMM phase in 6.5k cycles, waves 4-7 in 11.9k): static priority for that half
    if (wave >= NW / 2 && !(a.ablate & 16)) __builtin_amdgcn_s_setprio(1);
    constexpr int LPR = PRS / 16;        // 16 threads per row piece
    // Everything derived from the lane id is tile-invariant, and in a persistent loop the compiler hoists ALL of it
    // (every epilogue store / LDS address, ~145 registers) out of the loop and spills it: the first build ran at
    // 4.2 ms on scratch reloads.  refresh() makes the thread id opaque again, at the top of every tile and before
    // every epilogue, so the address arithmetic is redone where it is used (a few dozen VALU per tile).
    int lane, n, h, x, kx, hx, prow, pcol, po;
    auto refresh = [&]() {
        asm volatile("" : "+v"(tid));
        lane = tid & 63;
        n = lane & 31;
        h = lane >> 5;
        x = lane & 15;
        kx = x >> 1;
        hx = (h ^ (x & 1)) << 4;
        prow = tid / LPR;
        pcol = tid % LPR;
        po = 4 * h;   // this lane's feature offset into the bias / LayerNorm parameter vectors
    };
    refresh();

    // input panels: thread (prow, pcol) moves 16 bytes of rows prow + 32 i per panel
    constexpr int RPP = NW * 64 / LPR;   // 32 rows per pass
    constexpr int NP = TE / RPP;         // 4 passes
    const int np = a.K1 / PK;
    const int p1 = a.seg_width[0] / PK;
    const int p2 = p1 + (a.n_seg > 1 ? a.seg_width[1] / PK : np);
    const long long n_tiles = (a.M + TE - 1) / TE;

    auto fetch_index = [&](long long e0) {   // this thread's entry of the [3][TE] gather-row table of tile e0
        int r = 0;
        if (tid < 3 * TE) {
            const int sgm = tid / TE;
            long long e = e0 + (tid % TE);
            if (e >= a.M) e = a.M - 1;
            r = (int)e;
            if (sgm < a.n_seg && a.seg_index[sgm] != nullptr) r = a.seg_index[sgm][e];
        }
        return r < 0 ? 0 : r;
    };
    auto load_indices = [&](long long e0) {
        const int r = fetch_index(e0);
        if (tid < 3 * TE) idx[tid] = r;
    };
    u16x8 st[NP];
    auto load_panel = [&](int p) {
        const int sgm = p < p1 ? 0 : (p < p2 ? 1 : 2);
        const int pk = p - (sgm == 0 ? 0 : (sgm == 1 ? p1 : p2));
        const unsigned short* tb = a.seg_table[sgm] + pk * PK + pcol * 8;
        const size_t wdt = (size_t)a.seg_width[sgm];
#pragma unroll
        for (int i = 0; i < NP; ++i) {
            const int r = idx[sgm * TE + i * RPP + prow];
            st[i] = *(const u16x8*)(tb + (size_t)r * wdt);
        }
    };
    auto store_panel = [&](int buf) {
#pragma unroll
        for (int i = 0; i < NP; ++i)
            *(u16x8*)(smem + buf * PANEL + (i * RPP + prow) * PRS + ((pcol ^ (prow & 15)) << 4)) = st[i];
    };

    long long tile = blockIdx.x;
    if (tile >= n_tiles) return;
    load_indices(tile * TE);
    __syncthreads();
    load_panel(0);
    for (int it = 0; tile < n_tiles; tile += gridDim.x, ++it) {
        HGNN_STAMP(0);
        const long long e0 = tile * TE;
        const bool has_next = tile + gridDim.x < n_tiles;
        refresh();

        // ---------------- layer 1: B = input panels
        f32x16 acc1[NFH][NR];
        init_bias<NFH>(acc1, par + wave * NFH * 32 + po);
        {
            const int total = a.K1 / 16;
            const int wstride = NW * NFH * 64;
            const u16x8* wp = (const u16x8*)a.W[0] + (size_t)(wave * NFH) * 64 + lane;
            u16x8 w[D1 * NFH];
            ring_fill<NFH, D1>(w, wp, wstride, total);
            store_panel(0);
            __syncthreads();
            const char* brow = smem + n * PRS + hx;
            for (int p = 0; p < np; ++p) {
                const bool more = p + 1 < np;
                if (more && !(a.ablate & 4)) load_panel(p + 1);
                gemm_lds<NFH, D1, PRS>(acc1, w, wp, wstride, p * SPP, total, brow + (p & 1) * PANEL, kx, SPP, a.ablate);
                if (more) store_panel((p + 1) & 1);
                if (!(a.ablate & 8)) __syncthreads();
            }
        }
        HGNN_STAMP(1);
        dump_pre<NFH, H>(acc1, a.save_pre[0], a.M, e0, wave * NFH, n, h);
        float rstd[NR], shift[NR];
        if constexpr (NL == 3) {
            // ---------------- middle layer: B = hidden rows, same width
            const int wstride = NW * NFH * 64;
            const u16x8* wp = (const u16x8*)a.W[1] + (size_t)(wave * NFH) * 64 + lane;
            u16x8 w[D1 * NFH];
            ring_fill<NFH, D1>(w, wp, wstride, H / 16);   // in flight under the epilogue
            refresh();
            ln_partials<NFH>(acc1, red, wave, n, h);
            __syncthreads();   // (also: every wave is done reading the panels)
            ln_finish(red, n, 1.0f / (float)H, a.eps, rstd, shift);
            apply_write_hidden<NFH, ACT_H, HRS>(acc1, par + H + wave * NFH * 32 + po, par + 2 * H + wave * NFH * 32 + po,
                                                a.act[0], rstd, shift, smem, wave * NFH, n, h, x, a.ablate);
            __syncthreads();
            init_bias<NFH>(acc1, par + 3 * H + wave * NFH * 32 + po);
            gemm_lds<NFH, D1, HRS>(acc1, w, wp, wstride, 0, H / 16, smem + n * HRS + hx, kx, H / 16, a.ablate);
            dump_pre<NFH, H>(acc1, a.save_pre[1], a.M, e0, wave * NFH, n, h);
        }
        // output staging: thread (srow, spc) moves 16-byte piece spc of rows srow + 16 i (skip rows in, result rows out)
        constexpr int ORS = O * 2;          // staged output row stride (bytes)
        constexpr int OPR = ORS / 16;       // 16-byte pieces per row (32)
        constexpr int ORP = NW * 64 / OPR;  // rows per pass (16)
        constexpr int ONP = TE / ORP;       // passes (8)
        char* stage = smem + 2 * PANEL;     // clear of panel buffers 0 / 1, which the next tile fills first
        const bool has_skip = a.skip != nullptr;
        u16x8 sk[ONP];
        int srow = tid / OPR, spc = tid % OPR;
        constexpr int LH = NL - 2;  // index of the last hidden layer's parameters
        constexpr int LO = NL - 1;
        f32x16 acc2[NFO][NR];
        {
            const int wstride = NW * NFO * 64;
            const u16x8* wp = (const u16x8*)a.W[LO] + (size_t)(wave * NFO) * 64 + lane;
            u16x8 w[DO * NFO];
            refresh();
            // the next tile's gather rows: requested here, stored to LDS at the end of this epilogue (the input panels
            // of THIS tile are consumed), so that the next tile's first panel can be requested at the start of the
            // last epilogue with nothing to wait for
            const int r_next = has_next ? fetch_index((tile + gridDim.x) * TE) : 0;
            ring_fill<NFO, DO>(w, wp, wstride, H / 16);   // in flight under the epilogue
            ln_partials<NFH>(acc1, red, wave, n, h);
            __syncthreads();   // (also: every wave is done reading the panels / the previous hidden rows)
            ln_finish(red, n, 1.0f / (float)H, a.eps, rstd, shift);
            apply_write_hidden<NFH, ACT_H, HRS>(acc1, par + LH * 3 * H + H + wave * NFH * 32 + po,
                                                par + LH * 3 * H + 2 * H + wave * NFH * 32 + po,
                                                a.act[LH], rstd, shift, smem, wave * NFH, n, h, x, a.ablate);
            if (has_next && tid < 3 * TE) idx[tid] = r_next;
            __syncthreads();
            HGNN_STAMP(2);
            // D: the skip rows (whole rows, 16 bytes per thread and pass) fly under the output GEMM
            if (has_skip) {
#pragma unroll
                for (int i = 0; i < ONP; ++i) {
                    long long e = e0 + i * ORP + srow;
                    if (e >= a.M) e = a.M - 1;
                    sk[i] = *(const u16x8*)(a.skip + (size_t)e * O + spc * 8);
                }
            }
            // ---------------- output layer: B = hidden rows
            init_bias<NFO>(acc2, par + PAR_O + wave * NFO * 32 + po);
            gemm_lds<NFO, DO, HRS>(acc2, w, wp, wstride, 0, H / 16, smem + n * HRS + hx, kx, H / 16, a.ablate);
        }
        HGNN_STAMP(3);
        dump_pre<NFO, O>(acc2, a.save_pre[LO], a.M, e0, wave * NFO, n, h);
        refresh();
        srow = tid / OPR;
        spc = tid % OPR;
        if (has_next) load_panel(0);   // the next tile's first panel flies under the last epilogue
        ln_partials<NFO>(acc2, red, wave, n, h);
        __syncthreads();   // (also: every wave is done reading the hidden rows)
        if (has_skip) {
#pragma unroll
            for (int i = 0; i < ONP; ++i)
                *(u16x8*)(stage + (i * ORP + srow) * ORS + ((spc ^ (srow & 15)) << 4)) = sk[i];
        }
        ln_finish(red, n, 1.0f / (float)O, a.eps, rstd, shift);
        if (has_skip) __syncthreads();
        apply_stage_out<NFO, ACT_O, O>(acc2, par + PAR_O + O + wave * NFO * 32 + po, par + PAR_O + 2 * O + wave * NFO * 32 + po,
                                       a.act[LO], rstd, shift, stage, has_skip, wave * NFO, n, h, x, a.ablate);
        __syncthreads();
#pragma unroll
        for (int i = 0; i < ONP; ++i) {
            const long long e = e0 + i * ORP + srow;
            const u16x8 v = *(const u16x8*)(stage + (i * ORP + srow) * ORS + ((spc ^ (srow & 15)) << 4));
            if (e < a.M) *(u16x8*)(a.out + (size_t)e * O + spc * 8) = v;
        }
        HGNN_STAMP(4);
        // `red`, `stage`: next written after the next tile's first barriers (ln_partials / hidden rows): no hazard
    }
}

static int g_cus = 0;

template <int NL, int ACT_H, int ACT_O>
static int launch_act(const Args& a, hipStream_t s) {
    constexpr int NFH = 2, NFO = 1, D1 = 4, DO = 8;
    constexpr int HRS = NFH * NW * 32 * 2;
    const size_t lds_bytes = (size_t)cmax(TE * HRS, 2 * PANEL) + NW * TE * 8 + 3 * TE * 4 +
                             ((NL - 1) * 3 * NFH * NW * 32 + 3 * NFO * NW * 32) * sizeof(float);
    if (g_cus == 0) {
        int dev = 0;
        HGNN_CHECK_HIP(hipGetDevice(&dev));
        HGNN_CHECK_HIP(hipDeviceGetAttribute(&g_cus, hipDeviceAttributeMultiprocessorCount, dev));
        if (g_cus <= 0) g_cus = 256;
    }
    const long long n_tiles = ceil_div(a.M, TE);
    const unsigned grid = (unsigned)(n_tiles < g_cus ? n_tiles : g_cus);
    auto kern = k_mlp_bf16_rows128<NFH, NFO, NL, ACT_H, ACT_O, D1, DO>;
    HGNN_CHECK_HIP(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes));
    kern<<<grid, NW * 64, lds_bytes, s>>>(a);
    HGNN_CHECK_HIP(hipGetLastError());
    return HGNN_OK;
}

template <int NL>
static int launch(const Args& a, hipStream_t s) {
    bool hidden_gelu = true;
    for (int l = 0; l + 1 < NL; ++l) hidden_gelu = hidden_gelu && a.act[l] == HGNN_ACT_GELU;
    const int out = a.act[NL - 1];
    if (hidden_gelu && out == HGNN_ACT_TANH) return launch_act<NL, HGNN_ACT_GELU, HGNN_ACT_TANH>(a, s);
    if (hidden_gelu && out == HGNN_ACT_GELU) return launch_act<NL, HGNN_ACT_GELU, HGNN_ACT_GELU>(a, s);
    return launch_act<NL, -1, -1>(a, s);
}

}  // namespace fr
}  // namespace hgnn

using namespace hgnn;

extern "C" int hgnn_mlp_supported_bf16_rows128(const hgnn_mlp_desc* d) {
    if (d == nullptr) return 0;
    if (d->n_seg < 1 || d->n_seg > 3 || d->n_layers < 2 || d->n_layers > 3) return 0;
    int k = 0;
    for (int s = 0; s < d->n_seg; ++s) {
        if (d->seg_width[s] <= 0 || d->seg_width[s] % 128 != 0) return 0;
        k += d->seg_width[s];
    }
    if (k != d->width[0] || d->w0_cols != 0 || d->w_last_rows != 0 || d->n_pre != 0) return 0;
    const int n = d->n_layers;
    for (int l = 0; l < n; ++l)
        if (d->W[l] == nullptr || d->b[l] == nullptr || d->ln_w[l] == nullptr || d->ln_b[l] == nullptr) return 0;
    if (d->M < 0 || d->M > 0x7fffffffLL) return 0;
    if (d->width[1] != 512 || d->width[n] != 256) return 0;
    if (n == 3 && d->width[2] != 512) return 0;
    return 1;
}

extern "C" int hgnn_mlp_rows128_enabled(void) { return g_opt_mlp_rows128 != 0; }

extern "C" int hgnn_mlp_forward_bf16_rows128(const hgnn_mlp_desc* d, void* out, hgnn_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    HGNN_REQUIRE(d != nullptr && out != nullptr, "hgnn_mlp_forward_bf16_rows128: NULL argument");
    if (!hgnn_mlp_supported_bf16_rows128(d)) {
        set_error("hgnn_mlp_forward_bf16_rows128: unsupported shape (K -> 512 (-> 512) -> 256, LayerNorm on every "
                  "layer, every segment a multiple of 128 wide, no pre-projected segments)");
        return HGNN_ERR_UNSUPPORTED;
    }
    if (d->M == 0) return HGNN_OK;
    fr::Args a;
    for (int s = 0; s < 3; ++s) {
        const bool on = s < d->n_seg;
        a.seg_table[s] = on ? (const unsigned short*)d->seg_table[s] : (const unsigned short*)d->seg_table[0];
        a.seg_index[s] = on ? d->seg_index[s] : nullptr;
        a.seg_width[s] = on ? d->seg_width[s] : 0;
        if (on) {
            HGNN_REQUIRE(a.seg_table[s] != nullptr && (uintptr_t)a.seg_table[s] % 16 == 0,
                         "hgnn_mlp_forward_bf16_rows128: segment table %d is NULL or not 16-byte aligned", s);
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
        a.save_pre[l] = on ? (unsigned short*)d->save_pre[l] : nullptr;
        if (on) {
            HGNN_REQUIRE((uintptr_t)a.W[l] % 16 == 0 && (uintptr_t)a.b[l] % 16 == 0 &&
                             (uintptr_t)a.lnw[l] % 16 == 0 && (uintptr_t)a.lnb[l] % 16 == 0,
                         "hgnn_mlp_forward_bf16_rows128: layer %d parameters must be 16-byte aligned", l);
            HGNN_REQUIRE((uintptr_t)a.save_pre[l] % 8 == 0, "hgnn_mlp_forward_bf16_rows128: save_pre[%d] must be 8-byte aligned", l);
        }
    }
    a.eps = d->ln_eps;
    a.skip = (const unsigned short*)d->skip;
    a.out = (unsigned short*)out;
    a.M = d->M;
    a.ablate = g_opt_mlp_ablate;
    HGNN_REQUIRE((uintptr_t)out % 8 == 0 && (uintptr_t)a.skip % 8 == 0,
                 "hgnn_mlp_forward_bf16_rows128: out/skip must be 8-byte aligned");
    if (d->n_layers == 2) return fr::launch<2>(a, stream);
    return fr::launch<3>(a, stream);
}
