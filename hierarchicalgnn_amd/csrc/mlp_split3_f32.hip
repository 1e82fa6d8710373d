// fp32 fused  gather -> concat -> [Linear -> LayerNorm -> act] x {2,3} -> (+skip)  MLP whose GEMMs run on the bf16
// matrix pipe as SPLIT-bf16 products (hgnn_mlp_forward_f32_split3; the Python layer's default fp32 MLP path at latent
// 128 / 256, fused.set_fp32_split3).
//
// Every fp32 operand is written as  x = hi + mid (+ lo),  hi = bf16(x), mid = bf16(x - hi): 8 + 8 significand bits.
// Products of bf16 parts are exact in fp32 and v_mfma_f32_16x16x32_bf16 accumulates them in fp32, so
//     x . w  ~=  hi.hi + mid.hi + hi.mid          (dropped: mid.mid and the lo terms, ~2^-16 relative)
// costs 3 bf16 MFMAs = 3/16 of the time of the one fp32 MFMA (v_mfma_f32_16x16x4_f32 runs at 1/16 of the bf16 rate).
// Error budget, measured against the REFERENCE's scores on BASELINE config 2 (tests/test_split_bf16_study.py, oracle
// arithmetic): 2.0e-5 at model level, 5x inside the 1e-4 bar; bias, LayerNorm, activations (exact-erf GELU), skip and
// all rows in HBM stay fp32.
//
// Decomposition = the bf16 feature-split kernel's (mlp_split_bf16.hip): a workgroup of 8 waves owns 64 rows, wave w an
// eighth of every layer's features; weights stream from L2 in A-fragment order through the register ring (gemm_lds).
// What changes:
//   * input rows are fp32 in HBM; a 128-wide k-panel is split into a hi and a mid bf16 plane while it is written to
//     LDS; per 32-wide k-chunk the loop issues  W_mid.x_hi, W_hi.x_mid, W_hi.x_hi  (small terms first) from ONE fetch of
//     the chunk's W_hi and W_mid fragments -- the host interleaves them per chunk (fused._split3_weight);
//   * hidden rows are kept as two bf16 planes (hi, mid) in LDS, the hidden layers' GEMMs run the same loop;
//   * gathered node segments arrive pre-projected in fp32 (hgnn_mlp_desc.n_pre, exact fp32 library GEMMs) as whole
//     rows by LDS-DMA, as in the bf16 kernel;
//   * output / skip rows are fp32 (16 bytes per lane and tile).
#include "mlp_split_common.h"

namespace hgnn {
namespace f3 {
using namespace fs;

struct Args {
    const float* seg_table[3];
    const int32_t* seg_index[3];
    int seg_width[3];
    int n_seg;
    int K1;
    const unsigned short* W[3];  // bf16 split-3 stream in A-fragment order (see header)
    const float* b[3];
    const float* lnw[3];
    const float* lnb[3];
    int act[3];
    float eps;
    const float* skip;
    float* out;
    long long M;
    const float* pre_table[2];
    const int32_t* pre_index[2];
    int n_pre;
    float* save_pre[3];   // optional fp32 [M, width_l] dumps of each layer's pre-LayerNorm output (training forward)
    int stagger;          // shader-clock cycles between the start of workgroup slots (blockIdx % 16); 0 = all start together
#ifdef HGNN_SPLIT3_STAMPS
    unsigned long long* stamps;   // DIAGNOSTIC build (tools/split3_stamps.hip): [tiles][waves][9] shader-clock stamps
    long long stamp_tiles;
#endif
};

// phase boundaries of a tile, stamped by lane 0 of every wave in the diagnostic build only (no stamp executes in the
// product build): 0 tile start | 1 projected rows added | 2 first panel stored | 3 layer-1 GEMM done | 4 LayerNorm + GELU done
// | 5 hidden planes written | 6 output GEMM done | 7 output LayerNorm + tanh done | 8 rows stored
#ifdef HGNN_SPLIT3_STAMPS
#define HGNN_STAMP(k)                                                                                              \
    do {                                                                                                           \
        if (a.stamps != nullptr && tile < a.stamp_tiles && (threadIdx.x & 63) == 0)                                \
            a.stamps[((size_t)tile * NW + (threadIdx.x >> 6)) * 9 + (k)] = __builtin_amdgcn_s_memtime();             \
    } while (0)
#else
#define HGNN_STAMP(k) do { } while (0)
#endif

constexpr int NJ = 4;
constexpr int TE = 64;
constexpr int PK = 128;
constexpr int PRS = PK * 2 + 16;   // bf16 panel row stride (padded: 2-way conflicts at most)
constexpr int PANEL = TE * PRS;
constexpr int CPP = PK / 32;

// (hi, mid) planes of 4 fp32 values
__device__ __forceinline__ void split4(const f32x4 v, u16x4& h, u16x4& m) {
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        const unsigned short hb = bf16_bits(v[c]);
        h[c] = hb;
        m[c] = bf16_bits(v[c] - bf16_float(hb));
    }
}

// bias -> LayerNorm over ALL features of the layer -> activation (exact-erf GELU: this is the fp32 parity path)
// ACT >= 0: the activation code is a compile-time constant (a runtime switch compiles to a scalar branch tree PER
// ELEMENT -- ~600 branches per tile and wave, every element its own basic block, no overlap between the
// transcendental chains of neighbouring elements: PMC showed 3.9k vector instructions and 19 % VALU-active wave time
// per tile against 768 MFMAs); ACT = -1 keeps the runtime code for the rare combinations
template <int NT, int NW, int ACT>
__device__ __forceinline__ void layernorm_act(f32x4 (&acc)[NT][NJ], const float* __restrict__ lnw,
                                              const float* __restrict__ lnb, int act_rt, float eps, float* red, int wave,
                                              int ei, int g) {
    const int act = ACT >= 0 ? ACT : act_rt;
    constexpr float inv_n = 1.0f / (float)(NW * NT * 16);
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
        float s = 0.f, q = 0.f;
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            const f32x4 v = acc[t][j];
            s += (v.x + v.y) + (v.z + v.w);
            q = fmaf(v.x, v.x, q);
            q = fmaf(v.y, v.y, q);
            q = fmaf(v.z, v.z, q);
            q = fmaf(v.w, v.w, q);
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
    }
    __syncthreads();
    float rstd[NJ], shift[NJ];
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
        float s = 0.f, q = 0.f;
#pragma unroll
        for (int w = 0; w < NW; ++w) {
            const f32x2 sq = *(const f32x2*)(red + (w * TE + j * 16 + ei) * 2);
            s += sq.x;
            q += sq.y;
        }
        const float mean = s * inv_n;
        const float var = fmaxf(fmaf(-mean, mean, q * inv_n), 0.f);
        rstd[j] = 1.0f / sqrtf(var + eps);
        shift[j] = -mean * rstd[j];
    }
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        const f32x4 w4 = *(const f32x4*)(lnw + t * 16);
        const f32x4 b4 = *(const f32x4*)(lnb + t * 16);
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            f32x4 v = acc[t][j];
            v.x = act_apply(fmaf(fmaf(v.x, rstd[j], shift[j]), w4.x, b4.x), act);
            v.y = act_apply(fmaf(fmaf(v.y, rstd[j], shift[j]), w4.y, b4.y), act);
            v.z = act_apply(fmaf(fmaf(v.z, rstd[j], shift[j]), w4.z, b4.z), act);
            v.w = act_apply(fmaf(fmaf(v.w, rstd[j], shift[j]), w4.w, b4.w), act);
            acc[t][j] = v;
        }
    }
}

template <int NT>
__device__ __forceinline__ void init_bias(f32x4 (&acc)[NT][NJ], const float* __restrict__ b) {
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        const f32x4 bv = *(const f32x4*)(b + t * 16);
#pragma unroll
        for (int j = 0; j < NJ; ++j) acc[t][j] = bv;
    }
}

// training: this wave's slice of a layer's pre-LayerNorm rows -> fp32 [M, NOUT]
template <int NT, int NOUT>
__device__ __forceinline__ void dump_pre(const f32x4 (&acc)[NT][NJ], float* base, long long M, long long e0, int wave,
                                         int ei, int g) {
    if (base == nullptr) return;
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
        const long long e = e0 + j * 16 + ei;
        if (e >= M) continue;
        float* p = base + (size_t)e * NOUT + (size_t)(wave * NT * 16 + 4 * g);
#pragma unroll
        for (int t = 0; t < NT; ++t) *(f32x4*)(p + t * 16) = acc[t][j];
    }
}

// activated tile -> the hi and mid bf16 planes of the hidden rows (plane stride PLB bytes, row stride HRS)
template <int NT, int HRS, int PLB>
__device__ __forceinline__ void write_hidden2(const f32x4 (&acc)[NT][NJ], char* smem, int wave, int ei, int g) {
    char* lane0 = smem + ei * HRS + ((2 * wave * NT) << 4) + (g << 3);
#pragma unroll
    for (int t = 0; t < NT; ++t) {
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            u16x4 h, m;
            split4(acc[t][j], h, m);
            *(u16x4*)(lane0 + j * 16 * HRS + t * 32) = h;
            *(u16x4*)(lane0 + PLB + j * 16 * HRS + t * 32) = m;
        }
    }
}

// acc += x_hi.W_hi + x_mid.W_hi + x_hi.W_mid over n real 32-wide k-chunks.  The weight stream holds, per real chunk,
// the W_hi fragments followed by the W_mid fragments (virtual chunks 2c, 2c + 1: fused._split3_weight), so a W_hi
// fragment is fetched ONCE and feeds both operand planes (the first version streamed [W_hi | W_hi | W_mid] through
// three passes of the bf16 kernel's loop: 1.5 MiB of L2 reads per 64-row tile, L2-bound at 8.0 ms; now 1.0 MiB).
// Everything is double-buffered one chunk ahead: a chunk is 3 NT NJ MFMAs (768 cycles at NT = 4).
// FOUR: also the mid.mid product (2^-16 of a term: halves the truncation error; the data-gradient GEMMs of the
// training backward use it, gradients pass through a dozen of these before they reach the first cell)
template <int NT, int RS, int NW, bool FOUR = false>
__device__ __forceinline__ void gemm3s(f32x4 (&acc)[NT][NJ], const u16x8* __restrict__ wp, int gv0, int vtotal,
                                       const char* b_hi, const char* b_mid, int n) {
    constexpr int VS = NW * NT * 64;   // u16x8 units per virtual chunk
    u16x8 wh[2][NT], wm[2][NT], bh[2][NJ], bm[2][NJ];
    auto loadw = [&](int c, int sl) {
        int v = gv0 + 2 * c;
        if (v + 1 >= vtotal) v = vtotal - 2;
        const u16x8* p = wp + (size_t)v * VS;
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            wh[sl][t] = p[t * 64];
            wm[sl][t] = p[VS + t * 64];
        }
    };
    auto loadb = [&](int c, int sl) {
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            bh[sl][j] = *(const u16x8*)(b_hi + j * 16 * RS + c * 64);
            bm[sl][j] = *(const u16x8*)(b_mid + j * 16 * RS + c * 64);
        }
    };
    loadw(0, 0);
    loadb(0, 0);
    for (int c = 0; c < n; c += 2) {
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int cn = c + u + 1 < n ? c + u + 1 : n - 1;
            loadw(cn, (u + 1) & 1);
            loadb(cn, (u + 1) & 1);
            // keep the next chunk's loads AHEAD of this chunk's MFMAs (hipcc otherwise sinks them to their first use)
            __builtin_amdgcn_sched_group_barrier(0x020, 2 * NT, 0);
            __builtin_amdgcn_sched_group_barrier(0x100, 2 * NJ, 0);
#pragma unroll
            for (int t = 0; t < NT; ++t) {
#pragma unroll
                for (int j = 0; j < NJ; ++j) {
                    if constexpr (FOUR)
                        acc[t][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(as_bf16(wm[u][t]), as_bf16(bm[u][j]), acc[t][j], 0, 0, 0);
                    acc[t][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(as_bf16(wm[u][t]), as_bf16(bh[u][j]), acc[t][j], 0, 0, 0);
                    acc[t][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(as_bf16(wh[u][t]), as_bf16(bm[u][j]), acc[t][j], 0, 0, 0);
                    acc[t][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(as_bf16(wh[u][t]), as_bf16(bh[u][j]), acc[t][j], 0, 0, 0);
                }
            }
            __builtin_amdgcn_sched_group_barrier(0x008, (FOUR ? 4 : 3) * NT * NJ, 0);
        }
    }
}

// NW waves share the 64 rows (8: two per SIMD; 16: four per SIMD -- at latent 256 the two bf16 planes of the hidden
// rows leave room for ONE workgroup per CU, whose waves are phase-locked by the barriers: more of them hide more
// latency); NTH / NTO: 16-feature tiles per wave of the hidden / output layers; NL = 2 or 3 layers
template <int NW, int NTH, int NTO, int NL, int ACT_H, int ACT_O>
__global__ __launch_bounds__(NW * 64, NW == 4 ? 2 : 1) void k_mlp_f32_split3(const Args a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int NTHR = NW * 64;
    constexpr int H = NTH * NW * 16;
    constexpr int O = NTO * NW * 16;
    constexpr int HRS = H * 2 + 16;            // hidden row stride of one plane (padded)
    constexpr int PLB = TE * HRS;              // bytes of one hidden plane
    constexpr int PHB = H * 4;                 // bytes of an fp32 P row
    constexpr int REGION = cmax(cmax(2 * PLB, 4 * PANEL), 4 * 16 * PHB);
    constexpr int NC = H / 32;                 // k-chunks of a hidden layer (per plane)
    float* red = (float*)(smem + REGION);                 // [NW][TE][sum, sumsq]
    int32_t* pidx = (int32_t*)(red + NW * TE * 2);        // [2][TE]
    int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // Everything derived from the lane id is tile-invariant; in the persistent loop hipcc hoists all of it out of the
    // loop and spills it (136-257 registers in the first persistent build).  refresh() makes the thread id opaque at
    // the top of every tile, so the address arithmetic is redone where it is used (mlp_rows128_bf16.hip, same cure).
    int lane, ei, g, prow, pcol;
    constexpr int LPR = PK * 4 / 16;   // 32 threads move one row's 128 fp32 (16 bytes each)
    auto refresh = [&]() {
        asm volatile("" : "+v"(tid));
        lane = tid & 63;
        ei = lane & 15;
        g = lane >> 4;
        prow = tid / LPR;
        pcol = tid % LPR;
    };
    refresh();
    // gather rows of a tile, staged in LDS one tile ahead: [parity][3 segments + 2 pre-projected][TE]
    int32_t* tix = pidx + 2 * TE;
    const long long n_tiles = (a.M + TE - 1) / TE;
    constexpr int NIX = (5 * TE + NTHR - 1) / NTHR;   // table entries per thread (1 with 8 waves, 2 with 4)
    auto fetch_index = [&](long long tile, int k) {   // entry tid + k NTHR (< 5 TE) of the tile's table
        int r = 0;
        const int slot = tid + k * NTHR;
        if (slot < 5 * TE) {
            const int which = slot / TE;
            long long e = tile * TE + (slot % TE);
            if (e >= a.M) e = a.M - 1;
            r = (int)e;
            const int32_t* ix = which < 3 ? (which < a.n_seg ? a.seg_index[which] : nullptr)
                                          : (which - 3 < a.n_pre ? a.pre_index[which - 3] : nullptr);
            if (ix != nullptr) r = ix[e];
            if (which >= 3 && which - 3 >= a.n_pre) r = 0;
        }
        return r < 0 ? 0 : r;
    };
    // ---- input panels: 16 rows per pass with 8 waves
    constexpr int RPP = NTHR / LPR;
    constexpr int NP = TE / RPP;
    const int np = a.K1 / PK;
    const int p1 = a.seg_width[0] / PK;
    const int p2 = p1 + (a.n_seg > 1 ? a.seg_width[1] / PK : np);
    // pre-projected gathered segments (fp32 rows) by LDS-DMA, WAVE-PRIVATE: a wave fetches only its own 64-feature slice
    // (256 bytes) of the 16 rows of a pass into its own ring of four buffers, so the P phase needs no workgroup
    // barrier at all (the first version moved whole rows and paid one barrier per pass: 8 per tile in a workgroup whose
    // 8 waves are phase-locked anyway)
    static_assert(NTH == 4, "a wave's slice of a projected row is 16 pieces of 16 bytes");
    constexpr int FWB = NTH * 16 * 4;      // bytes of this wave's slice of a P row (256)
    constexpr int RPI = 1024 / FWB;        // rows per DMA instruction (4)
    constexpr int IPW = 16 / RPI;          // DMA instructions per wave and pass (4)
    static_assert(NW * 4 * 16 * FWB <= REGION, "P ring does not fit");
    const int npass = a.n_pre * NJ;
    char* pring = smem + wave * (4 * 16 * FWB);
    auto pissue = [&](int p, const int32_t* ti) {
        const int sgm = p / NJ, j = p % NJ;
#pragma unroll
        for (int i = 0; i < IPW; ++i) {
            const int row = i * RPI + (lane >> 4), pc = lane & 15;
            const int r = ti[(3 + sgm) * TE + 16 * j + row];
            const char* src = (const char*)(a.pre_table[sgm] + (size_t)r * H + wave * NTH * 16) + ((pc ^ row) << 4);
            dma_piece(src, lds_addr_of((float*)(pring + (p & 3) * 16 * FWB)) + (unsigned)i * 1024u);
        }
    };

    long long tile = blockIdx.x;
    if (tile >= n_tiles) return;
    if (a.stagger > 0) {   // DIAGNOSTIC (no effect measured, profiles/r03_split3_khalf_*): de-phase the persistent workgroups once
        const unsigned long long t0 = __builtin_amdgcn_s_memtime();
        const unsigned long long wait = (unsigned long long)(blockIdx.x % 16) * (unsigned)a.stagger;
        while (__builtin_amdgcn_s_memtime() - t0 < wait) __builtin_amdgcn_s_sleep(32);
    }
#pragma unroll
    for (int k = 0; k < NIX; ++k) {
        const int r = fetch_index(tile, k);
        if (tid + k * NTHR < 5 * TE) tix[tid + k * NTHR] = r;
    }
    __syncthreads();
    // Persistent workgroups: the hidden planes leave ONE workgroup per CU at latent 256, so nothing overlaps a tile's
    // start-up chain (gather indices -> projected rows -> first panel) unless the tile before requests it: the next
    // tile's indices are fetched under the output GEMM, its first three P passes under the output epilogue.
    for (int it = 0; tile < n_tiles; tile += gridDim.x, ++it) {
        const int32_t* ti = tix + (it & 1) * 5 * TE;
        int32_t* ti_next = tix + ((it + 1) & 1) * 5 * TE;
        const bool has_next = tile + gridDim.x < n_tiles;
        const long long e0 = tile * TE;
        refresh();
        HGNN_STAMP(0);
        const float* px[NP];
#pragma unroll
        for (int i = 0; i < NP; ++i)
            px[i] = a.seg_table[0] + (size_t)ti[i * RPP + prow] * (size_t)a.seg_width[0] + pcol * 4;
        f32x4 st[NP];
        int pl = 0;
        auto load_panel = [&]() {
#pragma unroll
            for (int i = 0; i < NP; ++i) {
                st[i] = *(const f32x4*)px[i];
                px[i] += PK;
            }
            ++pl;
            if (pl == p1) {
#pragma unroll
                for (int i = 0; i < NP; ++i)
                    px[i] = a.seg_table[1] + (size_t)ti[TE + i * RPP + prow] * (size_t)a.seg_width[1] + pcol * 4;
            }
            if (pl == p2) {
#pragma unroll
                for (int i = 0; i < NP; ++i)
                    px[i] = a.seg_table[2] + (size_t)ti[2 * TE + i * RPP + prow] * (size_t)a.seg_width[2] + pcol * 4;
            }
        };
        auto store_panel = [&](int buf) {   // buffers 2 buf (hi) and 2 buf + 1 (mid)
#pragma unroll
            for (int i = 0; i < NP; ++i) {
                u16x4 h, m;
                split4(st[i], h, m);
                char* dst = smem + (2 * buf) * PANEL + (i * RPP + prow) * PRS + pcol * 8;
                *(u16x4*)dst = h;
                *(u16x4*)(dst + PANEL) = m;
            }
        };

        // ---------------- layer 1
        f32x4 acc1[NTH][NJ];
        init_bias<NTH>(acc1, a.b[0] + wave * NTH * 16 + 4 * g);
        // the first input panel is requested BEFORE the (first tile's) pre-projected rows: it is older in the in-order
        // vector-memory queue, so the counted waits of the DMA ring below stay conservative
        const int vtotal = 2 * (a.K1 / 32);
        const u16x8* wp = (const u16x8*)a.W[0] + (size_t)(wave * NTH) * 64 + lane;
        load_panel();
        if (npass > 0) {
            if (it == 0) {   // later tiles: requested under the previous tile's output epilogue
                pissue(0, ti);
                if (npass > 1) pissue(1, ti);
                if (npass > 2) pissue(2, ti);
            }
#pragma unroll
            for (int p = 0; p < 2 * NJ; ++p) {
                if (p < npass) {
                    // at most the two younger passes' DMAs may still be out (everything else this wave issued since --
                    // stores, index and panel loads -- is younger still: the count is conservative)
                    if (p + 2 < npass) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * IPW) : "memory");
                    else if (p + 1 < npass) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(IPW) : "memory");
                    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    const char* buf = pring + (p & 3) * 16 * FWB + ei * FWB;
#pragma unroll
                    for (int t = 0; t < NTH; ++t)
                        acc1[t][p % NJ] += *(const f32x4*)(buf + (((4 * t + g) ^ ei) << 4));
                    if (p + 3 < npass) pissue(p + 3, ti);
                }
            }
            __syncthreads();
        }
        HGNN_STAMP(1);
        {
            store_panel(0);
            __syncthreads();
            HGNN_STAMP(2);
            const char* blane = smem + ei * PRS + (g << 4);
            for (int p = 0; p < np; ++p) {
                const bool more = p + 1 < np;
                if (more) load_panel();
                const char* bh = blane + (2 * (p & 1)) * PANEL;
                gemm3s<NTH, PRS, NW>(acc1, wp, 2 * p * CPP, vtotal, bh, bh + PANEL, CPP);
                if (more) store_panel((p + 1) & 1);
                __syncthreads();
            }
        }
        HGNN_STAMP(3);
        dump_pre<NTH, H>(acc1, a.save_pre[0], a.M, e0, wave, ei, g);
        layernorm_act<NTH, NW, ACT_H>(acc1, a.lnw[0] + wave * NTH * 16 + 4 * g, a.lnb[0] + wave * NTH * 16 + 4 * g, a.act[0],
                                      a.eps, red, wave, ei, g);
        // (the barrier inside layernorm_act also means: every wave is done reading the panels)
        HGNN_STAMP(4);
        write_hidden2<NTH, HRS, PLB>(acc1, smem, wave, ei, g);
        __syncthreads();
        HGNN_STAMP(5);
        const char* hlane = smem + ei * HRS + (g << 4);
        if constexpr (NL == 3) {
            // ---------------- middle layer (same width), hidden planes rewritten in place
            init_bias<NTH>(acc1, a.b[1] + wave * NTH * 16 + 4 * g);
            {
                const u16x8* wp1 = (const u16x8*)a.W[1] + (size_t)(wave * NTH) * 64 + lane;
                gemm3s<NTH, HRS, NW>(acc1, wp1, 0, 2 * NC, hlane, hlane + PLB, NC);
            }
            dump_pre<NTH, H>(acc1, a.save_pre[1], a.M, e0, wave, ei, g);
            layernorm_act<NTH, NW, ACT_H>(acc1, a.lnw[1] + wave * NTH * 16 + 4 * g, a.lnb[1] + wave * NTH * 16 + 4 * g,
                                          a.act[1], a.eps, red, wave, ei, g);
            write_hidden2<NTH, HRS, PLB>(acc1, smem, wave, ei, g);   // (barrier inside layernorm_act: all reads done)
            __syncthreads();
        }
        // ---------------- output layer
        refresh();
        constexpr int LO = NL - 1;
        f32x4 acc2[NTO][NJ];
        init_bias<NTO>(acc2, a.b[LO] + wave * NTO * 16 + 4 * g);
        int r_next[NIX];   // fly under the output GEMM
#pragma unroll
        for (int k = 0; k < NIX; ++k) r_next[k] = has_next ? fetch_index(tile + gridDim.x, k) : 0;
        {
            const u16x8* wpo = (const u16x8*)a.W[LO] + (size_t)(wave * NTO) * 64 + lane;
            gemm3s<NTO, HRS, NW>(acc2, wpo, 0, 2 * NC, hlane, hlane + PLB, NC);
        }
        HGNN_STAMP(6);
#pragma unroll
        for (int k = 0; k < NIX; ++k)
            if (has_next && tid + k * NTHR < 5 * TE) ti_next[tid + k * NTHR] = r_next[k];
        dump_pre<NTO, O>(acc2, a.save_pre[LO], a.M, e0, wave, ei, g);
        // skip rows BEFORE the next tile's DMAs: queued behind them they would wait for the projected rows
        f32x4 sk[NTO][NJ];
        size_t off[NJ];
        bool valid[NJ];
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            const long long e = e0 + j * 16 + ei;
            valid[j] = e < a.M;
            off[j] = (size_t)(valid[j] ? e : a.M - 1) * O + (size_t)(wave * NTO * 16 + 4 * g);
            if (a.skip != nullptr) {
#pragma unroll
                for (int t = 0; t < NTO; ++t) sk[t][j] = *(const f32x4*)(a.skip + off[j] + t * 16);
            }
        }
        layernorm_act<NTO, NW, ACT_O>(acc2, a.lnw[LO] + wave * NTO * 16 + 4 * g, a.lnb[LO] + wave * NTO * 16 + 4 * g, a.act[LO],
                                      a.eps, red, wave, ei, g);
        // (the barrier inside: every wave is past the hidden planes, the next tile's indices are visible)
        HGNN_STAMP(7);
        if (has_next && npass > 0) {
            pissue(0, ti_next);
            if (npass > 1) pissue(1, ti_next);
            if (npass > 2) pissue(2, ti_next);
        }
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            if (!valid[j]) continue;
#pragma unroll
            for (int t = 0; t < NTO; ++t) {
                f32x4 v = acc2[t][j];
                if (a.skip != nullptr) v += sk[t][j];
                *(f32x4*)(a.out + off[j] + t * 16) = v;
            }
        }
        HGNN_STAMP(8);
    }
}

// =====================================================================================================================
// k_mlp_f32_split3_r128 (round 3): the latent-256 edge / superedge update (K1 -> 512 -> 256, two layers) on 128-ROW tiles.
//
// Why: the 64-row kernel above streams 1 MiB of weight fragments per tile from L2 -- 30k of a tile's ~80k cycles at the
// ~35 B/clk a CU sustains from L2 -- and every per-tile fixed cost (barriers, the waves' arbitration skew, pipeline
// fill) is paid per 64 rows.  A fragment here feeds EIGHT row tiles (half the weight bytes per row), a tile has half as
// many barriers per row.  What made this impossible inside the 8-wave kernel is the register file: 128 rows x 512 hidden
// features of fp32 accumulators are 256 KB, half of a CU's registers.  So: ONE workgroup of FOUR waves per CU, one wave
// per SIMD, each with the full 512-register budget (256 VGPR + 256 AGPR; hipcc places the accumulators there):
//   * wave w owns hidden features [128 w, 128 w + 128) (8 tiles x 8 row tiles = 256 accumulator registers) and output
//     features [64 w, 64 w + 64);
//   * with one wave per SIMD nothing hides a load's latency but the wave's own instruction stream: weight fragments are
//     double-buffered one k-chunk ahead, LDS operands one HALF chunk ahead (four row tiles), and a full scheduling
//     barrier keeps hipcc from sinking the prefetch to its first use (it does: see the 64-row kernel's ISA);
//   * the hidden rows of 128 rows x 512 features x (hi, mid) are 256 KB -- more than LDS -- so the output GEMM runs in
//     two K-HALVES: the first half of every wave's hidden tiles is activated, split and written (128 KB), consumed,
//     then the second half, whose pre-activation accumulators simply stay in registers meanwhile;
//   * both 128-wide k-panels of the (pre-projected) first layer are resident at once: layer 1 is one uninterrupted
//     8-chunk stream per panel pair.
// Same arithmetic, same weight streams, same descriptor as the 64-row kernel; selected by the host for M >= 65,536 (at 40k rows its 313 tiles quantise worse over 256 CUs than 625 64-row tiles: 0.228 vs 0.210 ms).
#ifdef HGNN_SPLIT3_STAMPS
#define HGNN_STAMPW(k)                                                                                             \
    do {                                                                                                           \
        if (a.stamps != nullptr && tile < a.stamp_tiles && (threadIdx.x & 63) == 0)                                \
            a.stamps[((size_t)tile * WNW + (threadIdx.x >> 6)) * 12 + (k)] = __builtin_amdgcn_s_memtime();            \
    } while (0)
#else
#define HGNN_STAMPW(k) do { } while (0)
#endif
extern int g_opt_split3_one_wg;
#define HGNN_KH_NS r128
#define HGNN_KH_NW 8
#define HGNN_KH_NJ 8
#include "mlp_split3_khalf.h"
#undef HGNN_KH_NS
#undef HGNN_KH_NW
#undef HGNN_KH_NJ
// the same kernel on 64-row tiles with FOUR waves (128 hidden / 64 output features each) and TWO workgroups per CU
// (74 KB of LDS each): nothing inside a workgroup overlaps its barrier-separated phases -- the other workgroup does
#define HGNN_KH_NS r64x2
#define HGNN_KH_NW 4
#define HGNN_KH_NJ 4
#include "mlp_split3_khalf.h"
#undef HGNN_KH_NS
#undef HGNN_KH_NW
#undef HGNN_KH_NJ

// out[M, N] = x[M, K] . W^T (+ skip): ONE plain fp32 Linear without bias / LayerNorm on the same split-bf16 loop -- the
// M-row data-gradient GEMMs of the fp32 training backward (dz . W of Modules/utils.py:169-196's Linear under autograd;
// the library's fp32 GEMM runs them at ~130 TFLOP/s).  N = NT NW 16; K a multiple of 128; W as in the MLP kernel.
// blockIdx.y selects one of up to two (weight stream, output) pairs over the SAME input rows: the two pre-projections of
// an edge update (P_s = nodes . W_s^T for the source and the destination segment) run as one launch (hgnn_project_f32_split3)
template <int NW, int NT, bool FOUR>
__global__ __launch_bounds__(NW * 64, NW == 4 ? 2 : 1) void k_linear_f32_split3(const float* __restrict__ x, int K,
                                                                               const unsigned short* __restrict__ W0,
                                                                               const unsigned short* __restrict__ W1,
                                                                               const float* __restrict__ skip,
                                                                               float* __restrict__ out0,
                                                                               float* __restrict__ out1, long long M) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const unsigned short* __restrict__ W = blockIdx.y == 0 ? W0 : W1;
    float* __restrict__ out = blockIdx.y == 0 ? out0 : out1;
    constexpr int NTHR = NW * 64;
    constexpr int N = NT * NW * 16;
    constexpr int LPR = PK * 4 / 16;
    constexpr int RPP = NTHR / LPR;
    constexpr int NP = TE / RPP;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int ei = lane & 15, g = lane >> 4;
    const int prow = tid / LPR, pcol = tid % LPR;
    const long long e0 = (long long)blockIdx.x * TE;
    const float* px[NP];
#pragma unroll
    for (int i = 0; i < NP; ++i) {
        long long e = e0 + i * RPP + prow;
        if (e >= M) e = M - 1;
        px[i] = x + (size_t)e * (size_t)K + pcol * 4;
    }
    f32x4 st[NP];
    auto load_panel = [&]() {
#pragma unroll
        for (int i = 0; i < NP; ++i) {
            st[i] = *(const f32x4*)px[i];
            px[i] += PK;
        }
    };
    auto store_panel = [&](int buf) {
#pragma unroll
        for (int i = 0; i < NP; ++i) {
            u16x4 h, m;
            split4(st[i], h, m);
            char* dst = smem + (2 * buf) * PANEL + (i * RPP + prow) * PRS + pcol * 8;
            *(u16x4*)dst = h;
            *(u16x4*)(dst + PANEL) = m;
        }
    };
    f32x4 acc[NT][NJ];
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int j = 0; j < NJ; ++j) acc[t][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int np = K / PK;
    const int vtotal = 2 * (K / 32);
    const u16x8* wp = (const u16x8*)W + (size_t)(wave * NT) * 64 + lane;
    load_panel();
    store_panel(0);
    __syncthreads();
    const char* blane = smem + ei * PRS + (g << 4);
    for (int p = 0; p < np; ++p) {
        const bool more = p + 1 < np;
        if (more) load_panel();
        const char* bh = blane + (2 * (p & 1)) * PANEL;
        gemm3s<NT, PRS, NW, FOUR>(acc, wp, 2 * p * CPP, vtotal, bh, bh + PANEL, CPP);
        if (more) store_panel((p + 1) & 1);
        __syncthreads();
    }
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
        const long long e = e0 + j * 16 + ei;
        if (e >= M) continue;
        const size_t off = (size_t)e * N + (size_t)(wave * NT * 16 + 4 * g);
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            f32x4 v = acc[t][j];
            if (skip != nullptr) v += *(const f32x4*)(skip + off + t * 16);
            *(f32x4*)(out + off + t * 16) = v;
        }
    }
}

template <int NW, int NT, bool FOUR>
static int launch_linear(const float* x, int K, const unsigned short* W0, const unsigned short* W1, const float* skip,
                         float* out0, float* out1, long long M, hipStream_t s) {
    const size_t lds_bytes = (size_t)4 * PANEL;
    auto kern = k_linear_f32_split3<NW, NT, FOUR>;
    HGNN_CHECK_HIP(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes));
    kern<<<dim3((unsigned)ceil_div(M, TE), W1 != nullptr ? 2 : 1), NW * 64, lds_bytes, s>>>(x, K, W0, W1, skip, out0, out1, M);
    HGNN_CHECK_HIP(hipGetLastError());
    return HGNN_OK;
}

int g_opt_split3_one_wg = 0;    // hgnn_set_option("mlp_split3_one_wg"): DIAGNOSTIC, one persistent workgroup per CU where two fit
int g_opt_split3_rows128 = 1;   // hgnn_set_option("mlp_split3_rows128"): 1 (default) the 128-row kernel for K -> 512 -> 256 at M >= 65,536 (two tiles per CU)
static int g_cus = 0;

template <int NW, int NTH, int NTO, int NL, int ACT_H, int ACT_O>
static int launch_act(const Args& a, hipStream_t s) {
    constexpr int NTHR = NW * 64;
    constexpr int H = NTH * NW * 16;
    constexpr int HRS = H * 2 + 16;
    constexpr int REGION = cmax(cmax(2 * TE * HRS, 4 * PANEL), 4 * 16 * H * 4);
    const size_t lds_bytes = (size_t)REGION + NW * TE * 2 * sizeof(float) + 2 * TE * sizeof(int32_t) +
                             2 * 5 * TE * sizeof(int32_t);
    if (g_cus == 0) {
        int dev = 0;
        HGNN_CHECK_HIP(hipGetDevice(&dev));
        HGNN_CHECK_HIP(hipDeviceGetAttribute(&g_cus, hipDeviceAttributeMultiprocessorCount, dev));
        if (g_cus <= 0) g_cus = 256;
    }
    const long long n_tiles = ceil_div(a.M, TE);
    const long long resident = (long long)g_cus * (lds_bytes <= 80 * 1024 && !g_opt_split3_one_wg ? 2 : 1);   // persistent workgroups
    const unsigned grid = (unsigned)(n_tiles < resident ? n_tiles : resident);
    auto kern = k_mlp_f32_split3<NW, NTH, NTO, NL, ACT_H, ACT_O>;
    HGNN_CHECK_HIP(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes));
    kern<<<grid, NTHR, lds_bytes, s>>>(a);
    HGNN_CHECK_HIP(hipGetLastError());
    return HGNN_OK;
}

// activation codes as template parameters for the combinations the models use (hidden GELU + Tanh / GELU output:
// edge / node / supernode / superedge networks and encoders; Tanh + Tanh: the hidden layers of the score heads);
// anything else runs the runtime-switch instantiation
template <int NW, int NTH, int NTO, int NL>
static int launch(const Args& a, hipStream_t s) {
    bool hidden_gelu = true, hidden_tanh = true;
    for (int l = 0; l + 1 < NL; ++l) {
        hidden_gelu = hidden_gelu && a.act[l] == HGNN_ACT_GELU;
        hidden_tanh = hidden_tanh && a.act[l] == HGNN_ACT_TANH;
    }
    const int out = a.act[NL - 1];
    if (hidden_gelu && out == HGNN_ACT_TANH) return launch_act<NW, NTH, NTO, NL, HGNN_ACT_GELU, HGNN_ACT_TANH>(a, s);
    if (hidden_gelu && out == HGNN_ACT_GELU) return launch_act<NW, NTH, NTO, NL, HGNN_ACT_GELU, HGNN_ACT_GELU>(a, s);
    if (hidden_tanh && out == HGNN_ACT_TANH) return launch_act<NW, NTH, NTO, NL, HGNN_ACT_TANH, HGNN_ACT_TANH>(a, s);
    return launch_act<NW, NTH, NTO, NL, -1, -1>(a, s);
}

}  // namespace f3
}  // namespace hgnn

using namespace hgnn;

extern "C" int hgnn_mlp_supported_f32_split3(const hgnn_mlp_desc* d) {
    if (d == nullptr) return 0;
    if (d->n_seg < 1 || d->n_seg > 3 || d->n_layers < 2 || d->n_layers > 3) return 0;
    int k = 0;
    for (int s = 0; s < d->n_seg; ++s) {
        if (d->seg_width[s] <= 0 || d->seg_width[s] % 128 != 0) return 0;
        k += d->seg_width[s];
    }
    if (k != d->width[0] || d->w0_cols != 0 || d->w_last_rows != 0) return 0;
    const int n = d->n_layers;
    for (int l = 0; l < n; ++l)
        if (d->W[l] == nullptr || d->b[l] == nullptr || d->ln_w[l] == nullptr || d->ln_b[l] == nullptr) return 0;
    if (d->n_pre < 0 || d->n_pre > 2) return 0;
    for (int s = 0; s < d->n_pre; ++s)
        if (d->pre_table[s] == nullptr || d->pre_index[s] == nullptr) return 0;
    if (d->M < 0 || d->M > 0x7fffffffLL) return 0;
    const int h = d->width[1], o = d->width[n];
    if (n == 3 && d->width[2] != h) return 0;
    if (n == 2 && o == h) return (h == 256 || h == 512) ? 1 : 0;   // the two hidden layers of a score head
    if (h != 2 * o) return 0;
    return (o == 128 || o == 256) ? 1 : 0;
}

extern "C" int hgnn_mlp_forward_f32_split3(const hgnn_mlp_desc* d, float* out, hgnn_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    HGNN_REQUIRE(d != nullptr && out != nullptr, "hgnn_mlp_forward_f32_split3: NULL argument");
    if (!hgnn_mlp_supported_f32_split3(d)) {
        set_error("hgnn_mlp_forward_f32_split3: unsupported shape (K -> 2L (-> 2L) -> L, L in {128, 256}, or K -> H -> H, "
                  "H in {256, 512}; LayerNorm on every layer, every segment a multiple of 128 wide, save_pre optional)");
        return HGNN_ERR_UNSUPPORTED;
    }
    if (d->M == 0) return HGNN_OK;
    f3::Args a;
    for (int s = 0; s < 3; ++s) {
        const bool on = s < d->n_seg;
        a.seg_table[s] = on ? d->seg_table[s] : d->seg_table[0];
        a.seg_index[s] = on ? d->seg_index[s] : nullptr;
        a.seg_width[s] = on ? d->seg_width[s] : 0;
        if (on) {
            HGNN_REQUIRE(a.seg_table[s] != nullptr && (uintptr_t)a.seg_table[s] % 16 == 0,
                         "hgnn_mlp_forward_f32_split3: segment table %d is NULL or not 16-byte aligned", s);
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
        a.save_pre[l] = on ? d->save_pre[l] : nullptr;
        if (on) {
            HGNN_REQUIRE((uintptr_t)a.W[l] % 16 == 0 && (uintptr_t)a.b[l] % 16 == 0 &&
                             (uintptr_t)a.lnw[l] % 16 == 0 && (uintptr_t)a.lnb[l] % 16 == 0 &&
                             (uintptr_t)a.save_pre[l] % 16 == 0,
                         "hgnn_mlp_forward_f32_split3: layer %d parameters / save_pre must be 16-byte aligned", l);
        }
    }
    a.eps = d->ln_eps;
    a.skip = d->skip;
    a.out = out;
    a.M = d->M;
    a.stagger = 0;
    a.n_pre = d->n_pre;
    for (int s = 0; s < 2; ++s) {
        a.pre_table[s] = s < d->n_pre ? d->pre_table[s] : nullptr;
        a.pre_index[s] = s < d->n_pre ? d->pre_index[s] : nullptr;
        HGNN_REQUIRE((uintptr_t)a.pre_table[s] % 16 == 0, "hgnn_mlp_forward_f32_split3: pre_table[%d] must be 16-byte aligned", s);
    }
    HGNN_REQUIRE((uintptr_t)out % 16 == 0 && (uintptr_t)a.skip % 16 == 0,
                 "hgnn_mlp_forward_f32_split3: out/skip must be 16-byte aligned");
    const int o = d->width[d->n_layers];
    if (d->n_layers == 2 && o == d->width[1])
        return o == 512 ? f3::launch<8, 4, 4, 2>(a, stream) : f3::launch<4, 4, 4, 2>(a, stream);
    // latent 128: 4 waves (two 74-KiB workgroups per CU; 8 waves x 1/8 of 256 features left each wave 192 MFMAs per
    // tile against the tile's fixed costs)
    if (d->n_layers == 2 && o == 256 && f3::g_opt_split3_rows128 == 1 && d->M >= 65536) return f3::r128::launch_tile(a, stream);
    if (d->n_layers == 2 && o == 256 && f3::g_opt_split3_rows128 == 2 && d->M >= 65536) return f3::r64x2::launch_tile(a, stream);
    if (d->n_layers == 2) return o == 256 ? f3::launch<8, 4, 2, 2>(a, stream) : f3::launch<4, 4, 2, 2>(a, stream);
    return o == 256 ? f3::launch<8, 4, 2, 3>(a, stream) : f3::launch<4, 4, 2, 3>(a, stream);
}

extern "C" int hgnn_linear_f32_split3(const float* x, int64_t M, int32_t K, const void* w_split, int32_t N,
                                      const float* skip, float* out, hgnn_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    HGNN_REQUIRE(M >= 0 && M <= 0x7fffffffLL && K > 0 && K % 128 == 0 && (N == 256 || N == 512),
                 "hgnn_linear_f32_split3: M = %lld, K = %d (multiple of 128), N = %d (256 or 512)", (long long)M, K, N);
    if (M == 0) return HGNN_OK;
    HGNN_REQUIRE(x != nullptr && w_split != nullptr && out != nullptr, "hgnn_linear_f32_split3: NULL argument");
    HGNN_REQUIRE((uintptr_t)x % 16 == 0 && (uintptr_t)w_split % 16 == 0 && (uintptr_t)out % 16 == 0 &&
                     (uintptr_t)skip % 16 == 0,
                 "hgnn_linear_f32_split3: pointers must be 16-byte aligned");
    if (N == 512) return f3::launch_linear<8, 4, true>(x, K, (const unsigned short*)w_split, nullptr, skip, out, nullptr, M, stream);
    return f3::launch_linear<4, 4, true>(x, K, (const unsigned short*)w_split, nullptr, skip, out, nullptr, M, stream);
}

extern "C" int hgnn_project_f32_split3(const float* x, int64_t M, int32_t K, const void* w_split0, const void* w_split1,
                                       int32_t N, float* out0, float* out1, hgnn_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    HGNN_REQUIRE(M >= 0 && M <= 0x7fffffffLL && K > 0 && K % 128 == 0 && (N == 256 || N == 512),
                 "hgnn_project_f32_split3: M = %lld, K = %d (multiple of 128), N = %d (256 or 512)", (long long)M, K, N);
    if (M == 0) return HGNN_OK;
    HGNN_REQUIRE(x != nullptr && w_split0 != nullptr && out0 != nullptr && (w_split1 == nullptr) == (out1 == nullptr),
                 "hgnn_project_f32_split3: NULL argument (w_split1 / out1 are given together or not at all)");
    HGNN_REQUIRE((uintptr_t)x % 16 == 0 && (uintptr_t)w_split0 % 16 == 0 && (uintptr_t)w_split1 % 16 == 0 &&
                     (uintptr_t)out0 % 16 == 0 && (uintptr_t)out1 % 16 == 0,
                 "hgnn_project_f32_split3: pointers must be 16-byte aligned");
    if (N == 512)
        return f3::launch_linear<8, 4, false>(x, K, (const unsigned short*)w_split0, (const unsigned short*)w_split1, nullptr,
                                              out0, out1, M, stream);
    return f3::launch_linear<4, 4, false>(x, K, (const unsigned short*)w_split0, (const unsigned short*)w_split1, nullptr, out0,
                                          out1, M, stream);
}
