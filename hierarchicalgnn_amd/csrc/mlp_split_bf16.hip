// bf16 fused  gather -> concat -> [Linear -> LayerNorm -> act] x {2,3} -> (+skip)  MLP, feature-split
// decomposition (the wide-layer kernel: L in {128, 256, 512}; BASELINE config 4 is L = 512).
//
// mlp_fused_bf16.hip gives every wave 16-32 edges and ALL features, so each wave re-reads the whole
// weight chunk from LDS: at 16x the fp32 matrix rate that LDS stream (and the DMA issue feeding it)
// is what bounds it, and L = 512 does not fit a wave's accumulator registers at all.  Here a
// workgroup owns TE = 64 edges and wave w owns a QUARTER OF THE FEATURES of every layer:
//   * weights are private to a wave -> no LDS, no DMA: the host stores W in MFMA A-fragment order
//     (fragment (k-chunk c, 16-feature tile T) = 1 KiB, lane l = W[16T + l%16][32c + 8(l/16) .. +7]),
//     a wave streams its slice with plain 16-byte/lane global loads (always L2 hits: every
//     workgroup reads the same ~1 MB) through a register ring that runs >= 8 fragments ahead;
//   * one weight fragment feeds 4 MFMAs (the 4 edge tiles), a workgroup reads each weight byte once
//     for 64 edges;
//   * activations are what the waves share, so THEY go through LDS: the gathered input rows in
//     128-wide k-panels (coalesced 256-byte row pieces, double-buffered), and the hidden activations of each
//     layer as bf16 [64][H] (padded rows: conflict-free ds_write_b64 / ds_read_b128);
//   * LayerNorm statistics: in-register partial sums over the wave's features, exchanged through
//     LDS (2 floats per edge per wave), one-pass variance in fp32.
// GEMMs are transposed as in the other kernels (A = W fragment, B = activations, D[f][e]): lane
// (e = lane&15, g = lane>>4) holds features 16T + 4g + r of edge 16j + e.
#include "mlp_split_common.h"

namespace hgnn {
extern int g_opt_mlp_ablate;
int g_opt_mlp_split_variant = -1;  // hgnn_set_option("mlp_split_variant"): -1 auto, 0 counted waits, 2 burst
namespace fs {

struct Args {
    const unsigned short* seg_table[3];
    const int32_t* seg_index[3];
    int seg_width[3];
    int n_seg;
    int K1;
    const unsigned short* W[3];  // bf16, A-fragment order (see header)
    const float* b[3];
    const float* lnw[3];
    const float* lnb[3];
    int act[3];
    float eps;
    const unsigned short* skip;
    unsigned short* out;
    long long M;
    const unsigned short* pre_table[2];  // pre-projected gathered segments, bf16 [rows, width1] (hgnn_mlp_desc.n_pre)
    const int32_t* pre_index[2];
    int n_pre;
    unsigned short* save_pre[3];  // optional bf16 [M, width_l] dumps of each layer's pre-LayerNorm output: what the
                                  // bf16 training path's backward needs (hgnn_ln_act_backward_bf16, hgnn_wgrad_bf16)
    int ablate;  // DIAGNOSTIC (hgnn_set_option "mlp_ablate", wrong results): 1 = weights from chunk 0 only
                 // (L1-resident), 2 = no LayerNorm / activation, 4 = only the first input panel is loaded,
                 // 8 = no per-panel barriers, 16 = B reads from chunk 0 only
};

// bias -> LayerNorm over ALL features of the layer (this wave holds NT*16 of the NW*NT*16) -> activation
template <int NT, int ACT, int NW, int NJ>
__device__ __forceinline__ void layernorm_act(f32x4 (&acc)[NT][NJ], const float* __restrict__ lnw,
                                              const float* __restrict__ lnb, int act, float eps, float* red,
                                              int wave, int ei, int g, int ablate) {
    constexpr float inv_n = 1.0f / (float)(NW * NT * 16);
    constexpr int TE = 16 * NJ;
    if (ablate & 2) {
        __syncthreads();
        return;
    }
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
            v.x = act_t<ACT>(fmaf(fmaf(v.x, rstd[j], shift[j]), w4.x, b4.x), act);
            v.y = act_t<ACT>(fmaf(fmaf(v.y, rstd[j], shift[j]), w4.y, b4.y), act);
            v.z = act_t<ACT>(fmaf(fmaf(v.z, rstd[j], shift[j]), w4.z, b4.z), act);
            v.w = act_t<ACT>(fmaf(fmaf(v.w, rstd[j], shift[j]), w4.w, b4.w), act);
            acc[t][j] = v;
        }
    }
}

template <int NT, int NJ>
__device__ __forceinline__ void init_bias(f32x4 (&acc)[NT][NJ], const float* __restrict__ b) {
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        const f32x4 bv = *(const f32x4*)(b + t * 16);
#pragma unroll
        for (int j = 0; j < NJ; ++j) acc[t][j] = bv;
    }
}

// activated tile -> bf16 hidden rows in LDS (row = edge, stride HRS bytes, 16-byte pieces XOR-swizzled with the
// row: see gemm_lds).  `row0` = this lane's row 0 of the tile (smem + ei * HRS); the lane's 4 features of tile t are
// the (g & 1) half of logical piece pbase + 2 t + (g >> 1)
template <int NT, int HRS, int NJ, bool SWZ>
__device__ __forceinline__ void write_hidden(const f32x4 (&acc)[NT][NJ], char* row0, int pbase, int ei, int g) {
    if constexpr (!SWZ) {
        // padded rows (narrow shapes): one precomputed lane pointer, immediate offsets -- this variant sits exactly
        // on the 168-register boundary of 3 waves per SIMD
        char* hid_lane = row0 + ((pbase << 4) + (g << 3));
#pragma unroll
        for (int t = 0; t < NT; ++t) {
#pragma unroll
            for (int j = 0; j < NJ; ++j) {
                u16x4 o;
                o[0] = bf16_bits(acc[t][j].x);
                o[1] = bf16_bits(acc[t][j].y);
                o[2] = bf16_bits(acc[t][j].z);
                o[3] = bf16_bits(acc[t][j].w);
                *(u16x4*)(hid_lane + j * 16 * HRS + t * 32) = o;
            }
        }
        return;
    }
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        const int off = (((pbase + 2 * t + (g >> 1)) ^ ei) << 4) + ((g & 1) << 3);
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            u16x4 o;
            o[0] = bf16_bits(acc[t][j].x);
            o[1] = bf16_bits(acc[t][j].y);
            o[2] = bf16_bits(acc[t][j].z);
            o[3] = bf16_bits(acc[t][j].w);
            *(u16x4*)(row0 + j * 16 * HRS + off) = o;
        }
    }
}

template <int NT, int NW, int NJ>
__device__ __forceinline__ void store_out(const f32x4 (&acc)[NT][NJ], const Args& a, long long e0, int wave, int ei,
                                          int g) {
    constexpr int NOUT = NW * NT * 16;
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
        const long long e = e0 + j * 16 + ei;
        if (e >= a.M) continue;
        const size_t off = (size_t)e * NOUT + (size_t)(wave * NT * 16 + 4 * g);
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            f32x4 v = acc[t][j];
            if (a.skip != nullptr) {
                const u16x4 s = *(const u16x4*)(a.skip + off + t * 16);
                v.x += bf16_float(s[0]);
                v.y += bf16_float(s[1]);
                v.z += bf16_float(s[2]);
                v.w += bf16_float(s[3]);
            }
            u16x4 o;
            o[0] = bf16_bits(v.x);
            o[1] = bf16_bits(v.y);
            o[2] = bf16_bits(v.z);
            o[3] = bf16_bits(v.w);
            *(u16x4*)(a.out + off + t * 16) = o;
        }
    }
}

// training: this wave's slice of the layer's pre-LayerNorm rows z[e][f] -> bf16 [M, NW*NT*16] (wave-uniform base)
template <int NT, int NW, int NJ>
__device__ __forceinline__ void dump_pre(const f32x4 (&acc)[NT][NJ], unsigned short* base, long long M, long long e0,
                                         int wave, int ei, int g) {
    if (base == nullptr) return;
    constexpr int NOUT = NW * NT * 16;
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
        const long long e = e0 + j * 16 + ei;
        if (e >= M) continue;
        const size_t off = (size_t)e * NOUT + (size_t)(wave * NT * 16 + 4 * g);
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            u16x4 o;
            o[0] = bf16_bits(acc[t][j].x);
            o[1] = bf16_bits(acc[t][j].y);
            o[2] = bf16_bits(acc[t][j].z);
            o[3] = bf16_bits(acc[t][j].w);
            *(u16x4*)(base + off + t * 16) = o;
        }
    }
}


// NW waves x NJ 16-row tiles per workgroup; NTl: 16-feature tiles PER WAVE of layer l (= width_l / (16 NW));
// PK: k-panel width of the input rows
template <int NT1, int NT2, int NT3, int PK, int ACT_H, int ACT_O, int MINB, int VAR, int NW, int NJ>
__global__ __launch_bounds__(NW * 64, MINB) void k_mlp_bf16_split(const Args a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int TE = 16 * NJ;              // rows per workgroup
    constexpr int NTHR = NW * 64;
    constexpr int NC2 = NT1 * NW / 2;        // k-chunks of the hidden layers (width / 32)
    // Hidden width >= 512 (L >= 256): rows of whole 256-byte bank rows, 16-byte pieces XOR-swizzled with the row
    // (gemm_lds): SQ_LDS_BANK_CONFLICT 9.5 % of SQ_LDS_IDX_ACTIVE instead of 39 % (profiles/r02_edge_mlp_bf16_split_pmc.txt).
    // The kernel time does not move (2.75-2.83 ms at L=256, 8.6 ms at L=512: the LDS array is 14 % busy, it was never
    // the bound).  The L=128 variant sits exactly on the 168-register boundary of 3 waves per SIMD and the swizzled
    // addressing needs one register more (occupancy 3 -> 2: 1.30 vs 1.08 ms), so the narrow shape keeps the
    // +16-byte row padding.
    constexpr bool SWZ = NT1 * NW >= 32;
    constexpr int HRS = NT1 * NW * 32 + (SWZ ? 0 : 16);  // hidden row stride (bytes)
    constexpr int PRS = PK * 2 + (SWZ ? 0 : 16);         // panel row stride
    constexpr int PANEL = TE * PRS;
    constexpr int REGION = cmax(TE * HRS, 2 * PANEL);  // the panels alias the hidden rows
    constexpr int CPP = PK / 32;                      // k-chunks per panel
    float* red = (float*)(smem + REGION);             // [NW waves][TE][sum, sumsq]
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int ei = lane & 15;
    const int g = lane >> 4;
    const long long e0 = (long long)blockIdx.x * TE;

    // ---- input panels: thread (prow, pcol) moves 16 bytes of row prow (+RPP per pass) per panel
    constexpr int LPR = PK * 2 / 16;
    constexpr int RPP = NTHR / LPR;
    constexpr int NP = TE / RPP;
    const int prow = tid / LPR;
    const int pcol = tid % LPR;
    const unsigned short* px[NP];
    int r1[NP], r2[NP];
#pragma unroll
    for (int i = 0; i < NP; ++i) {
        long long e = e0 + i * RPP + prow;
        if (e >= a.M) e = a.M - 1;
        const long long r0 = a.seg_index[0] != nullptr ? (long long)a.seg_index[0][e] : e;
        px[i] = a.seg_table[0] + (size_t)r0 * (size_t)a.seg_width[0] + pcol * 8;
        r1[i] = a.n_seg > 1 ? (a.seg_index[1] != nullptr ? a.seg_index[1][e] : (int)e) : 0;
        r2[i] = a.n_seg > 2 ? (a.seg_index[2] != nullptr ? a.seg_index[2][e] : (int)e) : 0;
    }
    const int np = a.K1 / PK;
    const int p1 = a.seg_width[0] / PK;
    const int p2 = p1 + (a.n_seg > 1 ? a.seg_width[1] / PK : np);
    u16x8 st[NP];
    int pl = 0;  // next panel to load
    auto load_panel = [&]() {
#pragma unroll
        for (int i = 0; i < NP; ++i) {
            st[i] = *(const u16x8*)px[i];
            px[i] += PK;
        }
        ++pl;
        if (pl == p1) {
#pragma unroll
            for (int i = 0; i < NP; ++i) px[i] = a.seg_table[1] + (size_t)r1[i] * (size_t)a.seg_width[1] + pcol * 8;
        }
        if (pl == p2) {
#pragma unroll
            for (int i = 0; i < NP; ++i) px[i] = a.seg_table[2] + (size_t)r2[i] * (size_t)a.seg_width[2] + pcol * 8;
        }
    };
    auto store_panel = [&](int buf) {
#pragma unroll
        for (int i = 0; i < NP; ++i)
            *(u16x8*)(smem + buf * PANEL + (i * RPP + prow) * PRS + ((pcol ^ (SWZ ? (prow & 15) : 0)) << 4)) = st[i];
    };

    // ---------------- layer 1: B = input panels
    f32x4 acc1[NT1][NJ];
    init_bias<NT1, NJ>(acc1, a.b[0] + wave * NT1 * 16 + 4 * g);
    // pre-projected gathered segments: row e starts at b + sum_s P_s[idx_s[e]].  The P rows come in as WHOLE rows
    // (16 bytes per lane, 16 rows per pass) through a ring of LDS buffers and are picked up in the accumulator layout from
    // there: read straight from global memory that layout gives a lane 4 features of 16 different rows per instruction,
    // 8-byte pieces that are issue-bound (round 1 measured the projected path SLOWER for that reason: 3.7 vs 2.7 ms).
    if (a.n_pre > 0) {
        constexpr int HB = NT1 * NW * 32;          // bytes of a P row (= hidden width in bf16)
        constexpr int PPR = HB / 16;               // 16-byte pieces per row
        int32_t* pidx = (int32_t*)(red + NW * TE * 2);   // [2][TE] gather rows of this tile
        for (int i = tid; i < 2 * TE; i += NTHR) {
            const int sgm = i / TE;
            long long e = e0 + (i % TE);
            if (e >= a.M) e = a.M - 1;
            const int r = sgm < a.n_pre ? a.pre_index[sgm][e] : 0;
            pidx[i] = r < 0 ? 0 : r;
        }
        __syncthreads();
        const int npass = a.n_pre * NJ;            // one 16-row tile of one segment per pass
        // LDS-DMA (global_load_lds_dwordx4: 1 KiB per wave instruction, no staging registers) into a ring of four
        // pass buffers, three passes in flight: with register staging the ring was one or two passes deep and every
        // pass exposed most of a random-row HBM latency (projected 2.61-2.75 ms vs direct 2.77-2.83 ms)
        constexpr int NPI = 16 * HB / 1024;        // DMA instructions per pass
        constexpr int IPW = NPI / NW;              // ... per wave
        static_assert(NPI % NW == 0 && 4 * 16 * HB <= REGION, "P staging ring does not fit");
        auto pissue = [&](int p) {
            const int sgm = p / NJ, j = p % NJ;
#pragma unroll
            for (int i = 0; i < IPW; ++i) {
                const int k = wave + i * NW;                       // wave-uniform
                const int gp = k * 64 + lane;
                const int row = gp / PPR, pc = gp % PPR;
                const int r = pidx[sgm * TE + 16 * j + row];
                const char* src = (const char*)(a.pre_table[sgm] + (size_t)r * (HB / 2)) +
                                  (((pc & ~15) | ((pc ^ row) & 15)) << 4);
                dma_piece(src, lds_addr_of((float*)(smem + (p & 3) * 16 * HB)) + (unsigned)k * 1024u);
            }
        };
        pissue(0);
        if (npass > 1) pissue(1);
        if (npass > 2) pissue(2);
#pragma unroll
        for (int p = 0; p < 2 * NJ; ++p) {
            if (p < npass) {
                // this wave's pieces of pass p have landed once at most the younger passes' DMAs are outstanding
                if (p + 2 < npass) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * IPW) : "memory");
                else if (p + 1 < npass) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(IPW) : "memory");
                else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __syncthreads();   // every wave's pieces; and every wave is past its reads of pass p - 1
                const char* buf = smem + (p & 3) * 16 * HB + ei * HB + ((g & 1) << 3);
#pragma unroll
                for (int t = 0; t < NT1; ++t) {
                    const int q = 2 * (wave * NT1 + t) + (g >> 1);
                    const u16x4 v = *(const u16x4*)(buf + (((q & ~15) | ((q ^ ei) & 15)) << 4));
                    acc1[t][p % NJ].x += bf16_float(v[0]);
                    acc1[t][p % NJ].y += bf16_float(v[1]);
                    acc1[t][p % NJ].z += bf16_float(v[2]);
                    acc1[t][p % NJ].w += bf16_float(v[3]);
                }
                if (p + 3 < npass) pissue(p + 3);   // into the buffer of pass p - 1
            }
        }
        __syncthreads();   // the input panels reuse the buffers
    }
    {
        const int total = a.K1 / 32;
        const u16x8* wp = (const u16x8*)a.W[0] + (size_t)(wave * NT1) * 64 + lane;
        u16x8 w[Ring<NT1>::R];
        load_panel();
        ring_fill<NT1, NW>(w, wp, total);
        store_panel(0);
        __syncthreads();
        const char* blane = smem + ei * PRS + ((g ^ (SWZ ? (ei & 3) : 0)) << 4);
        const int kx = SWZ ? (ei >> 2) : 0;
        for (int p = 0; p < np; ++p) {
            const bool more = p + 1 < np && !(a.ablate & 4);
            if (more) load_panel();
            gemm_lds<NT1, PRS, VAR, NW, NJ>(acc1, w, wp, p * CPP, total, blane + (p & 1) * PANEL, kx, CPP, a.ablate);
            if (more) store_panel((p + 1) & 1);
            if (!(a.ablate & 8)) __syncthreads();
        }
    }
    dump_pre<NT1, NW, NJ>(acc1, a.save_pre[0], a.M, e0, wave, ei, g);
    if constexpr (NT2 == 0) {
        // single-layer launch (heads / encoder tails at latent 512, whose 1024-wide layers chain as separate launches)
        layernorm_act<NT1, ACT_O, NW, NJ>(acc1, a.lnw[0] + wave * NT1 * 16 + 4 * g, a.lnb[0] + wave * NT1 * 16 + 4 * g,
                                          a.act[0], a.eps, red, wave, ei, g, a.ablate);
        store_out<NT1, NW, NJ>(acc1, a, e0, wave, ei, g);
        return;
    }
    layernorm_act<NT1, ACT_H, NW, NJ>(acc1, a.lnw[0] + wave * NT1 * 16 + 4 * g, a.lnb[0] + wave * NT1 * 16 + 4 * g, a.act[0],
                              a.eps, red, wave, ei, g, a.ablate);
    // (the barrier inside layernorm_act also means: every wave is done reading the panels)
    write_hidden<NT1, HRS, NJ, SWZ>(acc1, smem + ei * HRS, 2 * wave * NT1, ei, g);
    __syncthreads();

    // ---------------- layer 2: B = hidden rows
    constexpr int N2 = NT2 > 0 ? NT2 : 2;   // (NT2 == 0 returned above; N2 only keeps the dead code well-formed)
    f32x4 acc2[N2][NJ];
    init_bias<N2, NJ>(acc2, a.b[1] + wave * N2 * 16 + 4 * g);
    {
        const u16x8* wp = (const u16x8*)a.W[1] + (size_t)(wave * N2) * 64 + lane;
        u16x8 w[Ring<N2>::R];
        ring_fill<N2, NW>(w, wp, NC2);
        gemm_lds<N2, HRS, VAR, NW, NJ>(acc2, w, wp, 0, NC2, smem + ei * HRS + ((g ^ (SWZ ? (ei & 3) : 0)) << 4), SWZ ? (ei >> 2) : 0, NC2, a.ablate);
    }
    dump_pre<N2, NW, NJ>(acc2, a.save_pre[1], a.M, e0, wave, ei, g);
    if constexpr (NT3 == 0) {
        layernorm_act<N2, ACT_O, NW, NJ>(acc2, a.lnw[1] + wave * N2 * 16 + 4 * g, a.lnb[1] + wave * N2 * 16 + 4 * g,
                                  a.act[1], a.eps, red, wave, ei, g, a.ablate);
        store_out<N2, NW, NJ>(acc2, a, e0, wave, ei, g);
    } else {
        layernorm_act<N2, ACT_H, NW, NJ>(acc2, a.lnw[1] + wave * N2 * 16 + 4 * g, a.lnb[1] + wave * N2 * 16 + 4 * g,
                                  a.act[1], a.eps, red, wave, ei, g, a.ablate);
        // in place: the barrier inside layernorm_act came after every wave's layer-2 reads
        write_hidden<N2, HRS, NJ, SWZ>(acc2, smem + ei * HRS, 2 * wave * N2, ei, g);
        __syncthreads();
        f32x4 acc3[NT3][NJ];
        init_bias<NT3, NJ>(acc3, a.b[2] + wave * NT3 * 16 + 4 * g);
        {
            const u16x8* wp = (const u16x8*)a.W[2] + (size_t)(wave * NT3) * 64 + lane;
            u16x8 w[Ring<NT3>::R];
            ring_fill<NT3, NW>(w, wp, NC2);
            gemm_lds<NT3, HRS, VAR, NW, NJ>(acc3, w, wp, 0, NC2, smem + ei * HRS + ((g ^ (SWZ ? (ei & 3) : 0)) << 4), SWZ ? (ei >> 2) : 0, NC2, a.ablate);
        }
        dump_pre<NT3, NW, NJ>(acc3, a.save_pre[2], a.M, e0, wave, ei, g);
        layernorm_act<NT3, ACT_O, NW, NJ>(acc3, a.lnw[2] + wave * NT3 * 16 + 4 * g, a.lnb[2] + wave * NT3 * 16 + 4 * g,
                                  a.act[2], a.eps, red, wave, ei, g, a.ablate);
        store_out<NT3, NW, NJ>(acc3, a, e0, wave, ei, g);
    }
}

template <int NT1, int NT2, int NT3, int MINB, int ACT_H, int ACT_O, int VAR, int NW, int NJ>
static int launch_act(const Args& a, hipStream_t s) {
    constexpr int PK = 128;
    constexpr int TE = 16 * NJ;
    constexpr bool SWZ = NT1 * NW >= 32;
    constexpr int HRS = NT1 * NW * 32 + (SWZ ? 0 : 16);
    constexpr int PRS = PK * 2 + (SWZ ? 0 : 16);
    const size_t lds_bytes = (size_t)cmax(TE * HRS, 2 * TE * PRS) + NW * TE * 2 * sizeof(float) + 2 * TE * sizeof(int32_t);
    const unsigned grid = (unsigned)ceil_div(a.M, TE);
    auto kern = k_mlp_bf16_split<NT1, NT2, NT3, PK, ACT_H, ACT_O, MINB, VAR, NW, NJ>;
    if (lds_bytes > 64 * 1024) {
        HGNN_CHECK_HIP(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize,
                                           (int)lds_bytes));
    }
    kern<<<grid, NW * 64, lds_bytes, s>>>(a);
    HGNN_CHECK_HIP(hipGetLastError());
    return HGNN_OK;
}

template <int NT1, int NT2, int NT3, int MINB, int NW = 4, int NJ = 4>
static int launch(const Args& a, hipStream_t s) {
    const int n = NT3 == 0 ? 2 : 3;
    bool hidden_gelu = true;
    for (int l = 0; l + 1 < n; ++l) hidden_gelu = hidden_gelu && a.act[l] == HGNN_ACT_GELU;
    const int out = a.act[n - 1];
    // burst schedule where a chunk is 32 MFMAs and two waves share a SIMD (see gemm_lds)
    constexpr int AUTO = (NT1 == 8 && NW == 4 && MINB == 2) ? 2 : 0;
    const int var = g_opt_mlp_split_variant < 0 ? AUTO : g_opt_mlp_split_variant;
    if (var == 2) {
        if (hidden_gelu && out == HGNN_ACT_TANH) return launch_act<NT1, NT2, NT3, MINB, HGNN_ACT_GELU, HGNN_ACT_TANH, 2, NW, NJ>(a, s);
        if (hidden_gelu && out == HGNN_ACT_GELU) return launch_act<NT1, NT2, NT3, MINB, HGNN_ACT_GELU, HGNN_ACT_GELU, 2, NW, NJ>(a, s);
        return launch_act<NT1, NT2, NT3, MINB, -1, -1, 2, NW, NJ>(a, s);
    }
    if (hidden_gelu && out == HGNN_ACT_TANH) return launch_act<NT1, NT2, NT3, MINB, HGNN_ACT_GELU, HGNN_ACT_TANH, 0, NW, NJ>(a, s);
    if (hidden_gelu && out == HGNN_ACT_GELU) return launch_act<NT1, NT2, NT3, MINB, HGNN_ACT_GELU, HGNN_ACT_GELU, 0, NW, NJ>(a, s);
    return launch_act<NT1, NT2, NT3, MINB, -1, -1, 0, NW, NJ>(a, s);
}

// one Linear -> LayerNorm -> act (+ skip) layer (n_layers = 1): N = NW * NT * 16 in {256, 512, 1024}
template <int NT, int MINB, int NW>
static int launch_single(const Args& a, hipStream_t s) {
    constexpr int AUTO = (NT == 8 && NW == 4 && MINB == 2) ? 2 : 0;
    const int var = g_opt_mlp_split_variant < 0 ? AUTO : g_opt_mlp_split_variant;
    const int act = a.act[0];
#define HGNN_SINGLE(V)                                                                                          \
    do {                                                                                                        \
        if (act == HGNN_ACT_GELU) return launch_act<NT, 0, 0, MINB, HGNN_ACT_GELU, HGNN_ACT_GELU, V, NW, 4>(a, s); \
        if (act == HGNN_ACT_TANH) return launch_act<NT, 0, 0, MINB, HGNN_ACT_TANH, HGNN_ACT_TANH, V, NW, 4>(a, s); \
        return launch_act<NT, 0, 0, MINB, -1, -1, V, NW, 4>(a, s);                                               \
    } while (0)
    if (var == 2) HGNN_SINGLE(2);
    HGNN_SINGLE(0);
#undef HGNN_SINGLE
}

}  // namespace fs
}  // namespace hgnn

using namespace hgnn;

extern "C" int hgnn_mlp_supported_bf16_split(const hgnn_mlp_desc* d) {
    if (d == nullptr) return 0;
    if (d->n_seg < 1 || d->n_seg > 3 || d->n_layers < 1 || d->n_layers > 3) return 0;
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
    const int h = d->width[1];
    const int o = d->width[n];
    if (n == 1) return (o == 256 || o == 512 || o == 1024) && d->n_pre == 0 ? 1 : 0;   // single layers (chains)
    if (n == 3 && d->width[2] != h) return 0;
    if (h != 2 * o) return 0;
    return (o == 128 || o == 256 || o == 512) ? 1 : 0;
}

extern "C" int hgnn_mlp_forward_bf16_split(const hgnn_mlp_desc* d, void* out, hgnn_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    HGNN_REQUIRE(d != nullptr && out != nullptr, "hgnn_mlp_forward_bf16_split: NULL argument");
    if (!hgnn_mlp_supported_bf16_split(d)) {
        set_error("hgnn_mlp_forward_bf16_split: unsupported shape (K -> 2L (-> 2L) -> L, LayerNorm on every layer, "
                  "L in {128,256,512}, every segment a multiple of 128 wide)");
        return HGNN_ERR_UNSUPPORTED;
    }
    if (d->M == 0) return HGNN_OK;
    fs::Args a;
    for (int s = 0; s < 3; ++s) {
        a.seg_table[s] = s < d->n_seg ? (const unsigned short*)d->seg_table[s] : nullptr;
        a.seg_index[s] = s < d->n_seg ? d->seg_index[s] : nullptr;
        a.seg_width[s] = s < d->n_seg ? d->seg_width[s] : 0;
        if (s < d->n_seg) {
            HGNN_REQUIRE(a.seg_table[s] != nullptr && (uintptr_t)a.seg_table[s] % 16 == 0,
                         "hgnn_mlp_forward_bf16_split: segment table %d is NULL or not 16-byte aligned", s);
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
                         "hgnn_mlp_forward_bf16_split: layer %d parameters must be 16-byte aligned", l);
        }
    }
    a.eps = d->ln_eps;
    a.skip = (const unsigned short*)d->skip;
    a.out = (unsigned short*)out;
    a.M = d->M;
    a.n_pre = d->n_pre;
    for (int s = 0; s < 2; ++s) {
        a.pre_table[s] = s < d->n_pre ? (const unsigned short*)d->pre_table[s] : nullptr;
        a.pre_index[s] = s < d->n_pre ? d->pre_index[s] : nullptr;
        HGNN_REQUIRE((uintptr_t)a.pre_table[s] % 16 == 0, "hgnn_mlp_forward_bf16_split: pre_table[%d] must be 16-byte aligned", s);
    }
    a.ablate = g_opt_mlp_ablate;
    for (int l = 0; l < 3; ++l) {
        a.save_pre[l] = l < d->n_layers ? (unsigned short*)d->save_pre[l] : nullptr;   // bf16 rows here
        HGNN_REQUIRE((uintptr_t)a.save_pre[l] % 8 == 0, "hgnn_mlp_forward_bf16_split: save_pre[%d] must be 8-byte aligned", l);
    }
    HGNN_REQUIRE((uintptr_t)out % 8 == 0 && (uintptr_t)a.skip % 8 == 0,
                 "hgnn_mlp_forward_bf16_split: out/skip must be 8-byte aligned");
    const int o = d->width[d->n_layers];
    if (d->n_layers == 1) {
        if (o == 256) return fs::launch_single<4, 2, 4>(a, stream);
        if (o == 512) return fs::launch_single<8, 2, 4>(a, stream);
        return fs::launch_single<8, 1, 8>(a, stream);
    }
    if (d->n_layers == 2) {
        switch (o) {
            case 128: return fs::launch<4, 2, 0, 2>(a, stream);
            case 256:
                // (round-2 null results, removed: 8 waves x 128 rows in this kernel, 3.11 vs 2.7 ms;
                // 4 waves x 128 rows -- one weight fragment feeds 8 MFMAs, half
                // the L1 weight traffic per FLOP, 256 accumulators per lane, one wave per SIMD -- spills 219 registers
                // under hipcc and runs at 4.35 ms vs 2.9 ms)
                return fs::launch<8, 4, 0, 2>(a, stream);
            case 512:
                // 8 waves share the 64 rows (2 per SIMD instead of 1): 8.6 vs 9.6 ms at M = 2M
                return fs::launch<8, 4, 0, 1, 8, 4>(a, stream);
        }
    } else {
        switch (o) {
            case 128: return fs::launch<4, 4, 2, 2>(a, stream);
            case 256: return fs::launch<8, 8, 4, 2>(a, stream);
            case 512:
                return fs::launch<8, 8, 4, 1, 8, 4>(a, stream);
        }
    }
    set_error("hgnn_mlp_forward_bf16_split: no instantiation");
    return HGNN_ERR_UNSUPPORTED;
}
