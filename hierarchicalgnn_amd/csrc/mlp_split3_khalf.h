// The K-HALF tile kernel of mlp_split3_f32.hip, included once per tile configuration (HGNN_KH_NS: namespace, HGNN_KH_NW:
// waves per workgroup, HGNN_KH_NJ: 16-row tiles per workgroup tile).  No include guard on purpose.
namespace HGNN_KH_NS {
constexpr int WNJ = HGNN_KH_NJ;
constexpr int WTE = 16 * WNJ;
constexpr int WPANEL = WTE * PRS;
constexpr int WNW = HGNN_KH_NW, WNTH = 32 / WNW, WNTO = 16 / WNW;   // a wave: 512 / WNW hidden, 256 / WNW output features of every row
constexpr int WNHJ = WNJ / 4;            // groups of four row tiles
constexpr int WCW = WNTH / 4;            // 32-feature chunks of a wave in one K-half of the hidden rows
constexpr int WBPC = WNW == 8 ? 1 : 2;   // workgroups per CU (LDS)
constexpr int WRD = WNW == 8 ? 4 : 2;    // passes in a wave's projected-row ring
constexpr int WAHEAD = WRD == 4 ? 3 : 2; // passes requested ahead of the one being added
constexpr int WH = WNTH * WNW * 16;      // 512
constexpr int WO = WNTO * WNW * 16;      // 256
constexpr int WHH = WH / 2;              // hidden features per K-half
constexpr int WHRS = WHH * 2 + 16;       // row stride of one half plane (padded)
constexpr int WPLB = WTE * WHRS;         // bytes of one half plane
constexpr int WFWB = WNTH * 16 * 4;      // bytes of a wave's slice of a projected row (256)
constexpr int WRING = WRD * 16 * WFWB;   // a wave's ring: WRD passes of 16 rows
constexpr int WREGION = cmax(cmax(2 * WPLB, 4 * WPANEL), WNW * WRING);
constexpr int WLDS = WREGION + WNW * WTE * 2 * 4 + 2 * 5 * WTE * 4;

// acc += x_hi.W_hi + x_mid.W_hi + x_hi.W_mid over n k-chunks; chunk c's weight fragments are virtual chunks VMAP(c),
// VMAP(c) + 1 of the stream.  ONE register set per operand, refilled IN PLACE: a weight tile's fragments are re-requested
// (for the next chunk) right after the 24 MFMAs that consume them -- seven tiles = 2.7k cycles ahead of their next use,
// an L2 round trip with room to spare -- and an LDS operand right after its last MFMA of the chunk (in the last weight
// tile; 8 row tiles later it is needed again: LDS latency).  A full scheduling barrier after every group keeps hipcc
// from sinking the refills to their first use.  Operand registers: 64 + 64 beside the 128 / 256 accumulators.
template <int NT, int RS, typename VMAP>
__device__ __forceinline__ void gemm3w(f32x4 (&acc)[NT][WNJ], const u16x8* __restrict__ wp, VMAP vmap, const char* b_hi,
                                       const char* b_mid, int n) {
    // ONE register set per operand, refilled IN PLACE: the LDS operands of four row tiles (a half chunk) right after
    // their last MFMA of the half (in the last weight tile), a weight tile's fragments right after the second half has
    // consumed them.  64 operand registers beside the accumulators (which hipcc keeps in AGPRs).
    constexpr int VS = WNW * NT * 64;
    u16x8 wh[NT], wm[NT], bh[4], bm[4];
    {
        const u16x8* p = wp + (size_t)vmap(0) * VS;
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            wh[t] = p[t * 64];
            wm[t] = p[VS + t * 64];
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            bh[j] = *(const u16x8*)(b_hi + j * 16 * RS);
            bm[j] = *(const u16x8*)(b_mid + j * 16 * RS);
        }
    }
    for (int c = 0; c < n; ++c) {
        const int cn = c + 1 < n ? c + 1 : n - 1;
        const u16x8* pn = wp + (size_t)vmap(cn) * VS;
#pragma unroll
        for (int half = 0; half < WNHJ; ++half) {
            // after this group of four row tiles: the next group of this chunk, or the first group of the next chunk
            const int nc = half + 1 < WNHJ ? c : cn, nh = (half + 1) % WNHJ;
#pragma unroll
            for (int t = 0; t < NT; ++t) {
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    f32x4& d = acc[t][half * 4 + j];
                    d = __builtin_amdgcn_mfma_f32_16x16x32_bf16(as_bf16(wm[t]), as_bf16(bh[j]), d, 0, 0, 0);
                    d = __builtin_amdgcn_mfma_f32_16x16x32_bf16(as_bf16(wh[t]), as_bf16(bm[j]), d, 0, 0, 0);
                    d = __builtin_amdgcn_mfma_f32_16x16x32_bf16(as_bf16(wh[t]), as_bf16(bh[j]), d, 0, 0, 0);
                    if (t == NT - 1) {
                        bh[j] = *(const u16x8*)(b_hi + (nh * 4 + j) * 16 * RS + nc * 64);
                        bm[j] = *(const u16x8*)(b_mid + (nh * 4 + j) * 16 * RS + nc * 64);
                        __builtin_amdgcn_sched_barrier(0);
                    }
                }
                if (half == WNHJ - 1) {
                    wh[t] = pn[t * 64];
                    wm[t] = pn[VS + t * 64];
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    }
}

template <int NT, int NWV, int ACT>
__device__ __forceinline__ void ln_stats(const f32x4 (&acc)[NT][WNJ], float eps, float* red, int wave, int ei, int g,
                                         float (&rstd)[WNJ], float (&shift)[WNJ]) {
    constexpr float inv_n = 1.0f / (float)(NWV * NT * 16);
#pragma unroll
    for (int j = 0; j < WNJ; ++j) {
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
            *(f32x2*)(red + (wave * WTE + j * 16 + ei) * 2) = sq;
        }
    }
    __syncthreads();
    // Two stages.  Every lane summing its 8 rows over the 8 waves itself is 64 LDS reads in flight (128 registers: hipcc
    // spilled the prefetched skip rows for them) and 8 divisions + square roots per lane.  Instead 4 threads reduce ONE row
    // (2 waves' partials each, combined by lane shuffles), the first of them writes (rstd, shift) over the row's wave-0
    // partial -- which only these 4 threads read -- and after a second barrier every lane reads its 8 rows' pairs.
    static_assert(NWV * 64 == 4 * WTE && NWV % 4 == 0, "4 threads per row");
    {
        const int row = (int)threadIdx.x >> 2, part = (int)threadIdx.x & 3;
        float s = 0.f, q = 0.f;
#pragma unroll
        for (int w = 0; w < NWV / 4; ++w) {
            const f32x2 sq = *(const f32x2*)(red + ((part * (NWV / 4) + w) * WTE + row) * 2);
            s += sq.x;
            q += sq.y;
        }
        s += __shfl_xor(s, 1);
        q += __shfl_xor(q, 1);
        s += __shfl_xor(s, 2);
        q += __shfl_xor(q, 2);
        const float mean = s * inv_n;
        const float var = fmaxf(fmaf(-mean, mean, q * inv_n), 0.f);
        f32x2 rs;
        rs.x = 1.0f / sqrtf(var + eps);
        rs.y = -mean * rs.x;
        if (part == 0) *(f32x2*)(red + row * 2) = rs;
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < WNJ; ++j) {
        const f32x2 rs = *(const f32x2*)(red + (j * 16 + ei) * 2);
        rstd[j] = rs.x;
        shift[j] = rs.y;
    }
}

// LayerNorm (given the row statistics) + activation of ONE accumulator tile value (4 features of one row)
template <int ACT>
__device__ __forceinline__ f32x4 ln_act4(f32x4 v, float rs, float sh, const f32x4 w4, const f32x4 b4, int act_rt) {
    const int act = ACT >= 0 ? ACT : act_rt;
    v.x = act_apply(fmaf(fmaf(v.x, rs, sh), w4.x, b4.x), act);
    v.y = act_apply(fmaf(fmaf(v.y, rs, sh), w4.y, b4.y), act);
    v.z = act_apply(fmaf(fmaf(v.z, rs, sh), w4.z, b4.z), act);
    v.w = act_apply(fmaf(fmaf(v.w, rs, sh), w4.w, b4.w), act);
    return v;
}

// hidden tiles [T0, T0 + WNTH / 2) of this wave: accumulators -> LayerNorm -> activation -> (hi, mid) bf16 planes of the K-half in
// LDS, in one pass (the accumulators live in AGPRs: every value is read out once and never written back)
template <int T0, int ACT>
__device__ __forceinline__ void act_write_half(const f32x4 (&acc)[WNTH][WNJ], const float* __restrict__ lnw,
                                               const float* __restrict__ lnb, int act_rt, const float (&rstd)[WNJ],
                                               const float (&shift)[WNJ], char* lane0) {
#pragma unroll
    for (int t = 0; t < WNTH / 2; ++t) {
        const f32x4 w4 = *(const f32x4*)(lnw + (T0 + t) * 16);
        const f32x4 b4 = *(const f32x4*)(lnb + (T0 + t) * 16);
#pragma unroll
        for (int j = 0; j < WNJ; ++j) {
            const f32x4 v = ln_act4<ACT>(acc[T0 + t][j], rstd[j], shift[j], w4, b4, act_rt);
            u16x4 h, m;
            split4(v, h, m);
            *(u16x4*)(lane0 + j * 16 * WHRS + t * 32) = h;
            *(u16x4*)(lane0 + WPLB + j * 16 * WHRS + t * 32) = m;
        }
    }
}

template <int ACT_H, int ACT_O>
__global__ __launch_bounds__(WNW * 64, 2) void k_mlp_f32_split3_khalf(const Args a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int NTHR = WNW * 64;
    float* red = (float*)(smem + WREGION);                // [WNW][WTE][sum, sumsq]
    int32_t* tix = (int32_t*)(red + WNW * WTE * 2);       // [parity][3 segments + 2 pre-projected][WTE]
    int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
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
    const long long n_tiles = (a.M + WTE - 1) / WTE;
    constexpr int NIX = (5 * WTE + NTHR - 1) / NTHR;   // 3
    auto fetch_index = [&](long long tile, int k) {
        int r = 0;
        const int slot = tid + k * NTHR;
        if (slot < 5 * WTE) {
            const int which = slot / WTE;
            long long e = tile * WTE + (slot % WTE);
            if (e >= a.M) e = a.M - 1;
            r = (int)e;
            const int32_t* ix = which < 3 ? (which < a.n_seg ? a.seg_index[which] : nullptr)
                                          : (which - 3 < a.n_pre ? a.pre_index[which - 3] : nullptr);
            if (ix != nullptr) r = ix[e];
            if (which >= 3 && which - 3 >= a.n_pre) r = 0;
        }
        return r < 0 ? 0 : r;
    };
    constexpr int RPP = NTHR / LPR;    // 8 rows per pass
    constexpr int NP = WTE / RPP;      // 16 passes per panel
    const int np = a.K1 / PK;
    const int p1 = a.seg_width[0] / PK;
    const int p2 = p1 + (a.n_seg > 1 ? a.seg_width[1] / PK : np);
    // projected rows: wave-private DMA ring, four passes of 16 rows x this wave's 512-byte slice
    constexpr int RPI = 1024 / WFWB;   // 2 rows per DMA instruction
    constexpr int IPW = 16 / RPI;      // 8 DMA instructions per pass
    constexpr int PPR = WFWB / 16;     // 32 pieces per slice row
    const int npass = a.n_pre * WNJ;
    char* pring = smem + wave * WRING;
    auto pissue = [&](int p, const int32_t* ti) {
        const int sgm = p / WNJ, j = p % WNJ;
#pragma unroll
        for (int i = 0; i < IPW; ++i) {
            const int row = i * RPI + lane / PPR, pc = lane % PPR;
            const int r = ti[(3 + sgm) * WTE + 16 * j + row];
            const char* src = (const char*)(a.pre_table[sgm] + (size_t)r * WH + wave * WNTH * 16) + ((pc ^ row) << 4);
            dma_piece(src, lds_addr_of((float*)(pring + (p % WRD) * 16 * WFWB)) + (unsigned)i * 1024u);
        }
    };

    long long tile = blockIdx.x;
    if (tile >= n_tiles) return;
    if (a.stagger > 0) {   // persistent workgroups with equal work stay in phase: de-phase their HBM bursts once, here
        const unsigned long long t0 = __builtin_amdgcn_s_memtime();
        const unsigned long long wait = (unsigned long long)(blockIdx.x % 16) * (unsigned)a.stagger;
        while (__builtin_amdgcn_s_memtime() - t0 < wait) __builtin_amdgcn_s_sleep(32);
    }
#pragma unroll
    for (int k = 0; k < NIX; ++k) {
        const int r = fetch_index(tile, k);
        if (tid + k * NTHR < 5 * WTE) tix[tid + k * NTHR] = r;
    }
    __syncthreads();
    for (int it = 0; tile < n_tiles; tile += gridDim.x, ++it) {
        const int32_t* ti = tix + (it & 1) * 5 * WTE;
        int32_t* ti_next = tix + ((it + 1) & 1) * 5 * WTE;
        const bool has_next = tile + gridDim.x < n_tiles;
        const long long e0 = tile * WTE;
        refresh();
        HGNN_STAMPW(0);
        // ---- input panels (two k-panels per group, both resident)
        int pl = 0;              // panels loaded so far
        auto panel_src = [&](int i) -> const float* {
            const int row = i * RPP + prow;
            if (pl < p1) return a.seg_table[0] + (size_t)ti[row] * (size_t)a.seg_width[0] + (size_t)pl * PK + pcol * 4;
            if (pl < p2) return a.seg_table[1] + (size_t)ti[WTE + row] * (size_t)a.seg_width[1] + (size_t)(pl - p1) * PK + pcol * 4;
            return a.seg_table[2] + (size_t)ti[2 * WTE + row] * (size_t)a.seg_width[2] + (size_t)(pl - p2) * PK + pcol * 4;
        };
        // (AGPRs hold the 256 accumulators; everything the vector ALU touches shares the 256 architectural VGPRs, so
        // staging is kept small: one panel = 16 x 16 bytes per lane at a time)
        f32x4 st[NP];
        auto load_panel = [&]() {
#pragma unroll
            for (int i = 0; i < NP; ++i) st[i] = *(const f32x4*)panel_src(i);
            ++pl;
        };
        auto store_panel = [&](int buf) {   // registers -> (hi, mid) planes in buffers 2 buf, 2 buf + 1
#pragma unroll
            for (int i = 0; i < NP; ++i) {
                u16x4 h, m;
                split4(st[i], h, m);
                char* dst = smem + (2 * buf) * WPANEL + (i * RPP + prow) * PRS + pcol * 8;
                *(u16x4*)dst = h;
                *(u16x4*)(dst + WPANEL) = m;
            }
        };

        // ---------------- layer 1
        f32x4 acc1[WNTH][WNJ];
        {
            const float* b = a.b[0] + wave * WNTH * 16 + 4 * g;
#pragma unroll
            for (int t = 0; t < WNTH; ++t) {
                const f32x4 bv = *(const f32x4*)(b + t * 16);
#pragma unroll
                for (int j = 0; j < WNJ; ++j) acc1[t][j] = bv;
            }
        }
        const int vtotal = 2 * (a.K1 / 32);
        const u16x8* wp = (const u16x8*)a.W[0] + (size_t)(wave * WNTH) * 64 + lane;
        if (npass > 0) {
            if (it == 0) {
                pissue(0, ti);
                if (npass > 1) pissue(1, ti);
                if (WAHEAD > 2 && npass > 2) pissue(2, ti);
            }
#pragma unroll
            for (int p = 0; p < 2 * WNJ; ++p) {
                if (p < npass) {
                    // conservative counts: at most the two younger passes may still be out (everything else issued since
                    // is younger still or already waited for)
                    if (WAHEAD > 2 && p + 2 < npass) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * IPW) : "memory");
                    else if (p + 1 < npass) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(IPW) : "memory");
                    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    const char* buf = pring + (p % WRD) * 16 * WFWB + ei * WFWB;
#pragma unroll
                    for (int t = 0; t < WNTH; ++t)
                        acc1[t][p % WNJ] += *(const f32x4*)(buf + (((4 * t + g) ^ ei) << 4));
                    if (p + WAHEAD < npass) {
                        // a two-pass ring refills the slot just read: its LDS reads have to be back first
                        if (WAHEAD == WRD) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                        pissue(p + WAHEAD, ti);
                    }
                }
            }
            __syncthreads();
        }
        HGNN_STAMPW(1);
        if constexpr (WNW == 8) {
            {
                const char* blane = smem + ei * PRS + (g << 4);
                load_panel();
                store_panel(0);
                __syncthreads();
                HGNN_STAMPW(2);
                for (int q = 0; q < np; ++q) {
                    const bool more = q + 1 < np;
                    if (more) load_panel();           // the next panel flies under this panel's GEMM
                    const char* bh = blane + (2 * (q & 1)) * WPANEL;
                    gemm3w<WNTH, PRS>(acc1, wp, [&](int c) { int v = 2 * (q * CPP + c); return v + 1 >= vtotal ? vtotal - 2 : v; },
                                      bh, bh + WPANEL, CPP);
                    if (more) store_panel((q + 1) & 1);
                    __syncthreads();
                }
            }
        } else {
            // four waves: 128 + 96 accumulator and operand registers leave no room to carry a staged panel across the GEMM
            // (hipcc spilled it, and scratch traffic retires in order with everything else).  Panels in resident PAIRS
            // instead: one after the other requested and split into LDS, one barrier, two GEMMs; the CU's other workgroup covers the wait.
            const char* blane = smem + ei * PRS + (g << 4);
            for (int q = 0; q < np; q += 2) {
                const bool two = q + 1 < np;
                load_panel();
                store_panel(0);
                if (two) {
                    load_panel();
                    store_panel(1);
                }
                __syncthreads();
                if (q == 0) HGNN_STAMPW(2);
                gemm3w<WNTH, PRS>(acc1, wp, [&](int c) { int v = 2 * (q * CPP + c); return v + 1 >= vtotal ? vtotal - 2 : v; },
                                  blane, blane + WPANEL, CPP);
                if (two)
                    gemm3w<WNTH, PRS>(acc1, wp, [&](int c) { int v = 2 * ((q + 1) * CPP + c); return v + 1 >= vtotal ? vtotal - 2 : v; },
                                      blane + 2 * WPANEL, blane + 3 * WPANEL, CPP);
                if (q + 2 < np) __syncthreads();   // the next pair overwrites the panels
            }
        }
        HGNN_STAMPW(3);
        refresh();   // lane-derived addresses are recomputed per phase, not carried (and spilled) across the GEMMs
        // training: this wave's slice of the pre-LayerNorm rows
        if (a.save_pre[0] != nullptr) {
#pragma unroll
            for (int j = 0; j < WNJ; ++j) {
                const long long e = e0 + j * 16 + ei;
                if (e < a.M) {
                    float* p = a.save_pre[0] + (size_t)e * WH + (size_t)(wave * WNTH * 16 + 4 * g);
#pragma unroll
                    for (int t = 0; t < WNTH; ++t) *(f32x4*)(p + t * 16) = acc1[t][j];
                }
            }
        }
        float rstd[WNJ], shift[WNJ];
        ln_stats<WNTH, WNW, ACT_H>(acc1, a.eps, red, wave, ei, g, rstd, shift);
        HGNN_STAMPW(4);
        // (the barrier inside also means: every wave is done reading the panels)

        // ---------------- output layer, two K-halves of the hidden rows
        constexpr int LO = 1;
        const u16x8* wpo = (const u16x8*)a.W[LO] + (size_t)(wave * WNTO) * 64 + lane;
        const char* hlane = smem + ei * WHRS + (g << 4);
        int r_next[NIX];
        // half 0: tiles 0-1 of every wave = hidden features 64 w + [0, 32) = stream chunk 2 w
        act_write_half<0, ACT_H>(acc1, a.lnw[0] + wave * WNTH * 16 + 4 * g, a.lnb[0] + wave * WNTH * 16 + 4 * g, a.act[0], rstd, shift, smem + ei * WHRS + wave * (WNTH * 16) + (g << 3));
        __syncthreads();
        HGNN_STAMPW(5);
        f32x4 acc2[WNTO][WNJ];
        {
            const float* b = a.b[LO] + wave * WNTO * 16 + 4 * g;
#pragma unroll
            for (int t = 0; t < WNTO; ++t) {
                const f32x4 bv = *(const f32x4*)(b + t * 16);
#pragma unroll
                for (int j = 0; j < WNJ; ++j) acc2[t][j] = bv;
            }
        }
        // LDS slot s (32 features) of a half holds stream chunk 2 s + half
        gemm3w<WNTO, WHRS>(acc2, wpo, [](int s) { return 2 * ((s / WCW) * 2 * WCW + s % WCW); }, hlane, hlane + WPLB, 8);
#pragma unroll
        for (int k = 0; k < NIX; ++k) r_next[k] = has_next ? fetch_index(tile + gridDim.x, k) : 0;
        HGNN_STAMPW(6);
        __syncthreads();                 // every wave is done reading half 0
        // the row statistics are re-read from LDS (still there: the next statistics pass comes after the output GEMM) instead
        // of living in 16 registers across the first half's GEMM -- hipcc spilled them (0.5 GB of scratch writes per launch)
        refresh();
        float rstd_b[WNJ], shift_b[WNJ];
#pragma unroll
        for (int j = 0; j < WNJ; ++j) {
            const f32x2 rs = *(const f32x2*)(red + (j * 16 + ei) * 2);
            rstd_b[j] = rs.x;
            shift_b[j] = rs.y;
        }
        act_write_half<WNTH / 2, ACT_H>(acc1, a.lnw[0] + wave * WNTH * 16 + 4 * g, a.lnb[0] + wave * WNTH * 16 + 4 * g, a.act[0], rstd_b, shift_b, smem + ei * WHRS + wave * (WNTH * 16) + (g << 3));
        __syncthreads();
        HGNN_STAMPW(7);
        gemm3w<WNTO, WHRS>(acc2, wpo, [](int s) { return 2 * ((s / WCW) * 2 * WCW + WCW + s % WCW); }, hlane, hlane + WPLB, 8);
        HGNN_STAMPW(8);
        refresh();
#pragma unroll
        for (int k = 0; k < NIX; ++k)
            if (has_next && tid + k * NTHR < 5 * WTE) ti_next[tid + k * NTHR] = r_next[k];
        if (a.save_pre[LO] != nullptr) {
#pragma unroll
            for (int j = 0; j < WNJ; ++j) {
                const long long e = e0 + j * 16 + ei;
                if (e < a.M) {
                    float* p = a.save_pre[LO] + (size_t)e * WO + (size_t)(wave * WNTO * 16 + 4 * g);
#pragma unroll
                    for (int t = 0; t < WNTO; ++t) *(f32x4*)(p + t * 16) = acc2[t][j];
                }
            }
        }
        // skip rows BEFORE the next tile's DMAs (queued behind them they would wait for the projected rows) and before the
        // statistics barrier, so that their latency hides under it
        auto row_off = [&](int j) {
            long long e = e0 + j * 16 + ei;
            if (e >= a.M) e = a.M - 1;
            return (size_t)e * WO + (size_t)(wave * WNTO * 16 + 4 * g);
        };
        // requested here, consumed after the statistics (two barriers): the LayerNorm parameters of this wave's features and
        // the first SKD skip row tiles
        f32x4 w2[WNTO], b2[WNTO];
#pragma unroll
        for (int t = 0; t < WNTO; ++t) {
            w2[t] = *(const f32x4*)(a.lnw[LO] + wave * WNTO * 16 + 4 * g + t * 16);
            b2[t] = *(const f32x4*)(a.lnb[LO] + wave * WNTO * 16 + 4 * g + t * 16);
        }
        // a ring of SKD row tiles (all 8 = 64 registers were spilled by hipcc straight after loading, and every reload then
        // queued behind the row stores: vmcnt retires in order, scratch included)
        constexpr int SKD = WNJ >= 8 ? 4 : 2;
        f32x4 sk[SKD][WNTO] = {};   // (zeros: an undefined no-skip value costs hipcc 8 scratch slots for the phi)
        if (a.skip != nullptr) {
#pragma unroll
            for (int j = 0; j < SKD; ++j)
#pragma unroll
                for (int t = 0; t < WNTO; ++t) sk[j][t] = *(const f32x4*)(a.skip + row_off(j) + t * 16);
        }
        float rstd2[WNJ], shift2[WNJ];
        ln_stats<WNTO, WNW, ACT_O>(acc2, a.eps, red, wave, ei, g, rstd2, shift2);
        // (the barriers inside: every wave is past the hidden planes, the next tile's indices are visible)
        if (has_next && npass > 0) {
            pissue(0, ti_next);
            if (npass > 1) pissue(1, ti_next);
            if (WAHEAD > 2 && npass > 2) pissue(2, ti_next);
        }
        HGNN_STAMPW(9);
        // (the barrier inside: every wave is past the hidden planes, the next tile's indices are visible)
        // output rows: accumulators -> LayerNorm -> activation -> + skip -> HBM in one pass; skip rows one row tile ahead
        // of their use (16 registers, not 128: the vector ALU's 256 VGPRs are shared with everything else)
#pragma unroll
        for (int j = 0; j < WNJ; ++j) {
            const bool ok = e0 + j * 16 + ei < a.M;
            const size_t o = row_off(j);
#pragma unroll
            for (int t = 0; t < WNTO; ++t) {
                f32x4 v = ln_act4<ACT_O>(acc2[t][j], rstd2[j], shift2[j], w2[t], b2[t], a.act[LO]);
                if (a.skip != nullptr) v += sk[j % SKD][t];
                if (ok) *(f32x4*)(a.out + o + t * 16) = v;
            }
            if (a.skip != nullptr && j + SKD < WNJ) {
#pragma unroll
                for (int t = 0; t < WNTO; ++t) sk[j % SKD][t] = *(const f32x4*)(a.skip + row_off(j + SKD) + t * 16);
            }
        }
        HGNN_STAMPW(10);
    }
}

static int g_cus_w = 0;

template <int ACT_H, int ACT_O>
static int launch_tile_act(const Args& a, hipStream_t s) {
    if (g_cus_w == 0) {
        int dev = 0;
        HGNN_CHECK_HIP(hipGetDevice(&dev));
        HGNN_CHECK_HIP(hipDeviceGetAttribute(&g_cus_w, hipDeviceAttributeMultiprocessorCount, dev));
        if (g_cus_w <= 0) g_cus_w = 256;
    }
    const long long n_tiles = ceil_div(a.M, WTE);
    const long long resident = (long long)g_cus_w * (g_opt_split3_one_wg ? 1 : WBPC);
    const unsigned grid = (unsigned)(n_tiles < resident ? n_tiles : resident);
    auto kern = k_mlp_f32_split3_khalf<ACT_H, ACT_O>;
    HGNN_CHECK_HIP(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)WLDS));
    kern<<<grid, WNW * 64, WLDS, s>>>(a);
    HGNN_CHECK_HIP(hipGetLastError());
    return HGNN_OK;
}

static int launch_tile(const Args& a, hipStream_t s) {
    if (a.act[0] == HGNN_ACT_GELU && a.act[1] == HGNN_ACT_TANH) return launch_tile_act<HGNN_ACT_GELU, HGNN_ACT_TANH>(a, s);
    if (a.act[0] == HGNN_ACT_GELU && a.act[1] == HGNN_ACT_GELU) return launch_tile_act<HGNN_ACT_GELU, HGNN_ACT_GELU>(a, s);
    return launch_tile_act<-1, -1>(a, s);
}
}  // namespace HGNN_KH_NS
