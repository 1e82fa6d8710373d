// HBM-bound row kernels of the message-passing path (gfx950, wave64):
//
//   k_seg_reduce  : destination-sorted segmented reduce = scatter_add without atomics
//                   (K1..K5: Modules/gnn_utils.py:50,124,125,142,143;
//                    BipartiteClassification/Models/HGNN_GMM.py:269)
//   k_gather_rows : out[e] = w[e]*rs[idx[e]]*table[idx[e]]   (K6 and scatter_add backward)
//   k_edge_dot    : out[e] = <A[ai[e]], B[bi[e]]>            (d/dweight of K2..K5)
//
// Layout: a feature row of F floats is read as F/4 float4 "columns".  RL lanes
// (a power of two, <= 64) cover one row with one 16-B load each, so a wave
// covers G = 64/RL rows per load instruction (F=256: one whole 1-KiB row per
// wave instruction, the widest coalesced access the hardware has).  Rows wider
// than 64 float4 use VPL loads per lane.  U independent row loads are issued
// before the first add so that each wave keeps U*G rows in flight.
//
// One wave owns one work item (= one destination, or one chunk of a long
// list): no atomics, every output row is written exactly once with 16-B
// stores, and the summation order is fixed by the plan.
#include "common.h"
#include <cstdlib>
#include <cstring>

namespace hgnn {

typedef float f32x4 __attribute__((ext_vector_type(4)));

extern int g_opt_mlp_ablate;
extern int g_opt_mlp_split_variant;
namespace f3 { extern int g_opt_split3_rows128; extern int g_opt_split3_one_wg; }

static int g_opt_nt_loads = 1;   // non-temporal loads for once-read source rows
static int g_opt_nt_stores = 0;  // non-temporal stores for gather output

__device__ __forceinline__ f32x4 ld4(const float* p, bool nt) {
    return nt ? __builtin_nontemporal_load((const f32x4*)p) : *(const f32x4*)p;
}

// TAG distinguishes the main pass (0), the partial-sum combine pass (1) and the main pass on a
// destination-sorted index (2, identity row ids = streaming reads) in profiles.
template <int RL, int VPL, int U, bool HAS_W, bool HAS_RS, bool NT, int TAG, int WPB, bool XCD>
__global__ __launch_bounds__(WPB * 64) void k_seg_reduce(
    const float* __restrict__ src, int F, int nvec, const int32_t* __restrict__ src_row,
    const int32_t* __restrict__ perm, const float* __restrict__ weight,
    const float* __restrict__ row_scale, const int32_t* __restrict__ wi_begin,
    const int32_t* __restrict__ wi_end, const int32_t* __restrict__ wi_target,
    const int32_t* __restrict__ n_items_ptr, int64_t max_items, float* __restrict__ out,
    float* __restrict__ partial) {
    constexpr int G = 64 / RL;
    const int lane = threadIdx.x & 63;
    int64_t bid = blockIdx.x;
    if (XCD) {
        // blocks are dealt round-robin over the 8 XCDs: give each XCD a contiguous run of
        // work items so that neighbouring lists (which share index cache lines) share an L2.
        const int64_t per_xcd = (gridDim.x + 7) / 8;
        bid = (bid % 8) * per_xcd + bid / 8;
    }
    const int64_t item = bid * WPB + (threadIdx.x >> 6);
    const int n_items = *n_items_ptr;
    if (item >= n_items || item >= max_items) return;
    const int begin = __builtin_amdgcn_readfirstlane(wi_begin[item]);
    const int end = __builtin_amdgcn_readfirstlane(wi_end[item]);
    const int target = __builtin_amdgcn_readfirstlane(wi_target[item]);
    const int g = lane / RL;
    const int c = lane % RL;

    f32x4 acc[VPL];
#pragma unroll
    for (int v = 0; v < VPL; ++v) acc[v] = f32x4{0.f, 0.f, 0.f, 0.f};

    for (int base = begin; base < end; base += 64) {
        const int n = (end - base) < 64 ? (end - base) : 64;
        int my_row = 0;
        float my_w = 1.f;
        if (lane < n) {
            const int p = base + lane;
            my_row = src_row != nullptr ? src_row[p] : p;
            if (HAS_W) my_w = weight[perm != nullptr ? perm[p] : p];
            if (HAS_RS) my_w *= row_scale[my_row];
        }
        for (int j = 0; j < n; j += G * U) {
            f32x4 val[U][VPL];
            float w[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int e = j + u * G + g;
                int r;
                if (G == 1) {
                    r = __builtin_amdgcn_readlane(my_row, (j + u) & 63);
                    w[u] = (HAS_W || HAS_RS)
                               ? __builtin_bit_cast(float, __builtin_amdgcn_readlane(
                                                               __builtin_bit_cast(int, my_w), (j + u) & 63))
                               : 1.f;
                } else {
                    r = __shfl(my_row, e & 63);
                    w[u] = (HAS_W || HAS_RS) ? __shfl(my_w, e & 63) : 1.f;
                }
                const bool ok = e < n;
                const float* rp = src + (size_t)r * (size_t)F;
#pragma unroll
                for (int v = 0; v < VPL; ++v) {
                    const int cv = c + v * 64;
                    if (ok && cv < nvec)
                        val[u][v] = ld4(rp + cv * 4, NT);
                    else
                        val[u][v] = f32x4{0.f, 0.f, 0.f, 0.f};
                }
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
#pragma unroll
                for (int v = 0; v < VPL; ++v) {
                    if (HAS_W || HAS_RS)
                        acc[v] += val[u][v] * w[u];
                    else
                        acc[v] += val[u][v];
                }
            }
        }
    }
    if (G > 1) {
#pragma unroll
        for (int off = RL; off < 64; off <<= 1) {
#pragma unroll
            for (int v = 0; v < VPL; ++v) {
                acc[v].x += __shfl_xor(acc[v].x, off);
                acc[v].y += __shfl_xor(acc[v].y, off);
                acc[v].z += __shfl_xor(acc[v].z, off);
                acc[v].w += __shfl_xor(acc[v].w, off);
            }
        }
    }
    if (g == 0) {
        float* op = target >= 0 ? out + (size_t)target * (size_t)F
                                : partial + (size_t)(~target) * (size_t)F;
#pragma unroll
        for (int v = 0; v < VPL; ++v) {
            const int cv = c + v * 64;
            if (cv < nvec) *(f32x4*)(op + cv * 4) = acc[v];
        }
    }
}

// any F (including F % 4 != 0): lanes stride over single floats, 64 columns per pass.
template <bool HAS_W, bool HAS_RS>
__global__ __launch_bounds__(256) void k_seg_reduce_scalar(
    const float* __restrict__ src, int F, const int32_t* __restrict__ src_row,
    const int32_t* __restrict__ perm, const float* __restrict__ weight,
    const float* __restrict__ row_scale, const int32_t* __restrict__ wi_begin,
    const int32_t* __restrict__ wi_end, const int32_t* __restrict__ wi_target,
    const int32_t* __restrict__ n_items_ptr, int64_t max_items, float* __restrict__ out,
    float* __restrict__ partial) {
    const int lane = threadIdx.x & 63;
    const int64_t item = (int64_t)blockIdx.x * kWavesPerBlock + (threadIdx.x >> 6);
    const int n_items = *n_items_ptr;
    if (item >= n_items || item >= max_items) return;
    const int begin = wi_begin[item], end = wi_end[item], target = wi_target[item];
    float* op = target >= 0 ? out + (size_t)target * (size_t)F : partial + (size_t)(~target) * (size_t)F;
    for (int col0 = 0; col0 < F; col0 += 64) {
        const int col = col0 + lane;
        float acc = 0.f;
        for (int p = begin; p < end; ++p) {
            const int r = src_row != nullptr ? src_row[p] : p;
            float w = 1.f;
            if (HAS_W) w = weight[perm != nullptr ? perm[p] : p];
            if (HAS_RS) w *= row_scale[r];
            if (col < F) acc += w * src[(size_t)r * (size_t)F + col];
        }
        if (col < F) op[col] = acc;
    }
}

template <int RL, int VPL, int U, bool HAS_W, bool HAS_RS>
__global__ __launch_bounds__(256) void k_gather_rows(const float* __restrict__ table, int F, int nvec,
                                                     const int32_t* __restrict__ idx, int64_t M,
                                                     const float* __restrict__ weight,
                                                     const float* __restrict__ row_scale,
                                                     float* __restrict__ out, bool nt_store) {
    constexpr int G = 64 / RL;
    const int lane = threadIdx.x & 63;
    const int g = lane / RL;
    const int c = lane % RL;
    const int64_t n_tiles = (M + 63) / 64;
    for (int64_t tile = (int64_t)blockIdx.x * kWavesPerBlock + (threadIdx.x >> 6); tile < n_tiles;
         tile += (int64_t)gridDim.x * kWavesPerBlock) {
        const int64_t base = tile * 64;
        const int n = (M - base) < 64 ? (int)(M - base) : 64;
        int my_idx = -1;
        float my_w = 1.f;
        if (lane < n) {
            my_idx = idx[base + lane];
            if (HAS_W) my_w = weight[base + lane];
            if (HAS_RS) my_w *= my_idx >= 0 ? row_scale[my_idx] : 0.f;
        }
        for (int j = 0; j < n; j += G * U) {
            f32x4 val[U][VPL];
            float w[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int e = j + u * G + g;
                const int r = __shfl(my_idx, e & 63);
                w[u] = (HAS_W || HAS_RS) ? __shfl(my_w, e & 63) : 1.f;
                const bool ok = (e < n) && (r >= 0);
                const float* rp = table + (size_t)(ok ? r : 0) * (size_t)F;
#pragma unroll
                for (int v = 0; v < VPL; ++v) {
                    const int cv = c + v * 64;
                    if (ok && cv < nvec)
                        val[u][v] = *(const f32x4*)(rp + cv * 4);
                    else
                        val[u][v] = f32x4{0.f, 0.f, 0.f, 0.f};
                }
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int e = j + u * G + g;
                if (e < n) {
                    float* op = out + (size_t)(base + e) * (size_t)F;
#pragma unroll
                    for (int v = 0; v < VPL; ++v) {
                        const int cv = c + v * 64;
                        if (cv < nvec) {
                            f32x4 x = (HAS_W || HAS_RS) ? val[u][v] * w[u] : val[u][v];
                            if (nt_store)
                                __builtin_nontemporal_store(x, (f32x4*)(op + cv * 4));
                            else
                                *(f32x4*)(op + cv * 4) = x;
                        }
                    }
                }
            }
        }
    }
}

// Transpose of k_seg_reduce: walk the plan in destination order, read each table row once and
// write it (scaled) to every row of its list.  out[perm[p],:] = w[perm[p]] * table[dst,:].
template <int RL, int VPL, bool HAS_W, int WPB>
__global__ __launch_bounds__(WPB * 64) void k_spread_rows(
    const float* __restrict__ table, int F, int nvec, const int32_t* __restrict__ perm,
    const float* __restrict__ weight, const int32_t* __restrict__ wi_begin,
    const int32_t* __restrict__ wi_end, const int32_t* __restrict__ wi_dst,
    const int32_t* __restrict__ n_items_ptr, int64_t max_items, float* __restrict__ out, bool nt_store) {
    constexpr int G = 64 / RL;
    const int lane = threadIdx.x & 63;
    const int64_t item = (int64_t)blockIdx.x * WPB + (threadIdx.x >> 6);
    const int n_items = *n_items_ptr;
    if (item >= n_items || item >= max_items) return;
    const int begin = __builtin_amdgcn_readfirstlane(wi_begin[item]);
    const int end = __builtin_amdgcn_readfirstlane(wi_end[item]);
    if (begin >= end) return;
    const int dst = __builtin_amdgcn_readfirstlane(wi_dst[item]);
    const int g = lane / RL;
    const int c = lane % RL;
    f32x4 row[VPL];
    const float* rp = table + (size_t)dst * (size_t)F;
#pragma unroll
    for (int v = 0; v < VPL; ++v) {
        const int cv = c + v * 64;
        row[v] = cv < nvec ? *(const f32x4*)(rp + cv * 4) : f32x4{0.f, 0.f, 0.f, 0.f};
    }
    for (int base = begin; base < end; base += 64) {
        const int n = (end - base) < 64 ? (end - base) : 64;
        int my_e = 0;
        float my_w = 1.f;
        if (lane < n) {
            my_e = perm[base + lane];
            if (HAS_W) my_w = weight[my_e];
        }
        for (int j = 0; j < n; j += G) {
            const int k = j + g;
            const int e = __shfl(my_e, k & 63);
            const float w = HAS_W ? __shfl(my_w, k & 63) : 1.f;
            if (k < n) {
                float* op = out + (size_t)e * (size_t)F;
#pragma unroll
                for (int v = 0; v < VPL; ++v) {
                    const int cv = c + v * 64;
                    if (cv < nvec) {
                        const f32x4 x = HAS_W ? row[v] * w : row[v];
                        if (nt_store)
                            __builtin_nontemporal_store(x, (f32x4*)(op + cv * 4));
                        else
                            *(f32x4*)(op + cv * 4) = x;
                    }
                }
            }
        }
    }
}

template <bool HAS_W>
__global__ __launch_bounds__(256) void k_spread_rows_scalar(
    const float* __restrict__ table, int F, const int32_t* __restrict__ perm,
    const float* __restrict__ weight, const int32_t* __restrict__ wi_begin,
    const int32_t* __restrict__ wi_end, const int32_t* __restrict__ wi_dst,
    const int32_t* __restrict__ n_items_ptr, int64_t max_items, float* __restrict__ out) {
    const int lane = threadIdx.x & 63;
    const int64_t item = (int64_t)blockIdx.x * kWavesPerBlock + (threadIdx.x >> 6);
    const int n_items = *n_items_ptr;
    if (item >= n_items || item >= max_items) return;
    const int begin = wi_begin[item], end = wi_end[item], dst = wi_dst[item];
    for (int p = begin; p < end; ++p) {
        const int e = perm[p];
        const float w = HAS_W ? weight[e] : 1.f;
        for (int col = lane; col < F; col += 64)
            out[(size_t)e * F + col] = w * table[(size_t)dst * F + col];
    }
}

template <bool HAS_W, bool HAS_RS>
__global__ __launch_bounds__(256) void k_gather_rows_scalar(const float* __restrict__ table, int F,
                                                            const int32_t* __restrict__ idx, int64_t M,
                                                            const float* __restrict__ weight,
                                                            const float* __restrict__ row_scale,
                                                            float* __restrict__ out) {
    const int lane = threadIdx.x & 63;
    for (int64_t e = (int64_t)blockIdx.x * kWavesPerBlock + (threadIdx.x >> 6); e < M;
         e += (int64_t)gridDim.x * kWavesPerBlock) {
        const int r = idx[e];
        float w = 1.f;
        if (HAS_W) w = weight[e];
        if (HAS_RS) w *= r >= 0 ? row_scale[r] : 0.f;
        for (int col = lane; col < F; col += 64)
            out[(size_t)e * F + col] = r >= 0 ? w * table[(size_t)r * F + col] : 0.f;
    }
}

template <int RL, int VPL>
__global__ __launch_bounds__(256) void k_edge_dot(const float* __restrict__ A, const int32_t* __restrict__ ai,
                                                  const float* __restrict__ B, const int32_t* __restrict__ bi,
                                                  int F, int nvec, int64_t M, float* __restrict__ out) {
    constexpr int G = 64 / RL;
    const int lane = threadIdx.x & 63;
    const int g = lane / RL;
    const int c = lane % RL;
    const int64_t n_tiles = (M + 63) / 64;
    for (int64_t tile = (int64_t)blockIdx.x * kWavesPerBlock + (threadIdx.x >> 6); tile < n_tiles;
         tile += (int64_t)gridDim.x * kWavesPerBlock) {
        const int64_t base = tile * 64;
        const int n = (M - base) < 64 ? (int)(M - base) : 64;
        int my_a = -1, my_b = -1;
        if (lane < n) {
            my_a = ai != nullptr ? ai[base + lane] : (int)(base + lane);
            my_b = bi != nullptr ? bi[base + lane] : (int)(base + lane);
        }
        for (int j = 0; j < n; j += G * 2) {
            float s[2];
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const int e = j + u * G + g;
                const int ra = __shfl(my_a, e & 63);
                const int rb = __shfl(my_b, e & 63);
                const bool ok = (e < n) && ra >= 0 && rb >= 0;
                s[u] = 0.f;
                const float* pa = A + (size_t)(ok ? ra : 0) * (size_t)F;
                const float* pb = B + (size_t)(ok ? rb : 0) * (size_t)F;
#pragma unroll
                for (int v = 0; v < VPL; ++v) {
                    const int cv = c + v * 64;
                    if (ok && cv < nvec) {
                        f32x4 x = *(const f32x4*)(pa + cv * 4);
                        f32x4 y = *(const f32x4*)(pb + cv * 4);
                        s[u] += x.x * y.x + x.y * y.y + x.z * y.z + x.w * y.w;
                    }
                }
            }
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                float t = s[u];
#pragma unroll
                for (int off = 1; off < RL; off <<= 1) t += __shfl_xor(t, off);
                const int e = j + u * G + g;
                if (c == 0 && e < n) out[base + e] = t;
            }
        }
    }
}

__global__ __launch_bounds__(256) void k_edge_dot_scalar(const float* __restrict__ A, const int32_t* __restrict__ ai,
                                                         const float* __restrict__ B, const int32_t* __restrict__ bi,
                                                         int F, int64_t M, float* __restrict__ out) {
    const int lane = threadIdx.x & 63;
    for (int64_t e = (int64_t)blockIdx.x * kWavesPerBlock + (threadIdx.x >> 6); e < M;
         e += (int64_t)gridDim.x * kWavesPerBlock) {
        const int ra = ai != nullptr ? ai[e] : (int)e;
        const int rb = bi != nullptr ? bi[e] : (int)e;
        float t = 0.f;
        if (ra >= 0 && rb >= 0)
            for (int col = lane; col < F; col += 64) t += A[(size_t)ra * F + col] * B[(size_t)rb * F + col];
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) t += __shfl_xor(t, off);
        if (lane == 0) out[e] = t;
    }
}

// ------------------------------------------------------------------ dispatch
struct SegArgs {
    const float* src;
    int F;
    const int32_t* src_row;
    const int32_t* perm;
    const float* weight;
    const float* row_scale;
    const int32_t *wi_begin, *wi_end, *wi_target, *n_items;
    int64_t max_items;
    float *out, *partial;
};

// 1-KiB rows (F in (128, 256], the headline shape): 16 rows in flight per wave, 16 waves per workgroup, plain
// work-item order.  Round-1 sweep of 40 variants (U in {2,4,8,16} x waves in {4,8,16} x XCD-contiguous remap x
// non-temporal loads, profiles/r01_tune_k1_L256.txt): 421-457 us, this one fastest; the XCD remap is 3-4 % slower.

template <int RL, int VPL, int U, bool W, bool RS, bool NT, int TAG, int WPB, bool XCD>
static void launch_seg3(const SegArgs& a, hipStream_t s) {
    unsigned grid = (unsigned)ceil_div(a.max_items, WPB);
    if (grid == 0) return;
    if (XCD) grid = (grid + 7) / 8 * 8;
    const int nvec = a.F / 4;
    k_seg_reduce<RL, VPL, U, W, RS, NT, TAG, WPB, XCD><<<grid, WPB * 64, 0, s>>>(
        a.src, a.F, nvec, a.src_row, a.perm, a.weight, a.row_scale, a.wi_begin, a.wi_end,
        a.wi_target, a.n_items, a.max_items, a.out, a.partial);
}

template <int RL, int VPL, int U, bool W, bool RS, int TAG>
static void launch_seg(const SegArgs& a, hipStream_t s) {
    if (g_opt_nt_loads)
        launch_seg3<RL, VPL, U, W, RS, true, TAG, 4, false>(a, s);
    else
        launch_seg3<RL, VPL, U, W, RS, false, TAG, 4, false>(a, s);
}

template <int TAG>
static void launch_seg_headline(const SegArgs& a, hipStream_t s) {
    if (g_opt_nt_loads)
        launch_seg3<64, 1, 16, false, false, true, TAG, 16, false>(a, s);
    else
        launch_seg3<64, 1, 16, false, false, false, TAG, 16, false>(a, s);
}

template <bool W, bool RS, int TAG>
static int dispatch_seg(const SegArgs& a, hipStream_t s) {
    const int F = a.F;
    if (F % 4 != 0 || F > 1024) {
        const unsigned grid = (unsigned)ceil_div(a.max_items, kWavesPerBlock);
        if (grid)
            k_seg_reduce_scalar<W, RS><<<grid, kBlock, 0, s>>>(a.src, F, a.src_row, a.perm, a.weight,
                                                               a.row_scale, a.wi_begin, a.wi_end,
                                                               a.wi_target, a.n_items, a.max_items,
                                                               a.out, a.partial);
        return HGNN_OK;
    }
    const int nvec = F / 4;
    if (nvec <= 4) launch_seg<4, 1, 4, W, RS, TAG>(a, s);
    else if (nvec <= 8) launch_seg<8, 1, 4, W, RS, TAG>(a, s);
    else if (nvec <= 16) launch_seg<16, 1, 4, W, RS, TAG>(a, s);
    else if (nvec <= 32) launch_seg<32, 1, 4, W, RS, TAG>(a, s);  // (U8 / 16-wave variants: within 1 %)
    else if (nvec <= 64) {
        if (!W && !RS) launch_seg_headline<TAG>(a, s);
        else launch_seg<64, 1, 8, W, RS, TAG>(a, s);
    } else if (nvec <= 128) launch_seg<64, 2, 4, W, RS, TAG>(a, s);
    else launch_seg<64, 4, 2, W, RS, TAG>(a, s);
    return HGNN_OK;
}

static unsigned stream_grid(int64_t n_tiles) {
    int64_t blocks = ceil_div(n_tiles, kWavesPerBlock);
    const int64_t cap = 256 * 8 * 4;  // 256 CUs x 8 blocks, x4 for balance
    if (blocks > cap) blocks = cap;
    return (unsigned)(blocks < 1 ? 1 : blocks);
}

}  // namespace hgnn

using namespace hgnn;

extern "C" int hgnn_set_option(const char* name, int value) {
    HGNN_REQUIRE(name != nullptr, "hgnn_set_option: name is NULL");
    if (!strcmp(name, "nt_loads")) g_opt_nt_loads = value;
    else if (!strcmp(name, "nt_stores")) g_opt_nt_stores = value;
    else if (!strcmp(name, "mlp_ablate")) g_opt_mlp_ablate = value & 31;
    else if (!strcmp(name, "mlp_split_variant")) g_opt_mlp_split_variant = value;
    else if (!strcmp(name, "mlp_split3_rows128")) {
        // 2 = the two-workgroup tile that returned wrong elements in one of two equivalent builds:
        // measurement tools only, behind an environment switch of its own
        if (value < 0 || value > 2 || (value == 2 && getenv("HGNN_EXPERIMENTAL") == nullptr)) {
            set_error("hgnn_set_option: mlp_split3_rows128 takes 0 or 1 (2 = experimental tile, needs HGNN_EXPERIMENTAL=1)");
            return HGNN_ERR_INVALID_ARG;
        }
        f3::g_opt_split3_rows128 = value;
    }
    else if (!strcmp(name, "mlp_split3_one_wg")) f3::g_opt_split3_one_wg = value;
    else {
        set_error("hgnn_set_option: unknown option '%s'", name);
        return HGNN_ERR_INVALID_ARG;
    }
    return HGNN_OK;
}

extern "C" int hgnn_segment_reduce_f32(const hgnn_plan* plan, const float* src, int32_t F,
                                       const float* weight, const float* row_scale, float* out,
                                       float* partial, hgnn_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    HGNN_REQUIRE(plan != nullptr, "hgnn_segment_reduce_f32: plan is NULL");
    HGNN_REQUIRE(F > 0, "hgnn_segment_reduce_f32: F must be positive (got %d)", F);
    if (plan->n_dst == 0) return HGNN_OK;
    HGNN_REQUIRE(out != nullptr, "hgnn_segment_reduce_f32: out is NULL");
    HGNN_REQUIRE(plan->n_rows == 0 || src != nullptr, "hgnn_segment_reduce_f32: src is NULL");
    HGNN_REQUIRE(plan->src_row != nullptr || !plan->has_gather,
                 "hgnn_segment_reduce_f32: src_row may only be NULL for a sorted plan without gather");
    HGNN_REQUIRE(partial != nullptr || plan->max_partial == 0, "hgnn_segment_reduce_f32: partial is NULL");
    HGNN_REQUIRE(((uintptr_t)src % 16 == 0 && (uintptr_t)out % 16 == 0 && (uintptr_t)partial % 16 == 0) || F % 4 != 0,
                 "hgnn_segment_reduce_f32: src/out/partial must be 16-byte aligned");
    SegArgs a;
    a.src = src;
    a.F = F;
    a.src_row = plan->src_row;
    a.perm = plan->perm;
    a.weight = weight;
    a.row_scale = row_scale;
    a.wi_begin = plan->wi_begin;
    a.wi_end = plan->wi_end;
    a.wi_target = plan->wi_target;
    a.n_items = plan->counts + HGNN_CNT_WORK;
    a.max_items = plan->max_work;
    a.out = out;
    a.partial = partial;
    int rc;
    if (weight && row_scale) rc = dispatch_seg<true, true, 0>(a, stream);
    else if (weight) rc = dispatch_seg<true, false, 0>(a, stream);
    else if (row_scale) {
        set_error("hgnn_segment_reduce_f32: row_scale requires weight");
        return HGNN_ERR_UNSUPPORTED;
    } else if (plan->src_row == nullptr) {
        rc = dispatch_seg<false, false, 2>(a, stream);  // TAG 2: sorted-layout (streaming) launches, named apart in profiles
    } else rc = dispatch_seg<false, false, 0>(a, stream);
    if (rc != HGNN_OK) return rc;
    // second pass: sum the partial rows of split destinations, in chunk order
    SegArgs b;
    b.src = partial;
    b.F = F;
    b.src_row = nullptr;
    b.perm = nullptr;
    b.weight = nullptr;
    b.row_scale = nullptr;
    b.wi_begin = plan->split_pbegin;
    b.wi_end = plan->split_pbegin + 1;
    b.wi_target = plan->split_dst;
    b.n_items = plan->counts + HGNN_CNT_SPLIT;
    b.max_items = plan->max_split;
    b.out = out;
    b.partial = partial;
    rc = dispatch_seg<false, false, 1>(b, stream);
    if (rc != HGNN_OK) return rc;
    HGNN_CHECK_HIP(hipGetLastError());
    return HGNN_OK;
}

template <bool W, bool RS>
static int dispatch_gather(const float* table, int F, const int32_t* idx, int64_t M, const float* weight,
                           const float* row_scale, float* out, hipStream_t s) {
    const int64_t n_tiles = ceil_div(M, 64);
    if (F % 4 != 0 || F > 1024) {
        k_gather_rows_scalar<W, RS><<<stream_grid(M), kBlock, 0, s>>>(table, F, idx, M, weight, row_scale, out);
        return HGNN_OK;
    }
    const int nvec = F / 4;
    const unsigned grid = stream_grid(n_tiles);
    const bool nt = g_opt_nt_stores != 0;
#define HGNN_G(RL, VPL, U) \
    k_gather_rows<RL, VPL, U, W, RS><<<grid, kBlock, 0, s>>>(table, F, nvec, idx, M, weight, row_scale, out, nt)
    if (nvec <= 4) HGNN_G(4, 1, 2);
    else if (nvec <= 8) HGNN_G(8, 1, 2);
    else if (nvec <= 16) HGNN_G(16, 1, 4);
    else if (nvec <= 32) HGNN_G(32, 1, 4);
    else if (nvec <= 64) HGNN_G(64, 1, 8);
    else if (nvec <= 128) HGNN_G(64, 2, 4);
    else HGNN_G(64, 4, 2);
#undef HGNN_G
    return HGNN_OK;
}

extern "C" int hgnn_gather_rows_f32(const float* table, int64_t table_rows, int32_t F, const int32_t* idx,
                                    int64_t M, const float* weight, const float* row_scale, float* out,
                                    hgnn_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    HGNN_REQUIRE(F > 0 && M >= 0 && table_rows >= 0, "hgnn_gather_rows_f32: bad sizes");
    if (M == 0) return HGNN_OK;
    HGNN_REQUIRE(table != nullptr && idx != nullptr && out != nullptr, "hgnn_gather_rows_f32: NULL pointer");
    HGNN_REQUIRE(((uintptr_t)table % 16 == 0 && (uintptr_t)out % 16 == 0) || F % 4 != 0,
                 "hgnn_gather_rows_f32: table/out must be 16-byte aligned");
    int rc;
    if (weight && row_scale) rc = dispatch_gather<true, true>(table, F, idx, M, weight, row_scale, out, stream);
    else if (weight) rc = dispatch_gather<true, false>(table, F, idx, M, weight, row_scale, out, stream);
    else if (row_scale) rc = dispatch_gather<false, true>(table, F, idx, M, weight, row_scale, out, stream);
    else rc = dispatch_gather<false, false>(table, F, idx, M, weight, row_scale, out, stream);
    if (rc != HGNN_OK) return rc;
    HGNN_CHECK_HIP(hipGetLastError());
    return HGNN_OK;
}

template <bool W>
static int dispatch_spread(const hgnn_plan* plan, const float* table, int F, const float* weight, float* out,
                           hipStream_t s) {
    const int64_t max_items = plan->max_work;
    const int32_t* n_items = plan->counts + HGNN_CNT_WORK;
    if (F % 4 != 0 || F > 1024) {
        const unsigned grid = (unsigned)ceil_div(max_items, kWavesPerBlock);
        if (grid)
            k_spread_rows_scalar<W><<<grid, kBlock, 0, s>>>(table, F, plan->perm, weight, plan->wi_begin,
                                                           plan->wi_end, plan->wi_dst, n_items, max_items, out);
        return HGNN_OK;
    }
    const int nvec = F / 4;
    const bool nt = g_opt_nt_stores != 0;
#define HGNN_S(RL, VPL, WPB)                                                                          \
    do {                                                                                              \
        const unsigned grid = (unsigned)ceil_div(max_items, WPB);                                     \
        if (grid)                                                                                     \
            k_spread_rows<RL, VPL, W, WPB><<<grid, WPB * 64, 0, s>>>(table, F, nvec, plan->perm, weight,   \
                                                                     plan->wi_begin, plan->wi_end,    \
                                                                     plan->wi_dst, n_items, max_items, out, nt); \
    } while (0)
    if (nvec <= 4) HGNN_S(4, 1, 4);
    else if (nvec <= 8) HGNN_S(8, 1, 4);
    else if (nvec <= 16) HGNN_S(16, 1, 4);
    else if (nvec <= 32) HGNN_S(32, 1, 4);
    else if (nvec <= 64) HGNN_S(64, 1, 8);
    else if (nvec <= 128) HGNN_S(64, 2, 8);
    else HGNN_S(64, 4, 8);
#undef HGNN_S
    return HGNN_OK;
}

extern "C" int hgnn_spread_rows_f32(const hgnn_plan* plan, const float* table, int32_t F, const float* weight,
                                    float* out, hgnn_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    HGNN_REQUIRE(plan != nullptr, "hgnn_spread_rows_f32: plan is NULL");
    HGNN_REQUIRE(F > 0, "hgnn_spread_rows_f32: F must be positive");
    if (plan->n_rows == 0 || plan->n_dst == 0) return HGNN_OK;
    HGNN_REQUIRE(!plan->has_gather, "hgnn_spread_rows_f32: plan must be a plain destination plan");
    HGNN_REQUIRE(table != nullptr && out != nullptr, "hgnn_spread_rows_f32: NULL pointer");
    HGNN_REQUIRE(((uintptr_t)table % 16 == 0 && (uintptr_t)out % 16 == 0) || F % 4 != 0,
                 "hgnn_spread_rows_f32: table/out must be 16-byte aligned");
    int rc = weight ? dispatch_spread<true>(plan, table, F, weight, out, stream)
                    : dispatch_spread<false>(plan, table, F, weight, out, stream);
    if (rc != HGNN_OK) return rc;
    HGNN_CHECK_HIP(hipGetLastError());
    return HGNN_OK;
}

extern "C" int hgnn_edge_dot_f32(const float* A, const int32_t* ai, int64_t a_rows, const float* B,
                                 const int32_t* bi, int64_t b_rows, int32_t F, int64_t M, float* out,
                                 hgnn_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    HGNN_REQUIRE(F > 0 && M >= 0, "hgnn_edge_dot_f32: bad sizes");
    if (M == 0) return HGNN_OK;
    HGNN_REQUIRE(A != nullptr && B != nullptr && out != nullptr, "hgnn_edge_dot_f32: NULL pointer");
    HGNN_REQUIRE((ai != nullptr || a_rows >= M) && (bi != nullptr || b_rows >= M),
                 "hgnn_edge_dot_f32: identity-indexed operand has fewer than M rows");
    const int64_t n_tiles = ceil_div(M, 64);
    if (F % 4 != 0 || F > 1024 || (uintptr_t)A % 16 != 0 || (uintptr_t)B % 16 != 0) {
        k_edge_dot_scalar<<<stream_grid(M), kBlock, 0, stream>>>(A, ai, B, bi, F, M, out);
    } else {
        const int nvec = F / 4;
        const unsigned grid = stream_grid(n_tiles);
#define HGNN_D(RL, VPL) k_edge_dot<RL, VPL><<<grid, kBlock, 0, stream>>>(A, ai, B, bi, F, nvec, M, out)
        if (nvec <= 4) HGNN_D(4, 1);
        else if (nvec <= 8) HGNN_D(8, 1);
        else if (nvec <= 16) HGNN_D(16, 1);
        else if (nvec <= 32) HGNN_D(32, 1);
        else if (nvec <= 64) HGNN_D(64, 1);
        else if (nvec <= 128) HGNN_D(64, 2);
        else HGNN_D(64, 4);
#undef HGNN_D
    }
    HGNN_CHECK_HIP(hipGetLastError());
    return HGNN_OK;
}
