// Fixed-radius k-nearest-neighbour search in a low-dimensional embedding space:
// what the reference obtains from frnn.frnn_grid_points (Modules/utils.py:228-239, called by
// DynamicGraphConstruction.forward, Modules/gnn_utils.py:194) for the bipartite graph
// (N hits -> S cluster centres, K=5) and the super graph (S -> S, K=10) in emb_dim = 8.
//
// frnn 0.0.0 is an un-vendored CUDA extension (grid-hashed search).  On MI355X the problem is
// small enough (N*S*D = 1e10 FMA at N=120k, S=10k, D=8) that an exact tiled brute force is both
// simpler and HBM-trivial: one query per lane, candidate points staged through LDS in tiles of
// 256, a sorted top-K kept in registers.  Results: for every query the <=K nearest points with
// squared distance < r^2, ascending by distance (ties: lower index first), -1 padded.
#include "common.h"

namespace hgnn {

constexpr int kKnnTile = 256;
constexpr int kKnnDMax = 16;

typedef float knn_f32x4 __attribute__((ext_vector_type(4)));

// DP = D padded to 4 / 8 / 16 with zeros (adds fmaf(0, 0, d2) = d2: same distances, no per-dimension
// conditionals, candidates read as broadcast 16-byte LDS vectors); BLOCK = 64 for few queries (the
// super graph: ~10k queries would otherwise occupy 36 of the 256 CUs), 256 otherwise
//
// SPLIT (few queries, e.g. the 10k x 10k super graph): blockIdx.y selects a slice of the candidate points, so
// that a query's candidates are searched by `gridDim.y` workgroups at once (157 workgroups of one wave
// each would leave most of the 256 CUs idle and every SIMD with a single wave: 1.7 ms for 0.8 GFLOP); each
// writes its sorted partial top-K (part_d / part_i [nq][slices][K]) and k_knn_merge combines the slices
// in index order, which reproduces the one-pass result exactly (ties: lower index first).
// radius: by value, or read from device memory (`r_dev`, the module's knn_radius buffer: no host read)
template <int K, int DP, int BLOCK, bool SPLIT>
__global__ __launch_bounds__(BLOCK) void k_knn_radius(const float* __restrict__ query, int64_t nq,
                                                      const float* __restrict__ points, int64_t np_all, int D,
                                                      float radius, const float* __restrict__ r_dev,
                                                      int64_t* __restrict__ idx_out, float* __restrict__ d2_out,
                                                      int64_t slice_len, float* __restrict__ part_d,
                                                      int* __restrict__ part_i) {
    __shared__ __attribute__((aligned(16))) float tile[kKnnTile * DP];
    const int64_t q = (int64_t)blockIdx.x * BLOCK + threadIdx.x;
    const bool active = q < nq;
    const float rr = r_dev != nullptr ? *r_dev : radius;
    const float r2 = rr * rr;
    const int64_t p_begin = SPLIT ? (int64_t)blockIdx.y * slice_len : 0;
    const int64_t np = SPLIT ? ((p_begin + slice_len) < np_all ? (p_begin + slice_len) : np_all) : np_all;
    float qv[DP];
#pragma unroll
    for (int d = 0; d < DP; ++d) qv[d] = (active && d < D) ? query[q * D + d] : 0.f;
    float best_d[K];
    int best_i[K];
#pragma unroll
    for (int k = 0; k < K; ++k) {
        best_d[k] = 3.0e38f;
        best_i[k] = -1;
    }
    for (int64_t base = p_begin; base < np; base += kKnnTile) {
        const int n = (np - base) < kKnnTile ? (int)(np - base) : kKnnTile;
        __syncthreads();
        for (int t = threadIdx.x; t < n * DP; t += BLOCK) {
            const int pt = t / DP, d = t % DP;
            tile[t] = d < D ? points[(base + pt) * D + d] : 0.f;
        }
        __syncthreads();
        if (active) {
            // 4 candidates per step: their distances first (independent LDS reads in flight together),
            // then the (rare) insertions in candidate order -- same result as one at a time
            for (int j0 = 0; j0 < n; j0 += 4) {
                float d2u[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    float d2 = 0.f;
#pragma unroll
                    for (int v = 0; v < DP / 4; ++v) {
                        const knn_f32x4 p4 = *(const knn_f32x4*)(tile + (j0 + u) * DP + v * 4);
                        float t = qv[v * 4 + 0] - p4.x;
                        d2 = fmaf(t, t, d2);
                        t = qv[v * 4 + 1] - p4.y;
                        d2 = fmaf(t, t, d2);
                        t = qv[v * 4 + 2] - p4.z;
                        d2 = fmaf(t, t, d2);
                        t = qv[v * 4 + 3] - p4.w;
                        d2 = fmaf(t, t, d2);
                    }
                    d2u[u] = (j0 + u) < n ? d2 : 3.0e38f;   // rows past the tile's end hold stale data
                }
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const float d2 = d2u[u];
                    if (d2 < r2 && d2 < best_d[K - 1]) {
                        // insertion into the sorted list (static indexing: stays in registers)
                        // STABLE insertion: the new entry goes behind every entry <= it, and from there on
                        // every entry moves down one slot (a carried entry that only TIES with its successor
                        // must still be put in front of it, or equal distances would swap places)
                        float cd = d2;
                        int ci = (int)(base + j0 + u);
                        bool ins = false;
#pragma unroll
                        for (int k = 0; k < K; ++k) {
                            if (ins || cd < best_d[k]) {
                                ins = true;
                                const float td = best_d[k];
                                const int ti = best_i[k];
                                best_d[k] = cd;
                                best_i[k] = ci;
                                cd = td;
                                ci = ti;
                            }
                        }
                    }
                }
            }
        }
    }
    if (active) {
        if constexpr (SPLIT) {
            const size_t o = ((size_t)q * gridDim.y + blockIdx.y) * K;
#pragma unroll
            for (int k = 0; k < K; ++k) {
                part_d[o + k] = best_d[k];
                part_i[o + k] = best_i[k];
            }
        } else {
#pragma unroll
            for (int k = 0; k < K; ++k) {
                idx_out[q * K + k] = (int64_t)best_i[k];
                if (d2_out != nullptr) d2_out[q * K + k] = best_i[k] >= 0 ? best_d[k] : -1.f;
            }
        }
    }
}

// merge the per-slice sorted lists of one query, slices in index order, strict '<' insertion: an entry of a
// later slice (higher indices) never displaces an equal distance of an earlier one
template <int K>
__global__ __launch_bounds__(256) void k_knn_merge(int64_t nq, int slices, const float* __restrict__ part_d,
                                                   const int* __restrict__ part_i, int64_t* __restrict__ idx_out,
                                                   float* __restrict__ d2_out) {
    const int64_t q = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (q >= nq) return;
    float best_d[K];
    int best_i[K];
#pragma unroll
    for (int k = 0; k < K; ++k) {
        best_d[k] = 3.0e38f;
        best_i[k] = -1;
    }
    for (int s = 0; s < slices; ++s) {
        const size_t o = ((size_t)q * slices + s) * K;
        for (int j = 0; j < K; ++j) {
            float cd = part_d[o + j];
            int ci = part_i[o + j];
            if (ci < 0 || !(cd < best_d[K - 1])) break;  // lists are ascending: nothing further can enter
            bool ins = false;
#pragma unroll
            for (int k = 0; k < K; ++k) {
                if (ins || cd < best_d[k]) {  // stable insertion, as in the search kernel
                    ins = true;
                    const float td = best_d[k];
                    const int ti = best_i[k];
                    best_d[k] = cd;
                    best_i[k] = ci;
                    cd = td;
                    ci = ti;
                }
            }
        }
    }
#pragma unroll
    for (int k = 0; k < K; ++k) {
        idx_out[q * K + k] = (int64_t)best_i[k];
        if (d2_out != nullptr) d2_out[q * K + k] = best_i[k] >= 0 ? best_d[k] : -1.f;
    }
}

// candidate slices for `nq` queries over `np` points: enough workgroups to fill the chip, whole tiles per slice
static inline void knn_slices(int64_t nq, int64_t np, int* slices, int64_t* slice_len) {
    *slices = 1;
    *slice_len = np;
    if (nq >= 65536 || np < 2 * kKnnTile) return;
    const int64_t blocks = ceil_div(nq, 64);
    int64_t want = ceil_div((int64_t)2048, blocks);
    if (want > 32) want = 32;
    if (want <= 1) return;
    int64_t len = ceil_div(ceil_div(np, want), (int64_t)kKnnTile) * kKnnTile;
    *slice_len = len;
    *slices = (int)ceil_div(np, len);
}

template <int K>
static void launch_knn(const float* query, int64_t nq, const float* points, int64_t np, int D, float radius,
                       const float* r_dev, int64_t* idx_out, float* d2_out, void* ws, hipStream_t stream) {
    const bool small = nq < 65536;
    int slices;
    int64_t slice_len;
    knn_slices(nq, np, &slices, &slice_len);
    if (ws == nullptr) slices = 1;
    const bool split = slices > 1;
    float* part_d = (float*)ws;
    int* part_i = (int*)(part_d + (split ? (size_t)nq * slices * K : 0));
    const dim3 grid((unsigned)ceil_div(nq, small ? 64 : 256), (unsigned)slices);
#define HGNN_KNN_DP(DP)                                                                                          \
    do {                                                                                                         \
        if (split)                                                                                               \
            k_knn_radius<K, DP, 64, true><<<grid, 64, 0, stream>>>(query, nq, points, np, D, radius, r_dev,      \
                                                                  idx_out, d2_out, slice_len, part_d, part_i);   \
        else if (small)                                                                                          \
            k_knn_radius<K, DP, 64, false><<<grid, 64, 0, stream>>>(query, nq, points, np, D, radius, r_dev,     \
                                                                   idx_out, d2_out, np, nullptr, nullptr);       \
        else                                                                                                     \
            k_knn_radius<K, DP, 256, false><<<grid, 256, 0, stream>>>(query, nq, points, np, D, radius, r_dev,   \
                                                                     idx_out, d2_out, np, nullptr, nullptr);     \
    } while (0)
    if (D <= 4) HGNN_KNN_DP(4);
    else if (D <= 8) HGNN_KNN_DP(8);
    else HGNN_KNN_DP(16);
#undef HGNN_KNN_DP
    if (split)
        k_knn_merge<K><<<(unsigned)ceil_div(nq, 256), 256, 0, stream>>>(nq, slices, part_d, part_i, idx_out, d2_out);
}

}  // namespace hgnn

using namespace hgnn;

static int knn_dispatch(const float* query, int64_t nq, const float* points, int64_t np, int32_t D, int32_t K,
                        float radius, const float* r_dev, int64_t* idx_out, float* dist2_out, void* ws,
                        size_t ws_bytes, hipStream_t stream, const char* who) {
    HGNN_REQUIRE(nq >= 0 && np >= 0 && np < ((int64_t)1 << 31), "%s: bad sizes", who);
    HGNN_REQUIRE(D >= 1 && D <= kKnnDMax, "%s: D must be in [1, %d] (got %d)", who, kKnnDMax, D);
    HGNN_REQUIRE(K >= 1 && K <= 32, "%s: K must be in [1, 32] (got %d)", who, K);
    HGNN_REQUIRE(r_dev != nullptr || radius >= 0.f, "%s: negative radius", who);
    if (nq == 0) return HGNN_OK;
    HGNN_REQUIRE(query != nullptr && idx_out != nullptr && (np == 0 || points != nullptr), "%s: NULL pointer", who);
    if (ws != nullptr) {
        int slices;
        int64_t slice_len;
        knn_slices(nq, np, &slices, &slice_len);
        const size_t need = slices > 1 ? (size_t)nq * slices * K * 8 : 0;
        HGNN_REQUIRE(ws_bytes >= need && (uintptr_t)ws % 16 == 0, "%s: workspace too small (%zu < %zu) or unaligned",
                     who, ws_bytes, need);
    }
#define HGNN_KNN(KK) launch_knn<KK>(query, nq, points, np, D, radius, r_dev, idx_out, dist2_out, ws, stream)
    switch (K) {
        case 1: HGNN_KNN(1); break;
        case 2: HGNN_KNN(2); break;
        case 3: HGNN_KNN(3); break;
        case 4: HGNN_KNN(4); break;
        case 5: HGNN_KNN(5); break;
        case 6: HGNN_KNN(6); break;
        case 8: HGNN_KNN(8); break;
        case 10: HGNN_KNN(10); break;
        case 12: HGNN_KNN(12); break;
        case 16: HGNN_KNN(16); break;
        case 20: HGNN_KNN(20); break;
        case 32: HGNN_KNN(32); break;
        default:
            set_error("%s: K=%d has no instantiation (1-6, 8, 10, 12, 16, 20, 32)", who, K);
            return HGNN_ERR_UNSUPPORTED;
    }
#undef HGNN_KNN
    HGNN_CHECK_HIP(hipGetLastError());
    return HGNN_OK;
}

extern "C" int hgnn_knn_radius_f32(const float* query, int64_t nq, const float* points, int64_t np, int32_t D,
                                   int32_t K, float radius, int64_t* idx_out, float* dist2_out,
                                   hgnn_stream_t stream_) {
    return knn_dispatch(query, nq, points, np, D, K, radius, nullptr, idx_out, dist2_out, nullptr, 0,
                        (hipStream_t)stream_, "hgnn_knn_radius_f32");
}

extern "C" int hgnn_knn_workspace_bytes(int64_t nq, int64_t np, int32_t K, size_t* bytes) {
    HGNN_REQUIRE(bytes != nullptr && nq >= 0 && np >= 0 && K >= 1 && K <= 32, "hgnn_knn_workspace_bytes: bad argument");
    int slices;
    int64_t slice_len;
    knn_slices(nq, np, &slices, &slice_len);
    *bytes = slices > 1 ? (size_t)nq * slices * K * 8 : 0;
    return HGNN_OK;
}

extern "C" int hgnn_knn_radius_ws_f32(const float* query, int64_t nq, const float* points, int64_t np, int32_t D,
                                      int32_t K, float radius, const float* radius_dev, int64_t* idx_out,
                                      float* dist2_out, void* workspace, size_t workspace_bytes,
                                      hgnn_stream_t stream_) {
    return knn_dispatch(query, nq, points, np, D, K, radius, radius_dev, idx_out, dist2_out, workspace,
                        workspace_bytes, (hipStream_t)stream_, "hgnn_knn_radius_ws_f32");
}
