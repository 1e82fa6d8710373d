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
template <int K, int DP, int BLOCK>
__global__ __launch_bounds__(BLOCK) void k_knn_radius(const float* __restrict__ query, int64_t nq,
                                                      const float* __restrict__ points, int64_t np, int D,
                                                      float r2, int64_t* __restrict__ idx_out,
                                                      float* __restrict__ d2_out) {
    __shared__ __attribute__((aligned(16))) float tile[kKnnTile * DP];
    const int64_t q = (int64_t)blockIdx.x * BLOCK + threadIdx.x;
    const bool active = q < nq;
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
    for (int64_t base = 0; base < np; base += kKnnTile) {
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
                        float cd = d2;
                        int ci = (int)(base + j0 + u);
#pragma unroll
                        for (int k = 0; k < K; ++k) {
                            if (cd < best_d[k]) {
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
#pragma unroll
        for (int k = 0; k < K; ++k) {
            idx_out[q * K + k] = (int64_t)best_i[k];
            if (d2_out != nullptr) d2_out[q * K + k] = best_i[k] >= 0 ? best_d[k] : -1.f;
        }
    }
}

template <int K>
static void launch_knn(const float* query, int64_t nq, const float* points, int64_t np, int D, float r2,
                       int64_t* idx_out, float* d2_out, hipStream_t stream) {
    const bool small = nq < 65536;
    const unsigned grid = (unsigned)ceil_div(nq, small ? 64 : 256);
#define HGNN_KNN_DP(DP)                                                                                      \
    do {                                                                                                     \
        if (small) k_knn_radius<K, DP, 64><<<grid, 64, 0, stream>>>(query, nq, points, np, D, r2, idx_out, d2_out); \
        else k_knn_radius<K, DP, 256><<<grid, 256, 0, stream>>>(query, nq, points, np, D, r2, idx_out, d2_out);     \
    } while (0)
    if (D <= 4) HGNN_KNN_DP(4);
    else if (D <= 8) HGNN_KNN_DP(8);
    else HGNN_KNN_DP(16);
#undef HGNN_KNN_DP
}

}  // namespace hgnn

using namespace hgnn;

extern "C" int hgnn_knn_radius_f32(const float* query, int64_t nq, const float* points, int64_t np, int32_t D,
                                   int32_t K, float radius, int64_t* idx_out, float* dist2_out,
                                   hgnn_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    HGNN_REQUIRE(nq >= 0 && np >= 0 && np < ((int64_t)1 << 31), "hgnn_knn_radius_f32: bad sizes");
    HGNN_REQUIRE(D >= 1 && D <= kKnnDMax, "hgnn_knn_radius_f32: D must be in [1, %d] (got %d)", kKnnDMax, D);
    HGNN_REQUIRE(K >= 1 && K <= 32, "hgnn_knn_radius_f32: K must be in [1, 32] (got %d)", K);
    HGNN_REQUIRE(radius >= 0.f, "hgnn_knn_radius_f32: negative radius");
    if (nq == 0) return HGNN_OK;
    HGNN_REQUIRE(query != nullptr && idx_out != nullptr && (np == 0 || points != nullptr),
                 "hgnn_knn_radius_f32: NULL pointer");
    const float r2 = radius * radius;
#define HGNN_KNN(KK) launch_knn<KK>(query, nq, points, np, D, r2, idx_out, dist2_out, stream)
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
            set_error("hgnn_knn_radius_f32: K=%d has no instantiation (1-6, 8, 10, 12, 16, 20, 32)", K);
            return HGNN_ERR_UNSUPPORTED;
    }
#undef HGNN_KNN
    HGNN_CHECK_HIP(hipGetLastError());
    return HGNN_OK;
}
