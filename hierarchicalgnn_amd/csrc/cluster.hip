// The hierarchy decision of HierarchicalGNNBlock.clustering (reference
// BipartiteClassification/Models/HGNN_GMM.py:162-234) as device code without host round trips:
//
//   hgnn_gmm2_fit_f32   2-component 1-D Gaussian mixture on the per-edge likelihoods (the reference ships
//                       them to sklearn GaussianMixture(2).fit on the CPU, :192): deterministic 2-means start,
//                       EM with the convergence test ON THE DEVICE -- every pass is one launch whose last
//                       workgroup reduces the partial sums and updates the mixture; once the lower bound
//                       moves by less than `tol` the remaining (already enqueued) passes return at once.
//   hgnn_gmm2_cut_f32   the cut where the right component is e^r times likelier than the left (:162-170,
//                       scipy fsolve on the host in the reference) by bisection in one thread, plus the
//                       score_cut initialisation / EMA (:195-208) on the device buffer itself.
//   hgnn_cc_labels      weakly connected components of the edges above the cut (:212-221, cugraph) by
//                       lock-free union-find: ONE pass over the edges (hook the larger root under the
//                       smaller with atomicCAS, retry from the returned parent), one pass of path
//                       compression.  No sweep-until-stable loop, hence no convergence read; the label of
//                       a component is its smallest vertex id whatever the interleaving (deterministic).
//
// Every loop a wave runs here has a strictly decreasing quantity (vertex id while climbing to a root,
// candidate root id after a failed CAS), so the grid always drains.
#include "common.h"

namespace hgnn {

constexpr int kGmmBlocks = HGNN_GMM_BLOCKS;
constexpr int kGmmThreads = 256;
// state (double[HGNN_GMM_STATE]) slots
enum { S_W0 = 0, S_W1, S_MU0, S_MU1, S_VAR0, S_VAR1, S_PREV, S_CONV, S_ITERS, S_MIN, S_MAX, S_C0, S_C1, S_CUT,
       S_LOWER, S_SPARE };

enum { PASS_MINMAX = 0, PASS_LLOYD = 1, PASS_HARD_M = 2, PASS_EM = 3 };

__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}
__device__ __forceinline__ double wave_min(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmin(v, __shfl_xor(v, o));
    return v;
}
__device__ __forceinline__ double wave_max(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmax(v, __shfl_xor(v, o));
    return v;
}

// One pass over v[M].  acc[0..6]: per-pass sums (meaning depends on the pass); block partials go to
// partials[block][8]; the LAST block to finish (ticket counter) folds them in block order (fixed order:
// deterministic) and updates the state.
template <int PASS>
__global__ __launch_bounds__(kGmmThreads) void k_gmm_pass(const float* __restrict__ v, int64_t M,
                                                         double* __restrict__ state, double* __restrict__ partials,
                                                         unsigned* __restrict__ ticket, float tol, float reg_covar) {
    __shared__ double red[kGmmThreads / 64][8];
    __shared__ bool is_last;
    if (PASS == PASS_EM && state[S_CONV] != 0.0) return;  // converged earlier: nothing to do (uniform)
    double a[7] = {0, 0, 0, 0, 0, 0, 0};
    if (PASS == PASS_MINMAX) {
        a[0] = 1e300;
        a[1] = -1e300;
    }
    float c0 = 0.f, c1 = 0.f, mu0 = 0.f, mu1 = 0.f, iv0 = 0.f, iv1 = 0.f, k0 = 0.f, k1 = 0.f;
    if (PASS == PASS_LLOYD || PASS == PASS_HARD_M) {
        c0 = (float)state[S_C0];
        c1 = (float)state[S_C1];
    }
    if (PASS == PASS_EM) {
        mu0 = (float)state[S_MU0];
        mu1 = (float)state[S_MU1];
        const float var0 = (float)state[S_VAR0], var1 = (float)state[S_VAR1];
        iv0 = 1.0f / var0;
        iv1 = 1.0f / var1;
        // log w_k - 0.5 log(2 pi var_k)
        k0 = logf((float)state[S_W0]) - 0.5f * logf(6.283185307179586f * var0);
        k1 = logf((float)state[S_W1]) - 0.5f * logf(6.283185307179586f * var1);
    }
    for (int64_t i = (int64_t)blockIdx.x * kGmmThreads + threadIdx.x; i < M; i += (int64_t)gridDim.x * kGmmThreads) {
        const float x = v[i];
        if (PASS == PASS_MINMAX) {
            a[0] = fmin(a[0], (double)x);
            a[1] = fmax(a[1], (double)x);
        } else if (PASS == PASS_LLOYD || PASS == PASS_HARD_M) {
            const bool hard = fabsf(x - c0) > fabsf(x - c1);  // component 1 iff strictly nearer to c1
            const double xd = (double)x;
            if (hard) {
                a[3] += 1.0;
                a[4] += xd;
                a[5] += xd * xd;
            } else {
                a[0] += 1.0;
                a[1] += xd;
                a[2] += xd * xd;
            }
        } else {
            const float d0 = x - mu0, d1 = x - mu1;
            const float lp0 = fmaf(-0.5f * d0 * d0, iv0, k0);
            const float lp1 = fmaf(-0.5f * d1 * d1, iv1, k1);
            const float m = fmaxf(lp0, lp1);
            const float e0 = __expf(lp0 - m), e1 = __expf(lp1 - m);
            const float s = e0 + e1;
            const float norm = m + __logf(s);
            const float r0 = e0 / s;
            const double r0d = (double)r0, r1d = 1.0 - (double)r0, xd = (double)x;
            a[0] += r0d;
            a[1] += r0d * xd;
            a[2] += r0d * xd * xd;
            a[3] += r1d;
            a[4] += r1d * xd;
            a[5] += r1d * xd * xd;
            a[6] += (double)norm;
        }
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int j = 0; j < 7; ++j) {
        double r = a[j];
        if (PASS == PASS_MINMAX) r = j == 0 ? wave_min(r) : (j == 1 ? wave_max(r) : 0.0);
        else r = wave_sum(r);
        if (lane == 0) red[wave][j] = r;
    }
    __syncthreads();
    if (threadIdx.x < 7) {
        double r = red[0][threadIdx.x];
        for (int w = 1; w < kGmmThreads / 64; ++w) {
            const double o = red[w][threadIdx.x];
            if (PASS == PASS_MINMAX) r = threadIdx.x == 0 ? fmin(r, o) : (threadIdx.x == 1 ? fmax(r, o) : 0.0);
            else r += o;
        }
        partials[(size_t)blockIdx.x * 8 + threadIdx.x] = r;
    }
    __threadfence();
    __syncthreads();
    if (threadIdx.x == 0) {
        const unsigned t = atomicAdd(ticket, 1u);
        is_last = (t == gridDim.x - 1);
    }
    __syncthreads();
    if (!is_last) return;
    __threadfence();
    // fold the block partials in a FIXED order (deterministic): thread t sums entry j = t & 7 of the blocks
    // b = t >> 3, t >> 3 + 32, ... (independent loads in flight), then the 32 strided sums are combined
    // in index order.  The partials were written by other workgroups during this launch: agent-scope
    // loads, not the per-CU vector cache.
    __shared__ double fold[32][8];
    {
        const int j = threadIdx.x & 7, g = threadIdx.x >> 3;
        double r = PASS == PASS_MINMAX ? (j == 0 ? 1e300 : -1e300) : 0.0;
        if (j < 7) {
            for (unsigned b = g; b < gridDim.x; b += 32) {
                const double o = __hip_atomic_load(partials + (size_t)b * 8 + j, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (PASS == PASS_MINMAX) r = j == 0 ? fmin(r, o) : (j == 1 ? fmax(r, o) : 0.0);
                else r += o;
            }
        }
        fold[g][j] = r;
    }
    __syncthreads();
    if (threadIdx.x < 7) {
        const int j = threadIdx.x;
        double r = fold[0][j];
        for (int g = 1; g < 32; ++g) {
            const double o = fold[g][j];
            if (PASS == PASS_MINMAX) r = j == 0 ? fmin(r, o) : (j == 1 ? fmax(r, o) : 0.0);
            else r += o;
        }
        red[0][j] = r;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        *ticket = 0;  // ready for the next pass (stream order)
        const double* s = red[0];
        if (PASS == PASS_MINMAX) {
            state[S_MIN] = s[0];
            state[S_MAX] = s[1];
            state[S_C0] = s[0];
            state[S_C1] = s[1];
        } else if (PASS == PASS_LLOYD) {
            state[S_C0] = s[1] / fmax(s[0], 1.0);
            state[S_C1] = s[4] / fmax(s[3], 1.0);
        } else {
            bool update = true;
            if (PASS == PASS_EM) {
                const double lower = s[6] / (double)M;
                const double iters = state[S_ITERS];
                state[S_LOWER] = lower;
                if (iters > 0.0 && fabs(lower - state[S_PREV]) < (double)tol) {
                    state[S_CONV] = 1.0;  // keep the mixture the lower bound was computed with
                    update = false;
                }
                state[S_PREV] = lower;
                state[S_ITERS] = iters + 1.0;
            }
            if (update) {
                const double eps10 = 10.0 * 1.1920928955078125e-07;  // sklearn: nk += 10 * eps(float32 data)
                const double n0 = s[0] + eps10, n1 = s[3] + eps10;
                const double m0 = s[1] / n0, m1 = s[4] / n1;
                state[S_W0] = n0 / (double)M;
                state[S_W1] = n1 / (double)M;
                state[S_MU0] = m0;
                state[S_MU1] = m1;
                state[S_VAR0] = fmax(s[2] / n0 - m0 * m0, 0.0) + (double)reg_covar;
                state[S_VAR1] = fmax(s[5] / n1 - m1 * m1, 0.0) + (double)reg_covar;
            }
        }
    }
}

__global__ void k_gmm_reset(double* state, unsigned* ticket) {
    if (threadIdx.x < HGNN_GMM_STATE) state[threadIdx.x] = 0.0;
    if (threadIdx.x == 0) *ticket = 0;
}

// root of  sigmoid(r) P(left | x) - sigmoid(-r) P(right | x)  between the two means (HGNN_GMM.py:162-170),
// then the block's score_cut bookkeeping (:195-208)
__global__ void k_gmm_cut(double* state, float granularity, int training, float momentum, float* score_cut) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    const double w[2] = {state[S_W0], state[S_W1]};
    const double mu[2] = {state[S_MU0], state[S_MU1]};
    const double var[2] = {state[S_VAR0], state[S_VAR1]};
    const int left = mu[0] <= mu[1] ? 0 : 1, right = 1 - left;
    const double sr = 1.0 / (1.0 + exp(-(double)granularity)), sl = 1.0 / (1.0 + exp((double)granularity));
    auto f = [&](double x) {
        double lp[2];
        for (int k = 0; k < 2; ++k)
            lp[k] = log(w[k]) - 0.5 * ((x - mu[k]) * (x - mu[k]) / var[k] + log(6.283185307179586 * var[k]));
        const double m = fmax(lp[0], lp[1]);
        const double p0 = exp(lp[0] - m), p1 = exp(lp[1] - m);
        const double p[2] = {p0 / (p0 + p1), p1 / (p0 + p1)};
        return sr * p[left] - sl * p[right];
    };
    double lo = mu[left], hi = mu[right];
    double cut = 0.5 * (lo + hi);
    if (f(lo) * f(hi) <= 0.0) {
        for (int it = 0; it < 60; ++it) {
            const double mid = 0.5 * (lo + hi);
            if (f(lo) * f(mid) <= 0.0) hi = mid;
            else lo = mid;
        }
        cut = 0.5 * (lo + hi);
    }
    state[S_CUT] = cut;
    const double mlo = fmin(mu[0], mu[1]), mhi = fmax(mu[0], mu[1]);
    float sc = *score_cut;
    if (isinf(sc)) sc = (float)(0.5 * (mlo + mhi));  // :196-197 first call: the middle of the two means
    if (training && cut > mlo && cut < mhi) sc = momentum * sc + (1.0f - momentum) * (float)cut;
    *score_cut = sc;
}

// ------------------------------------------------------------------------------------------- components
// parent[] is read and written by every workgroup while links are made by atomicCAS: relaxed agent-scope
// atomics keep the compiler from caching entries in registers and the loads out of the per-CU vector cache.
// A stale value would still be harmless (links are never removed and every link is validated by its CAS).
__device__ __forceinline__ int ld_parent(const int* parent, int v) {
    return __hip_atomic_load(parent + v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

__device__ __forceinline__ int uf_find(int* parent, int v) {
    // climb to the root; parents always have SMALLER ids, so the walk is finite.  Path halving writes are
    // benign races: they only ever replace the parent of a NON-root by one of its ancestors.
    int p = ld_parent(parent, v);
    while (p != v) {
        const int gp = ld_parent(parent, p);
        if (gp != p) __hip_atomic_store(parent + v, gp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        v = p;
        p = gp;
    }
    return v;
}

__global__ __launch_bounds__(256) void k_cc_init(int* __restrict__ parent, int* __restrict__ present, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) {
        parent[i] = (int)i;
        present[i] = 0;
    }
}

__global__ __launch_bounds__(256) void k_cc_hook(const int64_t* __restrict__ src, const int64_t* __restrict__ dst,
                                                 int64_t M, int64_t n, const float* __restrict__ score,
                                                 const float* __restrict__ cut, int* parent,
                                                 int* __restrict__ present) {
    const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (e >= M) return;
    if (score != nullptr && !(score[e] >= *cut)) return;  // the edge was cut (HGNN_GMM.py:212)
    const int64_t u64 = src[e], v64 = dst[e];
    if (u64 < 0 || v64 < 0 || u64 >= n || v64 >= n) return;
    present[u64] = 1;
    present[v64] = 1;
    int a = uf_find(parent, (int)u64);
    int b = uf_find(parent, (int)v64);
    while (a != b) {
        const int hi = a > b ? a : b, lo = a > b ? b : a;
        const int old = atomicCAS(&parent[hi], hi, lo);
        if (old == hi) break;  // hi was still a root: now it hangs under lo
        // hi had been linked under `old` (< hi) meanwhile: continue from there; max(a, b) strictly decreases
        a = uf_find(parent, old);
        b = lo;
    }
}

__global__ __launch_bounds__(256) void k_cc_compress(int* parent, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    int r = (int)i;
    int p = ld_parent(parent, r);
    while (p != r) {  // no more links are made: a read-only climb over strictly decreasing ids
        r = p;
        p = ld_parent(parent, r);
    }
    // races with other climbers only replace a parent by its root
    __hip_atomic_store(parent + i, r, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

}  // namespace hgnn

using namespace hgnn;

extern "C" int hgnn_gmm2_fit_f32(const float* v, int64_t M, int32_t max_iter, float tol, float reg_covar,
                                 double* state, double* partials, uint32_t* ticket, hgnn_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    HGNN_REQUIRE(M > 0 && v != nullptr && state != nullptr && partials != nullptr && ticket != nullptr,
                 "hgnn_gmm2_fit_f32: NULL argument or empty input");
    HGNN_REQUIRE(max_iter >= 1 && max_iter <= 1000 && tol >= 0.f && reg_covar >= 0.f, "hgnn_gmm2_fit_f32: bad parameters");
    int64_t want = ceil_div(M, (int64_t)kGmmThreads * 4);
    const unsigned grid = (unsigned)(want < 1 ? 1 : (want > kGmmBlocks ? kGmmBlocks : want));
    k_gmm_reset<<<1, 64, 0, stream>>>(state, ticket);
    k_gmm_pass<PASS_MINMAX><<<grid, kGmmThreads, 0, stream>>>(v, M, state, partials, ticket, tol, reg_covar);
    for (int i = 0; i < 8; ++i)
        k_gmm_pass<PASS_LLOYD><<<grid, kGmmThreads, 0, stream>>>(v, M, state, partials, ticket, tol, reg_covar);
    k_gmm_pass<PASS_HARD_M><<<grid, kGmmThreads, 0, stream>>>(v, M, state, partials, ticket, tol, reg_covar);
    for (int i = 0; i < max_iter; ++i)
        k_gmm_pass<PASS_EM><<<grid, kGmmThreads, 0, stream>>>(v, M, state, partials, ticket, tol, reg_covar);
    HGNN_CHECK_HIP(hipGetLastError());
    return HGNN_OK;
}

extern "C" int hgnn_gmm2_cut_f32(double* state, float granularity, int32_t training, float momentum,
                                 float* score_cut, hgnn_stream_t stream_) {
    HGNN_REQUIRE(state != nullptr && score_cut != nullptr, "hgnn_gmm2_cut_f32: NULL argument");
    k_gmm_cut<<<1, 64, 0, (hipStream_t)stream_>>>(state, granularity, training, momentum, score_cut);
    HGNN_CHECK_HIP(hipGetLastError());
    return HGNN_OK;
}

extern "C" int hgnn_cc_labels(const int64_t* src, const int64_t* dst, int64_t M, int64_t n, const float* score,
                              const float* cut, int32_t* labels, int32_t* present, hgnn_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    HGNN_REQUIRE(n >= 0 && n < ((int64_t)1 << 31) && M >= 0, "hgnn_cc_labels: bad sizes");
    if (n == 0) return HGNN_OK;
    HGNN_REQUIRE(labels != nullptr && present != nullptr && (M == 0 || (src != nullptr && dst != nullptr)),
                 "hgnn_cc_labels: NULL argument");
    HGNN_REQUIRE((score == nullptr) == (cut == nullptr), "hgnn_cc_labels: score and cut go together");
    k_cc_init<<<(unsigned)ceil_div(n, 256), 256, 0, stream>>>(labels, present, n);
    if (M > 0) k_cc_hook<<<(unsigned)ceil_div(M, 256), 256, 0, stream>>>(src, dst, M, n, score, cut, labels, present);
    k_cc_compress<<<(unsigned)ceil_div(n, 256), 256, 0, stream>>>(labels, n);
    HGNN_CHECK_HIP(hipGetLastError());
    return HGNN_OK;
}
