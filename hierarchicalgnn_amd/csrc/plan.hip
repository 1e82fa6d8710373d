// Destination-sorted aggregation plan (built once per event, amortised over every cell).
//
// int64 PyG edge_index row  ->  int32 CSR by destination over a STABLE sort of
// the rows, plus the work-item list the segmented-reduce kernel walks: one item
// per destination, except that lists longer than `chunk` rows are cut into
// balanced chunks whose partial sums are combined, in chunk order, by a second
// small launch (degree-skew handling; fixed order => bitwise reproducible).
//
// Replaces nothing in the reference directly: torch_scatter's CUDA scatter_add
// (called at Modules/gnn_utils.py:50,124,125,142,143) uses atomics on the
// unsorted index.  The plan is what lets the MI355X kernel read whole rows with
// plain 16-B loads and write each output row exactly once.
#include "common.h"
#include <cstring>
#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/device/device_scan.hpp>

namespace hgnn {

struct Tri {
    int a, b, c;
};
struct TriPlus {
    __host__ __device__ Tri operator()(const Tri& x, const Tri& y) const {
        return Tri{x.a + y.a, x.b + y.b, x.c + y.c};
    }
};

__global__ __launch_bounds__(256) void k_plan_prepare(const int64_t* __restrict__ dst_index,
                                                      const int64_t* __restrict__ gather_index,
                                                      int64_t M, int64_t N, int64_t R,
                                                      int32_t* __restrict__ keys,
                                                      int32_t* __restrict__ vals,
                                                      int32_t* __restrict__ dst32,
                                                      int32_t* __restrict__ counts) {
    int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= M) return;
    int64_t d = dst_index[e];
    bool ok = (d >= 0) && (d < N);
    if (gather_index != nullptr) {
        int64_t g = gather_index[e];
        ok = ok && (g >= 0) && (g < R);
    }
    keys[e] = ok ? (int32_t)d : (int32_t)N;  // sentinel N sorts invalid rows past every list
    vals[e] = (int32_t)e;
    dst32[e] = ok ? (int32_t)d : -1;
    if (!ok) atomicOr(&counts[HGNN_CNT_ERR], 1);
}

// p in [0, M]: writes rowptr[d] for every d in (key[p-1], key[p]] and src_row[p].
__global__ __launch_bounds__(256) void k_plan_rowptr(const int32_t* __restrict__ keys_sorted,
                                                     const int32_t* __restrict__ perm,
                                                     const int64_t* __restrict__ gather_index,
                                                     int64_t M, int64_t N,
                                                     int32_t* __restrict__ rowptr,
                                                     int32_t* __restrict__ src_row,
                                                     int32_t* __restrict__ counts) {
    int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p > M) return;
    int64_t prev = (p == 0) ? -1 : (int64_t)keys_sorted[p - 1];
    int64_t cur = (p == M) ? N : (int64_t)keys_sorted[p];
    for (int64_t d = prev + 1; d <= cur; ++d) rowptr[d] = (int32_t)p;
    if (p < M) {
        int32_t e = perm[p];
        // stable sort moved a row => the index was not sorted.  Plain store of a constant (benign
        // race, every writer stores 1): an atomic here would serialise ~M updates on one address.
        if (e != (int32_t)p) counts[HGNN_CNT_UNSORTED] = 1;
        int32_t r = e;
        if (gather_index != nullptr) r = (cur < N) ? (int32_t)gather_index[e] : 0;
        src_row[p] = r;
    }
}

__global__ __launch_bounds__(256) void k_plan_chunks(const int32_t* __restrict__ rowptr, int64_t N,
                                                     int32_t chunk, Tri* __restrict__ tri) {
    int64_t d = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (d >= N) return;
    int deg = rowptr[d + 1] - rowptr[d];
    int nch = deg <= chunk ? 1 : (deg + chunk - 1) / chunk;
    int split = deg > chunk ? 1 : 0;
    tri[d] = Tri{nch, split, split ? nch : 0};
}

__global__ __launch_bounds__(256) void k_plan_fill(const int32_t* __restrict__ rowptr,
                                                   const Tri* __restrict__ tri_scan, int64_t N,
                                                   int32_t chunk, hgnn_plan plan) {
    int64_t d = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (d >= N) return;
    const int begin = rowptr[d], end = rowptr[d + 1];
    const int deg = end - begin;
    const int nch = deg <= chunk ? 1 : (deg + chunk - 1) / chunk;
    const int split = deg > chunk ? 1 : 0;
    const int len = (deg + nch - 1) / nch;
    const Tri t = tri_scan[d];
    for (int k = 0; k < nch; ++k) {
        int b = begin + k * len;
        b = b < end ? b : end;
        int e = b + len;
        e = e < end ? e : end;
        int64_t item = (int64_t)t.a + k;
        if (item < plan.max_work) {
            plan.wi_begin[item] = b;
            plan.wi_end[item] = e;
            plan.wi_target[item] = split ? ~(t.c + k) : (int32_t)d;
            plan.wi_dst[item] = (int32_t)d;
        }
    }
    if (split && t.b < plan.max_split) {
        plan.split_dst[t.b] = (int32_t)d;
        plan.split_pbegin[t.b] = t.c;
    }
    if (d == N - 1) {
        int n_split = t.b + split;
        int n_partial = t.c + (split ? nch : 0);
        plan.counts[HGNN_CNT_WORK] = t.a + nch;
        plan.counts[HGNN_CNT_SPLIT] = n_split;
        plan.counts[HGNN_CNT_PARTIAL] = n_partial;
        plan.counts[HGNN_CNT_VALID] = rowptr[N];
        if (n_split <= plan.max_split) plan.split_pbegin[n_split] = n_partial;
    }
}

__global__ __launch_bounds__(256) void k_index_to_i32(const int64_t* __restrict__ idx, int64_t M,
                                                      int64_t limit, int32_t* __restrict__ out,
                                                      int32_t* __restrict__ err) {
    int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= M) return;
    int64_t v = idx[e];
    bool ok = v >= 0 && v < limit;
    out[e] = ok ? (int32_t)v : -1;
    if (!ok && err != nullptr) atomicOr(err, 1);
}

static int key_bits(int64_t N) {
    int bits = 1;
    while (bits < 31 && ((int64_t)1 << bits) <= N) ++bits;  // keys are in [0, N]
    return bits;
}

struct PlanScratch {
    size_t keys_in, keys_out, vals_in, tri_in, tri_out, temp, temp_bytes, total;
};

static int plan_scratch_layout(int64_t M, int64_t N, PlanScratch* s, hipStream_t stream) {
    size_t sort_bytes = 0, scan_bytes = 0;
    if (M > 0) {
        HGNN_CHECK_HIP(rocprim::radix_sort_pairs(nullptr, sort_bytes, (int32_t*)nullptr,
                                                 (int32_t*)nullptr, (int32_t*)nullptr,
                                                 (int32_t*)nullptr, (size_t)M, 0u,
                                                 (unsigned)key_bits(N), stream));
    }
    if (N > 0) {
        HGNN_CHECK_HIP(rocprim::exclusive_scan(nullptr, scan_bytes, (Tri*)nullptr, (Tri*)nullptr,
                                               Tri{0, 0, 0}, (size_t)N, TriPlus(), stream));
    }
    size_t off = 0;
    auto take = [&](size_t bytes) {
        size_t o = off;
        off = align_up(off + bytes, 256);
        return o;
    };
    s->keys_in = take((size_t)M * 4);
    s->keys_out = take((size_t)M * 4);
    s->vals_in = take((size_t)M * 4);
    s->tri_in = take((size_t)N * sizeof(Tri));
    s->tri_out = take((size_t)N * sizeof(Tri));
    s->temp_bytes = sort_bytes > scan_bytes ? sort_bytes : scan_bytes;
    s->temp = take(s->temp_bytes + 256);
    s->total = off;
    return HGNN_OK;
}

}  // namespace hgnn

using namespace hgnn;

extern "C" int hgnn_plan_dims(int64_t n_rows, int64_t n_dst, int64_t n_src, int32_t chunk,
                              hgnn_plan* plan) {
    HGNN_REQUIRE(plan != nullptr, "hgnn_plan_dims: plan is NULL");
    HGNN_REQUIRE(n_rows >= 0 && n_dst >= 0 && n_src >= 0, "hgnn_plan_dims: negative size");
    HGNN_REQUIRE(n_rows < ((int64_t)1 << 31) - 1024 && n_dst < ((int64_t)1 << 31) - 1024 &&
                     n_src < ((int64_t)1 << 31) - 1024,
                 "hgnn_plan_dims: sizes must fit int32 (rows=%lld dst=%lld src=%lld)",
                 (long long)n_rows, (long long)n_dst, (long long)n_src);
    if (chunk <= 0) {
        // a quarter of one wave's share of the rows, 4096 waves in flight
        // (256 CUs x 16 waves): cdna_hip_programming.md Appendix B, scatter/gather.
        int64_t share = n_rows / 4096 / 4;
        chunk = (int32_t)(share < 32 ? 32 : (share > 512 ? 512 : share));
    }
    plan->n_rows = n_rows;
    plan->n_dst = n_dst;
    plan->n_src = n_src;
    plan->chunk = chunk;
    plan->max_work = n_dst + n_rows / chunk + 1;
    plan->max_split = n_rows / (chunk + 1) + 1;
    plan->max_partial = 2 * (n_rows / chunk) + 2;
    HGNN_REQUIRE(plan->max_work < ((int64_t)1 << 31), "hgnn_plan_dims: too many work items");
    return HGNN_OK;
}

extern "C" int hgnn_plan_workspace_bytes(int64_t n_rows, int64_t n_dst, size_t* bytes) {
    HGNN_REQUIRE(bytes != nullptr, "hgnn_plan_workspace_bytes: bytes is NULL");
    PlanScratch s;
    int rc = plan_scratch_layout(n_rows, n_dst, &s, nullptr);
    if (rc != HGNN_OK) return rc;
    *bytes = s.total;
    return HGNN_OK;
}

extern "C" int hgnn_plan_build(const int64_t* dst_index, const int64_t* gather_index,
                               hgnn_plan* plan, void* workspace, size_t workspace_bytes,
                               hgnn_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    HGNN_REQUIRE(plan != nullptr, "hgnn_plan_build: plan is NULL");
    const int64_t M = plan->n_rows, N = plan->n_dst, R = plan->n_src;
    HGNN_REQUIRE(M == 0 || dst_index != nullptr, "hgnn_plan_build: dst_index is NULL");
    HGNN_REQUIRE(plan->chunk > 0 && plan->max_work >= N + M / plan->chunk + 1,
                 "hgnn_plan_build: plan dims not initialised (call hgnn_plan_dims)");
    HGNN_REQUIRE(plan->counts && plan->rowptr && (N == 0 || (plan->wi_begin && plan->wi_end && plan->wi_target && plan->wi_dst)) &&
                     plan->split_dst && plan->split_pbegin &&
                     (M == 0 || (plan->perm && plan->src_row && plan->dst32)),
                 "hgnn_plan_build: a plan array pointer is NULL");
    PlanScratch s;
    int rc = plan_scratch_layout(M, N, &s, stream);
    if (rc != HGNN_OK) return rc;
    if (workspace_bytes < s.total || (s.total > 0 && workspace == nullptr)) {
        set_error("hgnn_plan_build: workspace too small (%zu < %zu)", workspace_bytes, s.total);
        return HGNN_ERR_WORKSPACE;
    }
    plan->has_gather = gather_index != nullptr ? 1 : 0;
    char* ws = (char*)workspace;
    int32_t* keys_in = (int32_t*)(ws + s.keys_in);
    int32_t* keys_out = (int32_t*)(ws + s.keys_out);
    int32_t* vals_in = (int32_t*)(ws + s.vals_in);
    Tri* tri_in = (Tri*)(ws + s.tri_in);
    Tri* tri_out = (Tri*)(ws + s.tri_out);
    void* temp = ws + s.temp;

    HGNN_CHECK_HIP(hipMemsetAsync(plan->counts, 0, 8 * sizeof(int32_t), stream));
    HGNN_CHECK_HIP(hipMemsetAsync(plan->split_pbegin, 0, sizeof(int32_t), stream));
    if (M > 0) {
        k_plan_prepare<<<(unsigned)ceil_div(M, 256), 256, 0, stream>>>(
            dst_index, gather_index, M, N, R, keys_in, vals_in, plan->dst32, plan->counts);
        size_t tb = s.temp_bytes;
        HGNN_CHECK_HIP(rocprim::radix_sort_pairs(temp, tb, keys_in, keys_out, vals_in, plan->perm,
                                                 (size_t)M, 0u, (unsigned)key_bits(N), stream));
    }
    k_plan_rowptr<<<(unsigned)ceil_div(M + 1, 256), 256, 0, stream>>>(
        keys_out, plan->perm, gather_index, M, N, plan->rowptr, plan->src_row, plan->counts);
    if (N > 0) {
        k_plan_chunks<<<(unsigned)ceil_div(N, 256), 256, 0, stream>>>(plan->rowptr, N, plan->chunk, tri_in);
        size_t tb = s.temp_bytes;
        HGNN_CHECK_HIP(rocprim::exclusive_scan(temp, tb, tri_in, tri_out, Tri{0, 0, 0}, (size_t)N,
                                               TriPlus(), stream));
        k_plan_fill<<<(unsigned)ceil_div(N, 256), 256, 0, stream>>>(plan->rowptr, tri_out, N,
                                                                    plan->chunk, *plan);
    }
    HGNN_CHECK_HIP(hipGetLastError());
    return HGNN_OK;
}

extern "C" int hgnn_index_to_i32(const int64_t* idx, int64_t M, int64_t limit, int32_t* out,
                                 int32_t* err_flag, hgnn_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    HGNN_REQUIRE(M >= 0 && limit >= 0 && limit < ((int64_t)1 << 31), "hgnn_index_to_i32: bad sizes");
    if (M == 0) return HGNN_OK;
    HGNN_REQUIRE(idx != nullptr && out != nullptr, "hgnn_index_to_i32: NULL pointer");
    k_index_to_i32<<<(unsigned)ceil_div(M, 256), 256, 0, stream>>>(idx, M, limit, out, err_flag);
    HGNN_CHECK_HIP(hipGetLastError());
    return HGNN_OK;
}
