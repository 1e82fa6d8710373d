/*
 * hgnn_hip.h -- C ABI of libhgnn_hip.so: the MI355X (gfx950) message-passing
 * engine behind the reference's operator interface.
 *
 * Every entry point takes plain device pointers, sizes and a hipStream_t (as
 * void*), launches asynchronously on that stream, performs no allocation and no
 * host synchronisation (graph-capturable), and returns an int status
 * (HGNN_OK == 0).  hgnn_last_error() returns a thread-local message for the
 * last non-zero status.  No torch types appear here: a binding only needs
 * `tensor.data_ptr()` and the current stream handle.
 *
 * Reference interfaces these replace (paths relative to the reference root,
 * clairesonglee/HierarchicalGNN):
 *
 *   hgnn_plan_build / hgnn_segment_reduce_f32
 *       torch_scatter.scatter_add(src, index, dim=0, dim_size=N) as called at
 *       Modules/gnn_utils.py:50 and :125 (edge -> node aggregation, "K1"),
 *       Modules/gnn_utils.py:143 (weighted superedge -> supernode, "K4"),
 *       and, with a gather index, the fused expressions
 *       scatter_add(w * X[g], d, dim_size) at Modules/gnn_utils.py:124 ("K2"),
 *       :142 ("K3") and, with a per-row L1 scale,
 *       BipartiteClassification/Models/HGNN_GMM.py:269 ("K5").
 *   hgnn_spread_rows_f32
 *       backward of scatter_add (grad_src[e] = grad_out[index[e]]) and the gathers
 *       nodes[graph[0]], nodes[graph[1]] (Modules/gnn_utils.py:61) in destination order.
 *   hgnn_gather_rows_f32
 *       the row gathers nodes[graph[0]], nodes[graph[1]] at
 *       Modules/gnn_utils.py:61,134,152 ("K6"); also the backward of scatter_add.
 *   hgnn_edge_dot_f32
 *       backward of the weighted forms w.r.t. the weights (autograd of
 *       Modules/gnn_utils.py:124,142,143).
 *   hgnn_mlp_forward_f32
 *       the make_mlp Sequential (Modules/utils.py:169-196) of the edge / node networks
 *       fused with the concat + gathers feeding it and the skip connection
 *       (Modules/gnn_utils.py:52-53, :61-62): see the section at the end.
 */
#ifndef HGNN_HIP_H
#define HGNN_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define HGNN_OK 0
#define HGNN_ERR_INVALID_ARG 1
#define HGNN_ERR_HIP 2
#define HGNN_ERR_WORKSPACE 3
#define HGNN_ERR_UNSUPPORTED 4

#define HGNN_ABI_VERSION 23

typedef void* hgnn_stream_t; /* hipStream_t */

/* indices into hgnn_plan.counts (device int32[8]) */
#define HGNN_CNT_WORK 0     /* number of work items                         */
#define HGNN_CNT_SPLIT 1    /* number of destinations whose list was split  */
#define HGNN_CNT_PARTIAL 2  /* number of partial rows                       */
#define HGNN_CNT_ERR 3      /* !=0: an index was out of range (row dropped) */
#define HGNN_CNT_VALID 4    /* rows with a valid destination (and source)   */
#define HGNN_CNT_UNSORTED 5 /* !=0: the index was not already sorted by destination */

/*
 * A destination-sorted aggregation plan for one (index, dim_size) pair.  Built
 * once per event, reused by every cell (the topology is constant across the
 * 14 / 6+6 iterations of a forward: EdgeClassifier/Models/IN.py:87-88,
 * BipartiteClassification/Models/HGNN_GMM.py:93-94,:275-284).  All arrays are
 * device int32, allocated by the caller with the sizes hgnn_plan_dims() gives.
 */
typedef struct hgnn_plan {
    int64_t n_rows;    /* M: rows of `index` (edges)                              */
    int64_t n_dst;     /* N: dim_size                                             */
    int64_t n_src;     /* R: rows of the source table (== M when no gather index) */
    int32_t chunk;     /* longest list one wave sums; longer lists are split      */
    int32_t has_gather;
    int64_t max_work;     /* capacity of wi_*                                     */
    int64_t max_split;    /* capacity of split_dst; split_pbegin has +1           */
    int64_t max_partial;  /* rows of the partial-sum workspace                    */
    int32_t* perm;        /* [M]  original position of the p-th dst-sorted row (stable) */
    int32_t* src_row;     /* [M]  source-table row read at sorted position p; the caller may set it
                           *      to NULL when counts[HGNN_CNT_UNSORTED]==0 (index already sorted,
                           *      no gather): the reduce then streams rows p = begin..end contiguously */
    int32_t* dst32;       /* [M]  destination of ORIGINAL position e, -1 if invalid */
    int32_t* rowptr;      /* [N+1] CSR by destination over sorted positions       */
    int32_t* wi_begin;    /* [max_work]                                           */
    int32_t* wi_end;      /* [max_work]                                           */
    int32_t* wi_target;   /* [max_work] >=0: output row; <0: ~partial row         */
    int32_t* wi_dst;      /* [max_work] destination of the item (also for chunks) */
    int32_t* split_dst;   /* [max_split]                                          */
    int32_t* split_pbegin;/* [max_split+1] partial-row range of each split dst    */
    int32_t* counts;      /* [8]                                                  */
} hgnn_plan;

int hgnn_abi_version(void);
const char* hgnn_last_error(void);

/* sizeof(hgnn_plan) / sizeof(hgnn_mlp_desc) as compiled: lets a foreign-language binding
 * verify its struct mirror. */
int hgnn_sizeof_plan(void);
int hgnn_sizeof_mlp_desc(void);

/* Process-wide switches (diagnostics / A-B measurements).  NOT THREAD-SAFE: plain static ints read at launch time;
 * set them from the thread that launches, before the launches they should affect, never concurrently with one.
 * The defaults are the measured-best variants; everything a round measured slower has been removed from the library
 * (tools/experimental/ keeps the sources of the larger null results, DESIGN.md appendix A the numbers).
 *   "nt_loads"     1 (default): non-temporal loads of once-read source rows in K1..K5
 *   "nt_stores"    0 (default): non-temporal stores of gathered rows (K6)
 *   "mlp_split_variant" schedule of the feature-split bf16 MLP: -1 (default) per shape, 0 counted per-fragment
 *                  waits, 2 one wait per k-chunk ("burst")
 *   "mlp_ablate"   DIAGNOSTIC bits, results are WRONG.  fp32 / bf16 kernels: 1 skip LayerNorm/act, 2 skip weight
 *                  DMA, 4 skip barriers (tools/tune_mlp.py); feature-split bf16 kernel: 1 weights from chunk 0 only,
 *                  2 skip LayerNorm/act, 4 load only the first input panel, 8 skip the per-panel barriers, 16 LDS
 *                  operand reads from chunk 0 only (tools/tune_mlp_split.py)
 *   "mlp_split3_rows128" tile shape of hgnn_mlp_forward_f32_split3 for K -> 512 -> 256 at M >= 65,536: 1 (default)
 *                  128-row tiles, 8 waves, hidden rows consumed in two K-halves; 0 the 64-row kernel that serves every
 *                  other shape; 2 EXPERIMENTAL 64-row tiles, 4 waves, two workgroups per CU -- fastest, but one of two
 *                  equivalent builds returned wrong elements, cause not found: never a default, and
 *                  refused unless the environment variable HGNN_EXPERIMENTAL is set
 *   "mlp_split3_one_wg"  DIAGNOSTIC: 1 launches one persistent workgroup per CU where two fit (results must be
 *                  bitwise the same: tests/test_gpu_split3.py)
 * Any other name is an error (HGNN_ERR_INVALID_ARG). */
int hgnn_set_option(const char* name, int value);

/* Fills n_rows/n_dst/n_src/chunk/max_* of `plan` (pointers untouched).
 * chunk <= 0 selects the default rule (quarter of a wave's share of the rows,
 * clamped to [32, 512]). */
int hgnn_plan_dims(int64_t n_rows, int64_t n_dst, int64_t n_src, int32_t chunk, hgnn_plan* plan);

/* Bytes of scratch hgnn_plan_build needs (device memory, any 256-B aligned). */
int hgnn_plan_workspace_bytes(int64_t n_rows, int64_t n_dst, size_t* bytes);

/* dst_index: int64[M] destinations (PyG edge_index row / bipartite_graph row).
 * gather_index: int64[M] source-table rows, or NULL (row e of the table is edge e).
 * Out-of-range entries are dropped and flagged in counts[HGNN_CNT_ERR]. */
int hgnn_plan_build(const int64_t* dst_index, const int64_t* gather_index, hgnn_plan* plan,
                    void* workspace, size_t workspace_bytes, hgnn_stream_t stream);

/* out[d,:] = sum_{p in list(d)} weight[perm[p]] * row_scale[src_row[p]] * src[src_row[p],:]
 * weight (float[M], by ORIGINAL position) and row_scale (float[n_src]) may be NULL.
 * out: float[N,F] (every row written, zeros for empty lists);
 * partial: float[max_partial,F] scratch.  Deterministic: fixed summation order. */
int hgnn_segment_reduce_f32(const hgnn_plan* plan, const float* src, int32_t F,
                            const float* weight, const float* row_scale,
                            float* out, float* partial, hgnn_stream_t stream);

/* out[e,:] = weight[e] * row_scale[idx[e]] * table[idx[e],:]   (idx[e] < 0 -> zeros)
 * idx: int32[M]; weight float[M] / row_scale float[table_rows] may be NULL. */
int hgnn_gather_rows_f32(const float* table, int64_t table_rows, int32_t F,
                         const int32_t* idx, int64_t M,
                         const float* weight, const float* row_scale,
                         float* out, hgnn_stream_t stream);

/* The transpose of hgnn_segment_reduce_f32 (= backward of scatter_add, and the row gather
 * table[index] when the plan was built on `index`):
 *     out[e,:] = weight[e] * table[dst(e),:]      for every row e of the plan
 * walked in DESTINATION order: each table row is read once and written to the rows of its
 * list, so HBM sees one pass of 16-B stores over out[M,F] instead of M random row reads.
 * weight (float[M], by original position) may be NULL.  Rows whose index was out of range
 * are not written. */
int hgnn_spread_rows_f32(const hgnn_plan* plan, const float* table, int32_t F,
                         const float* weight, float* out, hgnn_stream_t stream);

/* bf16 feature rows (BASELINE config 4 dtype): same semantics as the _f32 entry points with
 * bf16 src/table/out (16-byte aligned, F a multiple of 8, F <= 512), fp32 accumulation and one
 * rounding per output element; weight / row_scale stay fp32; `partial` is fp32[max_partial,F]. */
int hgnn_segment_reduce_bf16(const hgnn_plan* plan, const void* src, int32_t F, const float* weight,
                             const float* row_scale, void* out, float* partial, hgnn_stream_t stream);
int hgnn_spread_rows_bf16(const hgnn_plan* plan, const void* table, int32_t F, const float* weight,
                          void* out, hgnn_stream_t stream);
int hgnn_gather_rows_bf16(const void* table, int64_t table_rows, int32_t F, const int32_t* idx, int64_t M,
                          const float* weight, void* out, hgnn_stream_t stream);

/* out[e] = sum_f A[ai[e],f] * B[bi[e],f];  ai/bi int32[M] or NULL (identity);
 * negative index -> 0. */
int hgnn_edge_dot_f32(const float* A, const int32_t* ai, int64_t a_rows,
                      const float* B, const int32_t* bi, int64_t b_rows,
                      int32_t F, int64_t M, float* out, hgnn_stream_t stream);

/* int64 -> int32 index conversion with range check (out-of-range -> -1, err flag set) */
int hgnn_index_to_i32(const int64_t* idx, int64_t M, int64_t limit, int32_t* out,
                      int32_t* err_flag, hgnn_stream_t stream);

/* Fixed-radius kNN (exact, tiled brute force) in a D<=16 dimensional space: for every query
 * the <=K nearest points with squared distance < radius^2, ascending (ties: lower index first),
 * idx -1 padded; dist2_out (may be NULL) holds squared distances, -1 for padding.
 * Replaces frnn.frnn_grid_points as called by find_neighbors (Modules/utils.py:228-239) from
 * DynamicGraphConstruction.forward (Modules/gnn_utils.py:194).  K in {1-6,8,10,12,16,20,32}. */
int hgnn_knn_radius_f32(const float* query, int64_t nq, const float* points, int64_t np, int32_t D,
                        int32_t K, float radius, int64_t* idx_out, float* dist2_out, hgnn_stream_t stream);

/* Same search with (i) the radius optionally read from DEVICE memory (`radius_dev` != NULL: the module's
 * knn_radius buffer, Modules/gnn_utils.py:181,205 -- no host read of it) and (ii) a workspace that lets a
 * search with few queries (the S x S super graph) split every query's candidates across workgroups and
 * merge the per-slice lists afterwards (same result, ties included).  workspace may be NULL (no split). */
int hgnn_knn_workspace_bytes(int64_t nq, int64_t np, int32_t K, size_t* bytes);
int hgnn_knn_radius_ws_f32(const float* query, int64_t nq, const float* points, int64_t np, int32_t D,
                           int32_t K, float radius, const float* radius_dev, int64_t* idx_out, float* dist2_out,
                           void* workspace, size_t workspace_bytes, hgnn_stream_t stream);

/* ------------------------------------------------------------------------
 * The hierarchy decision of HierarchicalGNNBlock.clustering
 * (BipartiteClassification/Models/HGNN_GMM.py:162-234) without host round trips.
 *
 * hgnn_gmm2_fit_f32: 2-component 1-D Gaussian mixture of v[M] (the atanh edge likelihoods, :188-189), what
 *   the reference gets from sklearn GaussianMixture(2).fit on the CPU (:192).  Deterministic 2-means start
 *   from the data extremes, then at most max_iter EM passes with sklearn's stopping rule (change of the mean
 *   log-likelihood < tol; reg_covar added to the variances) evaluated ON THE DEVICE: all passes are enqueued,
 *   the ones after convergence return immediately.  state: double[HGNN_GMM_STATE] =
 *   {w0, w1, mu0, mu1, var0, var1, previous lower bound, converged, EM passes run, min, max, c0, c1, cut,
 *    last lower bound, -};  partials: double[HGNN_GMM_BLOCKS * 8] scratch;  ticket: one uint32 scratch.
 * hgnn_gmm2_cut_f32: the cut x between the two means where sigmoid(r) P(left|x) = sigmoid(-r) P(right|x)
 *   (:162-170, scipy fsolve in the reference; bisection here) -> state[13]; then, on the module's DEVICE
 *   buffer score_cut[1] (:157): inf -> middle of the means (:196-197); in training mode, if the cut lies
 *   between the means, score_cut = momentum * score_cut + (1 - momentum) * cut (:201-208).
 * hgnn_cc_labels: weakly connected components over vertices 0..n-1 of the edges (src[e], dst[e]) whose
 *   score[e] >= *cut (score == cut == NULL: all edges), :212-221 (cugraph in the reference).  labels[v] =
 *   smallest vertex id of v's component (v itself if isolated); present[v] = 1 iff v is an endpoint of a
 *   kept edge.  Lock-free union-find: one pass over the edges, one compression pass; no iteration to
 *   convergence and therefore no host read.
 * ------------------------------------------------------------------------ */
#define HGNN_GMM_STATE 16
#define HGNN_GMM_BLOCKS 1024
int hgnn_gmm2_fit_f32(const float* v, int64_t M, int32_t max_iter, float tol, float reg_covar, double* state,
                      double* partials, uint32_t* ticket, hgnn_stream_t stream);
int hgnn_gmm2_cut_f32(double* state, float granularity, int32_t training, float momentum, float* score_cut,
                      hgnn_stream_t stream);
int hgnn_cc_labels(const int64_t* src, const int64_t* dst, int64_t M, int64_t n, const float* score,
                   const float* cut, int32_t* labels, int32_t* present, hgnn_stream_t stream);

/* ------------------------------------------------------------------------
 * Fused gather -> concat -> Linear -> LayerNorm -> act -> ... -> (+skip) MLP
 * (fp32 MFMA).  One descriptor describes up to 3 concatenated input segments,
 * each an optional row gather of a table, and up to 3 Linear layers with
 * LayerNorm + activation after each (make_mlp with layer_norm=True,
 * reference Modules/utils.py:169-196), as instantiated for the edge / node /
 * supernode / superedge networks at Modules/gnn_utils.py:22-41 and :77-115,
 * fused with the concat + gathers feeding them (:52,:61,:126,:134,:144,:152)
 * and the skip connection (:53,:62,:126,:134,:144,:152).  "K6+K7".
 * ------------------------------------------------------------------------ */
#define HGNN_ACT_NONE 0
#define HGNN_ACT_GELU 1 /* erf form, nn.GELU() default */
#define HGNN_ACT_TANH 2
#define HGNN_ACT_RELU 3

typedef struct hgnn_mlp_desc {
    int32_t n_seg;               /* 1..3 input segments, concatenated in order    */
    const float* seg_table[3];   /* [rows_i, seg_width[i]]                        */
    const int32_t* seg_index[3]; /* int32[M] row gather, or NULL (row e)          */
    int32_t seg_width[3];
    int32_t n_layers;            /* 1..3 (1: see hgnn_mlp_supported, single layers) */
    const float* W[3];           /* Linear weight [out_i, in_i] row-major (torch) */
    const float* b[3];           /* [out_i]                                       */
    const float* ln_w[3];        /* LayerNorm affine, NULL = no LayerNorm         */
    const float* ln_b[3];
    int32_t width[4];            /* in, h1, (h2), out                             */
    int32_t act[3];              /* HGNN_ACT_* after each layer                   */
    float ln_eps;
    const float* skip;           /* [M, out] added to the result, or NULL         */
    int64_t M;                   /* rows                                          */
    int32_t w0_cols;             /* columns stored per row of W[0]: 0 = width[0]; must be 16 (zero
                                  * padded) in small-K mode, i.e. when width[0] <= 16 is not a
                                  * multiple of 16 (node / edge encoders, K = 3 / 6: IN.py:26-46)  */
    int32_t w_last_rows;         /* rows stored in the last W / b (and ln_w / ln_b): 0 = width[n].  Must be 32
                                  * (zero padded) for a head = plain last layer of width[n] <= 32 real outputs
                                  * (width-1 classifiers IN.py:107-115, HGNN_GMM.py:313-321; emb_dim-wide
                                  * embedding head HGNN_GMM.py:74-82), and P = width[1] / 2 (zero padded) for a
                                  * LayerNorm'ed last layer of P-16 < width[n] < P outputs (supernode encoder,
                                  * L - emb_dim wide, HGNN_GMM.py:117): statistics and stores use width[n] */
    float* save_pre[3];          /* optional: [M, width[l+1]] buffers that receive layer l's output
                                  * BEFORE LayerNorm/activation (what a backward pass needs; hidden
                                  * activations are recomputed from it).  NULL = not saved.         */
    int32_t n_pre;               /* 0..2 PRE-PROJECTED gathered segments (fp32 kernel and bf16 split kernel): a gathered
                                  * segment table[idx] enters the first Linear linearly,
                                  *   W_s table[idx[e]] = (table W_s^T)[idx[e]],
                                  * so the caller may project the (few) table rows once, P_s = table W_s^T
                                  * [rows, width[1]], leave the segment out of seg_* / W[0] / width[0], and
                                  * hand P_s here: the kernel starts row e's accumulators at
                                  * b + sum_s P_s[pre_index[s][e]].  For nodes[graph[k]] (N rows, M = 16.7 N
                                  * edges) this removes 2/3 of the edge network's first-layer FLOPs.  */
    const float* pre_table[2];   /* [rows_s, width[1]], 16-byte aligned (bf16 rows for
                                  * hgnn_mlp_forward_bf16_split)                    */
    const int32_t* pre_index[2]; /* int32[M]                                      */
} hgnn_mlp_desc;

/* 1 if hgnn_mlp_forward_f32 has an instantiation for this descriptor (host-only check):
 *   cell networks / encoders: widths K -> 2L (-> 2L) -> L, LayerNorm on every layer,
 *       L in {32, 64, 128, 256}; every segment a multiple of 16 floats wide, or K <= 16 in
 *       small-K mode (W[0] zero-padded to 16 columns, w0_cols = 16);
 *   heads: K -> H -> H -> w, 1 <= w <= 32, LayerNorm + activation on the two hidden layers, plain last
 *       layer (ln_w[2] = NULL, act[2] = NONE) stored zero-padded as 32 rows (w_last_rows = 32),
 *       H in {64, 128, 256, 512}, no skip; out is float[M, w];
 *   single layers: n_layers = 1, K -> o with o in {512, 1024}, LayerNorm + activation (+ skip): the pieces of an
 *       fp32 MLP at latent 512 (its 1024-wide hidden layer is 256 accumulators per lane: one launch per layer, the
 *       hidden rows make one trip through HBM);
 *   narrow encoders: K -> 2P -> 2P -> o with P in {32, 64, 128, 256}, P-16 < o < P, o % 4 == 0, the last
 *       layer's W / b / ln_w / ln_b zero padded to P rows (w_last_rows = P), no skip, no save_pre;
 *       LayerNorm over the o real features; out is float[M, o]. */
int hgnn_mlp_supported(const hgnn_mlp_desc* d);

/* out[M, L] = MLP(cat_i seg_i[idx_i]) (+ skip).  No workspace; hidden activations stay in
 * registers.  Negative gather indices read row 0 (callers validate indices at plan build). */
int hgnn_mlp_forward_f32(const hgnn_mlp_desc* d, float* out, hgnn_stream_t stream);

/* bf16 variant (BASELINE config 4 dtype) on v_mfma_f32_16x16x32_bf16.  Same descriptor, read as:
 * seg_table / skip / out = bf16 rows; W[l] = bf16 [out][in] row-major, and for l >= 1 with its
 * COLUMNS stored in MFMA k-slot order: column 32c + 8g + j holds input feature
 * 32c + (j < 4 ? 4g + j : 16 + 4g + j - 4)  (c = k-block, g = 0..3, j = 0..7);
 * b / ln_w / ln_b fp32; accumulation and LayerNorm in fp32.  Supported: K -> 2L (-> 2L) -> L,
 * LayerNorm on every layer, L in {32, 64, 128, 256}, every segment a multiple of 32 wide. */
int hgnn_mlp_supported_bf16(const hgnn_mlp_desc* d);
int hgnn_mlp_forward_bf16(const hgnn_mlp_desc* d, void* out, hgnn_stream_t stream);

/* bf16, feature-split kernel for the wide layers (L in {128, 256, 512}; config 4 is L = 512): a
 * workgroup owns 64 rows, each of its 4 waves a quarter of every layer's features; hidden
 * activations cross waves through LDS, weights go straight from L2 to registers.  Same descriptor
 * and arithmetic as hgnn_mlp_forward_bf16, except that W[l] (bf16) is stored in MFMA A-FRAGMENT
 * ORDER, natural feature order for every layer:
 *   element index = ((c * (F/16) + T) * 64 + lane) * 8 + i  holds  W[16T + lane%16][32c + 8(lane/16) + i]
 * (F = out features, c = 32-wide k-chunk, T = 16-feature tile, lane = 0..63, i = 0..7).
 * Supported: K -> 2L (-> 2L) -> L, LayerNorm on every layer, every segment a multiple of 128 wide; and single layers
 * (n_layers = 1) K -> o, o in {256, 512, 1024}, LayerNorm + activation (+ skip): the pieces from which the heads and
 * the encoder tails (bf16 latent mode) are chained.
 * save_pre[l] (optional) receives layer l's pre-LayerNorm rows as BF16 [M, width[l+1]] (8-byte aligned): the
 * forward of the bf16 training path. */
int hgnn_mlp_supported_bf16_split(const hgnn_mlp_desc* d);
int hgnn_mlp_forward_bf16_split(const hgnn_mlp_desc* d, void* out, hgnn_stream_t stream);

/* fp32 rows, SPLIT-bf16 arithmetic (the Python layer's default path of the fp32 MLPs at latent 128 / 256: K -> 2L (-> 2L) -> L, and
 * K -> H -> H with H in {256, 512} = the two hidden layers of a score head, whose plain last Linear the caller applies;
 * LayerNorm on every layer, every segment a multiple of 128 wide, n_pre allowed, save_pre optional (fp32 dumps)): every fp32 operand of
 * the GEMMs is used as hi + mid with hi = bf16(x), mid = bf16(x - hi), and  x.w ~= hi.hi + mid.hi + hi.mid  runs as
 * three v_mfma_f32_16x16x32_bf16 (exact products, fp32 accumulation) instead of one fp32 MFMA at 1/16 of the rate.
 * Bias, LayerNorm, exact-erf GELU / tanh, skip and every row in HBM stay fp32.  Error at model level 2e-5 against the
 * reference's scores on BASELINE config 2 (north_star's bar: 1e-4).  Same descriptor as hgnn_mlp_forward_f32 (fp32
 * segment tables, fp32 pre_table rows of width[1] floats, fp32 skip / out), except W[l]: bf16, the layer's split
 * stream -- per 32-wide k-chunk c of the (kept) input columns the chunk's W_hi columns followed by its W_mid columns,
 * i.e. a [out, 2 K] matrix whose 32-column chunks 2c / 2c + 1 are hi / mid -- in the A-fragment order of
 * hgnn_mlp_forward_bf16_split. */
int hgnn_mlp_supported_f32_split3(const hgnn_mlp_desc* d);
int hgnn_mlp_forward_f32_split3(const hgnn_mlp_desc* d, float* out, hgnn_stream_t stream);

/* out[M, N] = x[M, K] . W^T (+ skip), all fp32 in HBM, the product as split-bf16 (same arithmetic and the same
 * weight stream layout as hgnn_mlp_forward_f32_split3: w_split = the split stream of W [N, K]): the M-row data-gradient
 * GEMMs of the fp32 training backward.  K a multiple of 128, N in {256, 512}; skip [M, N] or NULL. */
int hgnn_linear_f32_split3(const float* x, int64_t M, int32_t K, const void* w_split, int32_t N,
                           const float* skip, float* out, hgnn_stream_t stream);

/* The pre-projections of an edge update in ONE launch: out_s[M, N] = x[M, K] . W_s^T for s = 0 (and 1, when w_split1 /
 * out1 are given) over the same input rows (x = the node table, W_s = the first Linear's column block of gathered
 * segment s; hgnn_mlp_desc.pre_table).  Same split-bf16 arithmetic as hgnn_mlp_forward_f32_split3 itself (three
 * products), same weight stream layout as hgnn_linear_f32_split3.  K a multiple of 128, N in {256, 512}. */
int hgnn_project_f32_split3(const float* x, int64_t M, int32_t K, const void* w_split0, const void* w_split1,
                            int32_t N, float* out0, float* out1, hgnn_stream_t stream);

/* LayerNorm + activation of one make_mlp layer (Modules/utils.py:169-196: Linear -> LayerNorm ->
 * act) over rows z[M, W] (the Linear's output, as dumped by hgnn_mlp_forward_f32's save_pre), one
 * pass each -- the elementwise half of the fused MLP's backward (the GEMM half is the library's):
 *   forward :  out[r]    = act(gamma * (z[r] - mean_r) * rstd_r + beta)
 *   backward:  grad_z[r] = d/dz of the above applied to grad_out[r]; and per-workgroup column sums
 *              partials[b][0][c] = sum_r grad_y * xhat (dgamma), [b][1][c] = sum_r grad_y (dbeta),
 *              [b][2][c] = sum_r grad_z (gradient of the Linear's bias); the caller adds the
 *              HGNN_LN_ACT_BLOCKS partial rows (deterministic, no atomics).
 * W in {64, 128, 256, 512, 1024}; act = HGNN_ACT_*; exact-erf GELU as in the forward kernels. */
#define HGNN_LN_ACT_BLOCKS 1024
int hgnn_ln_act_forward_f32(const float* z, int64_t M, int32_t W, const float* gamma, const float* beta,
                            int32_t act, float eps, float* out, hgnn_stream_t stream);
int hgnn_ln_act_backward_f32(const float* z, const float* grad_out, int64_t M, int32_t W, const float* gamma,
                             const float* beta, int32_t act, float eps, float* grad_z,
                             float* partials /* [HGNN_LN_ACT_BLOCKS][3][W] */, hgnn_stream_t stream);

/* ------------------------------------------------------------------------
 * bf16 training path (BASELINE config 4 dtype): backward of the make_mlp Linear layers
 * (Modules/utils.py:169-196 as instantiated at Modules/gnn_utils.py:22-41, :77-115) under the reference's
 * autograd (edge_classifier_base.py:113-128 trains every configuration).
 *
 * hgnn_wgrad_bf16: weight gradient  out[ho, hi] = sum_m A[m, ho] * B[m, hi]  (A = dz [M, Ho], B = the layer's
 *   input rows [M, Hi], both bf16 row-major with row strides lda / ldb in ELEMENTS (multiples of 8, so that a
 *   column slice of a wider matrix can be passed); out fp32 [Ho, ldo]).  Hand-written split-K bf16-MFMA kernel,
 *   fp32 accumulation, partial sums combined in slice order (deterministic, no atomics).  workspace: device
 *   scratch of hgnn_wgrad_workspace_bytes(M, Ho, Hi) bytes.  Ho, Hi multiples of 8.
 * hgnn_ln_act_{forward,backward}_bf16: hgnn_ln_act_*_f32 with bf16 rows (z, grad_out, out / grad_z); fp32
 *   statistics and arithmetic, fp32 gamma / beta / partials.
 */
int hgnn_ln_act_forward_bf16(const void* z, int64_t M, int32_t W, const float* gamma, const float* beta,
                             int32_t act, float eps, void* out, hgnn_stream_t stream);
int hgnn_ln_act_backward_bf16(const void* z, const void* grad_out, int64_t M, int32_t W, const float* gamma,
                              const float* beta, int32_t act, float eps, void* grad_z,
                              float* partials /* [HGNN_LN_ACT_BLOCKS][3][W] */, hgnn_stream_t stream);
int hgnn_wgrad_workspace_bytes(int64_t M, int32_t Ho, int32_t Hi, size_t* bytes);
int hgnn_wgrad_bf16(const void* A, int64_t lda, const void* B, int64_t ldb, int64_t M, int32_t Ho, int32_t Hi,
                    float* out, int64_t ldo, float* colsum /* [Ho] = sum_m A[m, ho] (the bias gradient), or NULL */,
                    void* workspace, size_t workspace_bytes, hgnn_stream_t stream);

/* the same weight gradient for FP32 rows dz [M, Ho] / a [M, Hi] (lda / ldb in floats, multiples of 4, 16-byte aligned
 * rows), the products as split-bf16 (hi.hi + mid.hi + hi.mid, exact products, fp32 accumulation; see
 * hgnn_mlp_forward_f32_split3): the M-row weight-gradient GEMMs of the fp32 training backward.  Workspace as
 * hgnn_wgrad_workspace_bytes. */
int hgnn_wgrad_f32_split3(const float* A, int64_t lda, const float* B, int64_t ldb, int64_t M, int32_t Ho, int32_t Hi,
                          float* out, int64_t ldo, float* colsum, void* workspace, size_t workspace_bytes,
                          hgnn_stream_t stream);

/* hgnn_mlp_backward_layer_bf16: the hand-written DATA gradient of one Linear of the bf16 training path, fused with
 * what follows it in the backward:
 *   z_prev != NULL ("LayerNorm form"):  da = dz[M,K] . W[K,N];  out = dz' = dLayerNorm(act'(LN(z_prev)) * da)  with
 *       z_prev [M,N] the previous layer's pre-LayerNorm rows (bf16 dumps of the forward), ln_w / ln_b / act / eps that
 *       layer's LayerNorm + activation; a_prev (optional) receives act(LN(z_prev)) (the rows hgnn_wgrad_bf16 needs
 *       for W's gradient); partials: float [HGNN_MLP_BWD_BLOCKS][2][N] per-workgroup column sums of dgamma / dbeta
 *       (the caller adds them; the bias gradient = column sums of dz' comes from hgnn_wgrad_bf16's colsum);
 *   z_prev == NULL ("input form"):      out = dz . W (+ skip): gradient of a direct input segment of the first
 *       layer, the skip connection's gradient (skip [M,N] bf16, may be NULL) added in the epilogue.
 * Wt_frag: W^T (the [N][K] matrix) in the MFMA A-fragment order of hgnn_mlp_forward_bf16_split.
 * K a multiple of 128, N in {128, 256, 512}; all rows bf16, fp32 accumulation and LayerNorm arithmetic; deterministic. */
#define HGNN_MLP_BWD_BLOCKS 512
int hgnn_mlp_backward_layer_supported_bf16(int32_t K, int32_t N);
int hgnn_mlp_backward_layer_bf16(const void* dz, int64_t M, int32_t K, int32_t N, const void* Wt_frag,
                                 const void* z_prev, const float* ln_w, const float* ln_b, int32_t act, float eps,
                                 const void* skip, void* out, void* a_prev, float* partials, hgnn_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* HGNN_HIP_H */
