// DIAGNOSTIC build of hgnn_mlp_forward_f32_split3 (edge-update shape: node segments pre-projected, K1 = 256 -> 512 -> 256,
// tile shape argv[1] in {r64, r128, r64x2}; argv[2]: optional start stagger in cycles per workgroup slot;
// M = 2M rows) with shader-clock stamps at the phase boundaries of every tile (see HGNN_STAMP in mlp_split3_f32.hip).
// Random data; the product library contains no stamp.  Build and run on the GPU box:
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -DHGNN_SPLIT3_STAMPS tools/split3_stamps.hip -o tools/split3_stamps
//   tools/split3_stamps > profiles/r03_split3_stamps.json
#include "../hierarchicalgnn_amd/csrc/mlp_split3_f32.hip"
#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <vector>

namespace hgnn {
void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vfprintf(stderr, fmt, ap);
    va_end(ap);
    fputc('\n', stderr);
}
}  // namespace hgnn

#define CK(x)                                                      \
    do {                                                           \
        hipError_t e = (x);                                        \
        if (e != hipSuccess) {                                     \
            fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e)); \
            return 1;                                              \
        }                                                          \
    } while (0)

static unsigned short bf16_of(float x) {
    unsigned u;
    memcpy(&u, &x, 4);
    return (unsigned short)((u + 0x7fff + ((u >> 16) & 1)) >> 16);
}

int main(int argc, char** argv) {
    const bool x2 = argc > 1 && !strcmp(argv[1], "r64x2");   // the K-half kernel on 64-row tiles, 4 waves, two workgroups per CU
    const bool wide = x2 || (argc > 1 && !strcmp(argv[1], "r128"));   // argv[1]: "r128" | "r64"; argv[2]: start stagger in cycles per slot   // the 128-row kernel (11 stamps, 4 waves)
    using namespace hgnn;
    const long long N = 120000, M = 2000000;
    const int L = 256, H = 512;
    srand(1);
    auto rnd = [] { return (float)rand() / RAND_MAX * 2.f - 1.f; };
    std::vector<float> edges((size_t)M * L), P((size_t)N * H), par(H, 0.01f), one(H, 1.f);
    for (size_t i = 0; i < P.size(); ++i) P[i] = rnd();
    for (size_t i = 0; i < edges.size(); ++i) edges[i] = P[i % P.size()] * 0.7f;
    // split streams: [chunk c][hi | mid][tile][lane][8] -- random bf16 values (timing only)
    std::vector<unsigned short> w0((size_t)H * 2 * L), w1((size_t)L * 2 * H);
    for (auto& v : w0) v = bf16_of(rnd() * 0.05f);
    for (auto& v : w1) v = bf16_of(rnd() * 0.05f);
    std::vector<int32_t> g0(M), g1(M);
    for (long long e = 0; e < M; ++e) {
        g1[e] = (int32_t)(e * N / M);          // destination-sorted layout: ~17 consecutive rows share a destination
        g0[e] = rand() % N;
    }
    float *d_edges, *d_out, *d_P0, *d_P1, *d_zero, *d_one;
    unsigned short *d_w0, *d_w1;
    int32_t *d_g0, *d_g1;
    CK(hipMalloc(&d_edges, edges.size() * 4));
    CK(hipMalloc(&d_out, edges.size() * 4));
    CK(hipMalloc(&d_P0, P.size() * 4));
    CK(hipMalloc(&d_P1, P.size() * 4));
    CK(hipMalloc(&d_w0, w0.size() * 2));
    CK(hipMalloc(&d_w1, w1.size() * 2));
    CK(hipMalloc(&d_g0, M * 4));
    CK(hipMalloc(&d_g1, M * 4));
    CK(hipMalloc(&d_zero, H * 4));
    CK(hipMalloc(&d_one, H * 4));
    CK(hipMemcpy(d_edges, edges.data(), edges.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(d_P0, P.data(), P.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(d_P1, P.data(), P.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(d_w0, w0.data(), w0.size() * 2, hipMemcpyHostToDevice));
    CK(hipMemcpy(d_w1, w1.data(), w1.size() * 2, hipMemcpyHostToDevice));
    CK(hipMemcpy(d_g0, g0.data(), M * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(d_g1, g1.data(), M * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(d_zero, par.data(), H * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(d_one, one.data(), H * 4, hipMemcpyHostToDevice));
    const int TEH = wide && !x2 ? 128 : 64, NWH = x2 ? 4 : 8, NSH = wide ? 12 : 9, NPH = wide ? 10 : 8;
    const long long n_tiles = (M + TEH - 1) / TEH;
    unsigned long long* d_st;
    const size_t n_st = (size_t)n_tiles * NWH * NSH;
    CK(hipMalloc(&d_st, n_st * 8));
    CK(hipMemset(d_st, 0, n_st * 8));
    f3::Args a;
    memset(&a, 0, sizeof a);
    a.seg_table[0] = a.seg_table[1] = a.seg_table[2] = d_edges;
    a.seg_width[0] = L;
    a.n_seg = 1;
    a.K1 = L;
    a.W[0] = d_w0;
    a.W[1] = d_w1;
    for (int l = 0; l < 2; ++l) {
        a.b[l] = d_zero;
        a.lnw[l] = d_one;
        a.lnb[l] = d_zero;
    }
    a.act[0] = HGNN_ACT_GELU;
    a.act[1] = HGNN_ACT_TANH;
    a.eps = 1e-5f;
    a.skip = d_edges;
    a.out = d_out;
    a.M = M;
    a.pre_table[0] = d_P0;
    a.pre_table[1] = d_P1;
    a.pre_index[0] = d_g0;
    a.pre_index[1] = d_g1;
    a.n_pre = 2;
    a.stamps = nullptr;
    a.stamp_tiles = 0;
    a.stagger = argc > 2 ? atoi(argv[2]) : 0;
    for (int i = 0; i < 3; ++i)
        if ((x2 ? f3::r64x2::launch_tile(a, 0) : wide ? f3::r128::launch_tile(a, 0) : f3::launch<8, 4, 2, 2>(a, 0)) != HGNN_OK) return 1;   // warm-up, clocks settle
    CK(hipDeviceSynchronize());
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    CK(hipEventRecord(e0));
    for (int i = 0; i < 5; ++i)
        if ((x2 ? f3::r64x2::launch_tile(a, 0) : wide ? f3::r128::launch_tile(a, 0) : f3::launch<8, 4, 2, 2>(a, 0)) != HGNN_OK) return 1;
    CK(hipEventRecord(e1));
    CK(hipDeviceSynchronize());
    float ms_plain = 0;
    CK(hipEventElapsedTime(&ms_plain, e0, e1));
    a.stamps = d_st;
    a.stamp_tiles = n_tiles;
    CK(hipEventRecord(e0));
    if ((x2 ? f3::r64x2::launch_tile(a, 0) : wide ? f3::r128::launch_tile(a, 0) : f3::launch<8, 4, 2, 2>(a, 0)) != HGNN_OK) return 1;
    CK(hipEventRecord(e1));
    CK(hipDeviceSynchronize());
    float ms_st = 0;
    CK(hipEventElapsedTime(&ms_st, e0, e1));
    std::vector<unsigned long long> st(n_st);
    CK(hipMemcpy(st.data(), d_st, n_st * 8, hipMemcpyDeviceToHost));
    // per phase: cycles from the tile's earliest stamp k to its earliest / latest stamp k+1 over the 8 waves; medians over tiles
    const char* names8[8] = {"projected_rows_P_phase", "first_panel_store_and_barrier", "layer1_gemm", "layer1_layernorm_gelu",
                             "hidden_planes_write_and_barrier", "output_gemm", "output_layernorm_tanh", "skip_add_and_stores"};
    const char* names10[10] = {"projected_rows_P_phase", "panel_pair_load_store_barrier", "layer1_gemm", "layer1_stats",
                               "half0_activate_write_barrier", "output_gemm_half0", "half1_activate_write_barriers",
                               "output_gemm_half1", "output_stats", "activate_skip_store"};
    const char** names = wide ? names10 : names8;
    printf("{\n \"kernel\": \"%s with HGNN_SPLIT3_STAMPS\", \"M\": %lld, \"tiles\": %lld,\n",
           x2 ? "r64x2::k_mlp_f32_split3_khalf<GELU,TANH>" : wide ? "r128::k_mlp_f32_split3_khalf<GELU,TANH>" : "k_mlp_f32_split3<8,4,2,2,GELU,TANH>", M, n_tiles);
    printf(" \"stagger_cycles_per_slot\": %d, \"ms_per_launch_without_stamp_writes\": %.4f, \"ms_with_stamp_writes\": %.4f,\n", a.stagger, ms_plain / 5, ms_st);
    std::vector<double> tile_total;
    std::vector<std::vector<double>> ph(NPH), skew(NPH + 1);
    for (long long t = 0; t < n_tiles; ++t) {
        unsigned long long lo[12], hi[12];
        bool ok = true;
        for (int k = 0; k <= NPH; ++k) {
            lo[k] = ~0ull;
            hi[k] = 0;
            for (int w = 0; w < NWH; ++w) {
                const unsigned long long v = st[((size_t)t * NWH + w) * NSH + k];
                if (v == 0) ok = false;
                lo[k] = std::min(lo[k], v);
                hi[k] = std::max(hi[k], v);
            }
        }
        if (!ok || hi[NPH] < lo[0]) continue;
        tile_total.push_back((double)(hi[NPH] - lo[0]));
        for (int k = 0; k < NPH; ++k) ph[k].push_back((double)(hi[k + 1]) - (double)(hi[k]));
        for (int k = 0; k <= NPH; ++k) skew[k].push_back((double)(hi[k] - lo[k]));
    }
    auto med = [](std::vector<double>& v) {
        if (v.empty()) return 0.0;
        std::nth_element(v.begin(), v.begin() + v.size() / 2, v.end());
        return v[v.size() / 2];
    };
    printf(" \"tiles_with_complete_stamps\": %zu, \"median_tile_cycles\": %.0f,\n", tile_total.size(), med(tile_total));
    printf(" \"median_phase_cycles_last_wave_to_last_wave\": {");
    for (int k = 0; k < NPH; ++k) printf("%s\"%s\": %.0f", k ? ", " : "", names[k], med(ph[k]));
    printf("},\n \"median_wave_skew_cycles_at_each_stamp\": [");
    for (int k = 0; k <= NPH; ++k) printf("%s%.0f", k ? ", " : "", med(skew[k]));
    const int mf = (wide && !x2) ? 768 : (x2 ? 768 : 384);   // MFMAs per GEMM and wave
    printf("],\n \"ideal\": {\"mfma_cycles_per_wave_layer1\": %d, \"mfma_cycles_per_wave_output\": %d, \"note\": \"16 cycles x %d MFMAs per GEMM and wave; two waves share a SIMD\"}\n}\n", 16 * mf, 16 * mf, mf);
    return 0;
}
