// DIAGNOSTIC build of hgnn_mlp_forward_bf16_rows128 with shader-clock stamps at the phase boundaries of every tile
// (layer-1 GEMM | LayerNorm/act epilogue | output GEMM | output epilogue), random data, edge-update shape
// (N = 120k node rows, M = 2M rows, 768 -> 512 -> 256).  Build and run:
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -DHGNN_ROWS128_STAMPS tools/rows128_stamps.hip -o tools/rows128_stamps
//   tools/rows128_stamps [ablate bits]
#include "../hierarchicalgnn_amd/csrc/mlp_rows128_bf16.hip"
#include <algorithm>
#include <cstring>
#include <cstdlib>
#include <vector>

namespace hgnn {
int g_opt_mlp_ablate = 0;
void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vfprintf(stderr, fmt, ap);
    va_end(ap);
    fputc('\n', stderr);
}
}  // namespace hgnn

#define CK(x)                                                                \
    do {                                                                     \
        hipError_t e = (x);                                                  \
        if (e != hipSuccess) {                                               \
            fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e));           \
            return 1;                                                        \
        }                                                                    \
    } while (0)

static unsigned short bf16_of(float x) {
    unsigned u;
    memcpy(&u, &x, 4);
    return (unsigned short)((u + 0x7fff + ((u >> 16) & 1)) >> 16);
}

int main(int argc, char** argv) {
    hgnn::g_opt_mlp_ablate = argc > 1 ? atoi(argv[1]) : 0;
    const long long N = 120000, M = 2000000;
    const int L = 256, H = 512;
    srand(1);
    auto rnd = [] { return (float)rand() / RAND_MAX * 2.f - 1.f; };
    std::vector<unsigned short> nodes(N * L), edges(M * L), w0((size_t)H * 3 * L), w1((size_t)L * H);
    for (auto& v : nodes) v = bf16_of(rnd());
    for (size_t i = 0; i < edges.size(); ++i) edges[i] = nodes[i % nodes.size()] ^ (unsigned short)(i >> 20);
    for (auto& v : w0) v = bf16_of(rnd() * 0.05f);
    for (auto& v : w1) v = bf16_of(rnd() * 0.05f);
    std::vector<int32_t> g0(M), g1(M);
    for (long long e = 0; e < M; ++e) {
        g0[e] = rand() % N;
        g1[e] = rand() % N;
    }
    std::vector<float> par(H, 0.f), one(H, 1.f);
    unsigned short *d_nodes, *d_edges, *d_w0, *d_w1, *d_out;
    int32_t *d_g0, *d_g1;
    float *d_zero, *d_one;
    CK(hipMalloc(&d_nodes, nodes.size() * 2));
    CK(hipMalloc(&d_edges, edges.size() * 2));
    CK(hipMalloc(&d_out, edges.size() * 2));
    CK(hipMalloc(&d_w0, w0.size() * 2));
    CK(hipMalloc(&d_w1, w1.size() * 2));
    CK(hipMalloc(&d_g0, M * 4));
    CK(hipMalloc(&d_g1, M * 4));
    CK(hipMalloc(&d_zero, H * 4));
    CK(hipMalloc(&d_one, H * 4));
    CK(hipMemcpy(d_nodes, nodes.data(), nodes.size() * 2, hipMemcpyHostToDevice));
    CK(hipMemcpy(d_edges, edges.data(), edges.size() * 2, hipMemcpyHostToDevice));
    CK(hipMemcpy(d_w0, w0.data(), w0.size() * 2, hipMemcpyHostToDevice));
    CK(hipMemcpy(d_w1, w1.data(), w1.size() * 2, hipMemcpyHostToDevice));
    CK(hipMemcpy(d_g0, g0.data(), M * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(d_g1, g1.data(), M * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(d_zero, par.data(), H * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(d_one, one.data(), H * 4, hipMemcpyHostToDevice));
    hgnn_mlp_desc d;
    memset(&d, 0, sizeof d);
    d.n_seg = 3;
    d.seg_table[0] = (const float*)d_nodes;
    d.seg_table[1] = (const float*)d_nodes;
    d.seg_table[2] = (const float*)d_edges;
    d.seg_index[0] = d_g0;
    d.seg_index[1] = d_g1;
    d.seg_width[0] = d.seg_width[1] = d.seg_width[2] = L;
    d.n_layers = 2;
    d.W[0] = (const float*)d_w0;
    d.W[1] = (const float*)d_w1;
    for (int l = 0; l < 2; ++l) {
        d.b[l] = d_zero;
        d.ln_w[l] = d_one;
        d.ln_b[l] = d_zero;
    }
    d.width[0] = 3 * L;
    d.width[1] = H;
    d.width[2] = L;
    d.act[0] = HGNN_ACT_GELU;
    d.act[1] = HGNN_ACT_TANH;
    d.ln_eps = 1e-5f;
    d.skip = (const float*)d_edges;
    d.M = M;
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    float ms = 0.f;
    for (int rep = 0; rep < 6; ++rep) {
        CK(hipEventRecord(e0, 0));
        if (hgnn_mlp_forward_bf16_rows128(&d, d_out, nullptr) != 0) return 2;
        CK(hipEventRecord(e1, 0));
        CK(hipEventSynchronize(e1));
        CK(hipEventElapsedTime(&ms, e0, e1));
    }
    std::vector<unsigned long long> st(256 * 64 * 8 * 8);
    CK(hipMemcpyFromSymbol(st.data(), HIP_SYMBOL(hgnn::fr::g_stamps), st.size() * 8));
    // per-phase, per-wave medians over workgroups and tiles 1..58 (cycles of the shader clock)
    const char* names[5] = {"gemm1", "epi1", "gemm2", "epi2", "tile"};
    printf("{\"ablate\": %d, \"ms\": %.4f", hgnn::g_opt_mlp_ablate, ms);
    for (int k = 0; k < 5; ++k) {
        printf(", \"%s\": [", names[k]);
        for (int w = 0; w < 8; ++w) {
            std::vector<double> v;
            for (int b = 0; b < 256; ++b)
                for (int it = 1; it < 59; ++it) {
                    const unsigned long long* s = &st[((b * 64 + it) * 8) * 8 + w];
                    const unsigned long long* sn = &st[((b * 64 + it + 1) * 8) * 8 + w];
                    v.push_back(k < 4 ? (double)(s[(k + 1) * 8] - s[k * 8]) : (double)(sn[0] - s[0]));
                }
            std::sort(v.begin(), v.end());
            printf("%s%.0f", w ? ", " : "", v[v.size() / 2]);
        }
        printf("]");
    }
    printf("}\n");
    return 0;
}
