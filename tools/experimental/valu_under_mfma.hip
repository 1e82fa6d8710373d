// Hypothesis test for the fault of the experimental two-workgroup split3 tile: does vector-ALU
// code (LayerNorm FMA + exact-erf GELU + bf16 hi/mid split, the code of the activation phases) return different bits when a
// wave of ANOTHER workgroup runs MFMAs on the same SIMD?  Two workgroups of 4 waves per CU (72.5 KB of LDS each, so exactly
// two fit): "matrix" workgroups spin on v_mfma_f32_16x16x32_bf16, "vector" workgroups evaluate the activation chain on fixed
// inputs and XOR the result bits into a per-thread checksum.  Run 1: the matrix workgroups exit at once (reference).
// Run 2..: they spin for the whole duration.  Any checksum difference is a hardware-level (or hazard-level) finding.
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -I hierarchicalgnn_amd/csrc -o valu_under_mfma tools/experimental/valu_under_mfma.hip
#include "mlp_common.h"
#include <cstdio>
#include <cstdlib>
#include <vector>
namespace hgnn { void set_error(const char*, ...) {} }
using namespace hgnn;
typedef unsigned short u16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned short u16x8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ unsigned short bf16_bits(float x) { return __builtin_bit_cast(unsigned short, (__bf16)x); }
__device__ __forceinline__ float bf16_float(unsigned short b) { return __builtin_bit_cast(float, (unsigned)b << 16); }

__global__ __launch_bounds__(256, 2) void k(const float* __restrict__ x, unsigned* __restrict__ sums, int role_mode, int spin,
                                            int reps, float* __restrict__ sink) {
    extern __shared__ char smem[];
    const int tid = threadIdx.x;
    const bool matrix = role_mode == 0 ? (blockIdx.x & 1) : (blockIdx.x >= gridDim.x / 2);
    if (matrix) {
        if (!spin) return;
        f32x4 acc[8];
        for (int i = 0; i < 8; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
        u16x8 a8, b8;
        for (int i = 0; i < 8; ++i) { a8[i] = (unsigned short)(0x3c00 + tid + i); b8[i] = (unsigned short)(0x3b80 + 3 * tid + i); }
        for (int it = 0; it < spin; ++it)
#pragma unroll
            for (int i = 0; i < 8; ++i)
                acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a8), __builtin_bit_cast(bf16x8, b8), acc[i], 0, 0, 0);
        float s = 0.f;
        for (int i = 0; i < 8; ++i) s += acc[i].x + acc[i].y + acc[i].z + acc[i].w;
        if (s == 12345.678f) sink[0] = s;   // keep the loop
        return;
    }
    // vector workgroup: 64 values per thread, the activation chain of act_write_half + the tanh epilogue
    const int vb = role_mode == 0 ? (int)(blockIdx.x >> 1) : (int)blockIdx.x;   // mode 1: the FIRST half of the grid are the vector workgroups
    if (vb < 0 || vb >= (int)gridDim.x / 2) return;
    f32x4 v[4][4];
    const float* px = x + ((size_t)vb * 256 + tid) * 64;
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int j = 0; j < 4; ++j) v[t][j] = *(const f32x4*)(px + (t * 4 + j) * 4);
    unsigned sum = 0;
    char* lane0 = smem + (tid & 15) * 528 + (tid >> 6) * 128 + ((tid >> 4) & 3) * 8;
    for (int r = 0; r < reps; ++r) {
        const float rs = 1.0f + 0.001f * (float)(r & 63), sh = 0.01f * (float)((r & 15) - 8);
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const f32x4 w4 = f32x4{1.0f, 0.9f, 1.1f, 1.05f}, b4 = f32x4{0.01f, -0.02f, 0.03f, 0.0f};
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                f32x4 y;
                y.x = act_apply(fmaf(fmaf(v[t][j].x, rs, sh), w4.x, b4.x), HGNN_ACT_GELU);
                y.y = act_apply(fmaf(fmaf(v[t][j].y, rs, sh), w4.y, b4.y), HGNN_ACT_GELU);
                y.z = act_apply(fmaf(fmaf(v[t][j].z, rs, sh), w4.z, b4.z), HGNN_ACT_GELU);
                y.w = act_apply(fmaf(fmaf(v[t][j].w, rs, sh), w4.w, b4.w), HGNN_ACT_GELU);
                u16x4 h, m;
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    const unsigned short hb = bf16_bits(y[c]);
                    h[c] = hb;
                    m[c] = bf16_bits(y[c] - bf16_float(hb));
                }
                *(u16x4*)(lane0 + j * 16 * 528 + t * 32) = h;
                *(u16x4*)(lane0 + 33792 + j * 16 * 528 + t * 32) = m;
                sum ^= ((unsigned)h[0] | ((unsigned)h[1] << 16)) + 3u * ((unsigned)h[2] | ((unsigned)h[3] << 16));
                sum ^= (((unsigned)m[0] | ((unsigned)m[1] << 16)) + 5u * ((unsigned)m[2] | ((unsigned)m[3] << 16))) * 2654435761u;
                f32x4 z;
                z.x = act_apply(fmaf(y.x, rs, sh), HGNN_ACT_TANH);
                z.y = act_apply(fmaf(y.y, rs, sh), HGNN_ACT_TANH);
                z.z = act_apply(fmaf(y.z, rs, sh), HGNN_ACT_TANH);
                z.w = act_apply(fmaf(y.w, rs, sh), HGNN_ACT_TANH);
                sum ^= __builtin_bit_cast(unsigned, z.x) + 7u * __builtin_bit_cast(unsigned, z.y) + 11u * __builtin_bit_cast(unsigned, z.z) +
                       13u * __builtin_bit_cast(unsigned, z.w);
            }
        }
        __syncthreads();
        // read something back so that the LDS writes are not dead
        sum ^= *(const unsigned*)(smem + ((tid * 4 + r * 64) % 67000 & ~3)) & 0u;
    }
    sums[(size_t)vb * 256 + tid] = sum;
}

int main(int argc, char** argv) {
    const int n_vec_blocks = 256, reps = argc > 1 ? atoi(argv[1]) : 400;
    const size_t n = (size_t)n_vec_blocks * 256 * 64;
    std::vector<float> hx(n);
    srand(7);
    for (auto& f : hx) f = ((float)rand() / RAND_MAX * 2.f - 1.f) * 3.0f;
    float *dx, *dsink;
    unsigned *ds0, *ds1;
    hipMalloc(&dx, n * 4);
    hipMalloc(&dsink, 4);
    hipMalloc(&ds0, (size_t)n_vec_blocks * 256 * 4);
    hipMalloc(&ds1, (size_t)n_vec_blocks * 256 * 4);
    hipMemcpy(dx, hx.data(), n * 4, hipMemcpyHostToDevice);
    const int lds = 74240;
    hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    std::vector<unsigned> s0((size_t)n_vec_blocks * 256), s1(s0.size());
    for (int role = 0; role < 2; ++role) {
        k<<<2 * n_vec_blocks, 256, lds>>>(dx, ds0, role, 0, reps, dsink);
        hipDeviceSynchronize();
        hipMemcpy(s0.data(), ds0, s0.size() * 4, hipMemcpyDeviceToHost);
        for (int trial = 0; trial < 3; ++trial) {
            hipEvent_t e0, e1;
            hipEventCreate(&e0);
            hipEventCreate(&e1);
            hipEventRecord(e0);
            k<<<2 * n_vec_blocks, 256, lds>>>(dx, ds1, role, 40 * reps, reps, dsink);
            hipEventRecord(e1);
            hipError_t err = hipDeviceSynchronize();
            float ms = 0;
            hipEventElapsedTime(&ms, e0, e1);
            hipMemcpy(s1.data(), ds1, s1.size() * 4, hipMemcpyDeviceToHost);
            size_t bad = 0, bad_hi = 0;
            for (size_t i = 0; i < s0.size(); ++i)
                if (s0[i] != s1[i]) { ++bad; if ((i & 63) >= 48) ++bad_hi; }
            printf("role mode %d trial %d: %s, %.2f ms, threads with a different checksum under co-resident MFMA waves: %zu of %zu (in lanes 48-63: %zu)\n",
                   role, trial, hipGetErrorString(err), ms, bad, s0.size(), bad_hi);
        }
    }
    return 0;
}
