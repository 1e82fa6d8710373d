// LayerNorm + activation of one make_mlp layer (Modules/utils.py:169-196) as ONE pass over the rows,
// forward and backward -- the elementwise part of the fused MLP's training path (fused.py,
// _FusedMLPTrain).  The forward MFMA kernel keeps every layer's pre-LayerNorm output z; autograd
// through ATen then needs six passes over [M, W] per layer (native_layer_norm, act, act_backward,
// two layer_norm_backward kernels, bias-gradient sum); here:
//   forward :  a = act(LN(z))                                   read 1, write 1
//   backward:  dz = dLN(act'(LN(z)) * da),  dgamma, dbeta, dbias   read 2, write 1
// HBM-bound row kernels.  16 lanes own a row (W/16 values per lane as float4s, 256-byte coalesced
// pieces), so a wave works on 4 rows and row statistics are 4 cross-lane steps; the column sums
// (dgamma, dbeta, dbias) are carried in registers across all rows of a workgroup and written as
// per-workgroup partials [HGNN_LN_ACT_BLOCKS][3][W] that the caller adds up (deterministic: no atomics).
#include "mlp_common.h"

namespace hgnn {

constexpr int kLnActBlocks = HGNN_LN_ACT_BLOCKS;

// sum over the LPR lanes that own one row (LPR = 16: four rows per wave; LPR = 64: one 1024-wide row per wave)
template <int LPR>
__device__ __forceinline__ float row_sum(float v) {
#pragma unroll
    for (int o = 1; o < LPR; o <<= 1) v += __shfl_xor(v, o);
    return v;
}

// rows in fp32 (T = float) or bf16 (T = unsigned short: the training path of BASELINE config 4; statistics and
// all arithmetic stay fp32, one rounding per stored element)
typedef unsigned short u16x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ f32x4 ld_row4(const float* p) { return *(const f32x4*)p; }
__device__ __forceinline__ f32x4 ld_row4(const unsigned short* p) {
    const u16x4 v = *(const u16x4*)p;
    f32x4 o;
    o.x = __builtin_bit_cast(float, (unsigned)v[0] << 16);
    o.y = __builtin_bit_cast(float, (unsigned)v[1] << 16);
    o.z = __builtin_bit_cast(float, (unsigned)v[2] << 16);
    o.w = __builtin_bit_cast(float, (unsigned)v[3] << 16);
    return o;
}
__device__ __forceinline__ void st_row4(float* p, f32x4 v) { *(f32x4*)p = v; }
__device__ __forceinline__ void st_row4(unsigned short* p, f32x4 v) {
    u16x4 o;
    o[0] = __builtin_bit_cast(unsigned short, (__bf16)v.x);
    o[1] = __builtin_bit_cast(unsigned short, (__bf16)v.y);
    o[2] = __builtin_bit_cast(unsigned short, (__bf16)v.z);
    o[3] = __builtin_bit_cast(unsigned short, (__bf16)v.w);
    *(u16x4*)p = o;
}

template <int NV, bool BACKWARD, typename T, int LPR = 16>
__global__ __launch_bounds__(256) void k_ln_act(const T* __restrict__ z, const T* __restrict__ da,
                                                long long M, const float* __restrict__ gamma,
                                                const float* __restrict__ beta, int act, float eps,
                                                T* __restrict__ out, float* __restrict__ partials) {
    constexpr int W = NV * LPR * 4;
    constexpr int RPW = 64 / LPR;      // rows per wave
    constexpr int VS = LPR * 4;        // floats between a lane's consecutive vectors
    constexpr float inv_w = 1.0f / (float)W;
    __shared__ float red[BACKWARD ? 4 * 3 * W : 1];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int l16 = lane % LPR;  // this lane's 4-float column slot within a vector
    const int rg = lane / LPR;   // row within the wave's group of RPW
    f32x4 gm[NV], bt[NV];
#pragma unroll
    for (int v = 0; v < NV; ++v) {
        gm[v] = *(const f32x4*)(gamma + (v * LPR + l16) * 4);
        bt[v] = *(const f32x4*)(beta + (v * LPR + l16) * 4);
    }
    f32x4 s_dg[BACKWARD ? NV : 1], s_db[BACKWARD ? NV : 1], s_dz[BACKWARD ? NV : 1];
    if constexpr (BACKWARD) {
#pragma unroll
        for (int v = 0; v < NV; ++v) s_dg[v] = s_db[v] = s_dz[v] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    for (long long r0 = (long long)blockIdx.x * (4 * RPW); r0 < M; r0 += (long long)gridDim.x * (4 * RPW)) {
        const long long r = r0 + wave * RPW + rg;
        const bool valid = r < M;
        const size_t off = (size_t)(valid ? r : 0) * W + l16 * 4;
        f32x4 x[NV];
        float s = 0.f;
#pragma unroll
        for (int v = 0; v < NV; ++v) {
            x[v] = ld_row4(z + off + v * VS);
            s += (x[v].x + x[v].y) + (x[v].z + x[v].w);
        }
        f32x4 g[BACKWARD ? NV : 1];
        if constexpr (BACKWARD) {
#pragma unroll
            for (int v = 0; v < NV; ++v) g[v] = ld_row4(da + off + v * VS);
        }
        const float mean = row_sum<LPR>(s) * inv_w;
        float q = 0.f;
#pragma unroll
        for (int v = 0; v < NV; ++v) {
            x[v].x -= mean; x[v].y -= mean; x[v].z -= mean; x[v].w -= mean;
            q = fmaf(x[v].x, x[v].x, q);
            q = fmaf(x[v].y, x[v].y, q);
            q = fmaf(x[v].z, x[v].z, q);
            q = fmaf(x[v].w, x[v].w, q);
        }
        const float rstd = 1.0f / sqrtf(row_sum<LPR>(q) * inv_w + eps);
        if constexpr (!BACKWARD) {
#pragma unroll
            for (int v = 0; v < NV; ++v) {
                f32x4 o;
                o.x = act_apply(fmaf(x[v].x * rstd, gm[v].x, bt[v].x), act);
                o.y = act_apply(fmaf(x[v].y * rstd, gm[v].y, bt[v].y), act);
                o.z = act_apply(fmaf(x[v].z * rstd, gm[v].z, bt[v].z), act);
                o.w = act_apply(fmaf(x[v].w * rstd, gm[v].w, bt[v].w), act);
                if (valid) st_row4(out + off + v * VS, o);
            }
        } else {
            // x <- xhat, g <- dy * gamma;  column sums of dy*xhat and dy;  row sums of g and g*xhat
            float sg = 0.f, sgx = 0.f;
#pragma unroll
            for (int v = 0; v < NV; ++v) {
                float* xv = (float*)&x[v];
                float* gv = (float*)&g[v];
                const float* gmv = (const float*)&gm[v];
                const float* btv = (const float*)&bt[v];
                float* cdg = (float*)&s_dg[v];
                float* cdb = (float*)&s_db[v];
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    const float xh = xv[c] * rstd;
                    const float y = fmaf(xh, gmv[c], btv[c]);
                    const float a = act == HGNN_ACT_TANH ? fast_tanh(y) : 0.f;
                    const float dy = valid ? gv[c] * act_grad(y, a, act) : 0.f;
                    cdg[c] = fmaf(dy, xh, cdg[c]);
                    cdb[c] += dy;
                    const float gg = dy * gmv[c];
                    xv[c] = xh;
                    gv[c] = gg;
                    sg += gg;
                    sgx = fmaf(gg, xh, sgx);
                }
            }
            const float mg = row_sum<LPR>(sg) * inv_w;
            const float mgx = row_sum<LPR>(sgx) * inv_w;
#pragma unroll
            for (int v = 0; v < NV; ++v) {
                f32x4 o;
                o.x = rstd * (g[v].x - mg - x[v].x * mgx);
                o.y = rstd * (g[v].y - mg - x[v].y * mgx);
                o.z = rstd * (g[v].z - mg - x[v].z * mgx);
                o.w = rstd * (g[v].w - mg - x[v].w * mgx);
                if (valid) {
                    st_row4(out + off + v * VS, o);
                    s_dz[v].x += o.x; s_dz[v].y += o.y; s_dz[v].z += o.z; s_dz[v].w += o.w;
                }
            }
        }
    }
    if constexpr (BACKWARD) {
        // column partials: the wave's 4 row groups (lanes l16, l16+16, ...), then the 4 waves through LDS
#pragma unroll
        for (int v = 0; v < NV; ++v) {
            float* p[3] = {(float*)&s_dg[v], (float*)&s_db[v], (float*)&s_dz[v]};
#pragma unroll
            for (int k = 0; k < 3; ++k)
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    float t = p[k][c];
#pragma unroll
                    for (int o = LPR; o < 64; o <<= 1) t += __shfl_xor(t, o);
                    if (rg == 0) red[(wave * 3 + k) * W + (v * LPR + l16) * 4 + c] = t;
                }
        }
        __syncthreads();
        for (int i = tid; i < 3 * W; i += 256)
            partials[(size_t)blockIdx.x * 3 * W + i] = (red[i] + red[3 * W + i]) + (red[6 * W + i] + red[9 * W + i]);
    }
}

template <bool BACKWARD, typename T>
static int launch_ln_act(const T* z, const T* da, int64_t M, int W, const float* gamma, const float* beta,
                         int act, float eps, T* out, float* partials, hipStream_t s) {
    const unsigned grid = BACKWARD ? (unsigned)kLnActBlocks
                                   : (unsigned)(ceil_div(M, 16) < 4096 ? ceil_div(M, 16) : 4096);
    if (W == 1024) {   // one row per wave (64 lanes x 4 vectors): the hidden width at latent 512
        k_ln_act<4, BACKWARD, T, 64><<<grid, 256, 0, s>>>(z, da, M, gamma, beta, act, eps, out, partials);
        HGNN_CHECK_HIP(hipGetLastError());
        return HGNN_OK;
    }
    switch (W / 64) {
        case 1: k_ln_act<1, BACKWARD, T><<<grid, 256, 0, s>>>(z, da, M, gamma, beta, act, eps, out, partials); break;
        case 2: k_ln_act<2, BACKWARD, T><<<grid, 256, 0, s>>>(z, da, M, gamma, beta, act, eps, out, partials); break;
        case 4: k_ln_act<4, BACKWARD, T><<<grid, 256, 0, s>>>(z, da, M, gamma, beta, act, eps, out, partials); break;
        case 8: k_ln_act<8, BACKWARD, T><<<grid, 256, 0, s>>>(z, da, M, gamma, beta, act, eps, out, partials); break;
        default:
            set_error("hgnn_ln_act: width %d has no instantiation (64, 128, 256, 512, 1024)", W);
            return HGNN_ERR_UNSUPPORTED;
    }
    HGNN_CHECK_HIP(hipGetLastError());
    return HGNN_OK;
}

}  // namespace hgnn

using namespace hgnn;

static int check_ln_act(const void* z, int64_t M, int W, const void* gamma, const void* beta, int act,
                        const void* out, const char* who) {
    HGNN_REQUIRE(M >= 0, "%s: bad M", who);
    HGNN_REQUIRE(W == 64 || W == 128 || W == 256 || W == 512 || W == 1024,
                 "%s: width must be 64, 128, 256, 512 or 1024 (got %d)", who, W);
    HGNN_REQUIRE(act >= HGNN_ACT_NONE && act <= HGNN_ACT_RELU, "%s: unknown activation code %d", who, act);
    if (M == 0) return HGNN_OK;
    HGNN_REQUIRE(z != nullptr && gamma != nullptr && beta != nullptr && out != nullptr, "%s: NULL pointer", who);
    HGNN_REQUIRE((uintptr_t)z % 16 == 0 && (uintptr_t)gamma % 16 == 0 && (uintptr_t)beta % 16 == 0 &&
                     (uintptr_t)out % 16 == 0, "%s: pointers must be 16-byte aligned", who);
    return HGNN_OK;
}

extern "C" int hgnn_ln_act_forward_f32(const float* z, int64_t M, int32_t W, const float* gamma, const float* beta,
                                       int32_t act, float eps, float* out, hgnn_stream_t stream) {
    const int rc = check_ln_act(z, M, W, gamma, beta, act, out, "hgnn_ln_act_forward_f32");
    if (rc != HGNN_OK || M == 0) return rc;
    return launch_ln_act<false, float>(z, nullptr, M, W, gamma, beta, act, eps, out, nullptr, (hipStream_t)stream);
}

extern "C" int hgnn_ln_act_backward_f32(const float* z, const float* grad_out, int64_t M, int32_t W,
                                        const float* gamma, const float* beta, int32_t act, float eps,
                                        float* grad_z, float* partials, hgnn_stream_t stream) {
    const int rc = check_ln_act(z, M, W, gamma, beta, act, grad_z, "hgnn_ln_act_backward_f32");
    if (rc != HGNN_OK) return rc;
    HGNN_REQUIRE(partials != nullptr && (uintptr_t)partials % 16 == 0,
                 "hgnn_ln_act_backward_f32: partials must be a 16-byte aligned [HGNN_LN_ACT_BLOCKS][3][W] buffer");
    HGNN_REQUIRE(M == 0 || (grad_out != nullptr && (uintptr_t)grad_out % 16 == 0),
                 "hgnn_ln_act_backward_f32: grad_out is NULL or not 16-byte aligned");
    // M == 0 still runs: every workgroup writes its (zero) partials
    return launch_ln_act<true, float>(z, grad_out, M, W, gamma, beta, act, eps, grad_z, partials, (hipStream_t)stream);
}

extern "C" int hgnn_ln_act_forward_bf16(const void* z, int64_t M, int32_t W, const float* gamma, const float* beta,
                                        int32_t act, float eps, void* out, hgnn_stream_t stream) {
    const int rc = check_ln_act(z, M, W, gamma, beta, act, out, "hgnn_ln_act_forward_bf16");
    if (rc != HGNN_OK || M == 0) return rc;
    return launch_ln_act<false, unsigned short>((const unsigned short*)z, nullptr, M, W, gamma, beta, act, eps,
                                                (unsigned short*)out, nullptr, (hipStream_t)stream);
}

extern "C" int hgnn_ln_act_backward_bf16(const void* z, const void* grad_out, int64_t M, int32_t W,
                                         const float* gamma, const float* beta, int32_t act, float eps,
                                         void* grad_z, float* partials, hgnn_stream_t stream) {
    const int rc = check_ln_act(z, M, W, gamma, beta, act, grad_z, "hgnn_ln_act_backward_bf16");
    if (rc != HGNN_OK) return rc;
    HGNN_REQUIRE(partials != nullptr && (uintptr_t)partials % 16 == 0,
                 "hgnn_ln_act_backward_bf16: partials must be a 16-byte aligned [HGNN_LN_ACT_BLOCKS][3][W] buffer");
    HGNN_REQUIRE(M == 0 || (grad_out != nullptr && (uintptr_t)grad_out % 16 == 0),
                 "hgnn_ln_act_backward_bf16: grad_out is NULL or not 16-byte aligned");
    return launch_ln_act<true, unsigned short>((const unsigned short*)z, (const unsigned short*)grad_out, M, W, gamma,
                                               beta, act, eps, (unsigned short*)grad_z, partials, (hipStream_t)stream);
}
