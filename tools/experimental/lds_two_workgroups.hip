// Do two co-resident workgroups with > 64 KiB of dynamic LDS each keep disjoint LDS on gfx950?  Every workgroup fills its
// allocation with a pattern derived from its id, spins, then verifies.   hipcc -O2 --offload-arch=gfx950 -o lds2 lds_two_workgroups.hip
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ __launch_bounds__(256, 2) void k(unsigned* bad, int words, int spin) {
    extern __shared__ unsigned lds[];
    const unsigned tag = blockIdx.x * 0x9E3779B9u;
    for (int r = 0; r < 4; ++r) {
        for (int i = threadIdx.x; i < words; i += 256) lds[i] = tag + i + r;
        __syncthreads();
        const unsigned long long t0 = __builtin_amdgcn_s_memtime();
        while (__builtin_amdgcn_s_memtime() - t0 < (unsigned long long)spin) __builtin_amdgcn_s_sleep(8);
        unsigned n = 0;
        for (int i = threadIdx.x; i < words; i += 256) n += lds[i] != tag + i + r;
        if (n) atomicAdd(bad, n);
        __syncthreads();
    }
}
int main() {
    unsigned* bad;
    hipMalloc(&bad, 4);
    for (int bytes : {32768, 65536, 74240, 81920}) {
        for (int grid : {256, 512, 1024}) {
            hipMemset(bad, 0, 4);
            hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
            k<<<grid, 256, bytes>>>(bad, bytes / 4, 20000);
            hipError_t e = hipDeviceSynchronize();
            unsigned h = 0;
            hipMemcpy(&h, bad, 4, hipMemcpyDeviceToHost);
            printf("lds %d bytes, grid %d: %s, mismatching words %u\n", bytes, grid, hipGetErrorString(e), h);
        }
    }
    return 0;
}
