// In which order does the LDS serve the lanes of one wavefront instruction that add to the same address?
// Every lane does one returning atomicAdd(&cnt[d], 1) per round; the value it gets back is compared with the rank a
// lane-ordered service would give (count before the round + lanes below it with the same d).  Several wavefronts of a
// workgroup (and several workgroups per CU) run at once on rows of their own, so other wavefronts' traffic is in the way.
// Build: hipcc -O3 --offload-arch=gfx950 -o lds_atomic_order lds_atomic_order.hip
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>

__device__ __forceinline__ uint32_t mixu(uint32_t x)
{
    x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
    return x;
}

template <int WAVES>
__global__ void __launch_bounds__(WAVES * 64) k_order(uint32_t rounds, uint32_t spread, uint32_t seed, unsigned long long *bad,
                                                      unsigned long long *done)
{
    __shared__ uint32_t cnt[WAVES][256];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    for (int i = threadIdx.x; i < WAVES * 256; i += WAVES * 64) (&cnt[0][0])[i] = 0;
    __syncthreads();
    uint32_t expect_base[4] = {0, 0, 0, 0}; // unused; ranks are checked through ballots below
    (void)expect_base;
    unsigned long long nbad = 0;
    for (uint32_t r = 0; r < rounds; r++) {
        const uint32_t h = mixu(seed ^ (blockIdx.x * 7919u + wv) * 0x9E3779B1u ^ (r * 64u + lane) * 0x85ebca6bu);
        const uint32_t d = spread == 0 ? 0u : (h % spread);
        // count before the round, read by everybody before anybody adds
        const uint32_t before = __hip_atomic_load(&cnt[wv][d], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_s_waitcnt(0);
        const uint32_t got = atomicAdd(&cnt[wv][d], 1u);
        // lanes below me with the same d
        uint32_t below = 0;
        for (int b = 0; b < 64; b++) {
            const uint32_t db = __shfl(d, b);
            if (b < lane && db == d) below++;
        }
        if (got != before + below) nbad++;
        __builtin_amdgcn_wave_barrier();
    }
    if (nbad) atomicAdd(bad, nbad);
    if (threadIdx.x == 0) atomicAdd(done, (unsigned long long)rounds * WAVES * 64);
}

int main()
{
    unsigned long long *d_bad, *d_done, h[2];
    hipMalloc(&d_bad, 8); hipMalloc(&d_done, 8);
    const uint32_t spreads[] = {0, 1, 2, 3, 7, 16, 33, 64, 100, 256};
    for (uint32_t sp : spreads) {
        hipMemset(d_bad, 0, 8); hipMemset(d_done, 0, 8);
        k_order<8><<<2048, 512>>>(2000, sp, 12345u + sp, d_bad, d_done);
        k_order<16><<<1024, 1024>>>(2000, sp, 999u + sp, d_bad, d_done);
        k_order<1><<<8192, 64>>>(2000, sp, 77u + sp, d_bad, d_done);
        if (hipDeviceSynchronize() != hipSuccess) { printf("kernel failed\n"); return 1; }
        hipMemcpy(&h[0], d_bad, 8, hipMemcpyDeviceToHost);
        hipMemcpy(&h[1], d_done, 8, hipMemcpyDeviceToHost);
        printf("digits spread over %3u addresses: %llu returning adds, %llu out of lane order\n", sp, h[1], h[0]);
    }
    return 0;
}
