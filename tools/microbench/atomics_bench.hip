// Microbenchmark: rates of the memory operations the k-mer counting kernels are built from, on
// one MI355X: random 4-byte gathers, random 32-bit atomic adds (with and without a returned value),
// random 16-byte slot loads and 64-bit compare-and-swap, against tables of different footprints.
// Build: hipcc -O3 --offload-arch=gfx950 -o atomics_bench atomics_bench.hip
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { \
    fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1); } } while (0)

__device__ __forceinline__ uint64_t mix(uint64_t x)
{
    x += 0x9e3779b97f4a7c15ULL;
    x = (x ^ (x >> 30)) * 0xbf58476d1ce4e5b9ULL;
    x = (x ^ (x >> 27)) * 0x94d049bb133111ebULL;
    return x ^ (x >> 31);
}

template <int U>
__global__ void __launch_bounds__(256) k_gather(const uint32_t *__restrict__ tab, uint64_t mask, uint64_t n,
                                               uint32_t *__restrict__ out, uint64_t salt)
{
    uint64_t i0 = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) * U;
    uint64_t stride = (uint64_t)gridDim.x * blockDim.x * U;
    uint32_t acc = 0;
    for (uint64_t i = i0; i < n; i += stride) {
        uint32_t v[U];
#pragma unroll
        for (int u = 0; u < U; u++) v[u] = tab[mix(i + u + salt) & mask];
#pragma unroll
        for (int u = 0; u < U; u++) acc += v[u];
    }
    if (acc == 0x12345678u) out[0] = acc;
}

// bit gather: word index = key >> 5
template <int U>
__global__ void __launch_bounds__(256) k_add_ret(uint32_t *__restrict__ tab, uint64_t mask, uint64_t n,
                                                uint32_t *__restrict__ out, uint64_t salt)
{
    uint64_t i0 = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) * U;
    uint64_t stride = (uint64_t)gridDim.x * blockDim.x * U;
    uint32_t acc = 0;
    for (uint64_t i = i0; i < n; i += stride) {
        uint32_t v[U];
#pragma unroll
        for (int u = 0; u < U; u++) v[u] = atomicAdd(&tab[mix(i + u + salt) & mask], 1u);
#pragma unroll
        for (int u = 0; u < U; u++) acc += (v[u] >= 99u);
    }
    if (acc == 0x12345678u) out[0] = acc;
}

template <int U>
__global__ void __launch_bounds__(256) k_add_noret(uint32_t *__restrict__ tab, uint64_t mask, uint64_t n, uint64_t salt)
{
    uint64_t i0 = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) * U;
    uint64_t stride = (uint64_t)gridDim.x * blockDim.x * U;
    for (uint64_t i = i0; i < n; i += stride) {
#pragma unroll
        for (int u = 0; u < U; u++) atomicAdd(&tab[mix(i + u + salt) & mask], 1u);
    }
}

// gather a bit first, then add only if the bit is clear (the "saturated k-mers skip the add" shape)
template <int U>
__global__ void __launch_bounds__(256) k_bit_then_add(const uint32_t *__restrict__ bits, uint32_t *__restrict__ tab,
                                                     uint64_t mask, uint64_t n, uint32_t *__restrict__ out, uint64_t salt)
{
    uint64_t i0 = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) * U;
    uint64_t stride = (uint64_t)gridDim.x * blockDim.x * U;
    uint32_t acc = 0;
    for (uint64_t i = i0; i < n; i += stride) {
        uint64_t key[U];
        uint32_t b[U];
#pragma unroll
        for (int u = 0; u < U; u++) { key[u] = mix(i + u + salt) & mask; b[u] = bits[key[u] >> 5]; }
#pragma unroll
        for (int u = 0; u < U; u++) {
            if ((b[u] >> (key[u] & 31)) & 1u) acc++;
            else acc += (atomicAdd(&tab[key[u]], 1u) >= 99u);
        }
    }
    if (acc == 0x12345678u) out[0] = acc;
}

struct __attribute__((aligned(16))) Slot { uint64_t key; uint32_t cnt; uint32_t flg; };

template <int U>
__global__ void __launch_bounds__(256) k_slot_probe_add(Slot *__restrict__ tab, uint64_t mask, uint64_t n,
                                                       uint32_t *__restrict__ out, uint64_t salt, uint64_t keyspace)
{
    uint64_t i0 = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) * U;
    uint64_t stride = (uint64_t)gridDim.x * blockDim.x * U;
    uint32_t acc = 0;
    for (uint64_t i = i0; i < n; i += stride) {
#pragma unroll
        for (int u = 0; u < U; u++) {
            uint64_t key = (mix(i + u + salt) % keyspace) + 1;
            uint64_t h = mix(key) & mask;
            for (int probe = 0; probe < 64; probe++) {
                uint4 raw = *reinterpret_cast<const uint4 *>(&tab[h]);
                uint64_t cur = ((uint64_t)raw.y << 32) | raw.x;
                if (cur == 0) {
                    cur = atomicCAS((unsigned long long *)&tab[h].key, 0ULL, (unsigned long long)key);
                    if (cur == 0) cur = key;
                }
                if (cur == key) {
                    acc += (atomicAdd(&tab[h].cnt, 1u) >= 99u);
                    break;
                }
                h = (h + 1) & mask;
            }
        }
    }
    if (acc == 0x12345678u) out[0] = acc;
}

template <typename F>
static double timeit(F f, int reps = 3)
{
    hipEvent_t a, b;
    CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    f(0);  // warm
    CK(hipDeviceSynchronize());
    double best = 1e30;
    for (int r = 0; r < reps; r++) {
        CK(hipEventRecord(a));
        f(r + 1);
        CK(hipEventRecord(b));
        CK(hipEventSynchronize(b));
        float ms; CK(hipEventElapsedTime(&ms, a, b));
        if (ms < best) best = ms;
    }
    CK(hipGetLastError());
    return best;
}

int main(int argc, char **argv)
{
    uint64_t n = (argc > 1) ? strtoull(argv[1], 0, 10) : (1ULL << 28);
    uint32_t *out; CK(hipMalloc(&out, 64));
    const int grid = 256 * 8, blk = 256;
    size_t sizes_mb[] = {32, 128, 1024, 4096, 16384};
    printf("n=%llu ops per test, grid=%d x %d\n", (unsigned long long)n, grid, blk);
    for (size_t smb : sizes_mb) {
        uint64_t words = smb * 1024 * 1024 / 4;
        uint32_t *tab; CK(hipMalloc(&tab, words * 4)); CK(hipMemset(tab, 0, words * 4));
        uint64_t mask = words - 1;
        double t;
        t = timeit([&](int r) { hipLaunchKernelGGL(k_gather<4>, grid, blk, 0, 0, tab, mask, n, out, (uint64_t)r * n); });
        printf("table %6zu MiB  gather4B      U4: %8.3f ms  %7.2f Gop/s\n", smb, t, n / t * 1e-6);
        t = timeit([&](int r) { hipLaunchKernelGGL(k_gather<8>, grid, blk, 0, 0, tab, mask, n, out, (uint64_t)r * n); });
        printf("table %6zu MiB  gather4B      U8: %8.3f ms  %7.2f Gop/s\n", smb, t, n / t * 1e-6);
        t = timeit([&](int r) { hipLaunchKernelGGL(k_add_ret<1>, grid, blk, 0, 0, tab, mask, n, out, (uint64_t)r * n); });
        printf("table %6zu MiB  atomicAdd ret U1: %8.3f ms  %7.2f Gop/s\n", smb, t, n / t * 1e-6);
        t = timeit([&](int r) { hipLaunchKernelGGL(k_add_ret<4>, grid, blk, 0, 0, tab, mask, n, out, (uint64_t)r * n); });
        printf("table %6zu MiB  atomicAdd ret U4: %8.3f ms  %7.2f Gop/s\n", smb, t, n / t * 1e-6);
        t = timeit([&](int r) { hipLaunchKernelGGL(k_add_ret<8>, grid, blk, 0, 0, tab, mask, n, out, (uint64_t)r * n); });
        printf("table %6zu MiB  atomicAdd ret U8: %8.3f ms  %7.2f Gop/s\n", smb, t, n / t * 1e-6);
        t = timeit([&](int r) { hipLaunchKernelGGL(k_add_noret<4>, grid, blk, 0, 0, tab, mask, n, (uint64_t)r * n); });
        printf("table %6zu MiB  atomicAdd nor U4: %8.3f ms  %7.2f Gop/s\n", smb, t, n / t * 1e-6);
        CK(hipFree(tab));
        fflush(stdout);
    }
    {   // the shape of the direct-addressed k=15 path: 128 MiB bitmap (half the bits set) + 4 GiB counters
        uint64_t keys = 1ULL << 30;
        uint32_t *bits, *tab;
        CK(hipMalloc(&bits, keys / 8)); CK(hipMalloc(&tab, keys * 4));
        CK(hipMemset(tab, 0, keys * 4));
        for (int pat = 0; pat < 3; pat++) {
            int fill = pat == 0 ? 0x00 : pat == 1 ? 0x55 : 0xff;
            CK(hipMemset(bits, fill, keys / 8));
            double t = timeit([&](int r) { hipLaunchKernelGGL(k_bit_then_add<4>, grid, blk, 0, 0, bits, tab, keys - 1, n, out, (uint64_t)r * n); });
            printf("bitmap128MiB(fill %02x) then atomicAdd 4GiB U4: %8.3f ms  %7.2f Gop/s\n", fill, t, n / t * 1e-6);
        }
        CK(hipFree(bits)); CK(hipFree(tab));
    }
    {   // 16-byte slots, 16 GiB table, key space sized for ~50% load
        uint64_t slots = 1ULL << 30;
        Slot *tab; CK(hipMalloc(&tab, slots * sizeof(Slot))); CK(hipMemset(tab, 0, slots * sizeof(Slot)));
        uint64_t keyspace = slots / 2;
        for (int pass = 0; pass < 2; pass++) {
            double t = timeit([&](int r) { hipLaunchKernelGGL(k_slot_probe_add<2>, grid, blk, 0, 0, tab, slots - 1, n, out, (uint64_t)r * n, keyspace); }, 2);
            printf("slot16B 16GiB probe+CAS+add pass %d: %8.3f ms  %7.2f Gop/s\n", pass, t, n / t * 1e-6);
        }
        CK(hipFree(tab));
    }
    return 0;
}
