// microbenchmark for the index fill: 1e8 records scattered into an 0.8 GB array through a 0.54 GB offset table,
//  (a) records in random order (index_fill_kernel today), (b) records grouped into NP partitions of the bucket range (random inside)
// hipcc --offload-arch=gfx950 -O3 -o fill_bench fill_bench.hip && ./fill_bench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
#include <random>
typedef unsigned long long u64;
typedef unsigned int u32;
__global__ void fill(u64 n2, const ulonglong2 *__restrict__ rec, const u32 *__restrict__ bkt, u64 *__restrict__ ent)
{
    u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    for (; i < n2; i += (u64)gridDim.x * blockDim.x) {
        const ulonglong2 r = rec[i];
        ent[(u64)bkt[r.x >> 32] + (u32)r.x] = r.y;
    }
}
__global__ void gen(u64 n2, u64 T, ulonglong2 *rec, u32 *cnt, int np_shift, int grouped, u64 seed)
{
    u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    for (; i < n2; i += (u64)gridDim.x * blockDim.x) {
        u64 x = (i + seed) * 0x9E3779B97F4A7C15ull;
        x ^= x >> 29; x *= 0xBF58476D1CE4E5B9ull; x ^= x >> 32;
        u64 b = x & (T - 1);
        if (grouped) { // partition = position of the record in the array: i * NP / n2 ; bucket inside the partition random
            const u64 NP = T >> np_shift;
            const u64 P = (u64)((__uint128_t)i * NP / n2);
            b = (P << np_shift) | (b & ((1ull << np_shift) - 1));
        }
        const u32 slot = atomicAdd(&cnt[b], 1u);
        rec[i] = make_ulonglong2((b << 32) | slot, i);
    }
}
__global__ void scan_serial_blocks(u32 *cnt, u64 T, u64 *sums) { /* not timed: simple two-level scan */
    const u64 per = 1 << 16; u64 b = blockIdx.x; u64 s = 0;
    if (threadIdx.x == 0) { for (u64 i = b * per; i < (b + 1) * per && i < T; i++) { u32 c = cnt[i]; cnt[i] = (u32)s; s += c; } sums[b] = s; }
}
__global__ void add_base(u32 *cnt, u64 T, const u64 *base) { u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x; if (i < T) cnt[i] += (u32)base[i >> 16]; }
int main()
{
    const u64 n2 = 100000000ull, T = 1ull << 27;
    ulonglong2 *rec; u32 *bkt; u64 *ent, *sums;
    hipMalloc(&rec, n2 * 16); hipMalloc(&bkt, (T + 1) * 4); hipMalloc(&ent, n2 * 8); hipMalloc(&sums, (T >> 16) * 8);
    for (int grouped = 0; grouped <= 2; grouped++) {
        const int np_shift = grouped == 2 ? 17 : 19; // 1024 or 256 partitions
        hipMemset(bkt, 0, (T + 1) * 4);
        gen<<<4096, 256>>>(n2, T, rec, bkt, np_shift, grouped ? 1 : 0, 12345);
        scan_serial_blocks<<<(unsigned)(T >> 16), 64>>>(bkt, T, sums);
        std::vector<u64> h(T >> 16), base(T >> 16);
        hipMemcpy(h.data(), sums, h.size() * 8, hipMemcpyDeviceToHost);
        u64 s = 0; for (size_t i = 0; i < h.size(); i++) { base[i] = s; s += h[i]; }
        hipMemcpy(sums, base.data(), h.size() * 8, hipMemcpyHostToDevice);
        add_base<<<(unsigned)((T + 255) / 256), 256>>>(bkt, T, sums);
        hipDeviceSynchronize();
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        for (int rep = 0; rep < 3; rep++) {
            hipEventRecord(e0);
            fill<<<4096, 256>>>(n2, rec, bkt, ent);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            printf("%s: fill %.3f ms\n", grouped == 0 ? "random order" : (grouped == 1 ? "grouped, 256 partitions" : "grouped, 1024 partitions"), ms);
        }
    }
    // streaming reference: copy 1.6 GB
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0); hipMemcpyAsync(ent, rec, n2 * 8, hipMemcpyDeviceToDevice); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); printf("copy 0.8 GB: %.3f ms\n", ms);
    return 0;
}
