// Microbenchmark: what does a divergent global load cost per CU as a function of ACTIVE lanes and bytes per lane?
// build: hipcc --offload-arch=gfx950 -O3 scripts/micro/vmem_mask.hip -o gpurun_out/vmem_mask ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <cstdint>
template <int BYTES>
__global__ __launch_bounds__(256) void probe(const uint4 *__restrict__ table, uint32_t mask, uint32_t active, uint32_t iters, uint32_t *out)
{
    const uint32_t lane = threadIdx.x & 63;
    uint32_t idx = (blockIdx.x * 256 + threadIdx.x) * 2654435761u;
    uint32_t acc = 0;
    const bool on = lane < active;
    for (uint32_t i = 0; i < iters; ++i) {
        idx = idx * 1664525u + 1013904223u;
        if (on) {
            const uint32_t at = (idx >> 8) & mask;
            if (BYTES == 16) { const uint4 v = table[at]; acc += v.x ^ v.w; }
            else if (BYTES == 8) { const uint2 v = reinterpret_cast<const uint2 *>(table)[at * 2]; acc += v.x ^ v.y; }
            else { acc += reinterpret_cast<const uint32_t *>(table)[at * 4]; }
        }
    }
    if (acc == 0x12345678u) out[0] = acc;
}
int main()
{
    const size_t entries = 1u << 17; // 2 MiB of 16-B entries: L2 resident
    uint4 *table; uint32_t *out;
    hipMalloc(&table, entries * 16); hipMalloc(&out, 4);
    hipMemset(table, 1, entries * 16);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    const uint32_t iters = 2000, blocks = 256 * 8; // 8 workgroups per CU = 8 waves per SIMD
    for (int bytes : {4, 8, 16}) for (uint32_t active : {64u, 32u, 16u, 8u, 4u, 1u}) {
        for (int rep = 0; rep < 2; ++rep) {
            hipEventRecord(a);
            if (bytes == 16) hipLaunchKernelGGL(probe<16>, dim3(blocks), dim3(256), 0, 0, table, (uint32_t)entries - 1, active, iters, out);
            else if (bytes == 8) hipLaunchKernelGGL(probe<8>, dim3(blocks), dim3(256), 0, 0, table, (uint32_t)entries - 1, active, iters, out);
            else hipLaunchKernelGGL(probe<4>, dim3(blocks), dim3(256), 0, 0, table, (uint32_t)entries - 1, active, iters, out);
            hipEventRecord(b); hipEventSynchronize(b);
            float ms; hipEventElapsedTime(&ms, a, b);
            if (rep == 1) {
                const double instrPerCU = (double)blocks * 4 * iters / 256.0;
                printf("bytes/lane %2d active %2u: %.3f ms, %.1f ns per wave-load per CU (%.1f cycles @2.4GHz), %.2f ns per active lane\n",
                       bytes, active, ms, ms * 1e6 / instrPerCU, ms * 1e6 / instrPerCU * 2.4, ms * 1e6 / instrPerCU / active);
            }
        }
    }
    return 0;
}
