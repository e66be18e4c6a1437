// Microbenchmark: cost of a wave's divergent 64-byte record gather (four 16-byte loads per lane, as wf_trace_kernel does) by
// addressing form: 64-bit per-lane addresses vs scalar base + 32-bit per-lane offset, and 1..4 loads per record.  L2-resident table.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
template <int LOADS, bool WIDE>
__global__ __launch_bounds__(256) void probe(const uint4 *__restrict__ table, unsigned long long mask, uint32_t iters, uint32_t *out)
{
    uint32_t idx = (blockIdx.x * 256 + threadIdx.x) * 2654435761u;
    uint32_t acc = 0;
    for (uint32_t i = 0; i < iters; ++i) {
        idx = idx * 1664525u + 1013904223u;
        if (WIDE) { // record index as a 64-bit quantity the compiler cannot narrow: per-lane 64-bit address
            const unsigned long long rec = ((unsigned long long)(idx >> 8)) & mask;
            const uint4 *p = table + 4 * rec;
#pragma unroll
            for (int k = 0; k < LOADS; ++k) { const uint4 v = p[k]; acc += v.x ^ v.w; }
        } else {    // 32-bit record index < 2^26: byte offset fits 32 bits -> scalar base + per-lane offset
            const uint32_t rec = (idx >> 8) & (uint32_t)mask & 0x3ffffffu;
            const uint4 *p = table + 4u * rec;
#pragma unroll
            for (int k = 0; k < LOADS; ++k) { const uint4 v = p[k]; acc += v.x ^ v.w; }
        }
    }
    if (acc == 0x12345678u) out[0] = acc;
}
template <int LOADS, bool WIDE> void run(const uint4 *table, size_t records, uint32_t *out, const char *name)
{
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    const uint32_t iters = 1000, blocks = 256 * 5;
    float ms = 0;
    for (int rep = 0; rep < 2; ++rep) {
        hipEventRecord(a);
        hipLaunchKernelGGL((probe<LOADS, WIDE>), dim3(blocks), dim3(256), 0, 0, table, (unsigned long long)records - 1, iters, out);
        hipEventRecord(b); hipEventSynchronize(b);
        hipEventElapsedTime(&ms, a, b);
    }
    const double gathersPerCU = (double)blocks * 4 * iters / 256.0;
    printf("%-28s %d loads/record: %.3f ms, %.1f ns per wave-gather per CU, %.2f ns per lane-load\n", name, LOADS, ms, ms * 1e6 / gathersPerCU,
           ms * 1e6 / gathersPerCU / 64 / LOADS);
}
int main()
{
    const size_t records = 1u << 15; // 2 MiB: L2 resident
    uint4 *table; uint32_t *out;
    hipMalloc(&table, records * 64); hipMalloc(&out, 4);
    hipMemset(table, 1, records * 64);
    run<4, true>(table, records, out, "64-bit lane address");
    run<4, false>(table, records, out, "scalar base + 32-bit offset");
    run<3, true>(table, records, out, "64-bit lane address");
    run<3, false>(table, records, out, "scalar base + 32-bit offset");
    run<2, false>(table, records, out, "scalar base + 32-bit offset");
    run<1, true>(table, records, out, "64-bit lane address");
    run<1, false>(table, records, out, "scalar base + 32-bit offset");
    return 0;
}
