// Calibration for the fabric/HBM byte counters in the access pattern of wf_trace_kernel: every lane
// gathers a random 64-byte record (4 x dwordx4, like a pair record) or a random 8-byte word (like a cell range) from a table
// of a given size.  The number of records read is known, so TCC_EA0_RDREQ* / FETCH_SIZE per access and the achieved rate
// can be read off for a table that fits the 256 MiB Infinity Cache and for one that does not.
// build: hipcc --offload-arch=gfx950 -O3 scripts/micro/gather_calib.hip -o scripts/micro/bin/gather_calib
// run:   scripts/micro/bin/gather_calib            (prints ms and GB/s per case; each case is ONE kernel name)
//        rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_128B_sum ... -- scripts/micro/bin/gather_calib
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__device__ __forceinline__ uint32_t hash32(uint32_t v)
{
    v ^= v >> 16; v *= 0x7feb352du; v ^= v >> 15; v *= 0x846ca68bu; v ^= v >> 16;
    return v;
}

// records: `count` 64-byte records; every thread reads `per` records chosen by a hash of (thread, i)
template <int TAG>
__global__ __launch_bounds__(256) void gather64(const float4 *__restrict__ table, uint32_t count, uint32_t per, float *out)
{
    const uint32_t gid = blockIdx.x * 256 + threadIdx.x;
    float acc = 0.f;
    for (uint32_t i = 0; i < per; ++i) {
        const uint32_t r = (uint32_t)(((uint64_t)hash32(gid * 7919u + i * 104729u + TAG) * count) >> 32);
        const float4 *rec = table + 4 * (size_t)r;
        const float4 a = rec[0], b = rec[1], c = rec[2], d = rec[3];
        acc += a.x + b.y + c.z + d.w;
    }
    if (acc == 12345.678f) out[0] = acc;
}

template <int TAG>
__global__ __launch_bounds__(256) void gather8(const uint2 *__restrict__ table, uint32_t count, uint32_t per, float *out)
{
    const uint32_t gid = blockIdx.x * 256 + threadIdx.x;
    uint32_t acc = 0;
    for (uint32_t i = 0; i < per; ++i) {
        const uint32_t r = (uint32_t)(((uint64_t)hash32(gid * 7919u + i * 104729u + TAG) * count) >> 32);
        const uint2 v = table[r];
        acc += v.x ^ v.y;
    }
    if (acc == 0x12345678u) out[0] = 1.f;
}

// streaming read of the same bytes, for the x2 reference point (16 B per lane, coalesced)
__global__ __launch_bounds__(256) void stream16(const float4 *__restrict__ table, size_t quads, float *out)
{
    float acc = 0.f;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < quads; i += (size_t)gridDim.x * 256) acc += table[i].x;
    if (acc == 12345.678f) out[0] = acc;
}

int main()
{
    float *out;
    CHECK(hipMalloc(&out, 4));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    const size_t bigBytes = (size_t)2 << 30; // 2 GiB: 8x the Infinity Cache
    const size_t smallBytes = (size_t)96 << 20; // 96 MiB: fits the Infinity Cache (like pairRec of lambert_1m)
    float4 *big;
    CHECK(hipMalloc(&big, bigBytes));
    CHECK(hipMemset(big, 0, bigBytes));
    const uint32_t blocks = 256 * 16, per = 8; // 1 M threads x 8 = 8.4 M gathers per launch
    const double n = (double)blocks * 256 * per;
    auto timed = [&](const char *name, auto &&launch, double algBytes) {
        launch(); // warm-up (also warms the Infinity Cache for the small table)
        CHECK(hipEventRecord(e0));
        launch();
        CHECK(hipEventRecord(e1));
        CHECK(hipEventSynchronize(e1));
        float ms = 0.f;
        CHECK(hipEventElapsedTime(&ms, e0, e1));
        printf("%-28s %8.3f ms  %8.1f GB/s algorithmic  %6.2f ns per wave-gather per CU\n", name, ms, algBytes / ms / 1e6, ms * 1e6 / (n / 64 / 256));
    };
    timed("gather64 from 2 GiB", [&] { hipLaunchKernelGGL(gather64<1>, dim3(blocks), dim3(256), 0, 0, big, (uint32_t)(bigBytes / 64), per, out); }, n * 64);
    timed("gather64 from 96 MiB", [&] { hipLaunchKernelGGL(gather64<2>, dim3(blocks), dim3(256), 0, 0, big, (uint32_t)(smallBytes / 64), per, out); }, n * 64);
    timed("gather64 from 3 MiB", [&] { hipLaunchKernelGGL(gather64<3>, dim3(blocks), dim3(256), 0, 0, big, (uint32_t)((3u << 20) / 64), per, out); }, n * 64);
    timed("gather8 from 2 GiB", [&] { hipLaunchKernelGGL(gather8<1>, dim3(blocks), dim3(256), 0, 0, (const uint2 *)big, (uint32_t)(bigBytes / 8), per, out); }, n * 8);
    timed("gather8 from 12 MiB", [&] { hipLaunchKernelGGL(gather8<2>, dim3(blocks), dim3(256), 0, 0, (const uint2 *)big, (uint32_t)((12u << 20) / 8), per, out); }, n * 8);
    timed("stream16 over 2 GiB", [&] { hipLaunchKernelGGL(stream16, dim3(256 * 8), dim3(256), 0, 0, big, bigBytes / 16, out); }, (double)bigBytes);
    printf("gathers per launch: %.0f (x2 launches per case: one warm-up, one timed)\n", n);
    return 0;
}
