// rt_build_device.hip -- the camera candidate lists built on the GPU ("next" row 1; counterpart of
// CameraTriangleList::New, source/util/trianglelist.cpp:520-626).
//
// Membership is decided by the very functions the host builder uses (rt_build_shared.h: GetCameraPosition :74-90,
// FillRectangle :131-217), compiled for the device without fp contraction, so both builders put the same triangles into the
// same pixels; every pixel's entries are in ascending triangle order, which is the order the reference's sort on
// pixel*T+tri keys gives (:161,:565-574).  What the reference does with a 2 GiB key array and a quicksort is a count /
// exclusive scan / fill / per-pixel sort here:
//   cam_project      one thread per triangle: the three projected vertices
//   cam_rasterize    one thread per triangle (COUNT or FILL pass): triangles whose clipped rectangle has more than
//                    RT_BIG_RECT pixels are put on a list instead ...
//   cam_rasterize_big ... and rasterized by one workgroup each, the pixels of the rectangle dealt to its threads
//   cam_sort_pixels  one thread per pixel: insertion sort of its (short) list
// The reference's neighbour de-duplication (:580-613: a pixel whose list equals its left or upper neighbour's shares the
// storage) only changes where the entries live, not what a pixel's list holds; it is not done here, so Start/End never alias.
#include <hip/hip_runtime.h>
#include <hipcub/hipcub.hpp>

#include "raytrace_hip.h"
#include "rt_build_shared.h"

#include <chrono>
#include <cstdlib>

using rtbuild::Camera;
using rtbuild::F2;
using rtbuild::F3;
using rtbuild::RectSetup;

#define RT_BIG_RECT 1024u // rectangles with more pixels than this are shared out among a workgroup

namespace {

__global__ __launch_bounds__(256) void cam_project(const Camera cam, uint32_t T, const float4 *__restrict__ vertex, const int4 *__restrict__ triIndex,
                                                   F2 *__restrict__ pos)
{
    const uint32_t t = blockIdx.x * 256 + threadIdx.x;
    if (t >= T) return;
    const int4 vi = triIndex[t];
    const int idx[3] = { vi.x, vi.y, vi.z };
    for (int k = 0; k < 3; ++k) {
        const float4 v = vertex[idx[k]];
        pos[3 * (size_t)t + k] = rtbuild::camera_position(cam, F3{ v.x, v.y, v.z });
    }
}

// FILL = false: count[pixel] += 1 per member pixel;  FILL = true: list[start[pixel] + cursor[pixel]++] = triangle
template <bool FILL>
__device__ __forceinline__ void emit_pixel(uint64_t px, uint32_t tri, uint32_t *count, const uint32_t *start, uint32_t *list)
{
    const uint32_t slot = atomicAdd(&count[px], 1u);
    if (FILL) list[start[px] + slot] = tri;
}

template <bool FILL>
__global__ __launch_bounds__(256) void cam_rasterize(uint32_t W, uint32_t H, uint32_t T, const F2 *__restrict__ pos, uint32_t *count,
                                                     const uint32_t *__restrict__ start, uint32_t *list, uint32_t *bigList, uint32_t *bigCount)
{
    const uint32_t t = blockIdx.x * 256 + threadIdx.x;
    if (t >= T) return;
    const RectSetup s = rtbuild::rect_setup(W, H, pos[3 * (size_t)t], pos[3 * (size_t)t + 1], pos[3 * (size_t)t + 2]);
    const uint64_t area = (s.x1 >= s.x0 && s.y1 >= s.y0) ? (uint64_t)(s.x1 - s.x0 + 1) * (uint64_t)(s.y1 - s.y0 + 1) : 0;
    if (area > RT_BIG_RECT) { // left to a whole workgroup (same triangles in both passes: the decision is a function of the triangle)
        if (!FILL) bigList[atomicAdd(bigCount, 1u)] = t;
        return;
    }
    if (s.aOnScreen) emit_pixel<FILL>(rtbuild::rect_a_pixel(s, W), t, count, start, list);
    for (uint32_t x = s.x0; x <= s.x1; ++x)
        for (uint32_t y = s.y0; y <= s.y1; ++y) {
            if (x == s.ax && y == s.ay) continue;
            if (rtbuild::rect_pixel_test(s, x, y)) emit_pixel<FILL>((uint64_t)x + (uint64_t)y * (uint64_t)W, t, count, start, list);
        }
}

template <bool FILL>
__global__ __launch_bounds__(256) void cam_rasterize_big(uint32_t W, uint32_t H, const F2 *__restrict__ pos, uint32_t *count,
                                                         const uint32_t *__restrict__ start, uint32_t *list, const uint32_t *__restrict__ bigList,
                                                         const uint32_t *__restrict__ bigCount)
{
    for (uint32_t b = blockIdx.x; b < bigCount[0]; b += gridDim.x) {
        const uint32_t t = bigList[b];
        const RectSetup s = rtbuild::rect_setup(W, H, pos[3 * (size_t)t], pos[3 * (size_t)t + 1], pos[3 * (size_t)t + 2]);
        if (threadIdx.x == 0 && s.aOnScreen) emit_pixel<FILL>(rtbuild::rect_a_pixel(s, W), t, count, start, list);
        const uint64_t w = (uint64_t)(s.x1 - s.x0 + 1), h = (uint64_t)(s.y1 - s.y0 + 1);
        for (uint64_t i = threadIdx.x; i < w * h; i += 256) {
            const uint32_t x = s.x0 + (uint32_t)(i % w), y = s.y0 + (uint32_t)(i / w);
            if (x == s.ax && y == s.ay) continue;
            if (rtbuild::rect_pixel_test(s, x, y)) emit_pixel<FILL>((uint64_t)x + (uint64_t)y * (uint64_t)W, t, count, start, list);
        }
    }
}

__global__ __launch_bounds__(256) void cam_sort_pixels(uint64_t P, const uint32_t *__restrict__ start, const uint32_t *__restrict__ count, uint32_t *list,
                                                       uint32_t *__restrict__ end)
{
    const uint64_t p = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (p >= P) return;
    const uint32_t first = start[p], n = count[p];
    end[p] = first + n;
    uint32_t *l = list + first;
    for (uint32_t i = 1; i < n; ++i) { // lists are short (a few entries); entries are distinct triangles
        const uint32_t v = l[i];
        uint32_t j = i;
        while (j > 0 && l[j - 1] > v) { l[j] = l[j - 1]; --j; }
        l[j] = v;
    }
}

struct Buffers { // frees what it holds
    void *p[10] = { nullptr };
    int n = 0;
    template <class T> hipError_t alloc(T **dst, size_t count)
    {
        void *q = nullptr;
        const hipError_t e = hipMalloc(&q, (count ? count : 1) * sizeof(T));
        if (e == hipSuccess) { p[n++] = q; *dst = (T *)q; }
        return e;
    }
    ~Buffers() { for (int i = 0; i < n; ++i) (void)hipFree(p[i]); }
};

#define BUILD_OK(expr) do { if ((expr) != hipSuccess) return -4; } while (0)

} // namespace

extern "C" int rtHipBuildCameraListDevice(int device, cl_uint W, cl_uint H, const cl_float eye[4], const cl_float eyeToTopLeft[4],
                                          const cl_float leftToRight[4], const cl_float topToBottom[4], cl_float pixelSizeInv,
                                          cl_uint vertexCount, cl_uint triangleCount, const cl_float3 *vertex, const cl_int3 *triIndex,
                                          cl_uint **outStart, cl_uint **outEnd, cl_uint **outList, uint64_t *outListSize, double *deviceMs)
{
    if (!outStart || !outEnd || !outList || !outListSize || W == 0 || H == 0) return -1;
    int nDev = 0;
    if (hipGetDeviceCount(&nDev) != hipSuccess || device < 0 || device >= nDev) return -5; // no CPU fallback: use rtHipBuildCameraList for that
    BUILD_OK(hipSetDevice(device));
    const uint64_t P = (uint64_t)W * H;
    const uint32_t T = triangleCount;
    for (uint32_t t = 0; t < T; ++t)
        for (int k = 0; k < 3; ++k)
            if ((uint32_t)triIndex[t].s[k] >= vertexCount) return -6; // would be an out-of-bounds gather on the device
    Camera cam{ F3{ eye[0], eye[1], eye[2] }, F3{ eyeToTopLeft[0], eyeToTopLeft[1], eyeToTopLeft[2] },
                F3{ leftToRight[0], leftToRight[1], leftToRight[2] }, F3{ topToBottom[0], topToBottom[1], topToBottom[2] }, pixelSizeInv };
    Buffers buf;
    float4 *dVertex = nullptr; int4 *dIndex = nullptr; F2 *dPos = nullptr;
    uint32_t *dCount = nullptr, *dStart = nullptr, *dEnd = nullptr, *dList = nullptr, *dBigList = nullptr, *dBigCount = nullptr;
    BUILD_OK(buf.alloc(&dVertex, vertexCount)); BUILD_OK(buf.alloc(&dIndex, T)); BUILD_OK(buf.alloc(&dPos, (size_t)3 * T));
    BUILD_OK(buf.alloc(&dCount, P)); BUILD_OK(buf.alloc(&dStart, P + 1)); BUILD_OK(buf.alloc(&dEnd, P));
    BUILD_OK(buf.alloc(&dBigList, T)); BUILD_OK(buf.alloc(&dBigCount, 1));
    BUILD_OK(hipMemcpy(dVertex, vertex, (size_t)vertexCount * 16, hipMemcpyHostToDevice));
    BUILD_OK(hipMemcpy(dIndex, triIndex, (size_t)T * 16, hipMemcpyHostToDevice));
    hipEvent_t e0, e1;
    BUILD_OK(hipEventCreate(&e0)); BUILD_OK(hipEventCreate(&e1));
    BUILD_OK(hipEventRecord(e0, nullptr));
    const uint32_t tBlocks = (T + 255) / 256, bigBlocks = 1024;
    BUILD_OK(hipMemsetAsync(dCount, 0, P * 4, nullptr));
    BUILD_OK(hipMemsetAsync(dBigCount, 0, 4, nullptr));
    if (T) {
        hipLaunchKernelGGL(cam_project, dim3(tBlocks), dim3(256), 0, nullptr, cam, T, dVertex, dIndex, dPos);
        hipLaunchKernelGGL(cam_rasterize<false>, dim3(tBlocks), dim3(256), 0, nullptr, W, H, T, dPos, dCount, dStart, dList, dBigList, dBigCount);
        hipLaunchKernelGGL(cam_rasterize_big<false>, dim3(bigBlocks), dim3(256), 0, nullptr, W, H, dPos, dCount, dStart, dList, dBigList, dBigCount);
    }
    // exclusive scan of the counts -> start; the total is start[P-1] + count[P-1]
    void *tmp = nullptr; size_t tmpBytes = 0;
    BUILD_OK(hipcub::DeviceScan::ExclusiveSum(nullptr, tmpBytes, dCount, dStart, (int)P, nullptr));
    BUILD_OK(buf.alloc((char **)&tmp, tmpBytes));
    BUILD_OK(hipcub::DeviceScan::ExclusiveSum(tmp, tmpBytes, dCount, dStart, (int)P, nullptr));
    uint32_t lastStart = 0, lastCount = 0;
    BUILD_OK(hipMemcpy(&lastStart, dStart + (P - 1), 4, hipMemcpyDeviceToHost));
    BUILD_OK(hipMemcpy(&lastCount, dCount + (P - 1), 4, hipMemcpyDeviceToHost));
    const uint64_t total = (uint64_t)lastStart + lastCount; // (a sum above 2^32 has wrapped: checked below through the counts)
    BUILD_OK(buf.alloc(&dList, total));
    BUILD_OK(hipMemsetAsync(dCount, 0, P * 4, nullptr));
    if (T) {
        hipLaunchKernelGGL(cam_rasterize<true>, dim3(tBlocks), dim3(256), 0, nullptr, W, H, T, dPos, dCount, dStart, dList, dBigList, dBigCount);
        hipLaunchKernelGGL(cam_rasterize_big<true>, dim3(bigBlocks), dim3(256), 0, nullptr, W, H, dPos, dCount, dStart, dList, dBigList, dBigCount);
    }
    hipLaunchKernelGGL(cam_sort_pixels, dim3((uint32_t)((P + 255) / 256)), dim3(256), 0, nullptr, P, dStart, dCount, dList, dEnd);
    BUILD_OK(hipGetLastError());
    BUILD_OK(hipEventRecord(e1, nullptr));
    BUILD_OK(hipEventSynchronize(e1));
    float ms = 0.f;
    BUILD_OK(hipEventElapsedTime(&ms, e0, e1));
    (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
    if (deviceMs) *deviceMs = ms;

    cl_uint *start = (cl_uint *)std::malloc((size_t)P * 4), *end = (cl_uint *)std::malloc((size_t)P * 4);
    cl_uint *list = (cl_uint *)std::malloc((size_t)(total ? total : 1) * 4);
    if (!start || !end || !list) { std::free(start); std::free(end); std::free(list); return -2; }
    if (hipMemcpy(start, dStart, (size_t)P * 4, hipMemcpyDeviceToHost) != hipSuccess || hipMemcpy(end, dEnd, (size_t)P * 4, hipMemcpyDeviceToHost) != hipSuccess ||
        (total && hipMemcpy(list, dList, (size_t)total * 4, hipMemcpyDeviceToHost) != hipSuccess)) {
        std::free(start); std::free(end); std::free(list);
        return -4;
    }
    *outStart = start; *outEnd = end; *outList = list; *outListSize = total;
    return 0;
}
