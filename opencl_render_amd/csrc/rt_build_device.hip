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
//   cam_dedup_*      the reference's neighbour de-duplication (:580-613: a pixel whose list equals its left, else its upper
//                    neighbour's shares that neighbour's storage), so Start/End/list equal the host builder's arrays
#include <hip/hip_runtime.h>
#include <hipcub/hipcub.hpp>

#include "raytrace_hip.h"
#include "rt_build_shared.h"

#include <chrono>
#include <algorithm>
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

// ---- neighbour de-duplication (trianglelist.cpp:580-613) ---------------------------------------------------------------------
// The reference scans the pixels in order; a pixel whose list equals that of its left neighbour -- else of the neighbour above
// -- takes over that neighbour's (final) range, and the storage of the remaining lists is closed up.  Equality is a property of
// the lists' contents, so which pixels alias which does not depend on the scan: mark every pixel's parent (left, up or
// itself), compact the lists of the roots with an exclusive scan of their sizes, and resolve every pixel to the root of its
// chain by pointer jumping (chains are at most W + H long: ceil(log2(W + H)) rounds).
__global__ __launch_bounds__(256) void cam_dedup_mark(uint32_t W, uint64_t P, const uint32_t *__restrict__ start, const uint32_t *__restrict__ count,
                                                      const uint32_t *__restrict__ list, uint32_t *__restrict__ parent, uint32_t *__restrict__ keep)
{
    const uint64_t p = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (p >= P) return;
    const uint32_t x = (uint32_t)(p % W), n = count[p];
    const uint32_t *mine = list + start[p];
    auto same = [&](uint64_t q) {
        if (count[q] != n) return false;
        const uint32_t *other = list + start[q];
        for (uint32_t i = 0; i < n; ++i)
            if (mine[i] != other[i]) return false;
        return true;
    };
    uint32_t par = (uint32_t)p;
    if (x > 0 && same(p - 1)) par = (uint32_t)(p - 1);
    else if (p >= W && same(p - W)) par = (uint32_t)(p - W);
    parent[p] = par;
    keep[p] = (par == (uint32_t)p) ? n : 0u;
}

__global__ __launch_bounds__(256) void cam_dedup_jump(uint64_t P, uint32_t *parent)
{
    const uint64_t p = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (p >= P) return;
    const uint32_t a = parent[p];
    parent[p] = parent[a]; // in place: whatever value is read is an ancestor, so every round at least halves the distance to the root
}

__global__ __launch_bounds__(256) void cam_dedup_compact(uint64_t P, const uint32_t *__restrict__ start, const uint32_t *__restrict__ count,
                                                         const uint32_t *__restrict__ list, const uint32_t *__restrict__ parent,
                                                         const uint32_t *__restrict__ newStart, uint32_t *__restrict__ outStart,
                                                         uint32_t *__restrict__ outEnd, uint32_t *__restrict__ outList)
{
    const uint64_t p = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (p >= P) return;
    const uint32_t root = parent[p];
    const uint32_t at = newStart[root], n = count[root];
    outStart[p] = at;
    outEnd[p] = at + n;
    if (root == (uint32_t)p) {
        const uint32_t *src = list + start[p];
        for (uint32_t i = 0; i < n; ++i) outList[at + i] = src[i];
    }
}

struct Buffers { // frees what it holds
    void *p[24] = { nullptr };
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
    const uint32_t pBlocks = (uint32_t)((P + 255) / 256);
    hipLaunchKernelGGL(cam_sort_pixels, dim3(pBlocks), dim3(256), 0, nullptr, P, dStart, dCount, dList, dEnd);
    // neighbour de-duplication: parents, sizes kept, new starts of the roots, roots of everybody, compacted storage
    uint32_t *dParent = nullptr, *dKeep = nullptr, *dNewStart = nullptr, *dOutList = nullptr, *dOutStart = nullptr;
    BUILD_OK(buf.alloc(&dParent, P)); BUILD_OK(buf.alloc(&dKeep, P)); BUILD_OK(buf.alloc(&dNewStart, P)); BUILD_OK(buf.alloc(&dOutList, total));
    BUILD_OK(buf.alloc(&dOutStart, P));
    hipLaunchKernelGGL(cam_dedup_mark, dim3(pBlocks), dim3(256), 0, nullptr, W, P, dStart, dCount, dList, dParent, dKeep);
    BUILD_OK(hipcub::DeviceScan::ExclusiveSum(tmp, tmpBytes, dKeep, dNewStart, (int)P, nullptr));
    for (uint64_t reach = 1; reach < (uint64_t)W + H; reach *= 2) hipLaunchKernelGGL(cam_dedup_jump, dim3(pBlocks), dim3(256), 0, nullptr, P, dParent);
    hipLaunchKernelGGL(cam_dedup_compact, dim3(pBlocks), dim3(256), 0, nullptr, P, dStart, dCount, dList, dParent, dNewStart, dOutStart, dEnd, dOutList);
    BUILD_OK(hipGetLastError());
    BUILD_OK(hipEventRecord(e1, nullptr));
    BUILD_OK(hipEventSynchronize(e1));
    float ms = 0.f;
    BUILD_OK(hipEventElapsedTime(&ms, e0, e1));
    (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
    if (deviceMs) *deviceMs = ms;

    uint32_t lastNew = 0, lastKeep = 0;
    BUILD_OK(hipMemcpy(&lastNew, dNewStart + (P - 1), 4, hipMemcpyDeviceToHost));
    BUILD_OK(hipMemcpy(&lastKeep, dKeep + (P - 1), 4, hipMemcpyDeviceToHost));
    const uint64_t kept = (uint64_t)lastNew + lastKeep; // entries left after the de-duplication
    cl_uint *start = (cl_uint *)std::malloc((size_t)P * 4), *end = (cl_uint *)std::malloc((size_t)P * 4);
    cl_uint *list = (cl_uint *)std::malloc((size_t)(kept ? kept : 1) * 4);
    if (!start || !end || !list) { std::free(start); std::free(end); std::free(list); return -2; }
    if (hipMemcpy(start, dOutStart, (size_t)P * 4, hipMemcpyDeviceToHost) != hipSuccess || hipMemcpy(end, dEnd, (size_t)P * 4, hipMemcpyDeviceToHost) != hipSuccess ||
        (kept && hipMemcpy(list, dOutList, (size_t)kept * 4, hipMemcpyDeviceToHost) != hipSuccess)) {
        std::free(start); std::free(end); std::free(list);
        return -4;
    }
    *outStart = start; *outEnd = end; *outList = list; *outListSize = kept;
    return 0;
}

// ---- scene grid (counterpart of SceneTriangleList::New, trianglelist.cpp:655-737) ----------------------------------------
// Split planes at the vertex quantiles per axis (:657-678, index in 64 bits as in rt_builders.cpp) from radix-sorted
// coordinates; every triangle is flood-filled over the cells its clipped polygon touches, starting at the cell of vertex a
// (FillCube :452-503 with BoxIntersectsTriangle / Cull from rt_build_shared.h).  The fill finds a SET of cells (the face-
// connected component of cells that pass the test around the start cell), which does not depend on the order in which cells
// are visited -- so a thread can do a small triangle on its own, and a workgroup can share a big one level by level.
// (cell, triangle) pairs become 64-bit keys cell << 32 | triangle, radix-sorted: the lists come out cell-major with ascending
// triangles, which is what the reference's sort on cell*T+tri gives (:707).
#define RT_FILL_LOCAL 48u        // cells a thread's own fill may hold before the triangle is handed to a workgroup
#define RT_FILL_QUEUE (1u << 24) // cells per workgroup fill: every cell of the grid (a cell enters a fill once), so that a floor across a scene of a few vertices -- whose 256 planes per axis are a handful of distinct values -- cannot overflow it; 4 GB of build scratch of the 288
#define RT_FILL_GROUPS 64u

namespace {

constexpr uint32_t GRID_CELLS = (uint32_t)rtbuild::DIV * rtbuild::DIV * rtbuild::DIV;

__global__ __launch_bounds__(256) void grid_axis_values(uint32_t V, const float4 *__restrict__ vertex, float *__restrict__ vals)
{
    const uint32_t v = blockIdx.x * 256 + threadIdx.x;
    if (v >= V) return;
    const float4 p = vertex[v];
    vals[v] = p.x; vals[(size_t)V + v] = p.y; vals[2 * (size_t)V + v] = p.z;
}

__global__ void grid_planes(uint32_t V, const float *__restrict__ sorted, float *__restrict__ bm) // bm[257][4]
{
    const int i = threadIdx.x, w = blockIdx.x; // 257 threads x 3 axes
    if (i > rtbuild::DIV) return;
    const float *val = sorted + (size_t)w * V;
    const uint32_t index = (uint32_t)(((uint64_t)i * (uint64_t)(V - 1)) / (uint64_t)rtbuild::DIV);
    bm[4 * i + w] = (0 < index && index < V) ? (val[index] + val[index - 1]) / 2.f : val[index];
}

__device__ __forceinline__ F3 vtx(const float4 *vertex, int i) { const float4 v = vertex[i]; return F3{ v.x, v.y, v.z }; }

// wave-aggregated append of n keys per lane; returns the lane's first slot
__device__ __forceinline__ unsigned long long append_keys(unsigned long long *cursor, uint32_t n)
{
    uint32_t incl = n;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const uint32_t up = __shfl_up(incl, off, 64);
        if ((int)(threadIdx.x & 63) >= off) incl += up;
    }
    const uint32_t total = __shfl(incl, 63, 64);
    unsigned long long base = 0;
    if ((threadIdx.x & 63) == 63 && total) base = atomicAdd(cursor, (unsigned long long)total);
    base = __shfl(base, 63, 64);
    return base + incl - n;
}

__global__ __launch_bounds__(256) void grid_fill_small(uint32_t T, const float4 *__restrict__ vertex, const int4 *__restrict__ triIndex,
                                                       const float *__restrict__ bmGlobal, unsigned long long *keys, unsigned long long *keyCursor,
                                                       unsigned long long keyCap, uint32_t *bigList, uint32_t *bigCount, uint32_t *overflow)
{
    __shared__ float bm[rtbuild::DIV + 1][4];
    for (int i = threadIdx.x; i < 4 * (rtbuild::DIV + 1); i += 256) (&bm[0][0])[i] = bmGlobal[i];
    __syncthreads();
    const uint32_t t = blockIdx.x * 256 + threadIdx.x;
    uint32_t cells[RT_FILL_LOCAL];
    uint32_t n = 0;
    if (t < T) {
        const int4 vi = triIndex[t];
        const F3 a = vtx(vertex, vi.x), b = vtx(vertex, vi.y), c = vtx(vertex, vi.z);
        int cell[3];
        rtbuild::box_address(bm, a, cell);
        cells[n++] = (uint32_t)cell[0] + (uint32_t)cell[1] * rtbuild::DIV + (uint32_t)cell[2] * rtbuild::DIV * rtbuild::DIV;
        bool big = false;
        for (uint32_t cur = 0; cur < n && !big; ++cur) {
            const uint32_t id = cells[cur];
            cell[2] = (int)(id / (rtbuild::DIV * rtbuild::DIV));
            cell[1] = (int)((id % (rtbuild::DIV * rtbuild::DIV)) / rtbuild::DIV);
            cell[0] = (int)(id % rtbuild::DIV);
            float lo[3], hi[3];
            for (int i = 0; i < 3; ++i) { lo[i] = bm[cell[i]][i]; hi[i] = bm[cell[i] + 1][i]; }
            for (int i = 0; i < 3 && !big; ++i) {
                for (int j = -1; j <= 1 && !big; j += 2) {
                    cell[i] += j;
                    if (0 <= cell[i] && cell[i] < rtbuild::DIV) {
                        const uint32_t nid = (uint32_t)cell[0] + (uint32_t)cell[1] * rtbuild::DIV + (uint32_t)cell[2] * rtbuild::DIV * rtbuild::DIV;
                        bool seen = false;
                        for (uint32_t k = 0; k < n; ++k) seen |= (cells[k] == nid);
                        if (!seen) {
                            lo[i] = bm[cell[i]][i];
                            hi[i] = bm[cell[i] + 1][i];
                            if (rtbuild::box_hits_triangle(lo, hi, a, b, c)) {
                                if (n == RT_FILL_LOCAL) big = true;
                                else cells[n++] = nid;
                            }
                        }
                    }
                    cell[i] -= j;
                }
                lo[i] = bm[cell[i]][i];
                hi[i] = bm[cell[i] + 1][i];
            }
        }
        if (big) { bigList[atomicAdd(bigCount, 1u)] = t; n = 0; }
    }
    const unsigned long long at = append_keys(keyCursor, n); // every lane of the wave arrives here
    if (at + n > keyCap) { if (n) atomicExch(overflow, 1u); return; }
    for (uint32_t k = 0; k < n; ++k) keys[at + k] = ((unsigned long long)cells[k] << 32) | t;
}

// Cells whose closed interval [plane[i], plane[i+1]] meets [vmin, vmax] on one axis: first and last index.  A cell outside that
// range lies strictly beyond the triangle on this axis, so the first clip against it removes every point (comparisons only, no
// rounding): it cannot pass the box test.  (Planes may repeat -- zero-width cells -- which is why this is not "vertex cell +- 1".)
__device__ __forceinline__ void overlap_range(const float (*bm)[4], int axis, float vmin, float vmax, int &first, int &last)
{
    int lo = 0, hi = rtbuild::DIV - 1;
    while (lo < rtbuild::DIV - 1 && bm[lo + 1][axis] < vmin) ++lo;
    while (hi > 0 && vmax < bm[hi][axis]) --hi;
    first = lo; last = hi < lo ? lo : hi;
}

// Upper bound of the cells the big triangles can fill (their overlap boxes), so the key buffer can be sized before they are filled.
__global__ __launch_bounds__(256) void grid_big_bound(const float4 *__restrict__ vertex, const int4 *__restrict__ triIndex, const float *__restrict__ bmGlobal,
                                                      const uint32_t *__restrict__ bigList, const uint32_t *__restrict__ bigCount, unsigned long long *bound)
{
    __shared__ float bm[rtbuild::DIV + 1][4];
    for (int i = threadIdx.x; i < 4 * (rtbuild::DIV + 1); i += 256) (&bm[0][0])[i] = bmGlobal[i];
    __syncthreads();
    for (uint32_t bidx = blockIdx.x * 256 + threadIdx.x; bidx < bigCount[0]; bidx += gridDim.x * 256) {
        const int4 vi = triIndex[bigList[bidx]];
        const F3 a = vtx(vertex, vi.x), b = vtx(vertex, vi.y), c = vtx(vertex, vi.z);
        const float mn[3] = { fminf(a.x, fminf(b.x, c.x)), fminf(a.y, fminf(b.y, c.y)), fminf(a.z, fminf(b.z, c.z)) };
        const float mx[3] = { fmaxf(a.x, fmaxf(b.x, c.x)), fmaxf(a.y, fmaxf(b.y, c.y)), fmaxf(a.z, fmaxf(b.z, c.z)) };
        unsigned long long n = 1;
        for (int w = 0; w < 3; ++w) { int f, l; overlap_range(bm, w, mn[w], mx[w], f, l); n *= (unsigned long long)(l - f + 1); }
        atomicAdd(bound, n + 1ull); // + the cell of vertex a, which is taken untested
    }
}

// Big triangles.  The fill finds the face-connected set of cells that pass the box test around the cell of vertex a (FillCube,
// trianglelist.cpp:452-503).  The test of a cell does not depend on how the fill reached it, so every cell of a triangle's overlap
// box is tested ONCE, by the whole GPU (grid_test_big: blockIdx.y = the triangle's slot in the current batch), into a per-slot bitmap;
// the flood itself (grid_fill_big, one workgroup per triangle) then only follows bits, level by level.  (Testing inside the flood,
// as the reference does, tests a cell once per already-filled neighbour and serialises ~700 levels of box tests for a wall-sized
// triangle: 1 s for a 146-triangle room.)
__global__ __launch_bounds__(256) void grid_test_big(const float4 *__restrict__ vertex, const int4 *__restrict__ triIndex, const float *__restrict__ bmGlobal,
                                                     const uint32_t *__restrict__ bigList, uint32_t batchBase, uint32_t *passmaps)
{
    __shared__ float bm[rtbuild::DIV + 1][4];
    __shared__ int range[6];
    for (int i = threadIdx.x; i < 4 * (rtbuild::DIV + 1); i += 256) (&bm[0][0])[i] = bmGlobal[i];
    __syncthreads();
    const uint32_t t = bigList[batchBase + blockIdx.y];
    uint32_t *pass = passmaps + (size_t)blockIdx.y * (GRID_CELLS / 32);
    const int4 vi = triIndex[t];
    const F3 a = vtx(vertex, vi.x), b = vtx(vertex, vi.y), c = vtx(vertex, vi.z);
    if (threadIdx.x < 3) {
        const int w = threadIdx.x;
        const float va = w == 0 ? a.x : (w == 1 ? a.y : a.z), vb = w == 0 ? b.x : (w == 1 ? b.y : b.z), vc = w == 0 ? c.x : (w == 1 ? c.y : c.z);
        overlap_range(bm, w, fminf(va, fminf(vb, vc)), fmaxf(va, fmaxf(vb, vc)), range[2 * w], range[2 * w + 1]);
    }
    __syncthreads();
    const uint32_t nx = (uint32_t)(range[1] - range[0] + 1), ny = (uint32_t)(range[3] - range[2] + 1), nz = (uint32_t)(range[5] - range[4] + 1);
    const unsigned long long boxCells = (unsigned long long)nx * ny * nz;
    for (unsigned long long i = (unsigned long long)blockIdx.x * 256 + threadIdx.x; i < boxCells; i += (unsigned long long)gridDim.x * 256) {
        const int cx = range[0] + (int)(i % nx), cy = range[2] + (int)((i / nx) % ny), cz = range[4] + (int)(i / ((unsigned long long)nx * ny));
        const float lo[3] = { bm[cx][0], bm[cy][1], bm[cz][2] }, hi[3] = { bm[cx + 1][0], bm[cy + 1][1], bm[cz + 1][2] };
        if (rtbuild::box_hits_triangle(lo, hi, a, b, c)) {
            const uint32_t id = (uint32_t)cx + (uint32_t)cy * rtbuild::DIV + (uint32_t)cz * rtbuild::DIV * rtbuild::DIV;
            atomicOr(&pass[id >> 5], 1u << (id & 31));
        }
    }
}

__global__ __launch_bounds__(256) void grid_fill_big(const float4 *__restrict__ vertex, const int4 *__restrict__ triIndex, const float *__restrict__ bmGlobal,
                                                     unsigned long long *keys, unsigned long long *keyCursor, unsigned long long keyCap,
                                                     const uint32_t *__restrict__ bigList, uint32_t batchBase, uint32_t *bitmaps,
                                                     uint32_t *passmaps, uint32_t *queues, uint32_t *overflow)
{
    __shared__ float bm[rtbuild::DIV + 1][4];
    __shared__ uint32_t qCount, levelStart, levelEnd;
    __shared__ unsigned long long keyBase;
    __shared__ int range[6];
    for (int i = threadIdx.x; i < 4 * (rtbuild::DIV + 1); i += 256) (&bm[0][0])[i] = bmGlobal[i];
    uint32_t *bits = bitmaps + (size_t)blockIdx.x * (GRID_CELLS / 32); // all zero between triangles
    uint32_t *pass = passmaps + (size_t)blockIdx.x * (GRID_CELLS / 32); // this slot's test results (grid_test_big); cleared at the end
    uint32_t *queue = queues + (size_t)blockIdx.x * RT_FILL_QUEUE;
    __syncthreads();
    {
        const uint32_t t = bigList[batchBase + blockIdx.x];
        const int4 vi = triIndex[t];
        const F3 a = vtx(vertex, vi.x), b = vtx(vertex, vi.y), c = vtx(vertex, vi.z);
        if (threadIdx.x < 3) {
            const int w = threadIdx.x;
            const float va = w == 0 ? a.x : (w == 1 ? a.y : a.z), vb = w == 0 ? b.x : (w == 1 ? b.y : b.z), vc = w == 0 ? c.x : (w == 1 ? c.y : c.z);
            overlap_range(bm, w, fminf(va, fminf(vb, vc)), fmaxf(va, fmaxf(vb, vc)), range[2 * w], range[2 * w + 1]);
        }
        if (threadIdx.x == 0) {
            int cell[3];
            rtbuild::box_address(bm, a, cell);
            const uint32_t id = (uint32_t)cell[0] + (uint32_t)cell[1] * rtbuild::DIV + (uint32_t)cell[2] * rtbuild::DIV * rtbuild::DIV;
            bits[id >> 5] = 1u << (id & 31);
            queue[0] = id;
            qCount = 1; levelStart = 0; levelEnd = 1;
        }
        __syncthreads();
        const uint32_t nx = (uint32_t)(range[1] - range[0] + 1), ny = (uint32_t)(range[3] - range[2] + 1), nz = (uint32_t)(range[5] - range[4] + 1);
        const unsigned long long boxCells = (unsigned long long)nx * ny * nz;
        while (levelStart < levelEnd) { // workgroup-uniform
            for (uint32_t q = levelStart + threadIdx.x; q < levelEnd; q += 256) {
                const uint32_t id = queue[q];
                int cell[3] = { (int)(id % rtbuild::DIV), (int)((id % (rtbuild::DIV * rtbuild::DIV)) / rtbuild::DIV), (int)(id / (rtbuild::DIV * rtbuild::DIV)) };
                for (int i = 0; i < 3; ++i) {
                    for (int j = -1; j <= 1; j += 2) {
                        cell[i] += j;
                        if (0 <= cell[i] && cell[i] < rtbuild::DIV) {
                            const uint32_t nid = (uint32_t)cell[0] + (uint32_t)cell[1] * rtbuild::DIV + (uint32_t)cell[2] * rtbuild::DIV * rtbuild::DIV;
                            const uint32_t mask = 1u << (nid & 31);
                            if ((__atomic_load_n(&pass[nid >> 5], __ATOMIC_RELAXED) & mask) && !(__atomic_load_n(&bits[nid >> 5], __ATOMIC_RELAXED) & mask) &&
                                !(atomicOr(&bits[nid >> 5], mask) & mask)) {
                                const uint32_t pos = atomicAdd(&qCount, 1u);
                                if (pos < RT_FILL_QUEUE) queue[pos] = nid; else atomicExch(overflow, 2u);
                            }
                        }
                        cell[i] -= j;
                    }
                }
            }
            __threadfence_block();
            __syncthreads();
            if (threadIdx.x == 0) { levelStart = levelEnd; levelEnd = qCount < RT_FILL_QUEUE ? qCount : RT_FILL_QUEUE; }
            __syncthreads();
        }
        const uint32_t n = levelEnd;
        if (threadIdx.x == 0) keyBase = atomicAdd(keyCursor, (unsigned long long)n);
        __syncthreads();
        const bool fits = keyBase + n <= keyCap;
        if (!fits && threadIdx.x == 0) atomicExch(overflow, 1u);
        for (uint32_t q = threadIdx.x; q < n; q += 256) {
            const uint32_t id = queue[q];
            if (fits) keys[keyBase + q] = ((unsigned long long)id << 32) | t;
            bits[id >> 5] = 0u; // clear what this triangle set (a word may be cleared by several threads)
        }
        for (unsigned long long i = threadIdx.x; i < boxCells; i += 256) { // and the test results (a word may be cleared repeatedly)
            const uint32_t cx = (uint32_t)range[0] + (uint32_t)(i % nx), cy = (uint32_t)range[2] + (uint32_t)((i / nx) % ny), cz = (uint32_t)range[4] + (uint32_t)(i / ((unsigned long long)nx * ny));
            pass[(cx + cy * rtbuild::DIV + cz * rtbuild::DIV * rtbuild::DIV) >> 5] = 0u;
        }
    }
}

__global__ __launch_bounds__(256) void grid_count_cells(unsigned long long n, const unsigned long long *__restrict__ keys, uint32_t *count)
{
    const unsigned long long i = (unsigned long long)blockIdx.x * 256 + threadIdx.x;
    if (i < n) atomicAdd(&count[(uint32_t)(keys[i] >> 32)], 1u);
}

__global__ __launch_bounds__(256) void grid_write_list(unsigned long long n, const unsigned long long *__restrict__ keys, uint32_t *__restrict__ list,
                                                       uint32_t *__restrict__ start)
{
    const unsigned long long i = (unsigned long long)blockIdx.x * 256 + threadIdx.x;
    if (i < n) list[i] = (uint32_t)keys[i];
    if (i == 0) start[GRID_CELLS] = (uint32_t)n; // the entry after the last cell (raytrace.c:441 relies on the +1 layout)
}

} // namespace

extern "C" int rtHipBuildSceneGridDevice(int device, cl_uint vertexCount, cl_uint triangleCount, const cl_float3 *vertex, const cl_int3 *triIndex,
                                         cl_float3 outBoxMin[257], cl_uint **outStart, cl_uint **outList, uint64_t *outListSize, double *deviceMs)
{
    if (!outBoxMin || !outStart || !outList || !outListSize) return -1;
    int nDev = 0;
    if (hipGetDeviceCount(&nDev) != hipSuccess || device < 0 || device >= nDev) return -5; // no CPU fallback: rtHipBuildSceneGrid is the host builder
    BUILD_OK(hipSetDevice(device));
    const uint32_t V = vertexCount, T = triangleCount;
    for (uint32_t t = 0; t < T; ++t)
        for (int k = 0; k < 3; ++k)
            if ((uint32_t)triIndex[t].s[k] >= V) return -6;
    Buffers buf;
    float4 *dVertex = nullptr; int4 *dIndex = nullptr;
    float *dVals = nullptr, *dSorted = nullptr, *dBm = nullptr;
    unsigned long long *dKeys = nullptr, *dKeysSorted = nullptr, *dCursor = nullptr;
    uint32_t *dBigList = nullptr, *dBigCount = nullptr, *dOverflow = nullptr, *dBitmaps = nullptr, *dQueues = nullptr, *dStart = nullptr, *dCount = nullptr;
    const unsigned long long keyCap = std::max<unsigned long long>(32ull * T, 1ull << 22);
    BUILD_OK(buf.alloc(&dVertex, V)); BUILD_OK(buf.alloc(&dIndex, T));
    BUILD_OK(buf.alloc(&dVals, (size_t)3 * V)); BUILD_OK(buf.alloc(&dSorted, (size_t)3 * V)); BUILD_OK(buf.alloc(&dBm, 4 * (rtbuild::DIV + 1)));
    BUILD_OK(buf.alloc(&dKeys, keyCap)); BUILD_OK(buf.alloc(&dKeysSorted, keyCap)); BUILD_OK(buf.alloc(&dCursor, 1));
    BUILD_OK(hipMemcpy(dVertex, vertex, (size_t)V * 16, hipMemcpyHostToDevice));
    BUILD_OK(hipMemcpy(dIndex, triIndex, (size_t)T * 16, hipMemcpyHostToDevice));
    Buffers buf2; // (Buffers holds ten pointers)
    BUILD_OK(buf2.alloc(&dBigList, T)); BUILD_OK(buf2.alloc(&dBigCount, 1)); BUILD_OK(buf2.alloc(&dOverflow, 1));
    BUILD_OK(buf2.alloc(&dBitmaps, (size_t)RT_FILL_GROUPS * (GRID_CELLS / 32))); BUILD_OK(buf2.alloc(&dQueues, (size_t)RT_FILL_GROUPS * RT_FILL_QUEUE));
    BUILD_OK(buf2.alloc(&dStart, (size_t)GRID_CELLS + 1)); BUILD_OK(buf2.alloc(&dCount, (size_t)GRID_CELLS));
    BUILD_OK(hipMemset(dBitmaps, 0, (size_t)RT_FILL_GROUPS * (GRID_CELLS / 32) * 4));
    hipEvent_t e0, e1;
    BUILD_OK(hipEventCreate(&e0)); BUILD_OK(hipEventCreate(&e1));
    BUILD_OK(hipEventRecord(e0, nullptr));
    BUILD_OK(hipMemsetAsync(dBm, 0, sizeof(float) * 4 * (rtbuild::DIV + 1), nullptr));
    BUILD_OK(hipMemsetAsync(dCursor, 0, 8, nullptr));
    BUILD_OK(hipMemsetAsync(dBigCount, 0, 4, nullptr));
    BUILD_OK(hipMemsetAsync(dOverflow, 0, 4, nullptr));
    BUILD_OK(hipMemsetAsync(dCount, 0, (size_t)GRID_CELLS * 4, nullptr));
    Buffers buf3;
    if (V) {
        hipLaunchKernelGGL(grid_axis_values, dim3((V + 255) / 256), dim3(256), 0, nullptr, V, dVertex, dVals);
        void *tmp = nullptr; size_t tmpBytes = 0;
        BUILD_OK(hipcub::DeviceRadixSort::SortKeys(nullptr, tmpBytes, dVals, dSorted, (int)V, 0, 32, nullptr));
        BUILD_OK(buf3.alloc((char **)&tmp, tmpBytes));
        for (int w = 0; w < 3; ++w)
            BUILD_OK(hipcub::DeviceRadixSort::SortKeys(tmp, tmpBytes, dVals + (size_t)w * V, dSorted + (size_t)w * V, (int)V, 0, 32, nullptr));
        hipLaunchKernelGGL(grid_planes, dim3(3), dim3(320), 0, nullptr, V, dSorted, dBm);
    }
    unsigned long long n = 0;
    uint32_t overflow = 0;
    Buffers bufRetry;
    unsigned long long cap = keyCap;
    uint32_t *dPassmaps = nullptr;
    unsigned long long *dBound = nullptr;
    BUILD_OK(buf3.alloc(&dPassmaps, (size_t)RT_FILL_GROUPS * (GRID_CELLS / 32))); BUILD_OK(buf3.alloc(&dBound, 1));
    BUILD_OK(hipMemsetAsync(dPassmaps, 0, (size_t)RT_FILL_GROUPS * (GRID_CELLS / 32) * 4, nullptr));
    for (int attempt = 0; attempt < 2; ++attempt) {
        if (T) {
            hipLaunchKernelGGL(grid_fill_small, dim3((T + 255) / 256), dim3(256), 0, nullptr, T, dVertex, dIndex, dBm, dKeys, dCursor, cap, dBigList,
                               dBigCount, dOverflow);
            // the big triangles' cells are bounded by their overlap boxes: make room for them before they are filled
            BUILD_OK(hipMemsetAsync(dBound, 0, 8, nullptr));
            hipLaunchKernelGGL(grid_big_bound, dim3(64), dim3(256), 0, nullptr, dVertex, dIndex, dBm, dBigList, dBigCount, dBound);
            unsigned long long small = 0, bound = 0;
            BUILD_OK(hipMemcpy(&small, dCursor, 8, hipMemcpyDeviceToHost));
            BUILD_OK(hipMemcpy(&bound, dBound, 8, hipMemcpyDeviceToHost));
            if (small <= cap && small + bound > cap && small + bound <= (1ull << 31)) {
                unsigned long long *bigger = nullptr, *biggerSorted = nullptr;
                BUILD_OK(bufRetry.alloc(&bigger, (size_t)(small + bound))); BUILD_OK(bufRetry.alloc(&biggerSorted, (size_t)(small + bound)));
                if (small) BUILD_OK(hipMemcpy(bigger, dKeys, (size_t)small * 8, hipMemcpyDeviceToDevice));
                dKeys = bigger; dKeysSorted = biggerSorted; cap = small + bound;
            }
            uint32_t bigTotal = 0;
            BUILD_OK(hipMemcpy(&bigTotal, dBigCount, 4, hipMemcpyDeviceToHost));
            for (uint32_t base = 0; base < bigTotal; base += RT_FILL_GROUPS) { // RT_FILL_GROUPS bitmaps: that many big triangles at a time
                const uint32_t batch = std::min<uint32_t>(RT_FILL_GROUPS, bigTotal - base);
                hipLaunchKernelGGL(grid_test_big, dim3(128, batch), dim3(256), 0, nullptr, dVertex, dIndex, dBm, dBigList, base, dPassmaps);
                hipLaunchKernelGGL(grid_fill_big, dim3(batch), dim3(256), 0, nullptr, dVertex, dIndex, dBm, dKeys, dCursor, cap, dBigList, base,
                                   dBitmaps, dPassmaps, dQueues, dOverflow);
            }
        }
        BUILD_OK(hipMemcpy(&n, dCursor, 8, hipMemcpyDeviceToHost));
        BUILD_OK(hipMemcpy(&overflow, dOverflow, 4, hipMemcpyDeviceToHost));
        if (overflow != 1u || attempt == 1) break;
        // More pairs than the key buffer holds after all (very many mid-sized triangles).  The cursor has counted them all, like
        // the reference's own overflow pass (trianglelist.cpp:696-706): fill again into a buffer of exactly that size.
        if (n > 0xffffffffull) return -3;
        cap = n;
        BUILD_OK(bufRetry.alloc(&dKeys, (size_t)cap)); BUILD_OK(bufRetry.alloc(&dKeysSorted, (size_t)cap));
        BUILD_OK(hipMemsetAsync(dCursor, 0, 8, nullptr));
        BUILD_OK(hipMemsetAsync(dBigCount, 0, 4, nullptr));
        BUILD_OK(hipMemsetAsync(dOverflow, 0, 4, nullptr));
    }
    if (overflow || n > cap) return -7; // a single fill larger than the workgroup queue (2^22 cells), or a second overflow
    if (n > 0xffffffffull) return -3;
    uint32_t *dList = nullptr;
    BUILD_OK(buf3.alloc(&dList, (size_t)n));
    if (n) {
        void *tmp = nullptr; size_t tmpBytes = 0;
        BUILD_OK(hipcub::DeviceRadixSort::SortKeys(nullptr, tmpBytes, dKeys, dKeysSorted, (int)n, 0, 56, nullptr));
        BUILD_OK(buf3.alloc((char **)&tmp, tmpBytes));
        BUILD_OK(hipcub::DeviceRadixSort::SortKeys(tmp, tmpBytes, dKeys, dKeysSorted, (int)n, 0, 56, nullptr));
        hipLaunchKernelGGL(grid_count_cells, dim3((uint32_t)((n + 255) / 256)), dim3(256), 0, nullptr, n, dKeysSorted, dCount);
    }
    {
        void *tmp = nullptr; size_t tmpBytes = 0;
        BUILD_OK(hipcub::DeviceScan::ExclusiveSum(nullptr, tmpBytes, dCount, dStart, (int)GRID_CELLS, nullptr));
        BUILD_OK(buf3.alloc((char **)&tmp, tmpBytes));
        BUILD_OK(hipcub::DeviceScan::ExclusiveSum(tmp, tmpBytes, dCount, dStart, (int)GRID_CELLS, nullptr));
    }
    hipLaunchKernelGGL(grid_write_list, dim3((uint32_t)((std::max<unsigned long long>(n, 1) + 255) / 256)), dim3(256), 0, nullptr, n, dKeysSorted, dList, dStart);
    BUILD_OK(hipGetLastError());
    BUILD_OK(hipEventRecord(e1, nullptr));
    BUILD_OK(hipEventSynchronize(e1));
    float ms = 0.f;
    BUILD_OK(hipEventElapsedTime(&ms, e0, e1));
    (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
    if (deviceMs) *deviceMs = ms;

    cl_uint *start = (cl_uint *)std::malloc(((size_t)GRID_CELLS + 1) * 4), *list = (cl_uint *)std::malloc((size_t)(n ? n : 1) * 4);
    if (!start || !list) { std::free(start); std::free(list); return -2; }
    if (hipMemcpy(outBoxMin, dBm, sizeof(float) * 4 * (rtbuild::DIV + 1), hipMemcpyDeviceToHost) != hipSuccess ||
        hipMemcpy(start, dStart, ((size_t)GRID_CELLS + 1) * 4, hipMemcpyDeviceToHost) != hipSuccess ||
        (n && hipMemcpy(list, dList, (size_t)n * 4, hipMemcpyDeviceToHost) != hipSuccess)) {
        std::free(start); std::free(list);
        return -4;
    }
    *outStart = start; *outList = list; *outListSize = n;
    return 0;
}
