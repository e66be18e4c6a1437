// rt_builders.cpp -- host-side producers of the hot path's list inputs.
//
// Counterparts of the reference's CameraTriangleList::New (source/util/trianglelist.cpp:520-626) and
// SceneTriangleList::New (:655-737).  The membership tests are restated in the same fp32 arithmetic and
// evaluation order (FillRectangle :131-217, Cull :381-430, BoxIntersectsTriangle :433-449, FillCube :452-503,
// quantile planes :657-678), so the lists hold the same (cell, triangle) pairs in the same ascending order.
// What is NOT taken over is the reference's strategy: no 2 GiB u64 key scratch + quicksort (:522-523,:565,
// :681-682,:707) and no 2 MiB bitset memset per triangle (:457).  Pairs are binned with a count/prefix/fill
// pass, and the flood fill clears only the bits it set, which turns the 61 s grid build of a 1 M-triangle
// soup into seconds and lets triangles be processed on all host threads.
//
// Parity note: trianglelist.cpp cannot be compiled here (it needs the Maxon SDK's c4d.h), so these builders are
// pinned by restatement only ("parity unpinned" in DESIGN.md).  The trace kernel does not depend on that: any
// lists are just inputs to it, and the oracle consumes the very same arrays.
#include "raytrace_hip.h"
#include "rt_build_shared.h"

#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>

namespace {

using rtbuild::F2;
using rtbuild::F3;
using rtbuild::dot3;
using rtbuild::cross3;
using rtbuild::to_u32;
using rtbuild::Camera;
using rtbuild::camera_position;
using rtbuild::fill_rectangle;

inline F3 ld3(const cl_float3 &v) { return F3{ v.s[0], v.s[1], v.s[2] }; }

int hw_threads(int threads)
{
    if (threads > 0) return threads;
    unsigned n = std::thread::hardware_concurrency();
    return n ? (int)n : 1;
}

template <class Fn> void parallel_chunks(int threads, uint64_t n, Fn fn)
{
    threads = (int)std::min<uint64_t>((uint64_t)hw_threads(threads), std::max<uint64_t>(1, n));
    if (threads <= 1) { fn(0, (uint64_t)0, n); return; }
    std::vector<std::thread> pool;
    for (int t = 0; t < threads; ++t) {
        uint64_t lo = n * (uint64_t)t / (uint64_t)threads, hi = n * (uint64_t)(t + 1) / (uint64_t)threads;
        pool.emplace_back([=] { fn(t, lo, hi); });
    }
    for (auto &th : pool) th.join();
}

// ---- camera lists: the arithmetic lives in rt_build_shared.h (shared with the device builder) ----------------

// ---- scene grid ---------------------------------------------------------------------------------------------

constexpr int DIV = rtbuild::DIV; // trianglelist.h:110

// cull / box_hits_triangle / box_address: rt_build_shared.h (shared with the device builder)
using rtbuild::box_hits_triangle;
using rtbuild::box_address;

// trianglelist.cpp:452-503: face-connected flood fill from the cell of vertex a.  `cells` receives the ids;
// `bits` is a DIV^3-bit visited set, all-zero on entry and on exit.
void fill_cube(const float (*bm)[4], std::vector<uint8_t> &bits, std::vector<uint32_t> &cells, F3 a, F3 b, F3 c)
{
    int cell[3];
    box_address(bm, a, cell);
    uint32_t id = (uint32_t)cell[0] + (uint32_t)cell[1] * DIV + (uint32_t)cell[2] * DIV * DIV;
    cells.clear();
    bits[id >> 3] |= (uint8_t)(1u << (id & 7));
    cells.push_back(id);
    for (size_t cur = 0; cur < cells.size(); ++cur) {
        id = cells[cur];
        cell[2] = (int)(id / (DIV * DIV));
        cell[1] = (int)((id % (DIV * DIV)) / DIV);
        cell[0] = (int)(id % DIV);
        float lo[3], hi[3];
        for (int i = 0; i < 3; ++i) { lo[i] = bm[cell[i]][i]; hi[i] = bm[cell[i] + 1][i]; }
        for (int i = 0; i < 3; ++i) {
            for (int j = -1; j <= 1; j += 2) {
                cell[i] += j;
                if (0 <= cell[i] && cell[i] < DIV) {
                    uint32_t nid = (uint32_t)cell[0] + (uint32_t)cell[1] * DIV + (uint32_t)cell[2] * DIV * DIV;
                    uint8_t mask = (uint8_t)(1u << (nid & 7));
                    if (!(bits[nid >> 3] & mask)) {
                        lo[i] = bm[cell[i]][i];
                        hi[i] = bm[cell[i] + 1][i];
                        if (box_hits_triangle(lo, hi, a, b, c)) {
                            bits[nid >> 3] |= mask;
                            cells.push_back(nid);
                        }
                    }
                }
                cell[i] -= j;
            }
            lo[i] = bm[cell[i]][i];
            hi[i] = bm[cell[i] + 1][i];
        }
    }
    for (uint32_t v : cells) bits[v >> 3] = 0; // clear only what was touched (a byte may be cleared repeatedly)
}

template <class T> T *alloc_n(uint64_t n) { return (T *)std::malloc((size_t)std::max<uint64_t>(1, n) * sizeof(T)); }

} // namespace

extern "C" {

void rtHipFree(void *p) { std::free(p); }

int rtHipBuildCameraList(cl_uint W, cl_uint H, const cl_float eye[4], const cl_float eyeToTopLeft[4],
                         const cl_float leftToRight[4], const cl_float topToBottom[4], cl_float pixelSizeInv,
                         cl_uint triangleCount, const cl_float3 *vertex, const cl_int3 *triIndex, int threads,
                         cl_uint **outStart, cl_uint **outEnd, cl_uint **outList, uint64_t *outListSize)
{
    if (!outStart || !outEnd || !outList || !outListSize || W == 0 || H == 0) return -1;
    const uint64_t P = (uint64_t)W * H;
    Camera cam{ F3{ eye[0], eye[1], eye[2] }, F3{ eyeToTopLeft[0], eyeToTopLeft[1], eyeToTopLeft[2] },
                F3{ leftToRight[0], leftToRight[1], leftToRight[2] }, F3{ topToBottom[0], topToBottom[1], topToBottom[2] },
                pixelSizeInv };

    // project once
    std::vector<F2> pos((size_t)triangleCount * 3);
    parallel_chunks(threads, triangleCount, [&](int, uint64_t lo, uint64_t hi) {
        for (uint64_t t = lo; t < hi; ++t)
            for (int k = 0; k < 3; ++k) pos[3 * t + k] = camera_position(cam, ld3(vertex[triIndex[t].s[k]]));
    });

    // pass 1: count per pixel
    std::vector<std::atomic<uint32_t>> count(P);
    for (auto &c : count) c.store(0, std::memory_order_relaxed);
    parallel_chunks(threads, triangleCount, [&](int, uint64_t lo, uint64_t hi) {
        for (uint64_t t = lo; t < hi; ++t)
            fill_rectangle(W, H, pos[3 * t], pos[3 * t + 1], pos[3 * t + 2],
                           [&](uint64_t px) { count[px].fetch_add(1, std::memory_order_relaxed); });
    });
    cl_uint *start = alloc_n<cl_uint>(P), *end = alloc_n<cl_uint>(P);
    if (!start || !end) { std::free(start); std::free(end); return -2; }
    uint64_t total = 0;
    for (uint64_t p = 0; p < P; ++p) { start[p] = (cl_uint)total; total += count[p].load(std::memory_order_relaxed); end[p] = (cl_uint)total; }
    if (total > 0xffffffffull) { std::free(start); std::free(end); return -3; }
    cl_uint *list = alloc_n<cl_uint>(total);
    if (!list) { std::free(start); std::free(end); return -2; }

    // pass 2: fill (cursor = start + running count), then put each pixel's entries in ascending triangle order,
    // which is the order the reference's sort on pixel*T+tri keys gives (:161,:565-574)
    for (auto &c : count) c.store(0, std::memory_order_relaxed);
    parallel_chunks(threads, triangleCount, [&](int, uint64_t lo, uint64_t hi) {
        for (uint64_t t = lo; t < hi; ++t)
            fill_rectangle(W, H, pos[3 * t], pos[3 * t + 1], pos[3 * t + 2], [&](uint64_t px) {
                list[start[px] + count[px].fetch_add(1, std::memory_order_relaxed)] = (cl_uint)t;
            });
    });
    parallel_chunks(threads, P, [&](int, uint64_t lo, uint64_t hi) {
        for (uint64_t p = lo; p < hi; ++p) std::sort(list + start[p], list + end[p]);
    });

    // neighbour de-duplication: a pixel whose list equals its left (else upper) neighbour's aliases it (:580-613)
    uint64_t squeezed = 0;
    for (uint64_t p = 0; p < P; ++p) {
        const uint32_t x = (uint32_t)(p % W), y = (uint32_t)(p / W);
        const cl_uint n = end[p] - start[p];
        std::memmove(list + (start[p] - squeezed), list + start[p], (size_t)n * sizeof(cl_uint));
        start[p] -= (cl_uint)squeezed;
        end[p] -= (cl_uint)squeezed;
        bool aliased = false;
        if (0 < x && n == end[p - 1] - start[p - 1] && 0 == std::memcmp(list + start[p - 1], list + start[p], (size_t)n * sizeof(cl_uint))) {
            squeezed += n; start[p] = start[p - 1]; end[p] = end[p - 1]; aliased = true;
        }
        if (0 < y && !aliased && n == end[p - W] - start[p - W] && 0 == std::memcmp(list + start[p - W], list + start[p], (size_t)n * sizeof(cl_uint))) {
            squeezed += n; start[p] = start[p - W]; end[p] = end[p - W];
        }
    }
    *outStart = start; *outEnd = end; *outList = list; *outListSize = total - squeezed;
    return 0;
}

int rtHipBuildSceneGrid(cl_uint vertexCount, cl_uint triangleCount, const cl_float3 *vertex, const cl_int3 *triIndex,
                        int threads, cl_float3 outBoxMin[257], cl_uint **outStart, cl_uint **outList, uint64_t *outListSize)
{
    if (!outBoxMin || !outStart || !outList || !outListSize) return -1;
    static_assert(sizeof(cl_float3) == 16, "padded float3");
    float (*bm)[4] = reinterpret_cast<float (*)[4]>(outBoxMin);
    std::memset(bm, 0, sizeof(float) * 4 * (DIV + 1));

    // split planes at vertex quantiles, midway between neighbours (:657-678).  The reference computes the quantile index
    // (i * (vertexCount - 1)) / 256 in 32 bits, which wraps once vertexCount exceeds 2^24: its planes stop being monotone
    // and nearly all triangles end up in a few cells (measured: 5.4 s per 4K frame at 10 M triangles).  The index is
    // computed in 64 bits here -- the same value wherever the reference's does not overflow, the intended quantile where it
    // does.  The lists are inputs of the hot path: its parity does not depend on this choice.
    if (0 < vertexCount) {
        auto axis = [&](int w) { // the three axes are independent: one thread each when threads are allowed
            std::vector<float> val(vertexCount);
            for (cl_uint v = 0; v < vertexCount; ++v) val[v] = vertex[v].s[w];
            std::sort(val.begin(), val.end());
            for (int i = 0; i < DIV + 1; ++i) {
                const cl_uint index = (cl_uint)(((uint64_t)i * (uint64_t)(vertexCount - 1)) / (uint64_t)DIV);
                if (0 < index && index < vertexCount) bm[i][w] = (val[index] + val[index - 1]) / 2.f;
                else bm[i][w] = val[index];
            }
        };
        if (hw_threads(threads) >= 3) {
            std::thread tx(axis, 0), ty(axis, 1);
            axis(2);
            tx.join(); ty.join();
        } else for (int w = 0; w < 3; ++w) axis(w);
    }

    const uint64_t CELLS = (uint64_t)DIV * DIV * DIV;
    const int nthreads = (int)std::min<uint64_t>((uint64_t)hw_threads(threads), std::max<uint64_t>(1, triangleCount));
    // per-thread (cell, triangle) pairs; triangles are dealt in contiguous ranges so each thread's pairs are
    // already ascending in triangle id
    std::vector<std::vector<uint32_t>> cellOf(nthreads), triOf(nthreads);
    parallel_chunks(nthreads, triangleCount, [&](int t, uint64_t lo, uint64_t hi) {
        std::vector<uint8_t> bits(CELLS / 8, 0);
        std::vector<uint32_t> cells;
        for (uint64_t tri = lo; tri < hi; ++tri) {
            const cl_int3 &vi = triIndex[tri];
            fill_cube(bm, bits, cells, ld3(vertex[vi.s[0]]), ld3(vertex[vi.s[1]]), ld3(vertex[vi.s[2]]));
            for (uint32_t c : cells) { cellOf[t].push_back(c); triOf[t].push_back((uint32_t)tri); }
        }
    });

    cl_uint *start = alloc_n<cl_uint>(CELLS + 1);
    if (!start) return -2;
    std::memset(start, 0, (CELLS + 1) * sizeof(cl_uint));
    uint64_t total = 0;
    for (int t = 0; t < nthreads; ++t) { total += cellOf[t].size(); for (uint32_t c : cellOf[t]) ++start[c + 1]; }
    if (total > 0xffffffffull) { std::free(start); return -3; }
    for (uint64_t c = 1; c <= CELLS; ++c) start[c] += start[c - 1]; // :717-719
    cl_uint *list = alloc_n<cl_uint>(total);
    if (!list) { std::free(start); return -2; }
    {
        // threads hold ascending triangle ranges, so filling thread by thread keeps each cell's list ascending
        std::vector<cl_uint> cursor(start, start + CELLS);
        for (int t = 0; t < nthreads; ++t)
            for (size_t k = 0; k < cellOf[t].size(); ++k) list[cursor[cellOf[t][k]]++] = triOf[t][k];
    }
    *outStart = start; *outList = list; *outListSize = total;
    return 0;
}

} // extern "C"
