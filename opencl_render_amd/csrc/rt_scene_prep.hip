// rt_scene_prep.hip -- device side of a scene upload ("next" row 2): everything that used to be a host
// loop over the ABI arrays between the upload and the first frame now runs on the GPU, on the uploaded arrays themselves:
//   * validation of what the kernels will index with (vertex / material / triangle ids, camera list ranges, monotone grid
//     starts) -- a bad id would be an out-of-bounds gather; the reference does no such check (raytrace.c:344-489);
//   * the tile-major view of the per-pixel candidate ranges (index slot*128*128 + ly*128 + lx) for this instance's tiles;
//   * the dense view of the 256^3 grid for the wavefront trace kernel (rt_device.h): occupancy word and rank per 4x4x4 block,
//     pair order "first candidate of every non-empty cell at its dense id, then everybody's further candidates" with the
//     word each first record carries about its cell, the sparsely indexed block table.
// Each is a stream kernel over arrays that are in HBM anyway (67 MB of grid starts, the lists); on the host they were
// single-threaded loops of 30-100 ms per scene at 1 M triangles.
#include <hip/hip_runtime.h>
#include <hipcub/hipcub.hpp>

#include "rt_device.h"

namespace {

constexpr uint32_t BLOCKS = (RT_GRID_DIV / 4) * (RT_GRID_DIV / 4) * (RT_GRID_DIV / 4); // 262144

__global__ __launch_bounds__(256) void rtp_check_triangles(uint32_t T, uint32_t V, uint32_t M, const int4 *__restrict__ triIndex,
                                                           const int *__restrict__ triMaterial, uint32_t *err)
{
    const uint32_t t = blockIdx.x * 256 + threadIdx.x;
    if (t >= T) return;
    const int4 vi = triIndex[t];
    uint32_t bad = 0;
    if ((uint32_t)vi.x >= V || (uint32_t)vi.y >= V || (uint32_t)vi.z >= V) bad |= RT_PREP_ERR_TRI_INDEX;
    if (triMaterial[t] >= (int)M) bad |= RT_PREP_ERR_TRI_MATERIAL; // negative = no material (render.cpp:1098)
    if (bad) atomicOr(err, bad);
}

__global__ __launch_bounds__(256) void rtp_check_list(uint64_t n, const uint32_t *__restrict__ list, uint32_t T, uint32_t bit, uint32_t *err)
{
    const uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n && list[i] >= T) atomicOr(err, bit);
}

__global__ __launch_bounds__(256) void rtp_check_grid_start(const uint32_t *__restrict__ gridStart, uint32_t *err)
{
    const uint32_t c = blockIdx.x * 256 + threadIdx.x; // RT_GRID_DIV^3 threads
    if (gridStart[c] > gridStart[c + 1]) atomicOr(err, RT_PREP_ERR_GRID_MONOTONE);
}

// Tile-major candidate ranges of this instance's tiles; a range with end < start counts as empty, one that reaches past the
// list is an error.  Pixels of a tile that lie outside the image get the empty range 0..0.
__global__ __launch_bounds__(256) void rtp_camera_tile_major(uint32_t W, uint32_t H, uint32_t tilesX, const uint32_t *__restrict__ tileIds, uint32_t tileCount,
                                                             const uint32_t *__restrict__ camStart, const uint32_t *__restrict__ camEnd, uint64_t camListSize,
                                                             uint32_t *__restrict__ outStart, uint32_t *__restrict__ outEnd, uint32_t *err)
{
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i >= tileCount * RT_TILE_PIXELS) return;
    const uint32_t slot = i / RT_TILE_PIXELS, in = i % RT_TILE_PIXELS;
    const uint32_t tile = tileIds[slot];
    const uint32_t gx = (tile % tilesX) * RT_TILE + in % RT_TILE, gy = (tile / tilesX) * RT_TILE + in / RT_TILE;
    uint32_t a = 0, b = 0;
    if (gx < W && gy < H) {
        const uint64_t p = (uint64_t)gy * W + gx;
        a = camStart[p]; b = camEnd[p];
        if (b < a) b = a;
        if ((uint64_t)b > camListSize) { atomicOr(err, RT_PREP_ERR_CAM_RANGE); a = b = 0; }
    }
    outStart[i] = a; outEnd[i] = b;
}

// One thread per 4x4x4 block: occupancy word (bit (cx&3) | (cy&3)<<2 | (cz&3)<<4), non-empty cells and further candidates
// (pairs beyond the first of each cell) of the block.
__global__ __launch_bounds__(256) void rtp_block_words(const uint32_t *__restrict__ gridStart, unsigned long long *__restrict__ words,
                                                       uint32_t *__restrict__ blockCells, uint32_t *__restrict__ blockRest)
{
    const uint32_t b = blockIdx.x * 256 + threadIdx.x; // BLOCKS threads
    const uint32_t bx = b & 63u, by = (b >> 6) & 63u, bz = b >> 12;
    unsigned long long w = 0;
    uint32_t cells = 0, rest = 0;
    for (uint32_t z = 0; z < 4; ++z)
        for (uint32_t y = 0; y < 4; ++y) {
            const uint32_t row = (bx * 4) + RT_GRID_DIV * (by * 4 + y) + RT_GRID_DIV * RT_GRID_DIV * (bz * 4 + z);
            uint32_t s[5];
#pragma unroll
            for (int x = 0; x < 5; ++x) s[x] = gridStart[row + x];
#pragma unroll
            for (uint32_t x = 0; x < 4; ++x) {
                const uint32_t n = s[x + 1] - s[x];
                if (n) { w |= 1ull << (x | (y << 2) | (z << 4)); ++cells; rest += n - 1; }
            }
        }
    words[b] = w; blockCells[b] = cells; blockRest[b] = rest;
}

// One thread per block again, now that every block knows its first dense cell id (rank) and where its further candidates go:
// writes the sparsely indexed block table {word lo, word hi, rank} and the pair order (rt_device.h, pairRec).
__global__ __launch_bounds__(256) void rtp_block_emit(const uint32_t *__restrict__ gridStart, const uint32_t *__restrict__ gridList,
                                                      const unsigned long long *__restrict__ words, const uint32_t *__restrict__ rank,
                                                      const uint32_t *__restrict__ restBase, const uint32_t *__restrict__ blockCells,
                                                      uint32_t *__restrict__ sparse, uint32_t *__restrict__ pairOrder, uint32_t *__restrict__ pairCount)
{
    const uint32_t b = blockIdx.x * 256 + threadIdx.x;
    const uint32_t bx = b & 63u, by = (b >> 6) & 63u, bz = b >> 12;
    unsigned long long w = words[b];
    const uint32_t first = rank[b];
    const size_t at = (size_t)(bx | (by << 8) | (bz << 16)) * 3;
    sparse[at + 0] = (uint32_t)w; sparse[at + 1] = (uint32_t)(w >> 32); sparse[at + 2] = first;
    const uint32_t cellTotal = rank[BLOCKS - 1] + blockCells[BLOCKS - 1]; // the further candidates start after all first ones
    uint32_t dense = first, restAt = cellTotal + restBase[b];
    while (w) {
        const int bit = __ffsll((long long)w) - 1;
        w &= w - 1;
        const uint32_t cell = (bx * 4 + (bit & 3)) + RT_GRID_DIV * (by * 4 + ((bit >> 2) & 3)) + RT_GRID_DIV * RT_GRID_DIV * (bz * 4 + (bit >> 4));
        const uint32_t s = gridStart[cell], n = gridStart[cell + 1] - s;
        pairOrder[dense] = gridList[s];
        // what the first record says about its cell: candidates (15 = "15 or more") | where the further ones start << 4
        pairCount[dense] = (n < RT_PAIR_MANY ? n : RT_PAIR_MANY) | (restAt << 4);
        // the first of the further records carries the exact count (read only for cells with RT_PAIR_MANY or more)
        for (uint32_t i = 1; i < n; ++i) { pairOrder[restAt] = gridList[s + i]; pairCount[restAt] = (i == 1u) ? n : 0u; ++restAt; }
        ++dense;
    }
}

} // namespace

extern "C" hipError_t rtp_validate(uint32_t T, uint32_t V, uint32_t M, const void *triIndex, const int *triMaterial, uint64_t camListSize,
                                   const uint32_t *camList, const uint32_t *gridStart, uint64_t gridListSize, const uint32_t *gridList, uint32_t *err,
                                   hipStream_t stream)
{
    if (T && triIndex) hipLaunchKernelGGL(rtp_check_triangles, dim3((T + 255) / 256), dim3(256), 0, stream, T, V, M, (const int4 *)triIndex, triMaterial, err);
    if (camListSize) hipLaunchKernelGGL(rtp_check_list, dim3((uint32_t)((camListSize + 255) / 256)), dim3(256), 0, stream, camListSize, camList, T, RT_PREP_ERR_CAM_ENTRY, err);
    if (gridListSize) hipLaunchKernelGGL(rtp_check_list, dim3((uint32_t)((gridListSize + 255) / 256)), dim3(256), 0, stream, gridListSize, gridList, T, RT_PREP_ERR_GRID_ENTRY, err);
    if (gridStart) hipLaunchKernelGGL(rtp_check_grid_start, dim3((RT_GRID_DIV * RT_GRID_DIV * RT_GRID_DIV) / 256), dim3(256), 0, stream, gridStart, err);
    return hipGetLastError();
}

extern "C" hipError_t rtp_camera_ranges(uint32_t W, uint32_t H, uint32_t tilesX, const uint32_t *tileIds, uint32_t tileCount, const uint32_t *camStart,
                                        const uint32_t *camEnd, uint64_t camListSize, uint32_t *outStart, uint32_t *outEnd, uint32_t *err, hipStream_t stream)
{
    const uint32_t n = tileCount * RT_TILE_PIXELS;
    if (n) hipLaunchKernelGGL(rtp_camera_tile_major, dim3((n + 255) / 256), dim3(256), 0, stream, W, H, tilesX, tileIds, tileCount, camStart, camEnd,
                              camListSize, outStart, outEnd, err);
    return hipGetLastError();
}

// scratch: 4 x BLOCKS u32 + the scan's temporary storage; `scanBytes` in/out like hipcub (call with scratch == nullptr to size it)
extern "C" hipError_t rtp_dense_grid(const uint32_t *gridStart, const uint32_t *gridList, unsigned long long *words, uint32_t *sparse, uint32_t *pairOrder,
                                     uint32_t *pairCount, void *scratch, size_t *scratchBytes, hipStream_t stream)
{
    size_t scanBytes = 0;
    hipError_t e = hipcub::DeviceScan::ExclusiveSum(nullptr, scanBytes, (const uint32_t *)nullptr, (uint32_t *)nullptr, (int)BLOCKS, stream);
    if (e != hipSuccess) return e;
    const size_t need = (size_t)4 * BLOCKS * sizeof(uint32_t) + scanBytes;
    if (!scratch) { *scratchBytes = need; return hipSuccess; }
    if (*scratchBytes < need) return hipErrorInvalidValue;
    uint32_t *blockCells = (uint32_t *)scratch, *blockRest = blockCells + BLOCKS, *rank = blockRest + BLOCKS, *restBase = rank + BLOCKS;
    void *scanTmp = restBase + BLOCKS;
    hipLaunchKernelGGL(rtp_block_words, dim3(BLOCKS / 256), dim3(256), 0, stream, gridStart, words, blockCells, blockRest);
    e = hipcub::DeviceScan::ExclusiveSum(scanTmp, scanBytes, blockCells, rank, (int)BLOCKS, stream);
    if (e != hipSuccess) return e;
    e = hipcub::DeviceScan::ExclusiveSum(scanTmp, scanBytes, blockRest, restBase, (int)BLOCKS, stream);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(rtp_block_emit, dim3(BLOCKS / 256), dim3(256), 0, stream, gridStart, gridList, words, rank, restBase, blockCells, sparse, pairOrder,
                       pairCount);
    return hipGetLastError();
}
