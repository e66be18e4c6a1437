// rt_device.h -- device-side scene description shared by the HIP kernels (rt_kernels.hip) and the host API
// (rt_api.cpp).  Plain C++ POD, passed to kernels by value in the kernarg segment.
#ifndef RT_DEVICE_H
#define RT_DEVICE_H

#include <stdint.h>
#include <hip/hip_vector_types.h> // float4 / uint4 / uint2 for the host-side view of the device structs

#define RT_TILE 128            // tile edge (pixels); the reference's NDRange granule (raytrace.c:507)
#define RT_TILE_PIXELS (RT_TILE * RT_TILE)
#define RT_PATCH 16            // a 256-thread workgroup renders a 16x16 pixel patch; each wave an 8x8 quadrant
#define RT_RING 12             // ray queue slots per pixel (raytrace_opencl.c:404)
#define RT_GRID_DIV 256

// Layout of the device-resident scene.
//
//  triRec   [T][16 floats]  pre-resolved intersection record, one 64-byte line per triangle:
//                           a.xyz | ab.xyz | ac.xyz | n.xyz (= cross(ac,ab)) | abab abac acac 1/(abac^2-abab*acac)
//                           Every field is the value the reference recomputes per test (raytrace_opencl.c:131-149),
//                           produced once on the device with the same fp32 operations, so results are bit-identical
//                           while a candidate test costs one 64-B gather instead of 16 B index + 3 x 16 B vertices.
//  triShade [T][24 floats]  what only a shaded hit needs: b.xyz c.xyz | nA nB nC | uvA uvB uvC | materialId | pad
//  boxMin   [3][257]        split planes, one array per axis (staged into LDS by every workgroup)
//  camStart/camEnd          per pixel of this scene's tiles, TILE-MAJOR: index = slot*128*128 + ly*128 + lx
//  tileBuf  [slot][3][128*128] u16 planes R,G,B
struct RtDevScene {
    // camera (raytrace.h:61-66)
    float eye[3], topLeft[3], lr[3], tb[3];
    float pixelSizeInv;
    uint32_t width, height, sampleCount;
    // tiles rendered by this scene instance
    const uint32_t *tileIds;
    uint32_t tileCount, tilesX;
    const uint32_t *camStart, *camEnd, *camList;
    // geometry
    uint32_t triangleCount;
    const float *triRec;
    const float *triShade;
    // grid
    const float *boxMin;
    const uint32_t *gridStart, *gridList;
    // occupancy of the grid, one 64-bit word per 4x4x4 block of cells (2 MiB: L2-resident): word (cx>>2) + 64*(cy>>2)
    // + 4096*(cz>>2), bit (cx&3) | (cy&3)<<2 | (cz&3)<<4 set iff the cell's list is non-empty.  Lets the DDA walk
    // empty space without touching the 67 MB gridStart array; empty cells have no effect on the result.
    const unsigned long long *gridBits;
    // Dense view of the same grid for the wavefront trace kernel (no 67 MB sparse array, no list->record hop), shaped so
    // a cell visit costs as few divergent load instructions as possible (the texture addresser is the bottleneck):
    //   gridBlock[block]  {occupancy word lo, hi, rank}: rank = number of non-empty cells in all blocks before `block`
    //                     (12 B, one dwordx3 load per block entered; 3 MiB, L2-resident)
    //   dense cell id     k = rank + popcount(word & ((1<<bit)-1))
    //   cellRange[k]      {first, last} pair index range of the cell (one 8-byte load)
    //   pairRec[i]        64-byte record of pair i, replicated per (cell, triangle) pair so a cell's candidates are
    //                     contiguous: {a.xyz, triangleId} {n.xyz, -} {ab.xyz, abab} {ac.xyz, acac}
    //                     (plane test = first two float4; abac and 1/(abac^2-abab*acac) are recomputed when needed)
    const uint32_t *gridBlock;
    // the same 12-byte entries, indexed SPARSELY by (cx>>2) | (cy>>2)<<8 | (cz>>2)<<16 so that the byte offset of a cell's
    // block is 3*(cell & 0xFCFCFC) for a cell packed cx | cy<<8 | cz<<16 (wf_trace_sorted_kernel); 50 MB of address
    // space, the same 3 MiB of touched lines
    const uint32_t *gridBlockSparse;
    const uint2 *cellRange;
    const float *pairRec;
    // materials
    uint32_t materialCount, texelCount;
    const uint32_t *matSize; // 2 x 5 per material
    const int32_t *matStart; // 5 per material
    const uint8_t *textures; // 4 bytes per texel
    const float *bumpSin, *bumpCos; // [256*256] host-libm tables, index (hE<<8)|h0 (see rt_api.cpp)
    // lights
    uint32_t lightCount;
    const int32_t *lightType;
    const float *lightPos, *lightDir, *lightCol; // 4 floats each
    const float *lightRadius, *lightHalfAtt;
    const float *lightSpread; // per light: (float)(sin((r/2)*PI_F/180) * sqrt(|dir|^2)), host libm (raytrace_opencl.c:594)
    // outputs
    uint16_t *tileBuf;
    unsigned long long *stats; // 7 counters, only touched by the counted kernel variant
};

// ---- wavefront pipeline buffers (rt_wavefront.hip) -------------------------------------------------------------------
// A "path" is one sample of one pixel whose primary ray hit something; it gets a dense id `a` (allocated by the
// primary stage) and lives in HBM between stages as structure-of-arrays state.  Rays that need the grid are appended
// to a request queue; a persistent trace kernel consumes the queue with every lane busy (idle lanes refill from it).
//   round r:   logic(r)  reads  req[r&1].path + res[q]   for q < counts[r]      -> appends to req[(r+1)&1], counts[r+1]
//              trace(r+1) reads req[(r+1)&1][q]           for q < counts[r+1]    -> writes res[q]
#define RT_WF_MAX_ROUNDS 100000
#define RT_WF_SHARDS 256      // every queue is cut into this many independent slices, each with its own counter
#define RT_WF_PASSES 8        // at most this many trace passes per round; the last one runs every ray to its end
#define RT_WF_SORT_BINS 64    // walk-length classes of the sorted trace input: bin = (767 - predicted visits) / 12, 0 = longest
#define RT_WF_SORT_COPIES 4   // independent histograms (workgroup % copies) to spread the atomics
struct RtWavefront {
    uint32_t capacity;       // paths that fit (pixels of this instance's tiles x samplesInBatch)
    uint32_t sampleBase;     // samples sampleBase+1 .. sampleBase+samplesInBatch are in flight (1-based ids, raytrace.c:612-653)
    uint32_t samplesInBatch;
    // per-path state, indexed by path id
    unsigned long long *rng; // generator state (raytrace_opencl.c:474-481)
    unsigned long long *rngL; // second cursor into the same stream: where the CURRENT hit's light set-ups draw from (the
                             // main cursor has already been moved past all draws of that hit when its spawns were computed)
    uint4 *meta;             // x: output slot (localPixel*samplesInBatch + sb)  y: localPixel
                             // z: head | tail<<4 | stage<<8 | lookahead state<<10 | lookahead ring index<<12 | light<<16   w: hit triangle
    uint4 *laRes;            // answer of the path's look-ahead ray while it waits to be consumed
    float4 *outc;            // accumulated colour xyz, w: hit distance
    float4 *cur0, *cur1, *cur2; // ray in flight: o.xyz,tmin | d.xyz,excluded | weight.xyz, bounces<<1|fromCamera
    float4 *shN, *shWhere, *shF0, *shF1, *shAtt, *shToL; // n.xyz,l1 | where.xyz,l2 | face0.xyz,lmin | face1.xyz,lmax | atten | toLight
    float4 *shTex, *shTransp, *shRefl, *shLum;           // the hit's channel texels
    float4 *ring;            // [capacity][12][3] queued rays, same packing as cur0..2
    // ray requests / results
    float4 *reqO[2], *reqD[2]; // o.xyz,tmin | d.xyz,tmax
    uint2 *reqX[2];            // excluded triangle, path id | (1u<<31 for the look-ahead request, which sits right after its path's main request)
    uint4 *res;                // hit triangle (0xffffffff = none), t, l1, l2 (float bits)
    // Queues are SHARDED: a path is born into shard (primary workgroup % RT_WF_SHARDS) and everything it ever emits --
    // ray requests, continuation entries -- goes to the same shard's slice [shard*shardCap, (shard+1)*shardCap) of the
    // queue arrays.  Appends therefore hit RT_WF_SHARDS different counters instead of one (a single address sustains
    // only ~90 atomics/us, which was the whole cost of the primary and logic kernels), and a slice can never overflow:
    // it holds at most the paths born into it.  Counters are small rings zeroed in-stream by the logic kernel.
    uint32_t shardCap;         // PATHS per shard (multiple of 256): path-state arrays hold RT_WF_SHARDS*shardCap entries
    uint32_t queueStride;      // queue ENTRIES per shard = 2*shardCap: a path may have two rays in flight (its hit's shadow
                               // ray and a look-ahead trace of the next ring entry); req*/res/cont arrays use this stride
    uint32_t *counts;          // [3][RT_WF_SHARDS] fresh-request queue length; round r reads [r%3], appends to [(r+1)%3]
    uint32_t *contCounts;      // [RT_WF_PASSES][RT_WF_SHARDS] continuation entries appended by each trace pass
    uint4 *cont[2];            // [capacity][4], self-contained: {q, cell, endCell, excluded} {dx,dy,dz,tmin} {o.xyz,tmax} {d.xyz,-}
    // Length-sorted trace input (wf_setup_kernel / wf_scatter_kernel): a round's requests are turned into self-contained
    // entries (DDA start state computed once), keyed by the PREDICTED number of cell visits (exact for rays that hit
    // nothing) and counting-sorted longest first, so that a wave holds rays of similar length and the longest walks of
    // the round start first.  cont[0] is the staging area, cont[1] the sorted array (no continuation passes in this mode).
    uint32_t sortMode;         // 0: trace reads the sharded request queues directly; 1: setup -> scatter -> wf_trace_kernel<SORTED>;
                               // 2 (default): setup -> scatter -> wf_trace_sorted_kernel
    uint32_t lookAhead;        // 1: a path traces its next ring entry while the current hit's shadow ray is in flight
    uint32_t *sortHist;        // [RT_WF_SORT_COPIES][RT_WF_SORT_BINS] entries per (copy, bin) of the current round
    uint32_t *sortTotal;       // [1] entries in the sorted array
    float4 *sampleOut;         // [capacity] finished colour per output slot
};

#endif
