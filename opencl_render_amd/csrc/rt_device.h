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
    // a cell visit costs as few divergent load instructions as possible:
    //   gridBlockSparse[i] {occupancy word lo, hi, rank}: rank = number of non-empty cells in all blocks before this one, blocks
    //                     in the order (cx>>2) + 64*(cy>>2) + 4096*(cz>>2); 12 B per block, one dwordx2/x3 load.  The table
    //                     is indexed SPARSELY by (cx>>2) | (cy>>2)<<8 | (cz>>2)<<16, so that the byte offset of a cell's
    //                     block is 3*(cell & 0xFCFCFC) for a cell packed cx | cy<<8 | cz<<16 (two instructions):
    //                     50 MB of address space, 3 MiB of touched lines
    //   dense cell id     k = rank + popcount(word & ((1<<bit)-1)), bit = (cx&3) | (cy&3)<<2 | (cz&3)<<4
    //   pairRec[i]        64-byte record of a (cell, triangle) pair: {a.xyz, triangleId} {n.xyz, count} {ab.xyz, abab} {ac.xyz, acac}
    //                     (abac and 1/(abac^2-abab*acac) are recomputed in the test).  Records [0, cellCount) are the FIRST
    //                     candidate of cell k at index k itself, `count` = candidates of the cell: a cell visit is ONE
    //                     dependent gather (the typical cell of a fine scene holds one triangle), where a {first, last} range
    //                     table in between cost a second 128-byte fabric request per visit (profiles/r02_*: the trace kernel
    //                     runs at ~90 % of the chip's L2-miss request rate, two requests per occupied cell).  The further
    //                     candidates of cell k, in list order, sit at rest .. rest + count - 2 (records [cellCount, pairs)).
    //                     The first record's `count` word is min(count, RT_PAIR_MANY) | rest << 4; a cell with RT_PAIR_MANY
    //                     or more candidates has its exact count in the `count` word of its first FURTHER record.  Pair
    //                     indices therefore stay below 2^28 (build_grid refuses larger scenes: 17 GB of records).
    const uint32_t *gridBlockSparse;
    uint32_t planesTame;     // every grid plane is 0 or has 2^-60 <= |p| <= 2^39 (see wf_trace_kernel, tame_quotient)
    uint32_t cellCount; // non-empty cells
    const float *pairRec;
    // materials
    uint32_t materialCount, texelCount;
    // matRec [M][8 words = 32 B]: one descriptor per channel (colour, reflection, transparency, bump, luminance), loaded in one
    // burst when a hit is shaded instead of a chain of dependent loads per channel (size, then start, then texel):
    //   0                      the channel is absent (0 x 0)
    //   0x80000000 | b<<16|g<<8|r  a one-texel channel: the texel itself (what the reference's scenes are made of: render.cpp:1243-1275)
    //   0x40000000             an image: size and start are read from matSize / matStart as before
    const uint32_t *matRec;
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

// error bits of the device-side input validation (rt_scene_prep.hip)
#define RT_PREP_ERR_TRI_INDEX 1u     // a triangle references a vertex that does not exist
#define RT_PREP_ERR_TRI_MATERIAL 2u  // a triangle's material id is >= materialCount
#define RT_PREP_ERR_CAM_ENTRY 4u     // a camera list entry is not a triangle
#define RT_PREP_ERR_CAM_RANGE 8u     // a pixel's candidate range reaches past the camera list
#define RT_PREP_ERR_GRID_MONOTONE 16u // scenePixelTriangleListStart is not monotone
#define RT_PREP_ERR_GRID_ENTRY 32u   // a grid list entry is not a triangle

// ---- wavefront pipeline buffers (rt_wavefront.hip) -------------------------------------------------------------------
// A "path" is one sample of one pixel whose primary ray hit something; it gets a dense id `a` (allocated by the
// primary stage) and lives in HBM between stages as structure-of-arrays state.  Rays that need the grid are appended
// to a request queue.
//   round r:   logic(r)   reads  req[r&1][q].path + res[q]  for the entries of counts[r%3]   -> appends to req[(r+1)&1], counts[(r+1)%3]
//              sort(r+1)  turns the new requests into self-contained entries, longest predicted walk first
//              trace(r+1) walks the grid for every entry                                      -> writes res[q]
#define RT_WF_MAX_ROUNDS 100000
#define RT_WF_SHARDS 256      // paths are born into one of this many shards (primary workgroup % RT_WF_SHARDS)
#define RT_WF_QSHARDS (2 * RT_WF_SHARDS) // queue slices: [0,SHARDS) main requests (one per waiting path), [SHARDS,2*SHARDS) look-ahead requests
#define RT_WF_SORT_BINS 64    // walk-length classes of the sorted trace input (wf_setup_kernel), 0 = longest
#define RT_WF_SORT_COPIES 4   // independent histograms (workgroup % copies) to spread the atomics
#define RT_WF_ORDER_LENGTH 0u
#define RT_WF_ORDER_APPENDED 1u
#define RT_WF_ORDER_REGION 2u
#define RT_WF_REGION_SHIFT 6  // a region is 64 x 64 x 64 cells: 4 x 4 x 4 = 64 regions = RT_WF_SORT_BINS
// host-visible status words of a tile group (RtWavefront::hostStatus)
#define RT_WF_STATUS_ERROR 0  // RT_WF_ERR_* bits, sticky until the host clears them
#define RT_WF_STATUS_WAITING 1 // paths still waiting for the grid when the issued rounds of a batch were over, summed over the batches since the
                               // host last cleared it: non-zero = the frame is INCOMPLETE (the planned number of rounds was too small)
#define RT_WF_STATUS_BATCHES 2 // batches whose issued rounds are over (a progress count for the host)
#define RT_WF_STATUS_WORDS 16
#define RT_PAIR_MANY 15u            // a first pair record's count field saturates here (rt_device.h, pairRec)
#define RT_PAIR_LIMIT (1u << 28)    // pair indices must fit the record's rest field and the trace kernel's 29-bit key field
#define RT_WF_ROUND_LOG 64     // rounds of a batch whose trace-input size is logged for the launch plan (RtWavefront::roundLog)
#define RT_WF_ERR_SPIN 1u     // wf_trace_kernel's walk guard tripped: rays were abandoned, the frame is invalid
#define RT_WF_ERR_GRID 2u     // a planned frame's trace grid was smaller than the round's entries: entries were not traced; the host renders
                              // the frame again with the worst-case grid (rt_api.cpp, frame_finish)
struct RtWavefront {
    uint32_t capacity;       // paths that fit (pixels of this instance's tiles x samplesInBatch)
    uint32_t sampleBase;     // samples sampleBase+1 .. sampleBase+samplesInBatch are in flight (1-based ids, raytrace.c:612-653)
    uint32_t samplesInBatch;
    uint32_t lookAhead;      // 1: a path traces its next ring entry while the current hit's shadow ray is in flight
    uint32_t fastQuotient;   // 1: waves whose rays all have tame exponents skip the scaling / fix-up instructions of the quotients (RT_WF_FAST_QUOTIENT=0 turns it off)
    uint32_t spinLimit;      // walk phases a wave of wf_trace_kernel may run before it gives up and raises RT_WF_ERR_SPIN (RT_WF_SPIN_LIMIT, default 16384)
    uint32_t *hostStatus;    // pinned HOST words mapped into the device (RT_WF_STATUS_*): written by kernels, read by the host after a sync
    uint32_t *roundLog;      // [RT_WF_ROUND_LOG] entries (all segments) the trace kernel of round r found, for sizing later frames' launches
    uint32_t segLen[5];      // aimed-at cell visits per segment for rounds with >= segRays[0] | >= segRays[1] | >= segRays[2] | >= segRays[3] | fewer rays
    uint32_t segRays[4];     // (RT_WF_SEG="a,b,c,d,e", RT_WF_SEG_RAYS="a,b,c,d"; defaults 4096,384,96,64,16 and 700000,300000,100000,30000)
    // per-path state, indexed by path id
    unsigned long long *rng; // generator state (raytrace_opencl.c:474-481), already moved past the current hit's light draws
    unsigned long long *rngL; // lightCount > 1 only: where the current hit's NEXT light set-up draws from
    uint4 *meta;             // x: output slot (localPixel*samplesInBatch + sb)  y: localPixel  w: hit triangle
                             // z: head | tail<<4 | stage<<8 | attenuation stored<<9 | look-ahead state<<10 | look-ahead ring index<<12 | light<<16
    unsigned long long *laKey; // answer (hitKey) of the path's look-ahead ray while it waits to be consumed
    uint32_t *laSlot;        // queue index of the look-ahead request issued last round
    float4 *outc;            // accumulated colour xyz
    float4 *ring;            // [capacity][12][3] the pixel's ray queue (:459-468): o.xyz,tmin | d.xyz,excluded | weight.xyz, bounces<<1|fromCamera;
                             // slot `head` is the ray in flight
    // what a hit still needs once its spawns are made and a shadow ray is in flight (everything else was folded at shading time):
    float4 *shP;             // P.xyz = (1-out)*weight*(1-transparency)*texture, the factors of :649-651 in the reference's order; w: N.L of this light
    float4 *shFace;          // xyz: the face[] entry that :647 will select (the other one is dead); w: 1 if front facing
    float4 *shAtt;           // shadow attenuation so far (only once a transparent occluder was met, :616-625)
    float4 *shN;             // lightCount > 1 only: shading normal for the next light's set-up (the hit point rides in the request)
    // ray requests / results.  Queue slice s holds entries [s*shardCap, s*shardCap + counts[s]); a path born into shard s appends
    // its main requests to slice s and its look-ahead requests to slice RT_WF_SHARDS+s, so appends hit 512 different counters (a
    // single address sustains only ~90 atomics/us) and a slice can never overflow: it holds at most the paths of its shard.
    float4 *reqO[2], *reqD[2]; // o.xyz,tmin | d.xyz,tmax
    uint2 *reqX[2];            // excluded triangle, path id
    uint4 *res;                // round 0 only: the primary hit of the path -- triangle, t, l1, l2 (float bits)
    unsigned long long *hitKey; // per request: segment << 32 | pair index of the hit in the lowest segment that has one; all ones = no hit
    uint32_t shardCap;         // entries per queue slice = paths per shard (multiple of 256)
    uint32_t *counts;          // [3][RT_WF_QSHARDS] queue lengths; round r reads [r%3], appends to [(r+1)%3]; zeroed in-stream by the logic kernel
    // Length-sorted trace input (wf_setup_kernel / wf_scatter_kernel): a round's requests become self-contained 64-byte entries
    // {q, cell, endCell, excluded} {dx,dy,dz,tmin} {o.xyz,tmax} {d.xyz,-} (DDA start state computed once), keyed by the PREDICTED
    // number of cell visits (exact for rays that hit nothing) and counting-sorted longest first, so that a wave holds rays of
    // similar length and the longest walks of the round start first.
    // A long ray becomes several entries (SEGMENTS, rt_wavefront.hip): segment 0 sits at its request's queue index (region A,
    // [0, 2*capacity)), further segments are packed into region B ([2*capacity, 2*capacity + extraCap)).
    uint4 *stageEnt;           // [2*capacity + extraCap][4] entries, {.., cell | (bin | copy<<6)<<24, ..} .. {.., rank in workgroup | segment<<24}
    uint4 *sortedEnt;          // [2*capacity + extraCap][4] the entries of an APPENDED round (small rounds skip the sort), same layout
    uint32_t *sortedIdx;       // [2*capacity + extraCap] a SORTED round: index into stageEnt of the entry at each sorted position
    uint32_t *sortRank;        // [2*capacity + extraCap] rank of the entry inside its (bin, copy) class; 0xffffffff = an unused reservation
    uint8_t *sortTag;          // [2*capacity + extraCap] the class: bin | copy << 6 (what wf_scatter_kernel needs of an entry, 5 bytes instead of a 64-byte line)
    uint32_t *sortExtra;       // [1] entries in region B this round
    uint32_t extraCap;         // capacity of region B (multiple of 256)
    uint32_t *sortHist;        // [RT_WF_SORT_COPIES][RT_WF_SORT_BINS] entries per (copy, bin) of the current round
    uint32_t *sortTotal;       // [2] entries at the front of the sorted array | how the round was ordered: 0 by walk length, 1 appended (region B
                               // holds entries too), 2 by grid region (RT_WF_ORDER_*)
    uint32_t appendRays;       // rounds with fewer rays than this skip the counting sort (RT_WF_APPEND_RAYS, default 150000)
    uint32_t regionRays;       // rounds with at least this many rays are cut at REGION boundaries and sorted by region (RT_WF_REGION_RAYS;
                               // 0xffffffff = never): see wf_setup_kernel
    float4 *sampleOut;         // [capacity] finished colour per output slot
};

#endif
