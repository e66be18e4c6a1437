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
    //   pairRec[i]        64-byte record of a (cell, triangle) pair: {a.xyz, triangleId} {n.xyz, count} {ab.xyz, abac} {ac.xyz, 1/(abac^2-abab*acac)}
    //                     (dot(ab,ab) and dot(ac,ac) are recomputed in the test).  Records [0, cellCount) are the FIRST
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
    // [3][256] per axis: the cell that holds the middle of the i-th of 256 equal steps across the grid's box -- a one-read ESTIMATE of
    // a point's cell, good enough to predict how long a walk will be (the order in which rays are traced; never what they find)
    const uint8_t *cellLut;
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
// primary stage) and lives in HBM between stages as structure-of-arrays state.  A ray that needs the grid becomes a
// self-contained 64-byte TRACE ENTRY, written by the kernel that spawns it.
//   round r:   logic(r)   reads  pathOf[r&1][q] + hitKey[r&1][q] (+ the entry's ray) for the entries of counts[r%3]
//                         -> appends the entries of round r+1 to ent[(r+1)&1], counts[(r+1)%3], and -- for a round that will be
//                            ordered -- counts them into hist[(r+1)%3]
//              scatter(r+1)  (ordered rounds only) turns ranks into positions, longest predicted walk first
//              trace(r+1) walks the grid for every entry                                      -> atomicMin on hitKey[(r+1)&1][q]
#define RT_WF_MAX_ROUNDS 100000
#define RT_WF_SHARDS 256      // paths are born into one of this many shards (primary workgroup % RT_WF_SHARDS)
#define RT_WF_QSHARDS (2 * RT_WF_SHARDS) // queue slices: [0,SHARDS) main entries (one per waiting path), [SHARDS,2*SHARDS) look-ahead entries
#define RT_WF_SORT_BINS 64    // walk-length classes of an ordered trace input, 0 = longest
#define RT_WF_SORT_COPIES 32  // independent histograms (wave % copies): a hot class takes a few thousand returned atomics per round, ~90 per us and address
// host-visible status words of a tile group (RtWavefront::hostStatus)
#define RT_WF_STATUS_ERROR 0  // RT_WF_ERR_* bits, sticky until the host clears them
#define RT_WF_STATUS_WAITING 1 // paths still waiting for the grid when the issued rounds of a batch were over, summed over the batches since the
                               // host last cleared it: non-zero = the frame is INCOMPLETE (the planned number of rounds was too small)
#define RT_WF_STATUS_BATCHES 2 // batches whose issued rounds are over (a progress count for the host)
#define RT_WF_STATUS_WORDS 16
#define RT_PAIR_MANY 15u            // a first pair record's count field saturates here (rt_device.h, pairRec)
#define RT_PAIR_LIMIT (1u << 28)    // pair indices must fit the record's rest field and the trace kernel's 29-bit key field
#define RT_WF_ROUND_LOG 64     // rounds of a batch that are logged for the launch plan (RtWavefront::roundLog)
#define RT_WF_ERR_SPIN 1u     // wf_trace_kernel's walk guard tripped: rays were abandoned, the frame is invalid
#define RT_WF_ERR_GRID 2u     // a planned frame's trace grid was smaller than the round's entries: entries were not traced; the host renders
                              // the frame again with the worst-case grid (rt_api.cpp, frame_finish)
// per-round control words, three sets used in turn like the queue lengths (round r uses set r % 3): logic(r-1) fills set r,
// scatter(r) and trace(r) read it, logic(r) zeroes set (r + 2) % 3 for logic(r+1)
#define RT_WF_CTL_COUNTS 0                                   // [RT_WF_QSHARDS] queue lengths
#define RT_WF_CTL_HIST RT_WF_QSHARDS                         // [RT_WF_SORT_COPIES][RT_WF_SORT_BINS] entries per (copy, bin) of an ordered round
#define RT_WF_CTL_EXTRA (RT_WF_CTL_HIST + RT_WF_SORT_COPIES * RT_WF_SORT_BINS) // ordered round: entries in region B (further segments of cut rays)
#define RT_WF_CTL_TOTAL (RT_WF_CTL_EXTRA + 1)                // ordered round: entries in sortedIdx (written by wf_scatter_kernel)
#define RT_WF_CTL_WORDS (RT_WF_CTL_TOTAL + 3)                // (multiple of 4)

// How a round's entries are laid out and handed to the trace kernel.  Decided by the HOST before the round exists (from the launch
// plan the same frame left behind, or from a guess in a watched frame) and passed to the kernels as an argument: a wrong guess
// costs time, never correctness.
struct RtRoundMode {
    uint32_t ordered;   // 1: a round most paths spawn a ray for (the first one, or any big one) -- the logic kernel writes complete trace entries,
                        // cuts long rays into segments and counts the entries' walk-length classes, wf_scatter_kernel places them longest
                        // first; 0: a round few paths spawn a ray for -- the logic kernel writes the rays, the trace kernel's workgroups plan
                        // them, cut them into segments and trace those, in queue order
    uint32_t segLen;    // aimed-at cell visits per segment (>= 4096: rays are never cut)
    uint32_t groupRays; // round that is not ordered: rays per workgroup of the trace kernel (1..64; about 256 / expected segments per ray)
    uint32_t slices;    // queue slices in use per kind (power of two <= RT_WF_SHARDS, never more than the round before): a big round
                        // spreads its appends over 256 counters, a small one keeps its entries in a few dense stretches
};

struct RtWavefront {
    uint32_t capacity;       // paths that fit (pixels of this instance's tiles x samplesInBatch)
    uint32_t sampleBase;     // samples sampleBase+1 .. sampleBase+samplesInBatch are in flight (1-based ids, raytrace.c:612-653)
    uint32_t samplesInBatch;
    uint32_t lookAhead;      // 1: a path traces its next ring entry while the current hit's shadow ray is in flight
    uint32_t fastQuotient;   // 1: waves whose rays all have tame exponents skip the scaling / fix-up instructions of the quotients
    uint32_t spinLimit;      // walk phases a wave of wf_trace_kernel may run before it gives up and raises RT_WF_ERR_SPIN (default 16384)
    uint32_t *hostStatus;    // pinned HOST words mapped into the device (RT_WF_STATUS_*): written by kernels, read by the host after a sync
    uint4 *roundLog;         // [RT_WF_ROUND_LOG] per round: x rays, y longest queue slice, z entries in region B
    // per-path state, indexed by path id
    unsigned long long *rng; // generator state (raytrace_opencl.c:474-481), already moved past the current hit's light draws
    unsigned long long *rngL; // lightCount > 1 only: where the current hit's NEXT light set-up draws from
    uint4 *meta;             // x: output slot (localPixel*samplesInBatch + sb)  y: localPixel  w: hit triangle
                             // z: head | tail<<4 | stage<<8 | attenuation stored<<9 | look-ahead state<<10 | look-ahead ring index<<12 | light<<16
    unsigned long long *laKey; // answer (hitKey) of the path's look-ahead ray while it waits to be consumed
    float4 *outc;            // accumulated colour xyz
    float4 *ring;            // [capacity][12][3] the pixel's ray queue (:459-468): o.xyz,tmin | d.xyz,excluded | weight.xyz, bounces<<1|fromCamera;
                             // slot `head` is the ray in flight
    // what a hit still needs once its spawns are made and a shadow ray is in flight (everything else was folded at shading time):
    float4 *shP;             // P.xyz = (1-out)*weight*(1-transparency)*texture, the factors of :649-651 in the reference's order; w: N.L of this light
    float4 *shFace;          // xyz: the face[] entry that :647 will select (the other one is dead); w: 1 if front facing
    float4 *shAtt;           // shadow attenuation so far (only once a transparent occluder was met, :616-625)
    float4 *shN;             // lightCount > 1 only: shading normal for the next light's set-up (the hit point rides in the entry)
    // Trace entries, 64 bytes: {q, cell | class << 24, endCell, excluded} {dx,dy,dz,tmin} {o.xyz,tmax} {d.xyz, rank in wave | segment << 24}
    // -- the ray and, in an ordered round, its DDA start state (start / end cell, the three crossing parameters), computed by the
    // kernel that spawns the ray; a round that is not ordered carries the ray only (excluded, tmin, o, tmax, d) and has its start
    // states made by the trace kernel.  A long ray of an ordered round with few rays becomes several entries (SEGMENTS,
    // rt_wavefront.hip): segment 0 sits at the ray's queue index (region A, [0, 2*capacity)), further segments are packed into
    // region B ([2*capacity, 2*capacity + extraCap)).
    // Queue slice s of a round with `slices` slices per kind holds entries [s*sliceCap, s*sliceCap + counts[s]) with sliceCap =
    // capacity / slices for the main entries and the same range + capacity for the look-ahead entries (counts[RT_WF_SHARDS + s]);
    // a path born into shard b appends to slice b % slices, so appends hit up to 512 different counters (a single address sustains
    // only ~90 atomics/us) and a slice can never overflow: it holds at most the paths of its shards.
    uint4 *ent[2];             // [2*capacity + extraCap][4], by round parity
    uint2 *pathOf[2];          // [capacity] main queue entry q -> {path id, queue index of the look-ahead entry sent out with it or ~0}, by round parity
    unsigned long long *hitKey[2]; // [2*capacity] per ray: segment << 32 | pair index of the hit in the lowest segment that has one; all ones = no hit
    uint4 *res;                // round 0 only: the primary hit of the path -- triangle, t, l1, l2 (float bits)
    uint32_t shardCap;         // paths per shard (multiple of 256); capacity = RT_WF_SHARDS * shardCap
    uint32_t *ctl;             // [3][RT_WF_CTL_WORDS] per-round control words (RT_WF_CTL_*), zeroed in-stream by the logic kernel
    // An ORDERED round: entries keyed by the PREDICTED number of cell visits (exact for rays that hit nothing), counting-sorted
    // longest first, so that a wave holds rays of similar length and the longest walks of the round start first.
    uint32_t *sortedIdx;       // [2*capacity + extraCap] index into ent of the entry at each sorted position
    uint32_t *sortRank;        // [2*capacity + extraCap] rank of the entry inside its (bin, copy) class; 0xffffffff = an unused reservation
    uint16_t *sortTag;         // [2*capacity + extraCap] the class: bin | copy << 6 (what wf_scatter_kernel needs of an entry, 6 bytes instead of a 64-byte line)
    uint32_t extraCap;         // capacity of region B (multiple of 256)
    float4 *sampleOut;         // [capacity] finished colour per output slot
};

#endif
