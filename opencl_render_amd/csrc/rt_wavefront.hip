// rt_wavefront.hip -- the hot path as a staged wavefront pipeline for gfx950 (MI355X).
//
// Why: in a one-thread-per-pixel kernel only the pixels whose primary ray hit something (28 % in the headline scene)
// do any secondary work, and their grid walks differ in length by 10x, so a wave averages 4-5 busy lanes of 64
// (rocprofv3 PMC, profiles/r01_v1_*).  Here every ray that needs the grid becomes a queue entry.
//
//   wf_primary  one thread per (pixel, sample): jittered camera ray + per-pixel candidate list (raytrace_opencl.c:470-528).
//               Hits become PATHS: dense id from a wave-aggregated atomic, state written to HBM (rt_device.h).
//   wf_logic    one thread per waiting path: resumes the per-sample state machine of raytrace_opencl.c:532-724 where it
//               stopped, runs it until the next grid ray (shadow ray :611, or a queued ray :530) and appends that ray
//               to the next round's queue as a self-contained TRACE ENTRY (the ray + its DDA start state; a long ray of
//               a small round is cut into exact segments); paths with an empty ring retire their colour.
//   wf_scatter  big rounds only: orders the entries by predicted walk length (the logic kernel counted the classes).
//   wf_trace    one entry per lane: walks the non-uniform grid (:324-401) in blind phases, tests the occupied cells it
//               passed wave-cooperatively; a hit is published as (segment, pair) in hitKey and resolved by wf_logic.
//   wf_accum    frames with several samples: per pixel, samples in order, truncated saturating u16 accumulate into the
//               tile buffer (:726-741); a one-sample frame's pixels are written by wf_primary / wf_logic directly.
//
// Per path everything happens in the reference's order (RNG draws, ring FIFO, light loop), and paths never interact,
// so the planes are bit-identical to the single-launch kernel (rt_kernels.hip) and to the oracle.
#include "rt_devfuncs.h"

namespace {

enum { WS_RAY = 0, WS_SHADOW = 1 };

__device__ __forceinline__ float4 pack4(V3 v, float w) { return make_float4(v.x, v.y, v.z, w); }
__device__ __forceinline__ V3 xyz(float4 v) { return mk(v.x, v.y, v.z); }

// A frame with ONE sample per pixel needs no ordered accumulate (raytrace_opencl.c:726-741 adds to zeroed planes once): the
// kernel that finishes a pixel writes its three u16 values itself, wf_accum_kernel is not launched.
__device__ __forceinline__ void store_single_sample(const RtDevScene &S, uint32_t localPixel, V3 c)
{
    const uint32_t slot = localPixel / RT_TILE_PIXELS, inTile = localPixel % RT_TILE_PIXELS;
    uint16_t *planes = S.tileBuf + (size_t)slot * 3 * RT_TILE_PIXELS + inTile;
    uint32_t samples = S.sampleCount; // :728 (1 here).  Opaque, so that the quotient is made where it is used: hoisted to the top of the
    asm volatile("" : "+s"(samples)); // logic kernel it lived in a vector register across the state machine and was spilled
    const float scale = (float)(0xFFFF) / (float)samples;
    planes[0] = (uint16_t)sat_add_u16(0, c.x, scale);
    planes[RT_TILE_PIXELS] = (uint16_t)sat_add_u16(0, c.y, scale);
    planes[2 * RT_TILE_PIXELS] = (uint16_t)sat_add_u16(0, c.z, scale);
}

// Wave-aggregated queue append: one atomic per wave for all lanes that `want` a slot.  Must be reached by the lanes
// together (it ballots over the lanes that execute it).
__device__ __forceinline__ uint32_t wave_append(uint32_t *counter, bool want)
{
    const unsigned long long mask = __ballot(want);
    if (mask == 0) return 0;
    const uint32_t lane = __lane_id();
    const int leader = __ffsll((long long)mask) - 1;
    uint32_t base = 0;
    if ((int)lane == leader) base = atomicAdd(counter, (uint32_t)__popcll(mask));
    base = __shfl(base, leader, 64);
    return base + (uint32_t)__popcll(mask & ((1ull << lane) - 1ull));
}

// Two appends in one round trip: lanes 0 and 1 carry the two atomics of the wave in the same instruction.
__device__ __forceinline__ void wave_append2(uint32_t *counterA, bool wantA, uint32_t *counterB, bool wantB, uint32_t &posA, uint32_t &posB)
{
    const unsigned long long maskA = __ballot(wantA), maskB = __ballot(wantB);
    const uint32_t lane = __lane_id();
    uint32_t base = 0;
    const uint32_t n = (lane == 0u) ? (uint32_t)__popcll(maskA) : (uint32_t)__popcll(maskB);
    if (lane < 2u && n != 0u) base = atomicAdd(lane == 0u ? counterA : counterB, n);
    const unsigned long long below = (1ull << lane) - 1ull;
    posA = (uint32_t)__shfl((int)base, 0, 64) + (uint32_t)__popcll(maskA & below);
    posB = (uint32_t)__shfl((int)base, 1, 64) + (uint32_t)__popcll(maskB & below);
}

// Triangle test on a per-triangle record that is already in registers (rt_device.h: a | ab | ac | n | abab abac acac inv),
// branch-free like pair_test_flat below: same operations and operands as tri_test for every lane whose plane distance is in
// range, the others discard the second half.
__device__ __forceinline__ bool tri_rec_test_flat(const float4 r0, const float4 r1, const float4 r2, const float4 r3, V3 o, V3 d, float tmin,
                                                  float tmax, float &t, float &l1, float &l2)
{
    const V3 a = mk(r0.x, r0.y, r0.z), ab = mk(r0.w, r1.x, r1.y), ac = mk(r1.z, r1.w, r2.x), n = mk(r2.y, r2.z, r2.w);
    const V3 ao = sub3(o, a);
    t = -dot3(n, ao) / dot3(n, d);
    const V3 ap = sub3(along(o, t, d), a);
    const float ap_ab = dot3(ap, ab);
    const float ap_ac = dot3(ap, ac);
    l1 = (r3.y * ap_ac - r3.z * ap_ab) * r3.w;
    l2 = (r3.y * ap_ab - r3.x * ap_ac) * r3.w;
    return (tmin < t) & (t < tmax) & (0 <= l1) & (0 <= l2) & (l1 + l2 <= 1.f);
}

// Nearest hit among the pixel's candidate list (raytrace_opencl.c:514-528): running maximum, ties keep the earliest.
// Two candidates per step: their list entries were requested a step ahead, their records are requested together, so a
// pixel with one or two candidates (the common case) costs three dependent round trips: range, list, records.
__device__ __forceinline__ uint32_t camera_scan(const RtDevScene &S, uint32_t localPixel, V3 o, V3 d, float tmin, float tmax,
                                                uint32_t excluded, float &hit_t, float &hit_l1, float &hit_l2)
{
    uint32_t hit_tri = RT_NONE;
    hit_t = tmax;
    const uint32_t first = S.camStart[localPixel], last = S.camEnd[localPixel];
    const float4 *recs = reinterpret_cast<const float4 *>(S.triRec);
    uint32_t t0 = (first < last) ? S.camList[first] : RT_NONE;
    uint32_t t1 = (first + 1 < last) ? S.camList[first + 1] : RT_NONE;
    for (uint32_t i = first; i < last; i += 2) {
        const bool two = i + 1 < last;
        const float4 *ra = recs + 4 * (size_t)t0, *rb = recs + 4 * (size_t)(two ? t1 : t0);
        const float4 a0 = ra[0], a1 = ra[1], a2 = ra[2], a3 = ra[3];
        const float4 b0 = rb[0], b1 = rb[1], b2 = rb[2], b3 = rb[3];
        const uint32_t n0 = (i + 2 < last) ? S.camList[i + 2] : RT_NONE;
        const uint32_t n1 = (i + 3 < last) ? S.camList[i + 3] : RT_NONE;
        float t, l1, l2;
        if (tri_rec_test_flat(a0, a1, a2, a3, o, d, tmin, hit_t, t, l1, l2) & (excluded != t0)) {
            hit_t = t; hit_tri = t0; hit_l1 = l1; hit_l2 = l2;
        }
        if (tri_rec_test_flat(b0, b1, b2, b3, o, d, tmin, hit_t, t, l1, l2) & (excluded != t1) & two) {
            hit_t = t; hit_tri = t1; hit_l1 = l1; hit_l2 = l2;
        }
        t0 = n0; t1 = n1;
    }
    return hit_tri;
}

// The same scan without the pipelining, for the logic kernel's rare camera-type continuation rays (:707-722), where
// registers matter more than round trips.
__device__ __forceinline__ uint32_t camera_scan_compact(const RtDevScene &S, uint32_t localPixel, V3 o, V3 d, float tmin, float tmax,
                                                     uint32_t excluded, float &hit_t, float &hit_l1, float &hit_l2)
{
    uint32_t hit_tri = RT_NONE;
    hit_t = tmax;
    const uint32_t first = S.camStart[localPixel], last = S.camEnd[localPixel];
    for (uint32_t i = first; i < last; ++i) {
        const uint32_t tri = S.camList[i];
        if (excluded != tri) {
            float t, l1, l2;
            if (tri_test(S.triRec, tri, o, d, tmin, hit_t, t, l1, l2)) {
                hit_t = t; hit_tri = tri; hit_l1 = l1; hit_l2 = l2;
            }
        }
    }
    return hit_tri;
}

// Triangle test against a (cell, triangle) pair record (rt_device.h: {a,id}{n,-}{ab,abab}{ac,acac}), the arithmetic of
// tri_test / the reference (raytrace_opencl.c:124-172).
// The record is already in registers and the test is BRANCH-FREE: every lane computes both halves, and whether the
// plane distance was in range only enters the final predicate.  Lanes for which the reference would not have computed the
// second half (raytrace_opencl.c:143) discard it; for the others every operation and operand is the same, so t, l1, l2
// are bit-identical.  With no branch between the four 16-byte loads and their uses the compiler issues them together
// (one round trip per candidate instead of three dependent ones).
__device__ __forceinline__ bool pair_test_flat(const float4 r0, const float4 r1, const float4 r2, const float4 r3, V3 o, V3 d, float tmin,
                                               float tmax, uint32_t excluded, float &t, float &l1, float &l2)
{
    const uint32_t tri = __float_as_uint(r0.w);
    const V3 a = mk(r0.x, r0.y, r0.z), n = mk(r1.x, r1.y, r1.z);
    const V3 ao = sub3(o, a);
    t = -dot3(n, ao) / dot3(n, d);
    const V3 ab = mk(r2.x, r2.y, r2.z), ac = mk(r3.x, r3.y, r3.z);
    // The record's two spare words carry dot(ab,ac) and 1/(abac^2 - abab*acac) as rt_prepare_triangles worked them out (:147-149, the same
    // correctly rounded division); the two squared lengths are worked out here from the same operands.  (The other way round -- lengths
    // stored, the quotient taken here -- was a second full division per candidate: ~13 issue slots against 10 plain ones.)
    const float abab = dot3(ab, ab), abac = r2.w, acac = dot3(ac, ac);
    const float inv = r3.w;
    const V3 ap = sub3(along(o, t, d), a);
    const float ap_ab = dot3(ap, ab);
    const float ap_ac = dot3(ap, ac);
    l1 = (abac * ap_ac - acac * ap_ab) * inv;
    l2 = (abac * ap_ab - abab * ap_ac) * inv;
    return (tri != excluded) & (tmin < t) & (t < tmax) & (0 <= l1) & (0 <= l2) & (l1 + l2 <= 1.f);
}

// The answer of a traced ray from its hitKey: the winning (cell, triangle) pair is evaluated once more with the ray's own
// limits, which reproduces t, l1, l2 bit for bit (the running maximum of :366-379 only ever rejected other candidates).
__device__ __forceinline__ uint32_t resolve_hit(const RtDevScene &S, unsigned long long key, V3 o, V3 d, float tmin, float tmax,
                                                uint32_t excluded, float &t, float &l1, float &l2)
{
    if (key == ~0ull) return RT_NONE;
    const float4 *rec = reinterpret_cast<const float4 *>(S.pairRec) + 4 * (size_t)(uint32_t)key;
    const float4 r0 = rec[0], r1 = rec[1], r2 = rec[2], r3 = rec[3];
    pair_test_flat(r0, r1, r2, r3, o, d, tmin, tmax, excluded, t, l1, l2);
    return __float_as_uint(r0.w);
}

// Order of the hits of one ray inside a test phase: earlier recorded cell, then smaller t, then earlier candidate.
__device__ __forceinline__ unsigned long long hit_key(uint32_t cellOrder, float t, uint32_t pair)
{
    return ((unsigned long long)cellOrder << 60) | ((unsigned long long)__float_as_uint(t) << 29) | (unsigned long long)pair;
}

#if defined(RT_DIAG_STAMPS) || defined(RT_DIAG_LOGIC)
// Diagnostic build only (never shipped, outputs untouched): shader-clock stamps, summed per wave into S.stats.
__device__ __forceinline__ unsigned long long diag_stamp()
{
    unsigned long long t;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
    return t;
}
#endif

// GetSpherePoint (raytrace_opencl.c:30-45) split in two: the draws, and the scaling by the sphere's radius.  The number of
// draws does not depend on the radius, so a hit's light samples can be drawn before the radius is looked at.
struct SphereRaw { V3 p; float len, sq; };
// (forced inline: left to the inliner it became a call in the round-0 kernels once they grew, with the generator state in scratch)
__device__ __forceinline__ SphereRaw sphere_raw(uint64_t &s)
{
    SphereRaw r;
    do {
        r.p.x = rand11(s);
        r.p.y = rand11(s);
        r.p.z = rand11(s);
        r.len = sqrt_rn(dot3(r.p, r.p));
    } while (r.len <= 0.f);
    r.sq = sqrt_rn(rand01(s));
    return r;
}
__device__ __forceinline__ V3 sphere_scaled(const SphereRaw &r, float radius)
{
    const float scale = r.sq * radius / r.len; // :40
    return mk(scale * r.p.x, scale * r.p.y, scale * r.p.z);
}

} // namespace

// ---- stage 1: primary rays ---------------------------------------------------------------------------------------
// grid = (tileCount*64, samplesInBatch); workgroup = 16x16 pixel patch, wave = 8x8 quadrant.
__global__ __launch_bounds__(256) void wf_primary_kernel(const RtDevScene S, const RtWavefront W)
{
    const uint32_t slot = blockIdx.x >> 6, patch = blockIdx.x & 63;
    const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint32_t lx = (patch & 7) * RT_PATCH + (wave & 1) * 8 + (lane & 7);
    const uint32_t ly = (patch >> 3) * RT_PATCH + (wave >> 1) * 8 + (lane >> 3);
    const uint32_t tile = S.tileIds[slot];
    const uint32_t gx = (tile % S.tilesX) * RT_TILE + lx;
    const uint32_t gy = (tile / S.tilesX) * RT_TILE + ly;
    const uint32_t sb = blockIdx.y;
    const bool valid = gx < S.width && gy < S.height;

    uint32_t hit_tri = RT_NONE;
    float hit_t = 0.f, hit_l1 = 0.f, hit_l2 = 0.f;
    uint64_t rng = 0;
    V3 dir = mk(0.f, 0.f, 0.f);
    const uint32_t localPixel = slot * RT_TILE_PIXELS + ly * RT_TILE + lx;
    const uint32_t outSlot = localPixel * W.samplesInBatch + sb;
    if (valid) {
        const uint32_t pixel = gy * S.width + gx;
        rng = (uint64_t)pixel * (uint64_t)S.sampleCount + (uint64_t)(W.sampleBase + sb + 1); // :481
        const V3 lr = ld3(S.lr), tb = ld3(S.tb);
        dir = ld3(S.topLeft);
        float k = (float)gx + rand01(rng); // LR jitter first, then TB (:496-503)
        dir.x += lr.x * k; dir.y += lr.y * k; dir.z += lr.z * k;
        k = (float)gy + rand01(rng);
        dir.x += tb.x * k; dir.y += tb.y * k; dir.z += tb.z * k;
        hit_tri = camera_scan(S, localPixel, ld3(S.eye), dir, 0.f, RT_INF, RT_NONE, hit_t, hit_l1, hit_l2);
        if (hit_tri == RT_NONE) {
            if (S.sampleCount == 1u) store_single_sample(S, localPixel, mk(0.f, 0.f, 0.f));
            else W.sampleOut[outSlot] = make_float4(0.f, 0.f, 0.f, 0.f);
        }
    }
    const bool born = valid && hit_tri != RT_NONE;
    // workgroups are dealt to the shards round-robin: concurrently running groups append to different counters
    const uint32_t shard = (blockIdx.x + blockIdx.y * gridDim.x) % RT_WF_SHARDS;
    // the path id doubles as the index of its round-0 queue entry: the primary hit plays the answered request
    const uint32_t a = shard * W.shardCap + wave_append(&W.ctl[RT_WF_CTL_COUNTS + shard], born);
    if (born) {
        W.rng[a] = rng;
        W.meta[a] = make_uint4(outSlot, localPixel, 0u | (1u << 4) | ((uint32_t)WS_RAY << 8), hit_tri);
        // ring slot 0 = the camera ray (:490-508).  Only its direction is stored: round 0 of wf_logic_kernel, the slot's one
        // reader, knows the rest (origin = eye, weight 1, maxBounces 12, fromCamera), as it knows that nothing is collected yet.
        W.ring[(size_t)a * (RT_RING * 3) + 1] = pack4(dir, __uint_as_float(RT_NONE));
        W.res[a] = make_uint4(hit_tri, __float_as_uint(hit_t), __float_as_uint(hit_l1), __float_as_uint(hit_l2));
    }
}

// ---- trace entries: the DDA start state of a ray, and exact segments ------------------------------------------------
// A round lasts as long as its longest dependent chain: a ray that crosses the whole grid makes 766 cell visits one after
// the other, and measured round times are ~0.4 ms + 0.32 ms per million rays -- the constant is that chain.  The walk is a
// 3-way merge: per axis, the parameters T_a(i) = (plane_a[i] - o_a) / d_a at which the ray crosses successive planes form a
// non-decreasing sequence (the same rounded quotients the reference computes, :383-385), and every step takes the smallest
// head (:387-398).  So the state of the walk after all crossings with T <= tau is, per axis, simply the NUMBER of such
// crossings -- it can be computed without walking (a plane search plus two exact divides per axis), for any tau.  A long
// ray is therefore cut into up to RT_WF_MAXSEG SEGMENTS at parameters tau_k: segment k starts in the state at tau_k and ends
// when it has visited the start cell of segment k+1 (the existing end-cell rule, :380).  Segments are traced by different
// lanes; the ray's answer is the hit of its lowest segment that has one (atomicMin on hitKey), exactly the first cell
// with a hit in path order.  Segments after a hit are wasted work; the chain per lane is ~8x shorter.
//
// Who makes the DDA start state of a ray (no kernel of its own: a set-up kernel between logic and trace cost a launch, a second
// round trip of every ray through HBM and 30-50 us per round of a frame whose rounds are 60-600 us):
//   * an ORDERED round (many rays: bound by total work, never cut) -- wf_logic_kernel, where the ray is in registers and nearly
//     every lane spawns one; the entry carries the start state, and its walk-length class (predicted cell visits, exact for a
//     ray that hits nothing) is counted on the way (per wave in LDS, one global atomic instruction per wave); wf_scatter_kernel
//     turns ranks into positions, longest class first;
//   * any other round (few rays: bound by its longest chain, cut finely enough to occupy every SIMD) -- wf_trace_kernel<false>,
//     whose workgroups take a few rays each, plan them with every lane busy, and deal the segments to their lanes.  In a late
//     round only every tenth path spawns a ray: planning those in the logic kernel ran its whole tail at a tenth of the lanes
//     (measured: logic round 1 84 -> 167 us).
// Only the ORDER and GROUPING in which cells are visited changes.  Cutting costs work (every segment has a start-up and a test
// batch of its own, segments behind a hit are wasted), so the aimed-at cell visits per segment depend on how many rays the
// round has (RtRoundMode::segLen, chosen by the host from the launch plan).
#ifndef RT_WF_MAXSEG
#define RT_WF_MAXSEG 12
#endif
struct DdaState { uint32_t cell; float dx, dy, dz; };

// One axis of the state at parameter tau: c0 = cell coordinate of the walk's start, returns the coordinate after all
// crossings with T <= tau and, in `head`, the parameter of the next crossing.  `limit` = crossings that stay inside the grid.
// `lut` / `lutScale` (optional): this axis' 256 cell estimates and 256 / box width (RtDevScene::cellLut) -- the guess then costs one
// table read instead of the eight dependent reads of a plane search, unless the table is coarse where the point lies.
__device__ __forceinline__ uint32_t axis_state_at(const float *planes, uint32_t c0, float oa, float da, float tau, float &head,
                                                  const uint8_t *lut = nullptr, float lutScale = 0.f)
{
    const bool pos = (0.f <= da);
    const int limit = pos ? (int)(RT_GRID_DIV - 1 - c0) : (int)c0; // the crossing after these leaves the grid (tau is before it)
    // guess from the position (the plane search of GetBoxAddress), then settle it with the exact quotients
    const float p = oa + tau * da;
    int g = 0;
    bool search = true;
    if (lut) {
        const int i = min(255, max(0, (int)((p - planes[0]) * lutScale)));
        const int a = lut[max(i - 1, 0)], b = lut[min(i + 1, 255)];
        g = lut[i];
        search = b - a > 4; // (more than a few cells in three table steps: the settling loops below would walk them one by one)
    }
    if (search) {
        g = 0;
#pragma unroll
        for (int div = RT_GRID_DIV / 2; div >= 1; div /= 2)
            if (planes[g + div] < p) g += div;
    }
    int m = pos ? g - (int)c0 : (int)c0 - g;
    m = m < 0 ? 0 : (m > limit ? limit : m);
    // crossing number k (1-based) is plane c0+k going up, c0-k+1 going down
    while (m >= 1 && !((planes[pos ? (int)c0 + m : (int)c0 - m + 1] - oa) / da <= tau)) --m;
    float next = (planes[pos ? (int)c0 + m + 1 : (int)c0 - m] - oa) / da;
    while (m < limit && next <= tau) {
        ++m;
        next = (planes[pos ? (int)c0 + m + 1 : (int)c0 - m] - oa) / da;
    }
    head = next;
    return pos ? c0 + (uint32_t)m : c0 - (uint32_t)m;
}

// GetBoxAddress (:174-193) on the LDS planes: strict '<'; packed cx | cy << 8 | cz << 16
__device__ __forceinline__ uint32_t cell_of(const float *planes, V3 p)
{
    int cx = 0, cy = 0, cz = 0;
#pragma unroll
    for (int div = RT_GRID_DIV / 2; div >= 1; div /= 2) {
        if (planes[cx + div] < p.x) cx += div;
        if (planes[(RT_GRID_DIV + 1) + cy + div] < p.y) cy += div;
        if (planes[2 * (RT_GRID_DIV + 1) + cz + div] < p.z) cz += div;
    }
    return (uint32_t)cx | ((uint32_t)cy << 8) | ((uint32_t)cz << 16);
}

// What a ray's walk is made from: its DDA start state (:351-362, :383-385), where it ends (a ray with a finite range: the cell of
// its far end, :356-362), how many cells it will visit if it hits nothing, and where it leaves the grid (te).
struct EntryPlan { DdaState start; uint32_t endCell, visits; float te; };

// `haveStart`: the caller knows the start cell (a hit's shadow ray and its bounce ray start at the same point).  `lut` (optional; LDS
// copy of RtDevScene::cellLut followed by the three scales 256 / box width): where a ray WITHOUT an end cell leaves the grid -- which
// only predicts the length of its walk -- is estimated with one table read per axis instead of a plane search.
__device__ __forceinline__ EntryPlan plan_ray(const float *planes, V3 o, V3 d, float tmin, float tmax, bool haveStart = false, uint32_t startCell = 0u,
                                              const uint8_t *lut = nullptr, const float *lutScale = nullptr)
{
    EntryPlan p;
    const V3 lo = mk(planes[0], planes[RT_GRID_DIV + 1], planes[2 * (RT_GRID_DIV + 1)]);
    const V3 hi = mk(planes[RT_GRID_DIV], planes[2 * RT_GRID_DIV + 1], planes[3 * RT_GRID_DIV + 2]);
    // start / end cells (:351-362)
    V3 from = along(o, tmin, d);
    bind_in_cube(from, d, lo, hi);
    p.start.cell = startCell;
    if (!haveStart) p.start.cell = cell_of(planes, from);
    p.te = RT_INF;
    V3 to;
    if (tmax < RT_INF) {
        to = along(o, tmax, d);
        bind_in_cube(to, d, lo, hi);
    } else {
        // where the ray leaves the grid: the smallest of the three boundary crossings (same quotients as the walk's)
        if (d.x != 0.f) { const float t = (((0.f <= d.x) ? hi.x : lo.x) - o.x) / d.x; if (t < p.te) p.te = t; }
        if (d.y != 0.f) { const float t = (((0.f <= d.y) ? hi.y : lo.y) - o.y) / d.y; if (t < p.te) p.te = t; }
        if (d.z != 0.f) { const float t = (((0.f <= d.z) ? hi.z : lo.z) - o.z) / d.z; if (t < p.te) p.te = t; }
        to = (p.te < RT_INF) ? along(o, p.te, d) : from;
    }
    uint32_t last; // (where a ray without an end cell leaves the grid is a scheduling matter only)
    if (lut && !(tmax < RT_INF)) {
        const float fx = (to.x - lo.x) * lutScale[0], fy = (to.y - lo.y) * lutScale[1], fz = (to.z - lo.z) * lutScale[2];
        const int ix = min(255, max(0, (int)fx)), iy = min(255, max(0, (int)fy)), iz = min(255, max(0, (int)fz)); // (a NaN converts to 0)
        last = (uint32_t)lut[ix] | ((uint32_t)lut[256 + iy] << 8) | ((uint32_t)lut[512 + iz] << 16);
    } else last = cell_of(planes, to);
    p.endCell = (tmax < RT_INF) ? last : 0xffffffffu;
    const int cx = (int)(p.start.cell & 255u), cy = (int)((p.start.cell >> 8) & 255u), cz = (int)(p.start.cell >> 16);
    // distances from the ray ORIGIN to the next plane of each axis (:383-385)
    p.start.dx = (planes[cx + ((0 <= d.x) ? 1 : 0)] - o.x) / d.x;
    p.start.dy = (planes[(RT_GRID_DIV + 1) + cy + ((0 <= d.y) ? 1 : 0)] - o.y) / d.y;
    p.start.dz = (planes[2 * (RT_GRID_DIV + 1) + cz + ((0 <= d.z) ? 1 : 0)] - o.z) / d.z;
    // every step moves one axis by one cell in a fixed direction: visits = Manhattan distance + 1
    p.visits = (uint32_t)(abs((int)(last & 255u) - cx) + abs((int)((last >> 8) & 255u) - cy) + abs((int)(last >> 16) - cz)) + 1u;
    return p;
}

// Into how many segments a planned ray is cut when a segment should make about segLen cell visits.  Only rays without an end
// cell are cut, and only where every quotient involved is an ordinary number (a zero direction component makes heads infinite
// or NaN and the merge argument is not worth stretching to them); a ray only slightly over the aim is left whole.
__device__ __forceinline__ uint32_t segments_of(const EntryPlan &p, V3 d, float tmax, uint32_t segLen)
{
    const float ta = fminf(p.start.dx, fminf(p.start.dy, p.start.dz));
    const bool plain = !(tmax < RT_INF) && d.x != 0.f && d.y != 0.f && d.z != 0.f && p.te < RT_INF && -RT_INF < ta && ta < p.te &&
                       p.start.dx == p.start.dx && p.start.dy == p.start.dy && p.start.dz == p.start.dz && p.start.dx < RT_INF &&
                       p.start.dy < RT_INF && p.start.dz < RT_INF;
    if (!plain || p.visits <= segLen + segLen / 4) return 1u;
    const uint32_t n = (p.visits + segLen - 1) / segLen;
    return n > RT_WF_MAXSEG ? (uint32_t)RT_WF_MAXSEG : n;
}

// Cut k (1 <= k < n) of a ray: tau_k = ta + (te - ta) * k / n is non-decreasing in k, and a cut only exists where ta <= tau_k < te,
// so the cuts that exist are a prefix of 1 .. n - 1: whether cut k and cut k + 1 exist can be told from k alone.
__device__ __forceinline__ bool cut_at(float ta, float te, uint32_t k, uint32_t n, float &tau)
{
    tau = ta + (te - ta) * ((float)k / (float)n);
    return ta <= tau && tau < te;
}

// ---- stage 2: per-path state machine ---------------------------------------------------------------------------------
// Two rays of a path may be in flight at once: the current hit's shadow ray (or the ring ray being traced) and a
// LOOK-AHEAD trace of the next ring entry.  That is legal because nothing about a spawned ray depends on the shadow
// rays of the hit that spawned it -- only the ORDER in which colour is accumulated does, and that order is kept: the
// machine below is still strictly sequential per path, it merely finds some answers already there.  To know the spawned
// rays before the light loop has run, a hit's spawns are computed when it is shaded: the generator is moved past the light
// loop's draws first (their count depends on the generator alone, :30-45), the diffuse direction is drawn (:671), and the
// lights are set up later from the draws made on the way (light 0) or from a second cursor into the stream (rngL, further
// lights).  The draw order of raytrace_opencl.c is unchanged, so are the results.
//
// What a hit keeps while its shadow rays are out is folded to two vectors: `out` already holds the luminance term (:642-644,
// nothing touches `out` between a hit's shading and its :647-651), P is the product (1-out)*weight*(1-transparency)*texture
// of :649-651 in the reference's order, and of face[2] only the entry :647 will pick is tracked.
#ifndef RT_WF_LOGIC_WAVES
#define RT_WF_LOGIC_WAVES 3
#endif
#ifndef RT_WF_LOGIC_WAVES_FIRST
#define RT_WF_LOGIC_WAVES_FIRST 3
#endif
#ifndef RT_WF_LOGIC_WAVES_FIRST_ORDERED
#define RT_WF_LOGIC_WAVES_FIRST_ORDERED 3
#endif
#define RT_WF_LIGHTS_LDS 64
#ifdef RT_DIAG_LOGIC // diagnostic build: where a wave is at which time, round RT_DIAG_LOGIC (no waits added; scripts/diag_logic.py)
#define DG(i) dg[i] = diag_stamp()
#else
#define DG(i)
#endif
// What shading a hit reads of its triangle, fetched in one batch: the 24-float shading row (rt_device.h, triShade) and the
// first vertex (triRec[0]).
struct TriRow { float v[24]; float4 a; uint32_t tri; };
__device__ __forceinline__ void load_tri_row(const RtDevScene &S, uint32_t tri, TriRow &row)
{
    const float4 *sp = reinterpret_cast<const float4 *>(S.triShade + 24 * (size_t)tri);
    const float4 s0 = sp[0], s1 = sp[1], s2 = sp[2], s3 = sp[3], s4 = sp[4], s5 = sp[5];
    row.a = reinterpret_cast<const float4 *>(S.triRec)[4 * (size_t)tri];
    row.v[0] = s0.x; row.v[1] = s0.y; row.v[2] = s0.z; row.v[3] = s0.w; row.v[4] = s1.x; row.v[5] = s1.y; row.v[6] = s1.z; row.v[7] = s1.w;
    row.v[8] = s2.x; row.v[9] = s2.y; row.v[10] = s2.z; row.v[11] = s2.w; row.v[12] = s3.x; row.v[13] = s3.y; row.v[14] = s3.z; row.v[15] = s3.w;
    row.v[16] = s4.x; row.v[17] = s4.y; row.v[18] = s4.z; row.v[19] = s4.w; row.v[20] = s5.x; row.v[21] = s5.y; row.v[22] = s5.z; row.v[23] = s5.w;
    row.tri = tri;
}

// FIRST = round 0: every entry is a primary hit (stage, ring positions and the colour so far are known), a path's id is its queue index.
// ORDERED = the round this launch spawns is an ordered one (next.ordered): its entries are planned and classed here; the other
// instantiation carries none of that code (its registers are the state machine's).
// slicesIn = queue slices per kind of THIS round (what logic(round - 1) was told), next = how the round this launch spawns is laid out.
template <bool FIRST, bool ORDERED>
__global__ __launch_bounds__(256, FIRST ? (ORDERED ? RT_WF_LOGIC_WAVES_FIRST_ORDERED : RT_WF_LOGIC_WAVES_FIRST) : RT_WF_LOGIC_WAVES) void wf_logic_kernel(const RtDevScene S, const RtWavefront W, const uint32_t round,
                                                                                                             const uint32_t slicesIn, const RtRoundMode next)
{
    __shared__ Shared sh; // the texel/255 table and the split planes (for the entries of the rays spawned here)
    // the first RT_WF_LIGHTS_LDS lights, one LDS read away instead of a chain of small global loads per light and state
    __shared__ float4 ltPosRadius[RT_WF_LIGHTS_LDS], ltDirSpread[RT_WF_LIGHTS_LDS], ltColHalf[RT_WF_LIGHTS_LDS];
    __shared__ int ltType[RT_WF_LIGHTS_LDS];
    // an ordered round's classes, per wave: entries of this wave per walk-length class, then where the class's ranks of this wave start
    __shared__ uint32_t waveHist[ORDERED ? 4 : 1][RT_WF_SORT_BINS], waveBase[ORDERED ? 4 : 1][RT_WF_SORT_BINS];
    __shared__ uint8_t cellLut[ORDERED ? 3 * 256 : 4];
    // per wave: whose ray the i-th further segment belongs to, and its rank in the wave | class << 24
    __shared__ uint8_t itemOwner[ORDERED ? 4 : 1][2 * 64 * (RT_WF_MAXSEG - 1)];
    __shared__ uint32_t itemWord[ORDERED ? 4 : 1][ORDERED ? 2 * 64 * (RT_WF_MAXSEG - 1) : 1];
    __shared__ float lutScale[3];
    static_assert(RT_WF_SORT_BINS == 64, "one class per lane");
    sh.unit255[threadIdx.x] = (float)threadIdx.x / 255.f;
    if (ORDERED) {
        float *pl = &sh.planes[0][0];
        for (int i = threadIdx.x; i < 3 * (RT_GRID_DIV + 1); i += 256) pl[i] = S.boxMin[i];
        waveHist[threadIdx.x >> 6][threadIdx.x & 63] = 0u;
        if (threadIdx.x < 3 * 256 / 4) reinterpret_cast<uint32_t *>(cellLut)[threadIdx.x] = reinterpret_cast<const uint32_t *>(S.cellLut)[threadIdx.x];
        if (threadIdx.x < 3) lutScale[threadIdx.x] = 256.f / (S.boxMin[threadIdx.x * (RT_GRID_DIV + 1) + RT_GRID_DIV] - S.boxMin[threadIdx.x * (RT_GRID_DIV + 1)]);
    }
    if (threadIdx.x < RT_WF_LIGHTS_LDS && threadIdx.x < S.lightCount) {
        const uint32_t k = threadIdx.x;
        ltType[k] = S.lightType[k];
        ltPosRadius[k] = make_float4(S.lightPos[4 * k], S.lightPos[4 * k + 1], S.lightPos[4 * k + 2], S.lightRadius[k]);
        ltDirSpread[k] = make_float4(S.lightDir[4 * k], S.lightDir[4 * k + 1], S.lightDir[4 * k + 2], S.lightSpread[k]);
        ltColHalf[k] = make_float4(S.lightCol[4 * k], S.lightCol[4 * k + 1], S.lightCol[4 * k + 2], S.lightHalfAtt[k]);
    }
    __syncthreads();
    auto light_type = [&](uint32_t k) -> int { return k < RT_WF_LIGHTS_LDS ? ltType[k] : S.lightType[k]; };
    auto light_pos_radius = [&](uint32_t k) -> float4 {
        return k < RT_WF_LIGHTS_LDS ? ltPosRadius[k] : make_float4(S.lightPos[4 * k], S.lightPos[4 * k + 1], S.lightPos[4 * k + 2], S.lightRadius[k]);
    };
    auto light_dir_spread = [&](uint32_t k) -> float4 {
        return k < RT_WF_LIGHTS_LDS ? ltDirSpread[k] : make_float4(S.lightDir[4 * k], S.lightDir[4 * k + 1], S.lightDir[4 * k + 2], S.lightSpread[k]);
    };
    auto light_col_half = [&](uint32_t k) -> float4 {
        return k < RT_WF_LIGHTS_LDS ? ltColHalf[k] : make_float4(S.lightCol[4 * k], S.lightCol[4 * k + 1], S.lightCol[4 * k + 2], S.lightHalfAtt[k]);
    };

    const uint32_t in = round & 1, outq = in ^ 1;
    const uint32_t *ctlIn = W.ctl + (round % 3) * RT_WF_CTL_WORDS;
    uint32_t *ctlOut = W.ctl + ((round + 1) % 3) * RT_WF_CTL_WORDS;
    const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint32_t waveId = (blockIdx.x * 256 + threadIdx.x) >> 6, waves = (gridDim.x * 256) >> 6;
    Counters cn; // unused (COUNT=false instantiations below)
    // in-stream housekeeping: the control words two rounds ahead (queue lengths, class histogram, region B's fill)
    {
        const uint32_t gid = blockIdx.x * 256 + threadIdx.x;
        if (gid < RT_WF_CTL_WORDS) W.ctl[((round + 2) % 3) * RT_WF_CTL_WORDS + gid] = 0u;
    }
    // for the launch plan: this round's rays and its longest queue slice (one workgroup adds up the RT_WF_QSHARDS queue lengths)
    if (blockIdx.x == gridDim.x - 1 && round < RT_WF_ROUND_LOG) {
        __shared__ uint32_t sumWave[4], maxWave[4];
        static_assert(RT_WF_QSHARDS == 512, "two queue lengths per thread");
        const uint32_t c0 = ctlIn[RT_WF_CTL_COUNTS + threadIdx.x], c1 = ctlIn[RT_WF_CTL_COUNTS + 256 + threadIdx.x];
        uint32_t n = c0 + c1, m = max(c0, c1);
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) { n += __shfl_xor(n, off, 64); m = max(m, (uint32_t)__shfl_xor((int)m, off, 64)); }
        if (lane == 0) { sumWave[wave] = n; maxWave[wave] = m; }
        __syncthreads();
        if (threadIdx.x == 0) {
            uint32_t *log = reinterpret_cast<uint32_t *>(W.roundLog + round);
            log[0] = sumWave[0] + sumWave[1] + sumWave[2] + sumWave[3];
            log[1] = max(max(maxWave[0], maxWave[1]), max(maxWave[2], maxWave[3]));
        }
    }
    const bool multiLight = S.lightCount > 1u;
    const float *planes = &sh.planes[0][0];

    // main entries only (slices [0, slicesIn)): one per waiting path; look-ahead answers are picked up by index.
    // The grid is a whole number of waves per slice (rtw_launch_logic), so a wave stays in ONE slice and needs its length only:
    // slice = wave id % slicesIn, chunks of 64 entries dealt to the slice's waves in turn.
    // (wave-uniform values are told to be so: they live in scalar registers, not in one of the few vector registers left)
    const uint32_t shard = __builtin_amdgcn_readfirstlane(waveId % slicesIn);
    const uint32_t total = __builtin_amdgcn_readfirstlane(ctlIn[RT_WF_CTL_COUNTS + shard]);
    const uint32_t sliceCapIn = W.capacity / slicesIn, sliceCapOut = W.capacity / next.slices;
    const uint32_t outShard = shard % next.slices; // (slices never grow from one round to the next: a slice holds at most the paths of its shards)
    for (uint32_t localChunk = __builtin_amdgcn_readfirstlane(waveId / slicesIn); localChunk * 64 < total; localChunk += waves / slicesIn) {
        const uint32_t local = localChunk * 64 + lane;
        const uint32_t q = shard * sliceCapIn + local;
        const bool live = local < total;
#ifdef RT_DIAG_LOGIC
        unsigned long long dg[12];
        dg[0] = diag_stamp();
        for (int i = 1; i < 12; ++i) dg[i] = 0;
#endif
        bool emit = false, emitLa = false;
        V3 ro = mk(0, 0, 0), rd = mk(0, 0, 0), lo3 = mk(0, 0, 0), ld3v = mk(0, 0, 0);
        float rtmin = 0.f, rtmax = 0.f, latmin = 0.f;
        uint32_t rexcl = RT_NONE, laexcl = RT_NONE, a = 0;

        if (live) {
            // Round 0 issues its loads in as few dependent batches as the data allows -- a wave runs one chunk and has two
            // neighbours on its SIMD, so the chunk lasts as long as its chain of memory round trips (measured: ~8 of them at
            // 2-4 us each before the first shading instruction when every load sat next to its use).  Batch one: the primary hit,
            // the generator and the camera ray; batch two: the hit triangle's shading row.  Later rounds, whose paths are in
            // different stages, keep their loads next to the uses (batching them too measured 100 -> 120 us).  Both instantiations
            // fit 3 waves per SIMD without a spilled register (166 / 164 VGPRs) only because nothing that is wanted at the far end
            // of the state machine is carried across it: addresses, the sample scale, the output slot are made or read again
            // where they are used (a spill reload in this kernel is a wait for everything in flight).
            uint32_t res_tri = RT_NONE;
            float res_t = 0.f, res_l1 = 0.f, res_l2 = 0.f;
            unsigned long long key = ~0ull, laKeyEarly = ~0ull;
            float4 qo = make_float4(0.f, 0.f, 0.f, 0.f), qd = qo;
            if (FIRST) {
                a = q; // (a path is born at its own round-0 queue index)
                const uint4 r = W.res[q];
                res_tri = r.x; res_t = __uint_as_float(r.y); res_l1 = __uint_as_float(r.z); res_l2 = __uint_as_float(r.w);
            } else {
                // {path, queue slot of the look-ahead ray it sent out with this one}: the second answer is asked for right here, one
                // round trip earlier than through the path's own state (it is only USED if the path's flags say it is outstanding)
                const uint2 who = W.pathOf[in][q];
                a = who.x;
                key = W.hitKey[in][q];
                if (who.y != 0xffffffffu) laKeyEarly = W.hitKey[in][who.y];
            }
            float4 *ringA = W.ring + (size_t)a * (RT_RING * 3);
            uint64_t rng = W.rng[a];
            // (round 0 knows the path's flags and its hit triangle; the output slot and the pixel are wanted at the very end only and are
            // read then, instead of occupying two registers across the state machine)
            const uint4 meta = FIRST ? make_uint4(0u, 0u, 0u | (1u << 4) | ((uint32_t)WS_RAY << 8), res_tri) : W.meta[a];
            V3 out = mk(0.f, 0.f, 0.f); // (round 0: nothing collected yet)
            float4 c0 = qo, c1 = qo, c2 = qo;
            unsigned long long laKey = ~0ull;
            TriRow row;
            row.tri = RT_NONE;
            if (FIRST) {
                // slot 0 = the camera ray, of which wf_primary_kernel stores the direction only (:490-508: origin = eye, weight 1, maxBounces 12, fromCamera)
                c0 = pack4(ld3(S.eye), 0.f); c1 = ringA[1]; c2 = make_float4(1.f, 1.f, 1.f, __uint_as_float((12u << 1) | 1u));
                load_tri_row(S, res_tri, row);
            } else out = xyz(W.outc[a]);
            uint32_t hit_tri = meta.w;
            int head = (int)(meta.z & 15u), tail = (int)((meta.z >> 4) & 15u);
            const uint32_t stage = (meta.z >> 8) & 1u;
            bool attStored = ((meta.z >> 9) & 1u) != 0u;
            uint32_t laState = (meta.z >> 10) & 3u; // 0 none, 1 requested last round (answer at hitKey[pathOf.y]), 2 answer kept in laKey
            int laIndex = (int)((meta.z >> 12) & 15u);
            uint32_t j = meta.z >> 16;
            DG(1);
            bool laFetched = false;
            if (laState == 1u) { laKey = laKeyEarly; laState = 2u; laFetched = true; }
            else if (laState == 2u) laKey = W.laKey[a];

            // the ray in flight (ring slot `head`), loaded when its answer is here
            V3 cur_o = mk(0, 0, 0), cur_d = mk(0, 0, 0), cur_w = mk(0, 0, 0);
            float cur_tmin = 0.f;
            uint32_t cur_excl = RT_NONE;
            int cur_bounces = 0, cur_fromCamera = 0;
            // the hit being lit
            V3 n = mk(0, 0, 0), where = mk(0, 0, 0), P = mk(0, 0, 0), face = mk(0, 0, 0), atten = mk(1.f, 1.f, 1.f), toL = mk(0, 0, 0);
            float ndl = 0.f, lmin = 0.f, lmax = 0.f;
            bool front = false;
            uint64_t rngL = 0;
            SphereRaw raw0;
            raw0.p = mk(0, 0, 0); raw0.len = 1.f; raw0.sq = 0.f;

            enum { PC_RAY_RESULT, PC_SHADOW_RESULT, PC_LIGHT_SETUP, PC_LIGHT_ACCUM, PC_SHADE_END, PC_NEXT_RAY, PC_EXIT };
            int pc = PC_RAY_RESULT;
            if (!FIRST && stage == WS_SHADOW) {
                const float4 sp = W.shP[a], sf = W.shFace[a];
                P = xyz(sp); ndl = sp.w; face = xyz(sf); front = (sf.w != 0.f);
                if (attStored) atten = xyz(W.shAtt[a]);
                // the entry that was answered still holds the hit point and the direction to the light: {..} {.., tmin} {o, tmax} {d, ..}
                {
                    const float4 *e = reinterpret_cast<const float4 *>(W.ent[in]) + 4 * (size_t)q;
                    const float4 e1 = e[1];
                    qo = e[2]; qd = e[3];
                    where = xyz(qo); lmin = e1.w; toL = xyz(qd); lmax = qo.w;
                }
                if (multiLight) { n = xyz(W.shN[a]); rngL = W.rngL[a]; }
                res_tri = resolve_hit(S, key, where, toL, lmin, lmax, hit_tri, res_t, res_l1, res_l2);
                pc = PC_SHADOW_RESULT;
            } else {
                if (!FIRST) { c0 = ringA[head * 3 + 0]; c1 = ringA[head * 3 + 1]; c2 = ringA[head * 3 + 2]; }
                cur_o = xyz(c0); cur_tmin = c0.w; cur_d = xyz(c1); cur_excl = __float_as_uint(c1.w); cur_w = xyz(c2);
                cur_bounces = (int)(__float_as_uint(c2.w) >> 1);
                cur_fromCamera = (int)(__float_as_uint(c2.w) & 1u);
                if (!FIRST) res_tri = resolve_hit(S, key, cur_o, cur_d, cur_tmin, RT_INF, cur_excl, res_t, res_l1, res_l2);
            }
            DG(2);
            bool finished = false, shadedNow = false, rngDirty = false, outDirty = false;
            uint32_t emitStage = WS_RAY;

            // A spawned ray goes to the ring in HBM; the first one of this invocation also stays in registers, because it is the
            // ray the look-ahead below will ask for when the ring was empty (reading it back cost a round trip behind the stores).
            // Round 0 only: later rounds have no registers to spare.
            int firstSpawnSlot = -1;
            float4 firstSpawn0 = make_float4(0.f, 0.f, 0.f, 0.f), firstSpawn1 = firstSpawn0;
            uint32_t firstSpawnFlags = 0u;
            auto spawn = [&](float4 e0, float4 e1, float4 e2) {
                ringA[tail * 3 + 0] = e0; ringA[tail * 3 + 1] = e1; ringA[tail * 3 + 2] = e2;
                if (FIRST && firstSpawnSlot < 0) { firstSpawnSlot = tail; firstSpawn0 = e0; firstSpawn1 = e1; firstSpawnFlags = __float_as_uint(e2.w); }
                tail = (tail + 1) % RT_RING;
            };
            // SHADE_BEGIN (:532-561) and the hit's spawns, on the triangle's shading row
            auto shade_begin = [&](const float *shade, const float4 *firstVertex) {
                hit_tri = res_tri;
                const float hit_t = res_t, hit_l1 = res_l1, hit_l2 = res_l2;
                const int m = __float_as_int(shade[21]);
                if (m != 12345678) DG(3);
                const float *uv = shade + 15;
                where = along(cur_o, hit_t, cur_d);
                // the material's whole record in one burst (rt_device.h, matRec): the channel look-ups below are then no chain
                // of dependent loads, and one-texel channels need no further load at all
                MatRec mat;
                mat.desc[0] = mat.desc[1] = mat.desc[2] = mat.desc[3] = mat.desc[4] = 0u; mat.m = m;
                if (0 <= m) mat = load_mat(S, m);
                n = shading_normal<false>(S, sh, where, cur_o, cur_d, hit_tri, hit_l1, hit_l2, shade, m, cn, &mat, firstVertex);
                if (n.x != 12345.f) DG(4);
                V3 tex = mk(0, 0, 0), transp = mk(0, 0, 0), refl = mk(0, 0, 0), lum = mk(0, 0, 0);
                if (0 <= m) {
                    uint32_t raw;
                    if (mat.desc[CH_COLOR]) tex = texel_rec<false>(S, sh, mat, CH_COLOR, uv, hit_l1, hit_l2, raw, cn);
                    if (mat.desc[CH_TRANSPARENCY]) transp = texel_rec<false>(S, sh, mat, CH_TRANSPARENCY, uv, hit_l1, hit_l2, raw, cn);
                    if (mat.desc[CH_REFLECTION]) refl = texel_rec<false>(S, sh, mat, CH_REFLECTION, uv, hit_l1, hit_l2, raw, cn);
                    if (mat.desc[CH_LUMINANCE]) lum = texel_rec<false>(S, sh, mat, CH_LUMINANCE, uv, hit_l1, hit_l2, raw, cn);
                }
                // :642-644 now (the light loop does not touch `out`), then the light-independent factors of :649-651
                out.x += (1.f - out.x) * lum.x * cur_w.x;
                out.y += (1.f - out.y) * lum.y * cur_w.y;
                out.z += (1.f - out.z) * lum.z * cur_w.z;
                outDirty = true;
                front = (dot3(n, cur_d) <= 0.f);
                P.x = (1.f - out.x) * cur_w.x * (1.f - transp.x) * tex.x;
                P.y = (1.f - out.y) * cur_w.y * (1.f - transp.y) * tex.y;
                P.z = (1.f - out.z) * cur_w.z * (1.f - transp.z) * tex.z;
                face = mk(0.1f, 0.1f, 0.1f); // :540
                // The light loop (:563-637) draws one GetSpherePoint per light of a sampled type (:573,:595).  Make those
                // draws now: keep light 0's, remember where light 1's start, and leave the generator behind them all.
                rngL = rng;
                for (uint32_t k = 0; k < S.lightCount; ++k) {
                    const int type = light_type(k);
                    if (type >= 1 && type <= 9) {
                        const SphereRaw rr = sphere_raw(rng);
                        if (k == 0) raw0 = rr;
                    }
                    if (k == 0) rngL = rng;
                }
                rngDirty = true;
                // the hit's spawns (:656-722), ahead of its light loop: nothing below depends on the face lights
                if (cur_bounces > 0) {
                    const int frontI = front ? 1 : 0;
                    const float total_rt = RT_MAX2(RT_MAX2(refl.x + transp.x, refl.y + transp.y), refl.z + transp.z);
                    const float dif = (total_rt < 1.f) ? 1.f - total_rt : 0.f;
                    bool open = true;
                    V3 w = mk(cur_w.x * tex.x * dif, cur_w.y * tex.y * dif, cur_w.z * tex.z * dif);
                    if (3.f / 256.f <= w.x + w.y + w.z) { // diffuse bounce (:664-683)
                        V3 nd = sphere_scaled(sphere_raw(rng), 1.f);
                        if (frontI != ((0 <= dot3(nd, n)) ? 1 : 0)) { nd.x = -nd.x; nd.y = -nd.y; nd.z = -nd.z; }
                        spawn(pack4(where, 0.f), pack4(nd, __uint_as_float(hit_tri)), pack4(w, __uint_as_float(0u)));
                        if ((tail + 1) % RT_RING == head) open = false;
                    }
                    if (open) { // mirror (:686-705)
                        w = mk(cur_w.x * tex.x * refl.x, cur_w.y * tex.y * refl.y, cur_w.z * tex.z * refl.z);
                        if (3.f / 256.f <= w.x + w.y + w.z) {
                            const float two = -2.f * dot3(n, cur_d);
                            const V3 md = mk(cur_d.x + two * n.x, cur_d.y + two * n.y, cur_d.z + two * n.z);
                            spawn(pack4(where, 0.f), pack4(md, __uint_as_float(hit_tri)), pack4(w, __uint_as_float((uint32_t)(cur_bounces - 1) << 1)));
                            if ((tail + 1) % RT_RING == head) open = false;
                        }
                    }
                    if (open) { // see-through continuation (:707-722)
                        w = mk(cur_w.x * tex.x * transp.x, cur_w.y * tex.y * transp.y, cur_w.z * tex.z * transp.z);
                        if (3.f / 256.f <= w.x + w.y + w.z) {
                            spawn(pack4(cur_o, hit_t), pack4(cur_d, __uint_as_float(hit_tri)),
                                  pack4(w, __uint_as_float(((uint32_t)(cur_bounces - 1) << 1) | (uint32_t)cur_fromCamera)));
                        }
                    }
                }
                DG(5);
                j = 0;
                shadedNow = true;
                pc = PC_LIGHT_SETUP;
            };
            if (FIRST) { // the camera ray's hit: its row was requested with the path's state
                if (res_tri != RT_NONE) shade_begin(row.v, &row.a); else pc = PC_NEXT_RAY;
            }
            while (pc != PC_EXIT) {
                if (pc == PC_RAY_RESULT) {
                    if (res_tri == RT_NONE) { pc = PC_NEXT_RAY; continue; }
                    shade_begin(S.triShade + 24 * (size_t)res_tri, nullptr); // (read where it is used: registers, see above)
                } else if (pc == PC_LIGHT_SETUP) { // :563-607
                    if (j >= S.lightCount) { pc = PC_SHADE_END; continue; }
                    toL = mk(0.f, 0.f, 0.f); atten = mk(1.f, 1.f, 1.f); attStored = false;
                    lmin = 0.f; lmax = 0.f;
                    const int type = light_type(j);
                    if (type >= 1 && type <= 9) {
                        // light 0 is only ever set up in the invocation that shaded the hit, where its draws are at hand
                        const SphereRaw rr = (j == 0u) ? raw0 : sphere_raw(rngL);
                        if (type >= 3 && type <= 6) {
                            const float4 ld = light_dir_spread(j);
                            toL = sphere_scaled(rr, ld.w);
                            toL.x -= ld.x; toL.y -= ld.y; toL.z -= ld.z;
                            const float inv = 1.f / sqrt_rn(dot3(toL, toL));
                            toL.x *= inv; toL.y *= inv; toL.z *= inv;
                            lmax = RT_INF;
                        } else {
                            const float4 lp = light_pos_radius(j);
                            const V3 rp = sphere_scaled(rr, lp.w);
                            toL.x = rp.x + lp.x - where.x;
                            toL.y = rp.y + lp.y - where.y;
                            toL.z = rp.z + lp.z - where.z;
                            lmax = sqrt_rn(dot3(toL, toL));
                            const float inv = 1.f / lmax;
                            toL.x *= inv; toL.y *= inv; toL.z *= inv;
                        }
                    }
                    ndl = dot3(n, toL);
                    if (lmin < lmax) { // shadow ray (:608-611): leave the machine until the grid has answered
                        emit = true; emitStage = WS_SHADOW; ro = where; rd = toL; rtmin = lmin; rtmax = lmax; rexcl = hit_tri;
                        pc = PC_EXIT;
                    } else pc = PC_LIGHT_ACCUM;
                } else if (pc == PC_SHADOW_RESULT) { // :612-626
                    pc = PC_LIGHT_ACCUM;
                    if (res_tri != RT_NONE) {
                        const float *oshade = S.triShade + 24 * (size_t)res_tri;
                        const int om = __float_as_int(oshade[21]);
                        V3 tr = mk(0.f, 0.f, 0.f);
                        if (0 <= om) {
                            const MatRec omat = load_mat(S, om);
                            uint32_t raw;
                            if (omat.desc[CH_TRANSPARENCY]) tr = texel_rec<false>(S, sh, omat, CH_TRANSPARENCY, oshade + 15, res_l1, res_l2, raw, cn);
                        }
                        atten.x *= tr.x; atten.y *= tr.y; atten.z *= tr.z;
                        attStored = true;
                        if (0.f < atten.x && 0.f < atten.y && 0.f < atten.z) {
                            lmin = res_t;
                            emit = true; emitStage = WS_SHADOW; ro = where; rd = toL; rtmin = lmin; rtmax = lmax; rexcl = hit_tri;
                            pc = PC_EXIT;
                        }
                    }
                } else if (pc == PC_LIGHT_ACCUM) { // :628-636
                    const float mag = __builtin_fabsf(ndl);
                    const float4 lc = light_col_half(j);
                    const float x = lmax / lc.w;
                    const float e = mag * half_falloff(x);
                    if ((0.f <= ndl) == front) { // face[1] collects the lights in front of the normal, face[0] the others (:632-635)
                        face.x += (1.f - face.x) * atten.x * e * lc.x;
                        face.y += (1.f - face.y) * atten.y * e * lc.y;
                        face.z += (1.f - face.z) * atten.z * e * lc.z;
                    }
                    ++j;
                    pc = PC_LIGHT_SETUP;
                } else if (pc == PC_SHADE_END) { // :647-651
                    out.x += P.x * face.x;
                    out.y += P.y * face.y;
                    out.z += P.z * face.z;
                    outDirty = true;
                    pc = PC_NEXT_RAY;
                } else { // PC_NEXT_RAY (:509)
                    head = (head + 1) % RT_RING;
                    if (head == tail) { finished = true; pc = PC_EXIT; continue; }
                    const float4 c0 = ringA[head * 3 + 0], c1 = ringA[head * 3 + 1], c2 = ringA[head * 3 + 2];
                    cur_o = xyz(c0); cur_tmin = c0.w; cur_d = xyz(c1); cur_excl = __float_as_uint(c1.w); cur_w = xyz(c2);
                    cur_bounces = (int)(__float_as_uint(c2.w) >> 1);
                    cur_fromCamera = (int)(__float_as_uint(c2.w) & 1u);
                    if (cur_fromCamera) {
                        uint32_t localPixel = FIRST ? W.meta[a].y : meta.y; // (opaque: the list addresses are worked out here, not carried from the prologue)
                        asm volatile("" : "+v"(localPixel));
                        res_tri = camera_scan_compact(S, localPixel, cur_o, cur_d, cur_tmin, RT_INF, cur_excl, res_t, res_l1, res_l2);
                        pc = PC_RAY_RESULT;
                    } else if (laState == 2u && laIndex == head) { // traced ahead of time: the answer is already here
                        res_tri = resolve_hit(S, laKey, cur_o, cur_d, cur_tmin, RT_INF, cur_excl, res_t, res_l1, res_l2);
                        laState = 0u;
                        pc = PC_RAY_RESULT;
                    } else {
                        emit = true; emitStage = WS_RAY; ro = cur_o; rd = cur_d; rtmin = cur_tmin; rtmax = RT_INF; rexcl = cur_excl;
                        pc = PC_EXIT;
                    }
                }
            }

            DG(6);
            // Leaving with a request and no look-ahead outstanding: start the next ring entry's grid walk as well.
            if (!finished && laState == 0u && W.lookAhead) {
                const int nx = (head + 1) % RT_RING;
                if (nx != tail) {
                    float4 n0 = firstSpawn0, n1 = firstSpawn1;
                    uint32_t nflags = firstSpawnFlags;
                    if (!FIRST || nx != firstSpawnSlot) { n0 = ringA[nx * 3 + 0]; n1 = ringA[nx * 3 + 1]; nflags = __float_as_uint(ringA[nx * 3 + 2].w); }
                    if ((nflags & 1u) == 0u) { // a grid ray (camera-type rays are answered inline)
                        emitLa = true; lo3 = xyz(n0); latmin = n0.w; ld3v = xyz(n1); laexcl = __float_as_uint(n1.w);
                        laState = 1u; laIndex = nx;
                    }
                }
            }

            if (laexcl != 12345u) DG(7);
            if (finished) {
                const uint2 where2 = FIRST ? *reinterpret_cast<const uint2 *>(W.meta + a) : make_uint2(meta.x, meta.y); // {output slot, pixel}
                if (S.sampleCount == 1u) store_single_sample(S, where2.y, out);
                else W.sampleOut[where2.x] = pack4(out, 0.f);
            } else {
                // park the path in HBM until the grid has answered.  The index is made opaque here so that the store addresses are
                // worked out again (two instructions each) instead of being the prologue's load addresses kept alive across the
                // whole state machine: those were spilled, and every reload put an `s_waitcnt vmcnt(0)` -- i.e. "all stores so far
                // acknowledged" -- between two stores (ten of them in a row: a third of a later round's time).
                asm volatile("" : "+v"(a));
                if (rngDirty) W.rng[a] = rng;
                if (outDirty) W.outc[a] = pack4(out, 0.f);
                const uint32_t flags = (uint32_t)head | ((uint32_t)tail << 4) | (emitStage << 8) | ((attStored ? 1u : 0u) << 9) | (laState << 10) |
                                       ((uint32_t)laIndex << 12) | (j << 16);
                if (FIRST) reinterpret_cast<uint2 *>(W.meta + a)[1] = make_uint2(flags, hit_tri); // (slot and pixel stay as wf_primary_kernel wrote them)
                else W.meta[a] = make_uint4(meta.x, meta.y, flags, hit_tri);
                if (laState == 2u && laFetched) W.laKey[a] = laKey;
                if (emitStage == WS_SHADOW) {
                    W.shP[a] = pack4(P, ndl);
                    W.shFace[a] = pack4(face, front ? 1.f : 0.f);
                    if (attStored) W.shAtt[a] = pack4(atten, 0.f);
                    if (multiLight) {
                        W.rngL[a] = rngL;
                        if (shadedNow) W.shN[a] = pack4(n, 0.f);
                    }
                }
            }
        }
        // Every lane of the wave arrives here: one atomic instruction for the wave's two queue appends.
        uint32_t slot, slotLa;
        wave_append2(&ctlOut[RT_WF_CTL_COUNTS + outShard], emit, &ctlOut[RT_WF_CTL_COUNTS + RT_WF_SHARDS + outShard], emitLa, slot, slotLa);
        slot += outShard * sliceCapOut;
        slotLa += W.capacity + outShard * sliceCapOut;
        asm volatile("" : "+v"(a), "+v"(slot), "+v"(slotLa)); // (addresses made here, not carried across the state machine)
        if (slot != 0xfffffff0u) DG(8);
        // The rays of the next round go to their queue slots now.  An ORDERED round (most lanes have one): as complete trace entries --
        // DDA start state, segments, walk-length class -- so that nothing stands between this kernel and the walk but the placing of
        // the classes.  Any other round: the ray alone; wf_trace_kernel<false> plans and cuts it (stage "trace entries").
        const uint32_t copy = waveId % RT_WF_SORT_COPIES;
        if (!ORDERED) {
#pragma unroll 1
            for (int which = 0; which < 2; ++which) {
                const bool has = which ? emitLa : emit;
                if (!has) continue;
                const V3 o = which ? lo3 : ro, d = which ? ld3v : rd;
                const float tmin = which ? latmin : rtmin, tmax = which ? RT_INF : rtmax;
                const uint32_t excluded = which ? laexcl : rexcl, mine = which ? slotLa : slot;
                uint4 *e = W.ent[outq] + 4 * (size_t)mine;
                if (!which) W.pathOf[outq][mine] = make_uint2(a, emitLa ? slotLa : 0xffffffffu); // (only main entries are ever looked up: wf_logic_kernel's prologue)
                W.hitKey[outq][mine] = ~0ull; // no segment of this ray has a hit yet
                reinterpret_cast<uint32_t *>(e)[3] = excluded;
                reinterpret_cast<uint32_t *>(e)[7] = __float_as_uint(tmin);
                e[2] = make_uint4(__float_as_uint(o.x), __float_as_uint(o.y), __float_as_uint(o.z), __float_as_uint(tmax));
                e[3] = make_uint4(__float_as_uint(d.x), __float_as_uint(d.y), __float_as_uint(d.z), 0u);
            }
        } else {
            // walk-length class of an entry that will make v cell visits if it hits nothing (scheduling only).  Two scales: segments of a
            // finely cut round differ by a few visits, uncut rays by hundreds; class 0 = longest
            auto visit_class = [](uint32_t v) -> uint32_t {
                if ((int)v < 1) v = 1;
                if (v > 767u) v = 767u;
                return (v < 128u) ? 63u - (v >> 2) : 31u - (v - 128u) / 20u;
            };
            // Everything that can be known without touching memory comes first -- both rays' plans, their segments, every entry's class and
            // its rank inside the wave -- so that the one returned atomic every wave needs (where its classes' ranks start) goes out BEFORE
            // the entries' stores: a returned atomic waits for every store issued ahead of it, and behind 128 bytes of entry per lane the
            // wave sat out its own stores' round trip to HBM.  Nothing is read back either: a further segment's class and rank wait in LDS
            // for the lane that makes its entry.
            EntryPlan planM, planL;
            planM.visits = planL.visits = 1; planM.endCell = planL.endCell = 0xffffffffu; planM.te = planL.te = RT_INF;
            planM.start.cell = planL.start.cell = 0; planM.start.dx = planM.start.dy = planM.start.dz = 0.f; planL.start.dx = planL.start.dy = planL.start.dz = 0.f;
            uint32_t nsegM = 0, nsegL = 0;
            // (a round that is cut needs the exact far cell: the cut positions follow from the visit count)
            const uint8_t *lut = next.segLen >= 4096u ? cellLut : nullptr;
#pragma unroll 1
            for (int which = 0; which < 2; ++which) { // (one copy of the planning code: two made the kernel a sixth longer)
                const bool has = which ? emitLa : emit;
                if (__ballot(has) == 0ull) continue; // wave-uniform
                const V3 o = which ? lo3 : ro, d = which ? ld3v : rd;
                const float tmin = which ? latmin : rtmin, tmax = which ? RT_INF : rtmax;
                // (a hit's look-ahead ray starts where its shadow ray starts: one plane search for the two -- when the start is the origin
                // itself, tmin = 0, and lies inside the grid's box, so that BindInCube (:265-322) moves it for neither direction)
                const bool sameStart = which && emit && lo3.x == ro.x && lo3.y == ro.y && lo3.z == ro.z && latmin == 0.f && rtmin == 0.f &&
                                       planes[0] <= o.x && o.x <= planes[RT_GRID_DIV] && planes[RT_GRID_DIV + 1] <= o.y && o.y <= planes[2 * RT_GRID_DIV + 1] &&
                                       planes[2 * (RT_GRID_DIV + 1)] <= o.z && o.z <= planes[3 * RT_GRID_DIV + 2];
                if (has) {
                    const EntryPlan plan = plan_ray(planes, o, d, tmin, tmax, sameStart, planM.start.cell, lut, lutScale);
                    const uint32_t n = segments_of(plan, d, tmax, next.segLen);
                    if (which) { planL = plan; nsegL = n; } else { planM = plan; nsegM = n; }
                }
            }
            // items = further segments of this wave's rays: those of the main rays first (lane L's start at beforeM), then those of the
            // look-ahead rays (at itemsM + beforeL)
            const uint32_t extraM = nsegM > 1u ? nsegM - 1u : 0u, extraL = nsegL > 1u ? nsegL - 1u : 0u;
            uint32_t itemsM = 0, itemsL = 0, beforeM = 0, beforeL = 0;
            const bool cutting = __ballot((extraM | extraL) != 0u) != 0ull; // wave-uniform
            if (cutting) {
                uint32_t inclM = extraM, inclL = extraL;
#pragma unroll
                for (int off = 1; off < 64; off <<= 1) {
                    const uint32_t upM = __shfl_up(inclM, off, 64), upL = __shfl_up(inclL, off, 64);
                    if ((int)lane >= off) { inclM += upM; inclL += upL; }
                }
                itemsM = (uint32_t)__shfl((int)inclM, 63, 64); itemsL = (uint32_t)__shfl((int)inclL, 63, 64);
                beforeM = inclM - extraM; beforeL = inclL - extraL;
            }
            const uint32_t items = itemsM + itemsL;
            // room in region B for the wave's further segments (a wave that cuts anything: one atomic).  A reservation is never undone (an
            // add followed by a subtract is not atomic across waves); a count past extraCap just means "region B is full", every reader
            // clamps it; the one wave whose range straddles the end owns [at, extraCap) and marks those slots empty.  No room, no
            // cutting: the wave's rays stay whole -- decided BEFORE the classes are counted, so that every counted entry is placed.
            uint32_t extraBase = 0;
            bool room = true;
            if (items != 0u) { // wave-uniform
                if (lane == 0u) extraBase = atomicAdd(&ctlOut[RT_WF_CTL_EXTRA], items);
                extraBase = (uint32_t)__shfl((int)extraBase, 0, 64);
                if ((uint64_t)extraBase + items > (uint64_t)W.extraCap) {
                    room = false;
                    for (uint32_t i = extraBase + lane; i < W.extraCap; i += 64) {
                        W.ent[outq][4 * (size_t)(2u * W.capacity + i)].x = 0xffffffffu;
                        W.sortRank[2u * W.capacity + i] = 0xffffffffu;
                    }
                    if (nsegM > 1u) nsegM = 1u;
                    if (nsegL > 1u) nsegL = 1u;
                }
            }
            const uint32_t extraAt = 2u * W.capacity + extraBase; // region B of the entry array starts after the 2*capacity queue slots
            const float taM = fminf(planM.start.dx, fminf(planM.start.dy, planM.start.dz)), taL = fminf(planL.start.dx, fminf(planL.start.dy, planL.start.dz));
            const uint32_t perM = nsegM ? (planM.visits + nsegM - 1) / nsegM : 1u, perL = nsegL ? (planL.visits + nsegL - 1) / nsegL : 1u;
            // classes and ranks inside the wave: segment 0 of both rays, then every further segment (whose class follows from the cut
            // positions alone: the cuts that exist are a prefix of 1 .. nseg - 1, cut_at)
            uint32_t binM = 0, rankM = 0, binL = 0, rankL = 0;
            {
                float tau;
                if (emit) { binM = visit_class(nsegM > 1u && cut_at(taM, planM.te, 1u, nsegM, tau) ? perM : planM.visits); rankM = atomicAdd(&waveHist[wave][binM], 1u); }
                if (emitLa) { binL = visit_class(nsegL > 1u && cut_at(taL, planL.te, 1u, nsegL, tau) ? perL : planL.visits); rankL = atomicAdd(&waveHist[wave][binL], 1u); }
                if (cutting && room) {
#pragma unroll 1
                    for (int which = 0; which < 2; ++which) {
                        const uint32_t n = which ? nsegL : nsegM, per = which ? perL : perM, visits = which ? planL.visits : planM.visits;
                        const float ta = which ? taL : taM, te = which ? planL.te : planM.te;
                        const uint32_t at = which ? itemsM + beforeL : beforeM;
                        for (uint32_t k = 1; k < n; ++k) {
                            uint32_t word = 0xffffffffu; // (cut k does not exist: no entry)
                            if (cut_at(ta, te, k, n, tau)) {
                                const bool cutNext = k + 1u < n && cut_at(ta, te, k + 1u, n, tau);
                                const uint32_t bin = visit_class(cutNext ? per : visits - per * k);
                                word = atomicAdd(&waveHist[wave][bin], 1u) | (bin << 24);
                            }
                            itemWord[wave][at + k - 1] = word;
                            itemOwner[wave][at + k - 1] = (uint8_t)lane;
                        }
                    }
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
            __builtin_amdgcn_wave_barrier();
            // lane b fetches where class b's ranks of this wave start: the one returned atomic that stands between the plans and the stores
            {
                uint32_t l2 = lane;
                asm volatile("" : "+v"(l2)); // (the address is made here: hoisted out of the chunk loop it was spilled)
                const uint32_t n = waveHist[wave][l2];
                waveBase[wave][l2] = n ? atomicAdd(&ctlOut[RT_WF_CTL_HIST + copy * RT_WF_SORT_BINS + l2], n) : 0u;
                waveHist[wave][l2] = 0u; // (for this wave's next chunk)
            }
            __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
            __builtin_amdgcn_wave_barrier();
            // segment 0 of both rays, at their queue indices; where a cut ray's segment 0 ends is filled in by the lane that makes cut 1
#pragma unroll 1
            for (int which = 0; which < 2; ++which) {
                const bool has = which ? emitLa : emit;
                if (!has) continue;
                const V3 o = which ? lo3 : ro, d = which ? ld3v : rd;
                const float tmin = which ? latmin : rtmin, tmax = which ? RT_INF : rtmax;
                const uint32_t excluded = which ? laexcl : rexcl, mine = which ? slotLa : slot;
                const uint32_t bin = which ? binL : binM, rank = which ? rankL : rankM, cell = which ? planL.start.cell : planM.start.cell;
                const uint32_t endCell = which ? planL.endCell : planM.endCell, nseg = which ? nsegL : nsegM;
                uint4 *e = W.ent[outq] + 4 * (size_t)mine;
                if (!which) W.pathOf[outq][mine] = make_uint2(a, emitLa ? slotLa : 0xffffffffu); // (only main entries are ever looked up: wf_logic_kernel's prologue)
                W.hitKey[outq][mine] = ~0ull; // no segment of this ray has a hit yet
                if (nseg > 1u) { // (.z comes from another lane: not written here, so that the two stores cannot meet)
                    *reinterpret_cast<uint2 *>(e) = make_uint2(mine, cell);
                    reinterpret_cast<uint32_t *>(e)[3] = excluded;
                } else e[0] = make_uint4(mine, cell, endCell, excluded);
                e[1] = make_uint4(__float_as_uint(which ? planL.start.dx : planM.start.dx), __float_as_uint(which ? planL.start.dy : planM.start.dy),
                                  __float_as_uint(which ? planL.start.dz : planM.start.dz), __float_as_uint(tmin));
                e[2] = make_uint4(__float_as_uint(o.x), __float_as_uint(o.y), __float_as_uint(o.z), __float_as_uint(tmax));
                e[3] = make_uint4(__float_as_uint(d.x), __float_as_uint(d.y), __float_as_uint(d.z), 0u); // (segment 0)
                W.sortRank[mine] = waveBase[wave][bin] + rank;
                W.sortTag[mine] = (uint16_t)(bin | (copy << 6));
            }
            // The further segments, one per lane whoever's ray it is (a lane making its ray's up to 11 cuts one after the other while the
            // wave's other lanes wait was 40 % of a chunk's time).  Segment k goes from the walk's state at tau_k to the start cell of
            // segment k + 1; every lane can tell from k alone whether cut k and cut k + 1 exist, and writes where the segment BEFORE its
            // own ends.
            if (room) {
#pragma unroll 1
                for (int which = 0; which < 2; ++which) {
                    // (what the lanes hand out is picked by `which`, the same for the whole wave, BEFORE the shuffles: a shuffle reads nothing
                    // from a lane that sits the instruction out, so none of them may stand in a branch of its own)
                    const V3 so = which ? lo3 : ro, sd = which ? ld3v : rd;
                    const float stmin = which ? latmin : rtmin, stmax = which ? RT_INF : rtmax, sta = which ? taL : taM, ste = which ? planL.te : planM.te;
                    const uint32_t sexcl = which ? laexcl : rexcl, smine = which ? slotLa : slot, scell = which ? planL.start.cell : planM.start.cell;
                    const uint32_t snseg = which ? nsegL : nsegM, send = which ? planL.endCell : planM.endCell, sbefore = which ? beforeL : beforeM;
                    const uint32_t count = which ? itemsL : itemsM, base = which ? itemsM : 0u;
                    for (uint32_t i0 = 0; i0 < count; i0 += 64) {
                        const uint32_t i = i0 + lane;
                        const bool liveItem = i < count;
                        const uint32_t owner = liveItem ? itemOwner[wave][base + i] : 0u;
                        const uint32_t word = liveItem ? itemWord[wave][base + i] : 0xffffffffu;
                        const uint32_t k = i - (uint32_t)__shfl((int)sbefore, owner, 64) + 1u; // this lane makes cut k of `owner`'s ray
                        const V3 po = mk(__shfl(so.x, owner, 64), __shfl(so.y, owner, 64), __shfl(so.z, owner, 64));
                        const V3 pd = mk(__shfl(sd.x, owner, 64), __shfl(sd.y, owner, 64), __shfl(sd.z, owner, 64));
                        const float ptmin = __shfl(stmin, owner, 64), ptmax = __shfl(stmax, owner, 64), pta = __shfl(sta, owner, 64), pte = __shfl(ste, owner, 64);
                        const uint32_t pexcl = __shfl(sexcl, owner, 64), pmine = __shfl(smine, owner, 64), pcell = __shfl(scell, owner, 64);
                        const uint32_t pnseg = __shfl(snseg, owner, 64), pend = __shfl(send, owner, 64);
                        if (!liveItem) continue;
                        const uint32_t at = extraAt + base + i; // entry of segment k
                        uint32_t *prev = reinterpret_cast<uint32_t *>(W.ent[outq] + 4 * (size_t)(k == 1u ? pmine : at - 1u));
                        uint32_t *self = reinterpret_cast<uint32_t *>(W.ent[outq] + 4 * (size_t)at);
                        if (word == 0xffffffffu) { // rounding left no room for this cut: the segment before runs to the ray's end, this one does not exist
                            prev[2] = pend;
                            self[0] = 0xffffffffu;
                            W.sortRank[at] = 0xffffffffu;
                            continue;
                        }
                        float tau, tauNext;
                        (void)cut_at(pta, pte, k, pnseg, tau);
                        const bool cutNext = k + 1u < pnseg && cut_at(pta, pte, k + 1u, pnseg, tauNext);
                        DdaState st;
                        // (counting the crossings with T <= tau from the ray's start cell: the state does not depend on where counting begins)
                        const uint32_t nx = axis_state_at(planes, pcell & 255u, po.x, pd.x, tau, st.dx, cellLut, lutScale[0]);
                        const uint32_t ny = axis_state_at(planes + (RT_GRID_DIV + 1), (pcell >> 8) & 255u, po.y, pd.y, tau, st.dy, cellLut + 256, lutScale[1]);
                        const uint32_t nz = axis_state_at(planes + 2 * (RT_GRID_DIV + 1), pcell >> 16, po.z, pd.z, tau, st.dz, cellLut + 512, lutScale[2]);
                        st.cell = nx | (ny << 8) | (nz << 16);
                        prev[2] = st.cell; // the segment before ends where this one starts
                        *reinterpret_cast<uint2 *>(self) = make_uint2(pmine, st.cell);
                        self[3] = pexcl;
                        if (!cutNext) self[2] = pend; // the ray's last segment
                        uint4 *se = reinterpret_cast<uint4 *>(self);
                        se[1] = make_uint4(__float_as_uint(st.dx), __float_as_uint(st.dy), __float_as_uint(st.dz), __float_as_uint(ptmin));
                        se[2] = make_uint4(__float_as_uint(po.x), __float_as_uint(po.y), __float_as_uint(po.z), __float_as_uint(ptmax));
                        se[3] = make_uint4(__float_as_uint(pd.x), __float_as_uint(pd.y), __float_as_uint(pd.z), k << 24);
                        const uint32_t bin = word >> 24;
                        W.sortRank[at] = waveBase[wave][bin] + (word & 0xffffffu);
                        W.sortTag[at] = (uint16_t)(bin | (copy << 6));
                    }
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
            __builtin_amdgcn_wave_barrier(); // (waveBase, itemOwner and itemWord are rewritten by this wave's next chunk)
        }
        if (copy != 0xfffffff0u) DG(9);
        if (copy != 0xfffffff0u) DG(10);
#ifdef RT_DIAG_LOGIC
        if (round == RT_DIAG_LOGIC) {
            dg[11] = diag_stamp();
            // a stamp taken in a divergent branch belongs to the wave: take the latest any lane saw, and keep the sequence monotone
            for (int i = 1; i < 12; ++i) {
                unsigned long long v = dg[i];
                for (int off = 32; off >= 1; off >>= 1) { const unsigned long long o2 = __shfl_xor((long long)v, off, 64); v = v > o2 ? v : o2; }
                dg[i] = v > dg[i - 1] ? v : dg[i - 1];
            }
            if (lane == 0) { // machine | look-ahead pick | append atomics | entries | classes + ranks | rest | chunks | whole chunk
                atomicAdd(&S.stats[0], dg[6] - dg[0]); atomicAdd(&S.stats[1], dg[7] - dg[6]); atomicAdd(&S.stats[2], dg[8] - dg[7]);
                atomicAdd(&S.stats[3], dg[9] - dg[8]); atomicAdd(&S.stats[4], dg[10] - dg[9]); atomicAdd(&S.stats[5], dg[11] - dg[10]);
                atomicAdd(&S.stats[6], 1ull); atomicAdd(&S.stats[7], dg[11] - dg[0]);
            }
        }
#endif
    }
}

// ---- stage 2b: an ordered round's entries, longest predicted walk first ------------------------------------------------------
// Queue slice s of a round (s < slices: main entries, s >= slices: look-ahead entries): where it starts and how long it is.
__device__ __forceinline__ uint32_t slice_first(const RtWavefront &W, uint32_t slices, uint32_t s)
{
    const uint32_t kind = (s >= slices) ? 1u : 0u, shard = s - kind * slices;
    return kind * W.capacity + shard * (W.capacity / slices);
}
__device__ __forceinline__ uint32_t slice_count(const uint32_t *ctl, uint32_t slices, uint32_t s)
{
    const uint32_t kind = (s >= slices) ? 1u : 0u, shard = s - kind * slices;
    return ctl[RT_WF_CTL_COUNTS + kind * RT_WF_SHARDS + shard];
}

// Work items: the used 256-entry blocks of the queue slices (region A: segment 0 of every ray, in queue order), then the blocks of
// region B (further segments, densely packed); a fixed grid takes them in turn.  The logic kernel left a rank inside its (class,
// copy) and the class itself per entry; the classes' sizes are in the round's histogram.
__global__ __launch_bounds__(256) void wf_scatter_kernel(const RtWavefront W, const uint32_t round, const uint32_t slices)
{
    __shared__ uint32_t base[RT_WF_SORT_BINS * RT_WF_SORT_COPIES];
    __shared__ uint32_t longWave[4];
    uint32_t *ctl = W.ctl + (round % 3) * RT_WF_CTL_WORDS;
    const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    {
        const uint32_t c0 = ctl[RT_WF_CTL_COUNTS + threadIdx.x], c1 = ctl[RT_WF_CTL_COUNTS + 256 + threadIdx.x];
        uint32_t m = max(c0, c1);
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) m = max(m, (uint32_t)__shfl_xor((int)m, off, 64));
        if (lane == 0) longWave[wave] = m;
    }
    if (threadIdx.x < RT_WF_SORT_BINS) { // one wave: exclusive prefix over (bin, copy), bin-major
        uint32_t sum = 0;
        for (int c = 0; c < RT_WF_SORT_COPIES; ++c) sum += ctl[RT_WF_CTL_HIST + c * RT_WF_SORT_BINS + threadIdx.x];
        uint32_t incl = sum;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const uint32_t up = __shfl_up(incl, off, 64);
            if ((int)threadIdx.x >= off) incl += up;
        }
        uint32_t at = incl - sum;
        for (int c = 0; c < RT_WF_SORT_COPIES; ++c) { base[threadIdx.x * RT_WF_SORT_COPIES + c] = at; at += ctl[RT_WF_CTL_HIST + c * RT_WF_SORT_BINS + threadIdx.x]; }
        if (blockIdx.x == 0 && threadIdx.x == RT_WF_SORT_BINS - 1) ctl[RT_WF_CTL_TOTAL] = incl; // all entries are in [0, total)
    }
    __syncthreads();
    const uint32_t usedBlocks = (max(max(longWave[0], longWave[1]), max(longWave[2], longWave[3])) + 255u) >> 8;
    const uint32_t itemsA = 2u * slices * usedBlocks;
    const uint32_t extra = min(ctl[RT_WF_CTL_EXTRA], W.extraCap); // the count runs past the capacity when region B filled up (wf_logic_kernel)
    const uint32_t itemsB = (extra + 255u) >> 8;
    for (uint32_t item = blockIdx.x; item < itemsA + itemsB; item += gridDim.x) {
        uint32_t mine = 0;
        bool valid = false;
        if (item < itemsA) {
            const uint32_t sl = item % (2u * slices);
            const uint32_t local = (item / (2u * slices)) * 256 + threadIdx.x;
            valid = local < slice_count(ctl, slices, sl);
            mine = slice_first(W, slices, sl) + local;
        } else {
            const uint32_t local = (item - itemsA) * 256 + threadIdx.x;
            valid = local < extra;
            mine = 2u * W.capacity + local;
        }
        if (valid) {
            // only the ORDER is written: the trace kernel gathers its 64-byte entries through it
            const uint32_t rank = W.sortRank[mine];
            if (rank != 0xffffffffu) { // not an unused reservation
                const uint32_t tag = W.sortTag[mine]; // bin | copy << 6
                W.sortedIdx[base[(tag & 63u) * RT_WF_SORT_COPIES + (tag >> 6)] + rank] = mine;
            }
        }
    }
}

// ---- stage 3: grid traversal (raytrace_opencl.c:324-401) ------------------------------------------------------------------
// One sorted entry per lane.  WALK, THEN TEST: where a ray walks does not depend on what it hits -- only where it stops does:
// the reference resets its running maximum in every cell (:366) and ends at the first cell that produced any hit (:380).  So a
// lane walks freely and only RECORDS the occupied cells it passes in a small per-lane list in LDS; when lists fill up, or
// nobody can walk any further, the wave tests the recorded cells cooperatively (the items of all lanes are flattened, a lane
// takes one cell per round whoever recorded it, ray data comes from the owner lane by ds_bpermute); the first cell with a hit
// ends the ray, a ray without a hit carries on walking from where it stands.  Results do not depend on where a walk is cut.
// Waves of a workgroup never synchronise after the plane table is staged.  The walk is VALU-issue bound (a round's waves outnumber the
// wave slots), so the step is built to cost as few vector instructions as possible:
//   * per-ray constants of the step are precomputed per axis (signed cell increment, plane-table offset, exit coordinate);
//   * the occupancy word of a 4x4x4 block is found at byte offset 3*(cell & 0xFCFCFC) of a sparsely indexed copy of the
//     block table (two instructions instead of six: rt_device.h, gridBlockSparse), its bit with one multiply (bit-gather) and one bit-field extract;
//   * an occupied cell is recorded as its DENSE id (rank of its block, which arrives with the occupancy word, + occupied cells below
//     it): the index the test phase gathers the cell's record by.  Working it out in the test phase, where all 64 lanes have an item,
//     costs fewer instructions (6 of 64 lanes are on an occupied cell in the walk) but a second dependent gather of the block word
//     per item: one round trip per item instead of two was worth 5 % of the kernel (round 3).
#ifndef RT_WF_LEAN_WAVES
#define RT_WF_LEAN_WAVES 5
#endif
#ifndef RT_WF_LEAN_LIST
#define RT_WF_LEAN_LIST 16            // per-lane LDS slots: recorded occupied cells + the cells logged by the current blind phase
#endif
#ifndef RT_WF_BLIND
#define RT_WF_BLIND 5                 // cell visits per blind phase (4-7 are within 0.5 % of each other, 8 and 10 are 1 % and 3 % slower)
#endif
#ifndef RT_WF_MORE_ITEMS
#define RT_WF_MORE_ITEMS 128          // per-wave list of further candidates (beyond a cell's first) awaiting their test
#endif
#ifndef RT_WF_LEAN_STALL
#define RT_WF_LEAN_STALL 64           // test once (lanes without room for another phase) x this exceeds the lanes still walking
#endif
#define RT_WF_GROUP_RAYS 128          // rays a workgroup of wf_trace_kernel<false> plans at most (RtRoundMode::groupRays): its first two waves
template <bool ORDERED>
__global__ __launch_bounds__(256, RT_WF_LEAN_WAVES) void wf_trace_kernel(const RtDevScene S, const RtWavefront W, const uint32_t round, const RtRoundMode mode)
{
    __shared__ float planes[3 * (RT_GRID_DIV + 1)];
    __shared__ uint32_t cellList[RT_WF_LEAN_LIST][256];                 // [entry][thread] packed cells cx | cy<<8 | cz<<16
    __shared__ uint8_t ownerOf[4][RT_WF_LEAN_LIST * 64];                // per wave: lane that recorded item c
    __shared__ unsigned long long keyOf[4][64];                         // per wave and lane: (cell order, t, pair index) of the earliest hit
    __shared__ uint32_t moreOf[4][2][RT_WF_MORE_ITEMS];                 // per wave: further candidates under test {owner | cell order << 6}, {pair}
    __shared__ uint32_t moreCount[4];
    static_assert(RT_WF_LEAN_LIST <= 16, "hit_key: 4 bits of cell order");
    // <false> only: the workgroup's rays as planned by its first lanes {o,tmin} {d,tmax} {excluded, q, start cell, end cell} {dx,dy,dz,te},
    // the first segment of every ray in the workgroup's list of segments, and whose ray a segment is.  They live in the cell
    // lists' memory: the planning is over before the walk begins, and 5 KB more of LDS would cost the kernel its fifth workgroup per CU.
    static_assert(sizeof(float4) * 4 * RT_WF_GROUP_RAYS + 4 * (RT_WF_GROUP_RAYS + 1) + 256 <= sizeof(cellList) && RT_WF_GROUP_RAYS == 128, "the plan tables fit the cell lists; waves 0 and 1 plan");
    float4 *rayO = reinterpret_cast<float4 *>(&cellList[0][0]), *rayD = rayO + RT_WF_GROUP_RAYS, *rayT = rayD + RT_WF_GROUP_RAYS;
    uint4 *rayX = reinterpret_cast<uint4 *>(rayT + RT_WF_GROUP_RAYS);
    uint32_t *segFirst = reinterpret_cast<uint32_t *>(rayX + RT_WF_GROUP_RAYS);
    uint8_t *segOwner = reinterpret_cast<uint8_t *>(segFirst + RT_WF_GROUP_RAYS + 1); // [256]: a workgroup's segments are one per lane at most

    // Which rays this workgroup takes.
    //   ordered round:  256 entries, block i of the sorted order, through sortedIdx -- the hardware dispatches workgroups in order, so
    //                   the longest walks start first and a free slot always gets the longest work left;
    //   other rounds:   mode.groupRays rays of one queue slice, slice-minor (piece b of every slice before piece b + 1 of any: slices
    //                   fill evenly); the workgroup plans them, cuts them, and deals the segments to its lanes 256 at a time.
    // A planned frame's grid is sized from the same frame's previous rendering (rt_api.cpp); should it be too small the host is told
    // and renders the frame again with the worst-case grid (a loop that strides over the rest measured 4 % slower).
    const uint32_t *ctl = W.ctl + (round % 3) * RT_WF_CTL_WORDS;
    const uint32_t par = round & 1;
    const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#ifdef RT_DIAG_STAMPS
    const unsigned long long dgK0 = diag_stamp();
#endif
    uint32_t mine = 0, segments = 0;
    bool active = false;
    if (ORDERED) {
        const uint32_t total = ctl[RT_WF_CTL_TOTAL];
        const uint32_t blocksUsed = (total + 255u) >> 8;
        if (blockIdx.x == 0 && threadIdx.x == 0) {
            if (round < RT_WF_ROUND_LOG) reinterpret_cast<uint32_t *>(W.roundLog + round)[2] = min(ctl[RT_WF_CTL_EXTRA], W.extraCap);
            if (blocksUsed > gridDim.x) atomicOr(W.hostStatus + RT_WF_STATUS_ERROR, RT_WF_ERR_GRID);
        }
        if (blockIdx.x >= blocksUsed) return; // whole workgroup beyond the entries
        const uint32_t at = blockIdx.x * 256 + threadIdx.x;
        active = at < total;
        if (active) mine = W.sortedIdx[at];
        for (int i = threadIdx.x; i < 3 * (RT_GRID_DIV + 1); i += 256) planes[i] = S.boxMin[i];
        __syncthreads();
    } else {
        // Workgroup i takes piece `row` of queue slice `sl`, pieces numbered through the slices in turn -- found by adding up the
        // slices' piece counts, not by arithmetic on i: a grid striped "slice = i % slices" puts the workgroups of the empty slices (a
        // late round has no look-ahead rays) at a fixed stride, the hardware deals workgroups to XCDs and CUs round-robin, and half
        // of the CUs ended up with all of the round's work (measured: 64 -> 130 us).
        const uint32_t group = min(mode.groupRays, (uint32_t)RT_WF_GROUP_RAYS);
        __shared__ uint32_t pieceWave[4], pieceAt[2];
        uint32_t sl = 0, row = 0, count = 0;
        {
            static_assert(2 * RT_WF_SHARDS == 512, "two queue slices per thread");
            const uint32_t s0 = threadIdx.x, s1 = 256u + threadIdx.x;
            const uint32_t c0 = s0 < 2u * mode.slices ? slice_count(ctl, mode.slices, s0) : 0u, c1 = s1 < 2u * mode.slices ? slice_count(ctl, mode.slices, s1) : 0u;
            const uint32_t p0 = (c0 + group - 1u) / group, p1 = (c1 + group - 1u) / group;
            // pieces before slice s0 (slices 0..255 first, then 256..511): scan of p0 over the workgroup, then of p1 behind it
            uint32_t i0 = p0, i1 = p1;
#pragma unroll
            for (int off = 1; off < 64; off <<= 1) {
                const uint32_t u0 = __shfl_up(i0, off, 64), u1 = __shfl_up(i1, off, 64);
                if ((int)lane >= off) { i0 += u0; i1 += u1; }
            }
            if (lane == 63) { pieceWave[wave] = i0; }
            __syncthreads();
            uint32_t before0 = 0, all0 = 0;
            for (uint32_t w = 0; w < 4; ++w) { if (w < wave) before0 += pieceWave[w]; all0 += pieceWave[w]; }
            __syncthreads();
            if (lane == 63) { pieceWave[wave] = i1; }
            __syncthreads();
            uint32_t before1 = all0, all1 = all0;
            for (uint32_t w = 0; w < 4; ++w) { if (w < wave) before1 += pieceWave[w]; all1 += pieceWave[w]; }
            const uint32_t first0 = before0 + i0 - p0, first1 = before1 + i1 - p1;
            if (blockIdx.x == 0 && threadIdx.x == 0 && all1 > gridDim.x) atomicOr(W.hostStatus + RT_WF_STATUS_ERROR, RT_WF_ERR_GRID);
            if (blockIdx.x >= all1) return; // whole workgroup beyond the round's pieces
            if (first0 <= blockIdx.x && blockIdx.x < first0 + p0) { pieceAt[0] = s0; pieceAt[1] = blockIdx.x - first0; }
            if (first1 <= blockIdx.x && blockIdx.x < first1 + p1) { pieceAt[0] = s1; pieceAt[1] = blockIdx.x - first1; }
            __syncthreads();
            sl = pieceAt[0]; row = pieceAt[1];
            count = slice_count(ctl, mode.slices, sl);
        }
        for (int i = threadIdx.x; i < 3 * (RT_GRID_DIV + 1); i += 256) planes[i] = S.boxMin[i];
        __syncthreads();
        // The first lanes plan the workgroup's rays: every lane busy, where the logic kernel would have planned one ray in ten lanes.
        // A workgroup takes every rowsUsed-th ray of its slice, not a stretch of it: neighbours in the queue are neighbours in the
        // image, their rays are alike, and a stretch of long rays made one workgroup's list of segments two or three times the
        // others' (the round lasts as long as its slowest workgroup: 64 -> 160 us).
        const uint32_t rowsUsed = (count + group - 1u) / group;
        __shared__ uint32_t planWave[2];
        uint32_t n = 0;
        const uint32_t local = row + threadIdx.x * rowsUsed;
        const bool planner = threadIdx.x < group && local < count; // (group <= 128: waves 0 and 1)
        if (planner) {
            const uint32_t q = slice_first(W, mode.slices, sl) + local;
            const uint4 *e = W.ent[par] + 4 * (size_t)q;
            const uint32_t excl = reinterpret_cast<const uint32_t *>(e)[3];
            const float tmin = __uint_as_float(reinterpret_cast<const uint32_t *>(e)[7]);
            const uint4 c2 = e[2], c3 = e[3];
            const V3 o = mk(__uint_as_float(c2.x), __uint_as_float(c2.y), __uint_as_float(c2.z)), d = mk(__uint_as_float(c3.x), __uint_as_float(c3.y), __uint_as_float(c3.z));
            const float tmax = __uint_as_float(c2.w);
            const EntryPlan plan = plan_ray(planes, o, d, tmin, tmax);
            n = segments_of(plan, d, tmax, mode.segLen);
            rayO[threadIdx.x] = pack4(o, tmin); rayD[threadIdx.x] = pack4(d, tmax);
            rayX[threadIdx.x] = make_uint4(excl, q, plan.start.cell, plan.endCell);
            rayT[threadIdx.x] = make_float4(plan.start.dx, plan.start.dy, plan.start.dz, plan.te);
        }
        // one pass of 256 lanes must do: a workgroup whose rays would make more segments cuts them coarser
        if (wave < 2) {
            uint32_t sum = n;
#pragma unroll
            for (int off = 32; off >= 1; off >>= 1) sum += __shfl_xor(sum, off, 64);
            if (lane == 0) planWave[wave] = sum;
        }
        __syncthreads();
        const uint32_t all = planWave[0] + planWave[1];
        __syncthreads();
        if (wave < 2) {
            // (every ray keeps one segment at least: what is left of the 256 lanes is shared out in proportion)
            const uint32_t raysHere = min(group, (count - row + rowsUsed - 1u) / rowsUsed);
            if (all > 256u && n > 1u) n = max(1u, n * (256u - raysHere) / all);
            uint32_t incl = n;
#pragma unroll
            for (int off = 1; off < 64; off <<= 1) {
                const uint32_t up = __shfl_up(incl, off, 64);
                if ((int)lane >= off) incl += up;
            }
            if (lane == 63) planWave[wave] = incl;
            n = incl - n; // (this lane's first segment, inside its wave)
        }
        __syncthreads();
        if (wave < 2) {
            const uint32_t nextFirst = (uint32_t)__shfl_down((int)n, 1, 64); // (by every lane: a shuffle reads nothing from a lane that sits it out)
            const uint32_t first = n + (wave == 1 ? planWave[0] : 0u), mineN = (lane == 63 ? planWave[wave] : nextFirst) - n;
            segFirst[threadIdx.x] = first;
            if (threadIdx.x == 127) segFirst[128] = planWave[0] + planWave[1];
            for (uint32_t k = 0; k < mineN; ++k) segOwner[first + k] = (uint8_t)threadIdx.x;
        }
        __syncthreads();
        segments = segFirst[128];
    }

    // an ordered round: the workgroup's 256 entries; any other round: the workgroup's segments (256 at most, see above)
    for (uint32_t pass = 0; pass < 1u; ++pass) {
    uint32_t q = 0, excluded = RT_NONE, cell = 0, endCell = 0xffffffffu, seg = 0;
    V3 o = mk(0, 0, 0), d = mk(1, 1, 1);
    float tmin = 0.f, tmax = 0.f, dx = 0.f, dy = 0.f, dz = 0.f;
    if (ORDERED) {
        if (active) {
            const uint4 *e = W.ent[par] + 4 * (size_t)mine;
            const uint4 c0 = e[0], c1 = e[1], c2 = e[2], c3 = e[3];
            q = c0.x; cell = c0.y & 0xffffffu; endCell = c0.z; excluded = c0.w;
            dx = __uint_as_float(c1.x); dy = __uint_as_float(c1.y); dz = __uint_as_float(c1.z); tmin = __uint_as_float(c1.w);
            o = mk(__uint_as_float(c2.x), __uint_as_float(c2.y), __uint_as_float(c2.z)); tmax = __uint_as_float(c2.w);
            d = mk(__uint_as_float(c3.x), __uint_as_float(c3.y), __uint_as_float(c3.z));
            seg = c3.w >> 24;
        }
    } else {
        const uint32_t i = threadIdx.x;
        active = i < segments;
        if (active) {
            const uint32_t owner = segOwner[i];
            const uint32_t k = i - segFirst[owner], n = segFirst[owner + 1] - segFirst[owner];
            const float4 ro = rayO[owner], rd = rayD[owner], rt = rayT[owner];
            const uint4 rx = rayX[owner];
            o = xyz(ro); tmin = ro.w; d = xyz(rd); tmax = rd.w; excluded = rx.x; q = rx.y; seg = k;
            // segment k goes from the walk's state at tau_k (tau_0: the ray's own start) to the start cell of segment k + 1
            const float ta = fminf(rt.x, fminf(rt.y, rt.z)), te = rt.w;
            float tau;
            cell = rx.z; dx = rt.x; dy = rt.y; dz = rt.z;
            if (k > 0u) {
                if (cut_at(ta, te, k, n, tau)) {
                    // (counting the crossings with T <= tau from the ray's start cell: the state does not depend on where counting begins)
                    const uint32_t nx = axis_state_at(planes, rx.z & 255u, o.x, d.x, tau, dx);
                    const uint32_t ny = axis_state_at(planes + (RT_GRID_DIV + 1), (rx.z >> 8) & 255u, o.y, d.y, tau, dy);
                    const uint32_t nz = axis_state_at(planes + 2 * (RT_GRID_DIV + 1), rx.z >> 16, o.z, d.z, tau, dz);
                    cell = nx | (ny << 8) | (nz << 16);
                } else active = false; // rounding left no room for this cut: the segment before runs to the ray's end
            }
            endCell = rx.w;
            if (active && k + 1u < n && cut_at(ta, te, k + 1u, n, tau)) {
                float unused;
                const uint32_t nx = axis_state_at(planes, rx.z & 255u, o.x, d.x, tau, unused);
                const uint32_t ny = axis_state_at(planes + (RT_GRID_DIV + 1), (rx.z >> 8) & 255u, o.y, d.y, tau, unused);
                const uint32_t nz = axis_state_at(planes + 2 * (RT_GRID_DIV + 1), rx.z >> 16, o.z, d.z, tau, unused);
                endCell = nx | (ny << 8) | (nz << 16);
            }
        }
        __syncthreads(); // the plan tables become the cell lists
    }
    const bool fastWave = S.planesTame && W.fastQuotient && __ballot(active && !(tame_origin(o.x) && tame_origin(o.y) && tame_origin(o.z) &&
                                                                               tame_direction(d.x) && tame_direction(d.y) && tame_direction(d.z))) == 0ull;
    // per-axis step constants (:387-398): direction of travel is fixed per ray
    const bool px = (0.f <= d.x), py = (0.f <= d.y), pz = (0.f <= d.z);
    const uint32_t stepX = px ? 1u : (uint32_t)-1, stepY = py ? (1u << 8) : (uint32_t)-(1 << 8), stepZ = pz ? (1u << 16) : (uint32_t)-(1 << 16);
    // byte offset into `planes` of the plane that bounds the NEW cell ahead, counted from the OLD cell's coordinate c (which the exit test
    // has just extracted): the new cell is c +- 1 and its far plane c + 2 going up, c - 1 going down -> 4*(axisBase + (positive ? 2 : -1)),
    // in wrapping 32-bit arithmetic (c >= 1 when going down: c == 0 ended the walk)
    const uint32_t offX = 4u * (px ? 2u : (uint32_t)-1), offY = 4u * ((RT_GRID_DIV + 1) + (py ? 2u : (uint32_t)-1)),
                   offZ = 4u * (2 * (RT_GRID_DIV + 1) + (pz ? 2u : (uint32_t)-1));
    const char *__restrict__ blockTable = reinterpret_cast<const char *>(S.gridBlockSparse);
    const char *planeBytes = reinterpret_cast<const char *>(planes);

    uint32_t wordKey = 0xffffffffu, wordLo = 0, wordHi = 0, wordRank = 0; // occupancy word + rank of the last block looked up (key = cell & 0xFCFCFC)
    uint32_t listed = 0;
    bool walkEnded = !active;
    uint32_t spins = 0;
#ifdef RT_DIAG_STAMPS
    unsigned long long dgWalk = 0, dgTest = 0, dgWalkIters = 0, dgBatches = 0, dgSteps = 0, dgItems = 0;
    const unsigned long long dgStart = diag_stamp();
#endif

#pragma unroll 1
    for (;;) {
        // ---- walk: BLIND phases of RT_WF_BLIND cell visits (pure DDA stepping, every visited cell logged in LDS), each
        // followed by ONE batched look-up of the logged cells' occupancy words.  The look-ups of a phase are independent
        // loads, so a wave waits for memory once per phase instead of once per visit.
#ifdef RT_DIAG_STAMPS
        const unsigned long long dgW0 = diag_stamp();
#endif
#pragma unroll 1
        for (;;) {
            const bool canWalk = !walkEnded && listed + RT_WF_BLIND <= RT_WF_LEAN_LIST;
            const unsigned long long walkers = __ballot(canWalk);
            if (walkers == 0ull) break;
            const int stalled = __popcll(__ballot(!walkEnded && !canWalk));
            if (stalled * RT_WF_LEAN_STALL > __popcll(walkers)) break;
            if (++spins > W.spinLimit) break; // cannot happen (a ray makes at most 766 visits); keeps a logic error from hanging the GPU
#ifdef RT_DIAG_STAMPS
            dgWalkIters++;
#endif
            uint32_t logged = 0;
            // The quotient (plane - o) / d is the correctly rounded one, as the compiler expands it: v_div_scale x2, v_rcp, two
            // refinements of the reciprocal, q0 = n*r, two corrections, v_div_fmas, v_div_fixup.  When the exponents of n and d
            // are tame (tame_ray below) the scale instructions return their operands, v_div_fmas is a plain fma and the fix-up
            // returns its input: what is left is q0 = n*r1, e1 = fma(-d,q0,n), q1 = fma(e1,r1,q0), e2 = fma(-d,q1,n),
            // q2 = fma(e2,r1,q1) with r1 = the refined reciprocal, which depends on the axis only -- the same operations on the
            // same operands, bit for bit, in 5 issue slots instead of 14 (v_rcp_f32 is quarter rate).  A wave takes this path
            // when all of its rays are tame; the reciprocals live in registers only while the wave walks.
#define RT_WALK_STEP(FAST)                                                                                                          \
                if (canWalk && !walkEnded) {                                                                                        \
                    cellList[listed + u][threadIdx.x] = cell; /* (a lane that walks step u has walked every step before it: logged == u) */ \
                    ++logged;                                                                                                       \
                    /* axis choice (:387-398): x only if strictly smallest, else y if smaller than z, else z */                     \
                    const bool sxm = (dx < dy) & (dx < dz);                                                                         \
                    const bool sym = !sxm & (dy < dz);                                                                              \
                    const uint32_t shift = sxm ? 0u : (sym ? 8u : 16u);                                                             \
                    const uint32_t stepSel = sxm ? stepX : (sym ? stepY : stepZ);                                                   \
                    const uint32_t coord = (cell >> shift) & 255u;                                                                  \
                    /* the end cell ends the walk after it has been visited (:380-381); so does a step that would leave the grid  */ \
                    /* (:389,:393,:397): coordinate 255 going up, 0 going down.  ONE condition, one branch: nested, the compiler   */ \
                    /* carried "cell is the end cell" through three register moves per step                                         */ \
                    const bool done = (cell == endCell) | (coord == (((int32_t)stepSel > 0) ? 255u : 0u));                          \
                    if (!done) {                                                                                                    \
                        cell += stepSel;                                                                                            \
                        const uint32_t off = sxm ? offX : (sym ? offY : offZ);                                                      \
                        const float plane = *reinterpret_cast<const float *>(planeBytes + (uint32_t)((coord << 2) + off));          \
                        const float dd = sxm ? d.x : (sym ? d.y : d.z);                                                             \
                        const float oo = sxm ? o.x : (sym ? o.y : o.z);                                                             \
                        float nd;                                                                                                   \
                        if (FAST) {                                                                                                 \
                            const float rr = sxm ? rx : (sym ? ry : rz);                                                            \
                            nd = tame_quotient(plane - oo, dd, rr);                                                                 \
                        } else nd = (plane - oo) / dd;                                                                              \
                        dx = sxm ? nd : dx; dy = sym ? nd : dy; dz = (sxm | sym) ? dz : nd;                                         \
                    }                                                                                                               \
                    walkEnded = done;                                                                                               \
                }
            if (fastWave) {
                const float rx = refined_rcp(d.x), ry = refined_rcp(d.y), rz = refined_rcp(d.z);
#pragma unroll
                for (int u = 0; u < RT_WF_BLIND; ++u) { RT_WALK_STEP(true) }
            } else {
                const float rx = 0.f, ry = 0.f, rz = 0.f;
#pragma unroll
                for (int u = 0; u < RT_WF_BLIND; ++u) { RT_WALK_STEP(false) }
            }
#undef RT_WALK_STEP
            // look-up: which of the logged cells are occupied?  Keep those -- as their DENSE ids, which is what the test phase gathers
            // records by -- in path order, at the front of the list.  Only the
            // requested words stay in registers across the wait; the logged cells are read back from LDS on both sides of it
            // (what a word is needed for -- "the block differs from the one before" -- is recomputed the same way).
            uint3 lw[RT_WF_BLIND];
            {
                uint32_t prev = wordKey;
#pragma unroll
                for (int i = 0; i < RT_WF_BLIND; ++i) {
                    // EVERY lane loads: a lane that needs no word (its block is the one it has, or it logged fewer cells) reads the table's
                    // first entry -- one address for all of them, one access -- instead of sitting a branch out: no branch, and no
                    // registers to zero for the lanes that skipped it (six moves per slot; -2 % of the kernel)
                    const uint32_t key = cellList[listed + i][threadIdx.x] & 0xFCFCFCu;
                    const bool need = ((uint32_t)i < logged) & (key != prev);
                    lw[i] = *reinterpret_cast<const uint3 *>(blockTable + (size_t)(need ? key * 3u : 0u));
                    prev = need ? key : prev;
                }
            }
            const uint32_t listedBefore = listed;
#pragma unroll
            for (int i = 0; i < RT_WF_BLIND; ++i) {
                if ((uint32_t)i < logged) {
                    const uint32_t c = cellList[listedBefore + i][threadIdx.x]; // slot listedBefore + i >= listed: not overwritten yet
                    const uint32_t key = c & 0xFCFCFCu;
                    if (key != wordKey) { wordKey = key; wordLo = lw[i].x; wordHi = lw[i].y; wordRank = lw[i].z; }
                    // bit (cx&3) | (cy&3)<<2 | (cz&3)<<4 : gather the three 2-bit fields with one multiply
                    const uint32_t bit = (((c & 0x030303u) * 0x1041u) >> 12) & 63u;
                    const uint32_t half = (c & 0x20000u) ? wordHi : wordLo; // bit 5 of `bit` is bit 1 of cz
                    if (__builtin_amdgcn_ubfe(half, bit & 31u, 1u)) {
                        // the cell's dense id = rank of its block + occupied cells below it in the block: what the test phase gathers by
                        const uint32_t below = (1u << (bit & 31u)) - 1u;
                        const uint32_t inLo = __popc(wordLo & ((c & 0x20000u) ? 0xffffffffu : below));
                        cellList[listed][threadIdx.x] = wordRank + inLo + ((c & 0x20000u) ? __popc(wordHi & below) : 0u);
                        ++listed;
                    }
                }
            }
        }

#ifdef RT_DIAG_STAMPS
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const unsigned long long dgT0 = diag_stamp();
        dgWalk += dgT0 - dgW0;
        dgBatches++;
#endif
        // ---- test, wave-cooperative: items = recorded cells of the whole wave, one per lane and round.  The reference tests a
        // cell's candidates in list order against a running maximum that is reset per cell (:366-379) and takes the hit of the
        // EARLIEST cell that has one (:380).  A candidate's own t, l1, l2 do not depend on the running maximum -- it only picks
        // the smallest t, the earlier candidate on a tie -- so candidates are tested independently and an LDS atomicMin on
        // (cell order | t | pair index) finds the same winner (t > tmin >= 0: its bits order like the value; pair indices grow
        // in list order inside a cell).  Round one tests every cell's FIRST candidate (it sits at the cell's dense id and says
        // how many more there are and where), the further candidates of all cells are flattened into a second list and tested
        // one per lane afterwards (7 % of the cells of a fine scene have any: looping over them cell by cell ran 64-lane
        // instructions for a handful of lanes, behind two more dependent gathers).  The owner re-evaluates the winning pair.
        {
            // (the cached block entry stays across the test phase: a register or two more at the kernel's peak, one gather fewer per batch and lane)
            const uint32_t mineN = listed;
            uint32_t incl = mineN;
#pragma unroll
            for (int off = 1; off < 64; off <<= 1) {
                const uint32_t up = __shfl_up(incl, off, 64);
                if ((int)lane >= off) incl += up;
            }
            const uint32_t myBase = incl - mineN;
            const uint32_t items = __shfl(incl, 63, 64);
#ifdef RT_DIAG_STAMPS
            dgItems += items;
#endif
            if (items) { // wave-uniform
                // (the lists are addressed as the __shared__ arrays they are: through a generic `volatile` pointer every access became
                // a system-coherent FLAT instruction with a wait behind it.  One wave, in-order LDS: a wavefront-scope fence is all the
                // ordering the exchange between lanes needs, and it costs no instruction.)
                uint8_t (&owners)[RT_WF_LEAN_LIST * 64] = ownerOf[wave];
                unsigned long long (&keys)[64] = keyOf[wave];
                uint32_t (&moreWho)[RT_WF_MORE_ITEMS] = moreOf[wave][0], (&morePair)[RT_WF_MORE_ITEMS] = moreOf[wave][1];
                for (uint32_t j = 0; j < mineN; ++j) owners[myBase + j] = (uint8_t)lane;
                {   // (made here, not kept in a register pair across the walk: the compiler spilled the hoisted constant to scratch)
                    unsigned long long none = ~0ull;
                    asm volatile("" : "+v"(none));
                    keys[lane] = none;
                }
                if (lane == 0) moreCount[wave] = 0u;
                __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
                __builtin_amdgcn_wave_barrier(); // one wave: its LDS operations retire in order
                for (uint32_t c0 = 0; c0 < items; c0 += 64) {
                    const uint32_t c = c0 + lane;
                    const bool has = c < items;
                    const uint32_t owner = has ? owners[c] : 0u;
                    const uint32_t ownerBase = __shfl(myBase, owner, 64);
                    const V3 po = mk(__shfl(o.x, owner, 64), __shfl(o.y, owner, 64), __shfl(o.z, owner, 64));
                    const V3 pd = mk(__shfl(d.x, owner, 64), __shfl(d.y, owner, 64), __shfl(d.z, owner, 64));
                    const float ptmin = __shfl(tmin, owner, 64), ptmax = __shfl(tmax, owner, 64);
                    const uint32_t pexcl = __shfl(excluded, owner, 64);
                    if (has) {
                        const uint32_t j = c - ownerBase;
                        const uint32_t pc = cellList[j][(wave << 6) + owner];
                        const uint32_t dense = pc; // (made by the look-up that found the cell occupied)
                        const float4 *rec = reinterpret_cast<const float4 *>(S.pairRec) + 4 * (size_t)dense;
                        float4 r0 = rec[0], r1 = rec[1], r2 = rec[2], r3 = rec[3];
                        const uint32_t info = __float_as_uint(r1.w);
                        unsigned long long best = ~0ull;
                        float t, l1, l2;
                        if (pair_test_flat(r0, r1, r2, r3, po, pd, ptmin, ptmax, pexcl, t, l1, l2)) best = hit_key(j, t, dense);
                        uint32_t n = info & 15u;
                        if (n > 1u) { // further candidates: onto the second list; a crowded cell, or one the list has no room for, in place
                            uint32_t restAt = info >> 4, i = 0;
                            if (n < RT_PAIR_MANY) {
                                const uint32_t at = atomicAdd(&moreCount[wave], n - 1u);
                                for (; i + 1 < n && at + i < RT_WF_MORE_ITEMS; ++i) { moreWho[at + i] = owner | (j << 6); morePair[at + i] = restAt + i; }
                            }
                            if (i + 1 < n) {
                                rec = reinterpret_cast<const float4 *>(S.pairRec) + 4 * (size_t)(restAt + i);
                                r0 = rec[0]; r1 = rec[1]; r2 = rec[2]; r3 = rec[3];
                                if (n == RT_PAIR_MANY) n = __float_as_uint(r1.w); // (i == 0 here: the first further record has the exact count)
#pragma unroll 1
                                for (;;) {
                                    if (pair_test_flat(r0, r1, r2, r3, po, pd, ptmin, ptmax, pexcl, t, l1, l2)) best = min(best, hit_key(j, t, restAt + i));
                                    if (++i + 1 >= n) break;
                                    rec += 4;
                                    r0 = rec[0]; r1 = rec[1]; r2 = rec[2]; r3 = rec[3];
                                }
                            }
                        }
                        if (best != ~0ull) atomicMin(&keys[owner], best);
                    }
                }
                __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
                __builtin_amdgcn_wave_barrier();
                const uint32_t more = min(moreCount[wave], (uint32_t)RT_WF_MORE_ITEMS);
                for (uint32_t e0 = 0; e0 < more; e0 += 64) {
                    const uint32_t e = e0 + lane;
                    const bool has = e < more;
                    const uint32_t who = has ? moreWho[e] : 0u;
                    const uint32_t owner = who & 63u;
                    const V3 po = mk(__shfl(o.x, owner, 64), __shfl(o.y, owner, 64), __shfl(o.z, owner, 64));
                    const V3 pd = mk(__shfl(d.x, owner, 64), __shfl(d.y, owner, 64), __shfl(d.z, owner, 64));
                    const float ptmin = __shfl(tmin, owner, 64), ptmax = __shfl(tmax, owner, 64);
                    const uint32_t pexcl = __shfl(excluded, owner, 64);
                    if (has) {
                        const uint32_t pair = morePair[e];
                        const float4 *rec = reinterpret_cast<const float4 *>(S.pairRec) + 4 * (size_t)pair;
                        const float4 r0 = rec[0], r1 = rec[1], r2 = rec[2], r3 = rec[3];
                        float t, l1, l2;
                        if (pair_test_flat(r0, r1, r2, r3, po, pd, ptmin, ptmax, pexcl, t, l1, l2))
                            atomicMin(&keys[owner], hit_key(who >> 6, t, pair));
                    }
                }
                __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
                __builtin_amdgcn_wave_barrier();
                if (mineN) {
                    const unsigned long long key = keys[lane];
                    if (key != ~0ull) { // this segment's hit: the ray's answer is that of its lowest segment with one
                        atomicMin(&W.hitKey[par][q], ((unsigned long long)seg << 32) | (key & (unsigned long long)(2u * RT_PAIR_LIMIT - 1u)));
                        active = false;
                        walkEnded = true;
                    }
                }
            }
            listed = 0;
        }
#ifdef RT_DIAG_STAMPS
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        dgTest += diag_stamp() - dgT0;
#endif
        if (active && walkEnded) active = false; // walked to the end without a hit: hitKey[q] stays as it is
        if (spins > W.spinLimit || __ballot(active) == 0ull) break;
    }
    if (spins > W.spinLimit) { // the guard tripped: rays of this wave were abandoned -- tell the host, which fails the frame
        if (lane == 0) atomicOr(W.hostStatus + RT_WF_STATUS_ERROR, RT_WF_ERR_SPIN);
        break;
    }
#ifdef RT_DIAG_STAMPS
    if (lane == 0 && round == RT_DIAG_STAMPS) { // cycle anatomy of this wave in round RT_DIAG_STAMPS (scripts/diag_stamps.py); [5] = before the walk
        atomicAdd(&S.stats[0], diag_stamp() - dgStart); atomicAdd(&S.stats[1], dgWalk); atomicAdd(&S.stats[2], dgTest);
        atomicAdd(&S.stats[3], dgWalkIters); atomicAdd(&S.stats[4], dgBatches); atomicAdd(&S.stats[5], dgStart - dgK0);
        atomicAdd(&S.stats[6], 1ull); atomicAdd(&S.stats[7], dgItems);
    }
#endif
    } // passes
}

// ---- stage 4: samples -> u16 planes -----------------------------------------------------------------------------------
// One thread per pixel of the instance's tiles, row-major inside the tile (coalesced 2-byte stores).  Samples are added
// in order, each addend truncated on its own, saturating (raytrace_opencl.c:726-741); `first` starts from zero, later
// sample batches continue from the tile buffer.
__global__ __launch_bounds__(256) void wf_accum_kernel(const RtDevScene S, const RtWavefront W, const int first)
{
    const uint32_t localPixel = blockIdx.x * 256 + threadIdx.x;
    if (localPixel >= S.tileCount * RT_TILE_PIXELS) return;
    const uint32_t slot = localPixel / RT_TILE_PIXELS, inTile = localPixel % RT_TILE_PIXELS;
    const uint32_t tile = S.tileIds[slot];
    const uint32_t gx = (tile % S.tilesX) * RT_TILE + (inTile % RT_TILE), gy = (tile / S.tilesX) * RT_TILE + (inTile / RT_TILE);
    if (gx >= S.width || gy >= S.height) return;
    uint16_t *planes = S.tileBuf + (size_t)slot * 3 * RT_TILE_PIXELS + inTile;
    int r = 0, g = 0, b = 0;
    if (!first) { r = planes[0]; g = planes[RT_TILE_PIXELS]; b = planes[2 * RT_TILE_PIXELS]; }
    const float scale = (float)(0xFFFF) / (float)S.sampleCount; // :728
    for (uint32_t sb = 0; sb < W.samplesInBatch; ++sb) {
        const float4 c = W.sampleOut[localPixel * W.samplesInBatch + sb];
        r = sat_add_u16(r, c.x, scale);
        g = sat_add_u16(g, c.y, scale);
        b = sat_add_u16(b, c.z, scale);
    }
    planes[0] = (uint16_t)r;
    planes[RT_TILE_PIXELS] = (uint16_t)g;
    planes[2 * RT_TILE_PIXELS] = (uint16_t)b;
}

// ---- end of a batch's issued rounds: tell the host whether anybody is still waiting ------------------------------------------
// One workgroup.  The main queue slices of round `round` (what logic(round - 1) appended) hold one request per path that is not
// finished; their sum goes to the mapped host word, so the host can issue a frame's rounds without looking at the queue in
// between and check afterwards (rt_api.cpp).
__global__ __launch_bounds__(256) void wf_status_kernel(const RtWavefront W, const uint32_t round)
{
    __shared__ uint32_t part[4];
    uint32_t n = W.ctl[(round % 3) * RT_WF_CTL_WORDS + RT_WF_CTL_COUNTS + threadIdx.x]; // RT_WF_SHARDS == 256 main slices
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) n += __shfl_xor(n, off, 64);
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = n;
    __syncthreads();
    if (threadIdx.x == 0) {
        const uint32_t waiting = part[0] + part[1] + part[2] + part[3];
        if (waiting) atomicAdd(W.hostStatus + RT_WF_STATUS_WAITING, waiting);
        atomicAdd(W.hostStatus + RT_WF_STATUS_BATCHES, 1u);
    }
    // nothing of the batch is in flight any more: leave all three sets of control words zeroed for the next one (the host skips its memset --
    // a launch of the runtime's fill kernel with ~6 us of idle time in front of it -- when the batch before ended here)
    for (uint32_t i = threadIdx.x; i < 3u * RT_WF_CTL_WORDS; i += 256u) W.ctl[i] = 0u;
}

// ---- launch wrappers ------------------------------------------------------------------------------------------------
extern "C" hipError_t rtw_launch_primary(const RtDevScene *scene, const RtWavefront *wf, hipStream_t stream)
{
    if (scene->tileCount == 0) return hipSuccess;
    hipLaunchKernelGGL(wf_primary_kernel, dim3(scene->tileCount * 64, wf->samplesInBatch), dim3(256), 0, stream, *scene, *wf);
    return hipGetLastError();
}

static bool mode_ok(const RtRoundMode &m)
{
    return m.slices >= 1u && m.slices <= RT_WF_SHARDS && (m.slices & (m.slices - 1u)) == 0u && m.segLen >= 1u && m.groupRays >= 1u && m.groupRays <= RT_WF_GROUP_RAYS;
}

// slicesIn: the queue slices of the round this launch consumes; next: the layout of the round it spawns (next.slices <= slicesIn)
extern "C" hipError_t rtw_launch_logic(const RtDevScene *scene, const RtWavefront *wf, uint32_t round, uint32_t blocks, uint32_t slicesIn, const RtRoundMode *next, hipStream_t stream)
{
    if (blocks % (RT_WF_SHARDS / 4) != 0) return hipErrorInvalidValue; // a whole number of waves per queue slice (wf_logic_kernel)
    if (!mode_ok(*next) || slicesIn < next->slices || slicesIn > RT_WF_SHARDS || (slicesIn & (slicesIn - 1u)) != 0u) return hipErrorInvalidValue;
    if (round == 0u && next->ordered) hipLaunchKernelGGL((wf_logic_kernel<true, true>), dim3(blocks), dim3(256), 0, stream, *scene, *wf, round, slicesIn, *next);
    else if (round == 0u) hipLaunchKernelGGL((wf_logic_kernel<true, false>), dim3(blocks), dim3(256), 0, stream, *scene, *wf, round, slicesIn, *next);
    else if (next->ordered) hipLaunchKernelGGL((wf_logic_kernel<false, true>), dim3(blocks), dim3(256), 0, stream, *scene, *wf, round, slicesIn, *next);
    else hipLaunchKernelGGL((wf_logic_kernel<false, false>), dim3(blocks), dim3(256), 0, stream, *scene, *wf, round, slicesIn, *next);
    return hipGetLastError();
}

// an ordered round: ranks -> positions (fixed grid, the kernel strides over the blocks that are in use)
extern "C" hipError_t rtw_launch_scatter(const RtWavefront *wf, uint32_t round, uint32_t blocks, const RtRoundMode *mode, hipStream_t stream)
{
    if (!mode_ok(*mode)) return hipErrorInvalidValue;
    hipLaunchKernelGGL(wf_scatter_kernel, dim3(blocks), dim3(256), 0, stream, *wf, round, mode->slices);
    return hipGetLastError();
}

// one workgroup per 256 entries (a watched frame's grid is sized for the worst case; surplus workgroups exit at once)
extern "C" hipError_t rtw_launch_trace(const RtDevScene *scene, const RtWavefront *wf, uint32_t round, uint32_t blocks, const RtRoundMode *mode, hipStream_t stream)
{
    if (!mode_ok(*mode)) return hipErrorInvalidValue;
    if (mode->ordered) hipLaunchKernelGGL(wf_trace_kernel<true>, dim3(blocks), dim3(256), 0, stream, *scene, *wf, round, *mode);
    else hipLaunchKernelGGL(wf_trace_kernel<false>, dim3(blocks), dim3(256), 0, stream, *scene, *wf, round, *mode);
    return hipGetLastError();
}

extern "C" hipError_t rtw_launch_status(const RtWavefront *wf, uint32_t round, hipStream_t stream)
{
    static_assert(RT_WF_SHARDS == 256, "wf_status_kernel sums one main queue slice per thread");
    hipLaunchKernelGGL(wf_status_kernel, dim3(1), dim3(256), 0, stream, *wf, round);
    return hipGetLastError();
}

extern "C" hipError_t rtw_launch_accum(const RtDevScene *scene, const RtWavefront *wf, int first, hipStream_t stream)
{
    const uint32_t n = scene->tileCount * RT_TILE_PIXELS;
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL(wf_accum_kernel, dim3((n + 255) / 256), dim3(256), 0, stream, *scene, *wf, first);
    return hipGetLastError();
}
