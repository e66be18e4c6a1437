// rt_kernels.hip -- hand-written HIP kernels for gfx950 (MI355X): the ray-trace + shade hot path.
//
// What is computed is what the reference's kernel computes (source/opencl/raytrace_opencl.c:406-742, seeded and
// sequenced like its deterministic C path, source/opencl/raytrace.c:612-653); how it is computed is not:
//   * one launch renders every sample of every pixel of a tile set (the reference launches tiles x samples,
//     raytrace.c:535-556); a 256-thread workgroup owns a 16x16 pixel patch, each wave an 8x8 quadrant, so the
//     lanes of a wave share candidate lists, grid cells and cache lines;
//   * candidate tests read one pre-resolved 64-byte triangle record (rt_device.h) instead of gathering
//     16 B index + 3 x 16 B vertices and re-deriving edges, normal and the barycentric denominator per test;
//   * the 3-D DDA keeps the three plane distances in registers and re-divides only the axis it stepped
//     (one IEEE divide per cell instead of three; same expression, same bits);
//   * split planes (3 x 257 floats) and the texel/255 table live in LDS;
//   * the 12-slot ray queue is touched only on push/pop; the ray in flight stays in registers.
//
// Bit-exactness contract: fp32 everywhere with no contraction (build: -ffp-contract=off), IEEE divide and sqrt
// (-fhip-fp32-correctly-rounded-divide-sqrt), fp32 denormals kept (hipcc default).  The reference's C path
// detours through double for sqrt/floor/modf/fabs/sin/cos/pow:
//   sqrt  : (float)sqrt((double)x) == correctly rounded sqrtf(x) (53 >= 2*24+2 bits: no double-rounding error)
//   floor/fabs : exact in either width
//   modf  : done in double here too (positive_modf needs the 53-bit sum, see pos_modf)
//   sin/cos (bump) and the distant-light spread: host-libm tables / per-light constants uploaded with the scene
//   pow(0.5, x): exp2(-x) in double on the device (exact for the reference's own scenes where x is 0 or NaN)
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "rt_device.h"

namespace {

struct V3 { float x, y, z; };

__device__ __forceinline__ V3 mk(float x, float y, float z) { V3 r; r.x = x; r.y = y; r.z = z; return r; }
__device__ __forceinline__ V3 ld3(const float *p) { return mk(p[0], p[1], p[2]); }
// raytrace.c:18-20: (a0*b0 + a1*b1) + a2*b2
__device__ __forceinline__ float dot3(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
// raytrace.c:21-27
__device__ __forceinline__ V3 cross3(V3 a, V3 b) { return mk(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x); }
__device__ __forceinline__ V3 sub3(V3 a, V3 b) { return mk(a.x - b.x, a.y - b.y, a.z - b.z); }
__device__ __forceinline__ V3 along(V3 o, float t, V3 d) { return mk(o.x + t * d.x, o.y + t * d.y, o.z + t * d.z); }
__device__ __forceinline__ float sqrt_rn(float v) { return __builtin_sqrtf(v); }

#define RT_PI_F 3.14159265f // raytrace.h:33
#define RT_INF __builtin_inff()
#define RT_NONE 0xffffffffu

enum { CH_COLOR = 0, CH_REFLECTION = 1, CH_TRANSPARENCY = 2, CH_BUMP = 3, CH_LUMINANCE = 4, CH_COUNT = 5 }; // raytrace_opencl.h:14-22
enum { ST_SAMPLES = 0, ST_PCAND, ST_GRAYS, ST_GCELLS, ST_GCAND, ST_HITS, ST_TEXELS, ST_COUNT };

struct Counters { uint32_t v[ST_COUNT]; };

// ---- PRNG (raytrace_opencl.c:1-23) ---------------------------------------------------------------------------
__device__ __forceinline__ uint64_t rol64(uint64_t v, int n) { return (v << n) | (v >> (64 - n)); }
__device__ __forceinline__ uint64_t xs64star(uint64_t v)
{
    v ^= v >> 12;
    v ^= v << 25;
    v ^= v >> 27;
    return v * 2685821657736338717ULL;
}
// returns the raw [0,1] fraction: (float)((double)s / 2^64)
__device__ __forceinline__ float rand_unit(uint64_t &s)
{
    s ^= xs64star((rol64(s, 55) ^ rol64(s, 3)) * 0xc23f3c0ad9da6357ULL);
    s ^= xs64star((rol64(s, 35) ^ rol64(s, 3)) ^ 0xce84d6af03c16b89ULL);
    s ^= xs64star((rol64(s, 63) ^ rol64(s, 35)) * 0xf097ef8bbe03ddccULL);
    s ^= xs64star((rol64(s, 41) ^ rol64(s, 12)) ^ 0x48302294fbfe30bfULL);
    s ^= xs64star((rol64(s, 1) ^ rol64(s, 62)) * 0x79e7425e3f4f147dULL);
    s ^= xs64star((rol64(s, 42) ^ rol64(s, 29)) ^ 0x14d1d30856e5be9aULL);
    s ^= xs64star((rol64(s, 47) ^ rol64(s, 45)) * 0x24289d47a66617c3ULL);
    s ^= xs64star((rol64(s, 39) ^ rol64(s, 6)) ^ 0x5576fb2f80a05d14ULL);
    // u64 -> f64 (RN); the divide by (double)0xFFFFFFFFFFFFFFFF == 2^64 is an exact scaling; f64 -> f32 (RN)
    return __double2float_rn(__ull2double_rn(s) * 0x1p-64);
}
// randF(min,max) = min + (max-min)*u (:22): for (0,1) that is 0.f + 1.f*u, for (-1,1) it is -1.f + 2.f*u
__device__ __forceinline__ float rand01(uint64_t &s) { return 0.f + 1.f * rand_unit(s); }
__device__ __forceinline__ float rand11(uint64_t &s) { return -1.f + 2.f * rand_unit(s); }

// raytrace_opencl.c:30-45
__device__ V3 sphere_point(uint64_t &s, float radius)
{
    V3 p;
    float len;
    do {
        p.x = rand11(s);
        p.y = rand11(s);
        p.z = rand11(s);
        len = sqrt_rn(dot3(p, p));
    } while (len <= 0.f);
    float scale = sqrt_rn(rand01(s)) * radius / len;
    return mk(scale * p.x, scale * p.y, scale * p.z);
}

// raytrace_opencl.c:25-28.  Must stay in double: frac + 1.0 needs up to 53 bits (a tiny negative frac gives
// 1 - 2^-k, which survives the second modf and only then rounds to 1.0f).
__device__ __forceinline__ float pos_modf(float v)
{
    double ip;
    double f = modf((double)v, &ip);
    f = modf(f + 1.0, &ip);
    return __double2float_rn(f);
}

// ---- triangle test against a pre-resolved record (raytrace_opencl.c:124-172) ------------------------------------
// t is always produced; l1/l2 only when tmin < t < tmax (as in the reference).
__device__ __forceinline__ bool tri_test(const float *__restrict__ triRec, uint32_t tri, V3 o, V3 d, float tmin, float tmax,
                                         float &t, float &l1, float &l2)
{
    const float4 *rec = reinterpret_cast<const float4 *>(triRec) + 4 * (size_t)tri;
    const float4 r0 = rec[0], r1 = rec[1], r2 = rec[2];
    const V3 a = mk(r0.x, r0.y, r0.z), ab = mk(r0.w, r1.x, r1.y), ac = mk(r1.z, r1.w, r2.x), n = mk(r2.y, r2.z, r2.w);
    const V3 ao = sub3(o, a);
    t = -dot3(n, ao) / dot3(n, d);
    bool hit = false;
    if (tmin < t && t < tmax) {
        const float4 r3 = rec[3]; // abab abac acac inv
        const V3 ap = sub3(along(o, t, d), a);
        const float ap_ab = dot3(ap, ab);
        const float ap_ac = dot3(ap, ac);
        l1 = (r3.y * ap_ac - r3.z * ap_ab) * r3.w;
        l2 = (r3.y * ap_ab - r3.x * ap_ac) * r3.w;
        hit = (0 <= l1 && 0 <= l2 && l1 + l2 <= 1.f);
    }
    return hit;
}

// raytrace_opencl.c:83-101
__device__ __forceinline__ float point_line_sq(V3 o, V3 e, V3 p)
{
    V3 oe = sub3(e, o);
    float oe_sq = dot3(oe, oe);
    V3 op = sub3(p, o);
    float k = dot3(op, oe) / oe_sq;
    V3 foot = along(o, k, oe);
    V3 dd = sub3(foot, p);
    return dot3(dd, dd);
}

// Workgroup-shared tables.
struct Shared {
    float planes[3][RT_GRID_DIV + 1]; // split planes per axis
    float unit255[256];               // i / 255.f
};

// raytrace_opencl.c:174-193 on the LDS copy: strict '<'
__device__ __forceinline__ void box_address(const Shared &sh, V3 p, int &cx, int &cy, int &cz)
{
    cx = 0; cy = 0; cz = 0;
#pragma unroll
    for (int div = RT_GRID_DIV / 2; div >= 1; div /= 2) {
        if (sh.planes[0][cx + div] < p.x) cx += div;
        if (sh.planes[1][cy + div] < p.y) cy += div;
        if (sh.planes[2][cz + div] < p.z) cz += div;
    }
}

// raytrace_opencl.c:265-322 (the boolean result is ignored by both callers, :354,:360, so none is returned;
// the early exits still stop the remaining clamps)
__device__ __forceinline__ void bind_in_cube(V3 &p, V3 d, V3 lo, V3 hi)
{
    float t;
    if (p.x < lo.x) { if (d.x <= 0) return; t = (lo.x - p.x) / d.x; p.x += t * d.x; p.y += t * d.y; p.z += t * d.z; }
    if (hi.x < p.x) { if (0 <= d.x) return; t = (hi.x - p.x) / d.x; p.x += t * d.x; p.y += t * d.y; p.z += t * d.z; }
    if (p.y < lo.y) { if (d.y <= 0) return; t = (lo.y - p.y) / d.y; p.x += t * d.x; p.y += t * d.y; p.z += t * d.z; }
    if (hi.y < p.y) { if (0 <= d.y) return; t = (hi.y - p.y) / d.y; p.x += t * d.x; p.y += t * d.y; p.z += t * d.z; }
    if (p.z < lo.z) { if (d.z <= 0) return; t = (lo.z - p.z) / d.z; p.x += t * d.x; p.y += t * d.y; p.z += t * d.z; }
    if (hi.z < p.z) { if (0 <= d.z) return; t = (hi.z - p.z) / d.z; p.x += t * d.x; p.y += t * d.y; p.z += t * d.z; }
}

// ---- secondary rays: 3-D DDA over the non-uniform grid (raytrace_opencl.c:324-401) ------------------------------
template <bool COUNT>
__device__ uint32_t grid_trace(const RtDevScene &S, const Shared &sh, V3 o, V3 d, float tmin, float tmax, uint32_t excluded,
                               float &t_out, float &l1_out, float &l2_out, Counters &cn)
{
    const V3 lo = mk(sh.planes[0][0], sh.planes[1][0], sh.planes[2][0]);
    const V3 hi = mk(sh.planes[0][RT_GRID_DIV], sh.planes[1][RT_GRID_DIV], sh.planes[2][RT_GRID_DIV]);
    uint32_t best = RT_NONE;
    int cx, cy, cz, ex = -1, ey = -1, ez = -1;
    V3 from = along(o, tmin, d);
    bind_in_cube(from, d, lo, hi);
    box_address(sh, from, cx, cy, cz);
    if (tmax < RT_INF) {
        V3 to = along(o, tmax, d);
        bind_in_cube(to, d, lo, hi);
        box_address(sh, to, ex, ey, ez);
    }
    // plane selectors and step directions are fixed per ray (:383-398)
    const int px = (0 <= d.x) ? 1 : 0, py = (0 <= d.y) ? 1 : 0, pz = (0 <= d.z) ? 1 : 0;
    const int sx = px ? 1 : -1, sy = py ? 1 : -1, sz = pz ? 1 : -1;
    // distances are measured from the ray origin (:383-385); each depends only on its own axis' cell index,
    // so only the axis that stepped is re-divided
    float dx = (sh.planes[0][cx + px] - o.x) / d.x;
    float dy = (sh.planes[1][cy + py] - o.y) / d.y;
    float dz = (sh.planes[2][cz + pz] - o.z) / d.z;
    if (COUNT) cn.v[ST_GRAYS]++;
    // occupancy word of the 4x4x4 block the walk is in; reloaded only when the walk leaves the block
    uint32_t wordAt = (uint32_t)((cx >> 2) + 64 * (cy >> 2) + 4096 * (cz >> 2));
    unsigned long long word = S.gridBits[wordAt];
    for (;;) {
        float tbest = tmax; // reset per cell (:366)
        if (COUNT) cn.v[ST_GCELLS]++;
        if ((word >> ((cx & 3) | ((cy & 3) << 2) | ((cz & 3) << 4))) & 1ull) {
            const uint32_t id = (uint32_t)(cx + RT_GRID_DIV * cy + RT_GRID_DIV * RT_GRID_DIV * cz);
            const uint32_t first = S.gridStart[id], last = S.gridStart[id + 1];
            if (COUNT) cn.v[ST_GCAND] += last - first;
            for (uint32_t i = first; i < last; ++i) {
                const uint32_t tri = S.gridList[i];
                if (excluded != tri) {
                    float t, l1, l2;
                    if (tri_test(S.triRec, tri, o, d, tmin, tbest, t, l1, l2)) {
                        best = tri; tbest = t; l1_out = l1; l2_out = l2;
                    }
                }
            }
        }
        t_out = tbest;
        // first cell with any hit ends the walk, as does the end cell (:380-381)
        if (best != RT_NONE || (cx == ex && cy == ey && cz == ez)) break;
        if ((dx < dy) & (dx < dz)) {
            cx += sx;
            if (cx < 0 || RT_GRID_DIV <= cx) break;
            dx = (sh.planes[0][cx + px] - o.x) / d.x;
        } else if (dy < dz) {
            cy += sy;
            if (cy < 0 || RT_GRID_DIV <= cy) break;
            dy = (sh.planes[1][cy + py] - o.y) / d.y;
        } else {
            cz += sz;
            if (cz < 0 || RT_GRID_DIV <= cz) break;
            dz = (sh.planes[2][cz + pz] - o.z) / d.z;
        }
        const uint32_t at = (uint32_t)((cx >> 2) + 64 * (cy >> 2) + 4096 * (cz >> 2));
        if (at != wordAt) { wordAt = at; word = S.gridBits[at]; }
    }
    return best;
}

// ---- texture fetch (raytrace_opencl.c:103-122) --------------------------------------------------------------
// Returns the three channel BYTES' unit values; `raw` receives the red byte (bump height index).
template <bool COUNT>
__device__ __forceinline__ V3 texel(const RtDevScene &S, const Shared &sh, int start, uint32_t w, uint32_t h, const float *uv,
                                    float l1, float l2, uint32_t &raw, Counters &cn)
{
    const float pu = pos_modf(uv[0] + (uv[2] - uv[0]) * l1 + (uv[4] - uv[0]) * l2);
    const float pv = pos_modf(uv[1] + (uv[3] - uv[1]) * l1 + (uv[5] - uv[1]) * l2);
    const float lx = pu * (float)(w - 1u);
    const float ly = pv * (float)(h - 1u);
    const int fx = (int)__builtin_floorf(lx);
    const int fy = (int)__builtin_floorf(ly);
    const int at = (int)((uint32_t)fx + (uint32_t)fy * w);
    long long idx = (long long)start + (long long)at;
    // in-range for every finite uv (pu,pv in [0,1]); the clamp only guards the reference's own UB (NaN uv)
    if (idx < 0) idx = 0;
    if (idx >= (long long)S.texelCount) idx = (long long)S.texelCount - 1;
    const uchar4 px = reinterpret_cast<const uchar4 *>(S.textures)[idx];
    if (COUNT) cn.v[ST_TEXELS]++;
    raw = px.x;
    return mk(sh.unit255[px.x], sh.unit255[px.y], sh.unit255[px.z]);
}

// ---- shading normal (raytrace_opencl.c:195-263) -----------------------------------------------------------------
template <bool COUNT>
__device__ V3 shading_normal(const RtDevScene &S, const Shared &sh, V3 where, V3 ray_o, V3 ray_d, uint32_t tri, float l1, float l2,
                             const float *shade, int m, Counters &cn)
{
    const float4 *rec = reinterpret_cast<const float4 *>(S.triRec) + 4 * (size_t)tri;
    const float4 r0 = rec[0];
    const V3 a = mk(r0.x, r0.y, r0.z);
    const V3 b = ld3(shade + 0), c = ld3(shade + 3);
    const V3 na = ld3(shade + 6), nb = ld3(shade + 9), nc = ld3(shade + 12);
    const float dab = sqrt_rn(point_line_sq(a, b, where));
    const float dbc = sqrt_rn(point_line_sq(b, c, where));
    const float dca = sqrt_rn(point_line_sq(c, a, where));
    const float inv = 1.f / (dab + dbc + dca);
    V3 n;
    n.x = (dab * nc.x + dbc * na.x + dca * nb.x) * inv;
    n.y = (dab * nc.y + dbc * na.y + dca * nb.y) * inv;
    n.z = (dab * nc.z + dbc * na.z + dca * nb.z) * inv;

    if (0 <= m) {
        const uint32_t bw = S.matSize[2 * (CH_COUNT * m + CH_BUMP)];
        if (0 < bw) {
            const uint32_t bh = S.matSize[2 * (CH_COUNT * m + CH_BUMP) + 1];
            const int bstart = S.matStart[CH_COUNT * m + CH_BUMP];
            const float *uv = shade + 15;
            const V3 tb = ld3(S.tb), lr = ld3(S.lr);
            uint32_t h0, hs, he;
            float t, p1 = 0.f, p2 = 0.f;
            (void)texel<COUNT>(S, sh, bstart, bw, bh, uv, l1, l2, h0, cn);
            tri_test(S.triRec, tri, ray_o, mk(ray_d.x + tb.x, ray_d.y + tb.y, ray_d.z + tb.z), 0.f, RT_INF, t, p1, p2);
            (void)texel<COUNT>(S, sh, bstart, bw, bh, uv, p1, p2, hs, cn);
            tri_test(S.triRec, tri, ray_o, mk(ray_d.x + lr.x, ray_d.y + lr.y, ray_d.z + lr.z), 0.f, RT_INF, t, p1, p2);
            (void)texel<COUNT>(S, sh, bstart, bw, bh, uv, p1, p2, he, cn);
            // xPart = (float)sin((hE-h0)*PI_F/2), etc. (:251-253) depend only on the two height bytes: host-libm tables
            const float xp = S.bumpSin[(he << 8) | h0];
            const float yp = S.bumpSin[(hs << 8) | h0];
            const float np = S.bumpCos[(he << 8) | h0] * S.bumpCos[(hs << 8) | h0];
            n.x = np * n.x / S.pixelSizeInv + xp * lr.x + yp * tb.x;
            n.y = np * n.y / S.pixelSizeInv + xp * lr.y + yp * tb.y;
            n.z = np * n.z / S.pixelSizeInv + xp * lr.z + yp * tb.z;
            const float li = 1.f / sqrt_rn(dot3(n, n));
            n.x *= li; n.y *= li; n.z *= li;
        }
    }
    return n;
}

// One queued ray (raytrace_opencl.c:461-468).  maxDistance of every queued ray is INFINITY (:680,:702,:719), so it is
// not stored.
struct QRay {
    V3 o, d, w;
    float tmin;
    uint32_t excluded;
    int bounces;
    int fromCamera;
};

// x86-64 cvttss2si semantics (what the reference binary computes at :729-737): NaN / out of range -> INT_MIN
__device__ __forceinline__ int trunc_x86(float v)
{
    return (v > -2147483904.0f && v < 2147483648.0f) ? (int)v : (int)0x80000000;
}

__device__ __forceinline__ int sat_add_u16(int plane, float colour, float scale)
{
    int v = plane + trunc_x86(colour * scale);
    if (v < 0) v = 0;
    if (0xFFFF < v) v = 0xFFFF;
    return v;
}

#define RT_MAX2(a, b) (((a) > (b)) ? (a) : (b)) /* raytrace.h:30 */

// One sample of one pixel (raytrace_opencl.c:470-725); returns the sample's colour.
template <bool COUNT>
__device__ V3 trace_sample(const RtDevScene &S, const Shared &sh, uint32_t pixel, uint32_t localPixel, float fpx, float fpy,
                           uint32_t sampleId, Counters &cn)
{
    QRay ring[RT_RING];
    uint64_t rng = (uint64_t)pixel * (uint64_t)S.sampleCount + (uint64_t)sampleId; // :481
    const V3 lr = ld3(S.lr), tb = ld3(S.tb);
    V3 out = mk(0.f, 0.f, 0.f);

    // primary ray (:490-508): LR jitter, then TB; not normalised
    QRay cur;
    cur.bounces = 12;
    cur.excluded = RT_NONE;
    cur.o = ld3(S.eye);
    cur.d = ld3(S.topLeft);
    float k = fpx + rand01(rng);
    cur.d.x += lr.x * k; cur.d.y += lr.y * k; cur.d.z += lr.z * k;
    k = fpy + rand01(rng);
    cur.d.x += tb.x * k; cur.d.y += tb.y * k; cur.d.z += tb.z * k;
    cur.w = mk(1.f, 1.f, 1.f);
    cur.fromCamera = 1;
    cur.tmin = 0.f;
    float cur_tmax = RT_INF;
    int head = 0, tail = 1;
    if (COUNT) cn.v[ST_SAMPLES]++;

    for (;;) {
        float hit_t = cur_tmax;
        uint32_t hit_tri = RT_NONE;
        float hit_l1 = 0.f, hit_l2 = 0.f;

        if (cur.fromCamera) {
            // per-pixel candidate list, running max, ties keep the earliest entry (:514-528)
            const uint32_t first = S.camStart[localPixel], last = S.camEnd[localPixel];
            if (COUNT) cn.v[ST_PCAND] += (first < last) ? last - first : 0u;
            for (uint32_t i = first; i < last; ++i) {
                const uint32_t tri = S.camList[i];
                if (cur.excluded != tri) {
                    float t, l1, l2;
                    if (tri_test(S.triRec, tri, cur.o, cur.d, cur.tmin, hit_t, t, l1, l2)) {
                        hit_t = t; hit_tri = tri; hit_l1 = l1; hit_l2 = l2;
                    }
                }
            }
        } else {
            hit_tri = grid_trace<COUNT>(S, sh, cur.o, cur.d, cur.tmin, cur_tmax, cur.excluded, hit_t, hit_l1, hit_l2, cn);
        }

        if (hit_tri != RT_NONE) {
            const float *shade = S.triShade + 24 * (size_t)hit_tri;
            const int m = __float_as_int(shade[21]);
            const float *uv = shade + 15;
            V3 tex = mk(0, 0, 0), transp = mk(0, 0, 0), refl = mk(0, 0, 0), lum = mk(0, 0, 0);
            V3 face0 = mk(0.1f, 0.1f, 0.1f), face1 = mk(0.1f, 0.1f, 0.1f); // ambient floor (:540)
            const V3 where = along(cur.o, hit_t, cur.d);
            const V3 n = shading_normal<COUNT>(S, sh, where, cur.o, cur.d, hit_tri, hit_l1, hit_l2, shade, m, cn);
            if (COUNT) cn.v[ST_HITS]++;

            if (0 <= m) { // :550-561
                const int mc = CH_COUNT * m;
                uint32_t raw;
                uint32_t w;
                w = S.matSize[2 * (mc + CH_COLOR)];
                if (0 < w) tex = texel<COUNT>(S, sh, S.matStart[mc + CH_COLOR], w, S.matSize[2 * (mc + CH_COLOR) + 1], uv, hit_l1, hit_l2, raw, cn);
                w = S.matSize[2 * (mc + CH_TRANSPARENCY)];
                if (0 < w) transp = texel<COUNT>(S, sh, S.matStart[mc + CH_TRANSPARENCY], w, S.matSize[2 * (mc + CH_TRANSPARENCY) + 1], uv, hit_l1, hit_l2, raw, cn);
                w = S.matSize[2 * (mc + CH_REFLECTION)];
                if (0 < w) refl = texel<COUNT>(S, sh, S.matStart[mc + CH_REFLECTION], w, S.matSize[2 * (mc + CH_REFLECTION) + 1], uv, hit_l1, hit_l2, raw, cn);
                w = S.matSize[2 * (mc + CH_LUMINANCE)];
                if (0 < w) lum = texel<COUNT>(S, sh, S.matStart[mc + CH_LUMINANCE], w, S.matSize[2 * (mc + CH_LUMINANCE) + 1], uv, hit_l1, hit_l2, raw, cn);
            }

            for (uint32_t j = 0; j < S.lightCount; ++j) { // :563-637
                V3 toL = mk(0.f, 0.f, 0.f), atten = mk(1.f, 1.f, 1.f);
                float lmin = 0.f, lmax = 0.f;
                const int type = S.lightType[j];
                if (type == 1 || type == 2 || type == 7 || type == 8 || type == 9) { // SPOT SPOTRECT TUBE AREA PHOTOMETRIC (:567-584)
                    const V3 r = sphere_point(rng, S.lightRadius[j]);
                    const float *lp = S.lightPos + 4 * j;
                    toL.x = r.x + lp[0] - where.x;
                    toL.y = r.y + lp[1] - where.y;
                    toL.z = r.z + lp[2] - where.z;
                    lmax = sqrt_rn(dot3(toL, toL));
                    const float inv = 1.f / lmax;
                    toL.x *= inv; toL.y *= inv; toL.z *= inv;
                } else if (type >= 3 && type <= 6) { // DISTANT PARALLEL PARSPOT PARSPOTRECT (:589-606)
                    const float *ld = S.lightDir + 4 * j;
                    toL = sphere_point(rng, S.lightSpread[j]);
                    toL.x -= ld[0]; toL.y -= ld[1]; toL.z -= ld[2];
                    const float inv = 1.f / sqrt_rn(dot3(toL, toL));
                    toL.x *= inv; toL.y *= inv; toL.z *= inv;
                    lmax = RT_INF;
                } // OMNI (and unknown types): no shadow ray, zero direction (:585-588)

                if (lmin < lmax) { // shadow ray through transparent occluders (:608-627)
                    for (;;) {
                        float t, l1, l2;
                        const uint32_t occ = grid_trace<COUNT>(S, sh, where, toL, lmin, lmax, hit_tri, t, l1, l2, cn);
                        if (occ == RT_NONE) break;
                        const float *oshade = S.triShade + 24 * (size_t)occ;
                        const int om = __float_as_int(oshade[21]);
                        V3 tr = mk(0.f, 0.f, 0.f);
                        if (0 <= om) {
                            const uint32_t w = S.matSize[2 * (CH_COUNT * om + CH_TRANSPARENCY)];
                            uint32_t raw;
                            if (0 < w) tr = texel<COUNT>(S, sh, S.matStart[CH_COUNT * om + CH_TRANSPARENCY], w, S.matSize[2 * (CH_COUNT * om + CH_TRANSPARENCY) + 1], oshade + 15, l1, l2, raw, cn);
                        }
                        atten.x *= tr.x; atten.y *= tr.y; atten.z *= tr.z;
                        if (!(0.f < atten.x && 0.f < atten.y && 0.f < atten.z)) break;
                        lmin = t;
                    }
                }
                { // two-sided |N.L| with distance falloff, screen-blended into the facing side (:628-636)
                    const float ndl = dot3(n, toL);
                    const float mag = __builtin_fabsf(ndl);
                    const float x = lmax / S.lightHalfAtt[j];
                    // (float)pow(0.5f, x) in double: 0.5^x = 2^-x
                    const float fall = __double2float_rn(exp2(-(double)x));
                    const float e = mag * (fall == fall ? fall : 1.f);
                    const float *lc = S.lightCol + 4 * j;
                    if (0.f <= ndl) {
                        face1.x += (1.f - face1.x) * atten.x * e * lc[0];
                        face1.y += (1.f - face1.y) * atten.y * e * lc[1];
                        face1.z += (1.f - face1.z) * atten.z * e * lc[2];
                    } else {
                        face0.x += (1.f - face0.x) * atten.x * e * lc[0];
                        face0.y += (1.f - face0.y) * atten.y * e * lc[1];
                        face0.z += (1.f - face0.z) * atten.z * e * lc[2];
                    }
                }
            }

            // emission, then lit diffuse (:639-653)
            out.x += (1.f - out.x) * lum.x * cur.w.x;
            out.y += (1.f - out.y) * lum.y * cur.w.y;
            out.z += (1.f - out.z) * lum.z * cur.w.z;
            const int front = (dot3(n, cur.d) <= 0.f) ? 1 : 0;
            const V3 lit = front ? face1 : face0;
            out.x += (1.f - out.x) * cur.w.x * (1.f - transp.x) * tex.x * lit.x;
            out.y += (1.f - out.y) * cur.w.y * (1.f - transp.y) * tex.y * lit.y;
            out.z += (1.f - out.z) * cur.w.z * (1.f - transp.z) * tex.z * lit.z;

            if (cur.bounces > 0) { // :656
                const float total = RT_MAX2(RT_MAX2(refl.x + transp.x, refl.y + transp.y), refl.z + transp.z);
                const float dif = (total < 1.f) ? 1.f - total : 0.f;
                bool open = true; // false once the ring-full check fires (the reference `continue`s, :682,:704,:721)
                V3 w;
                // diffuse bounce (:664-683)
                w = mk(cur.w.x * tex.x * dif, cur.w.y * tex.y * dif, cur.w.z * tex.z * dif);
                if (3.f / 256.f <= w.x + w.y + w.z) {
                    QRay q;
                    q.bounces = 0;
                    q.excluded = hit_tri;
                    q.o = where;
                    q.d = sphere_point(rng, 1.f);
                    if (front != ((0 <= dot3(q.d, n)) ? 1 : 0)) { q.d.x = -q.d.x; q.d.y = -q.d.y; q.d.z = -q.d.z; }
                    q.w = w;
                    q.fromCamera = 0;
                    q.tmin = 0.f;
                    ring[tail] = q;
                    tail = (tail + 1) % RT_RING;
                    if ((tail + 1) % RT_RING == head) open = false;
                }
                // mirror (:686-705)
                if (open) {
                    w = mk(cur.w.x * tex.x * refl.x, cur.w.y * tex.y * refl.y, cur.w.z * tex.z * refl.z);
                    if (3.f / 256.f <= w.x + w.y + w.z) {
                        QRay q;
                        const float two = -2.f * dot3(n, cur.d);
                        q.bounces = cur.bounces - 1;
                        q.excluded = hit_tri;
                        q.o = where;
                        q.d = mk(cur.d.x + two * n.x, cur.d.y + two * n.y, cur.d.z + two * n.z);
                        q.w = w;
                        q.fromCamera = 0;
                        q.tmin = 0.f;
                        ring[tail] = q;
                        tail = (tail + 1) % RT_RING;
                        if ((tail + 1) % RT_RING == head) open = false;
                    }
                }
                // see-through continuation of the same ray (:707-722)
                if (open) {
                    w = mk(cur.w.x * tex.x * transp.x, cur.w.y * tex.y * transp.y, cur.w.z * tex.z * transp.z);
                    if (3.f / 256.f <= w.x + w.y + w.z) {
                        QRay q;
                        q.bounces = cur.bounces - 1;
                        q.excluded = hit_tri;
                        q.o = cur.o;
                        q.d = cur.d;
                        q.w = w;
                        q.fromCamera = cur.fromCamera;
                        q.tmin = hit_t;
                        ring[tail] = q;
                        tail = (tail + 1) % RT_RING;
                    }
                }
            }
        }

        head = (head + 1) % RT_RING;
        if (head == tail) break;
        cur = ring[head];
        cur_tmax = RT_INF;
    }
    return out;
}

} // namespace

// ---- kernels ---------------------------------------------------------------------------------------------------

// Trace: grid = tileCount * 64 workgroups of 256 threads; workgroup = 16x16 patch, wave = 8x8 quadrant.
template <bool COUNT>
#ifndef RT_TRACE_WAVES
#define RT_TRACE_WAVES 5
#endif
__global__ __launch_bounds__(256, RT_TRACE_WAVES) void rt_trace_kernel(const RtDevScene S)
{
    __shared__ Shared sh;
    for (int i = threadIdx.x; i < 3 * (RT_GRID_DIV + 1); i += 256) (&sh.planes[0][0])[i] = S.boxMin[i];
    sh.unit255[threadIdx.x] = (float)threadIdx.x / 255.f; // (tableValue.x) / 255.f (:118-120)
    __syncthreads();

    const uint32_t slot = blockIdx.x >> 6;        // 64 patches per tile
    const uint32_t patch = blockIdx.x & 63;
    const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint32_t lx = (patch & 7) * RT_PATCH + (wave & 1) * 8 + (lane & 7);
    const uint32_t ly = (patch >> 3) * RT_PATCH + (wave >> 1) * 8 + (lane >> 3);
    const uint32_t tile = S.tileIds[slot];
    const uint32_t gx = (tile % S.tilesX) * RT_TILE + lx;
    const uint32_t gy = (tile / S.tilesX) * RT_TILE + ly;

    Counters cn;
    if (COUNT) for (int i = 0; i < ST_COUNT; ++i) cn.v[i] = 0;

    if (gx < S.width && gy < S.height) {
        const uint32_t pixel = gy * S.width + gx;
        const uint32_t localPixel = slot * RT_TILE_PIXELS + ly * RT_TILE + lx;
        const float scale = (float)(0xFFFF) / (float)S.sampleCount; // :728
        int r = 0, g = 0, b = 0;
        // samples 1..S in order, each addend truncated on its own, saturating (:726-741; raytrace.c:612-653)
        for (uint32_t s = 1; s <= S.sampleCount; ++s) {
            const V3 c = trace_sample<COUNT>(S, sh, pixel, localPixel, (float)gx, (float)gy, s, cn);
            r = sat_add_u16(r, c.x, scale);
            g = sat_add_u16(g, c.y, scale);
            b = sat_add_u16(b, c.z, scale);
        }
        uint16_t *planes = S.tileBuf + (size_t)slot * 3 * RT_TILE_PIXELS + ly * RT_TILE + lx;
        planes[0] = (uint16_t)r;
        planes[RT_TILE_PIXELS] = (uint16_t)g;
        planes[2 * RT_TILE_PIXELS] = (uint16_t)b;
    }

    if (COUNT) {
        for (int i = 0; i < ST_COUNT; ++i) {
            unsigned long long v = cn.v[i];
            for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
            if (lane == 0 && v) atomicAdd(&S.stats[i], v);
        }
    }
}

// Builds triRec / triShade from the ABI arrays (one thread per triangle).  Same operations as
// raytrace_opencl.c:131-149 so every stored value equals what the reference recomputes per test.
__global__ __launch_bounds__(256) void rt_prepare_triangles(uint32_t triangleCount, const float4 *__restrict__ vertex,
                                                            const int4 *__restrict__ triIndex, const int *__restrict__ triMaterial,
                                                            const float2 *__restrict__ triUv, const float4 *__restrict__ triNormal,
                                                            float *__restrict__ triRec, float *__restrict__ triShade)
{
    const uint32_t t = blockIdx.x * 256 + threadIdx.x;
    if (t >= triangleCount) return;
    const int4 vi = triIndex[t];
    const float4 fa = vertex[vi.x], fb = vertex[vi.y], fc = vertex[vi.z];
    const V3 a = mk(fa.x, fa.y, fa.z), b = mk(fb.x, fb.y, fb.z), c = mk(fc.x, fc.y, fc.z);
    const V3 ab = sub3(b, a), ac = sub3(c, a);
    const V3 n = cross3(ac, ab);
    const float abab = dot3(ab, ab), abac = dot3(ab, ac), acac = dot3(ac, ac);
    const float inv = 1.f / (abac * abac - abab * acac);
    float4 *rec = reinterpret_cast<float4 *>(triRec) + 4 * (size_t)t;
    rec[0] = make_float4(a.x, a.y, a.z, ab.x);
    rec[1] = make_float4(ab.y, ab.z, ac.x, ac.y);
    rec[2] = make_float4(ac.z, n.x, n.y, n.z);
    rec[3] = make_float4(abab, abac, acac, inv);
    float *sh = triShade + 24 * (size_t)t;
    const float4 na = triNormal[3 * (size_t)t], nb = triNormal[3 * (size_t)t + 1], nc = triNormal[3 * (size_t)t + 2];
    const float2 ua = triUv[3 * (size_t)t], ub = triUv[3 * (size_t)t + 1], uc = triUv[3 * (size_t)t + 2];
    sh[0] = b.x; sh[1] = b.y; sh[2] = b.z; sh[3] = c.x; sh[4] = c.y; sh[5] = c.z;
    sh[6] = na.x; sh[7] = na.y; sh[8] = na.z; sh[9] = nb.x; sh[10] = nb.y; sh[11] = nb.z; sh[12] = nc.x; sh[13] = nc.y; sh[14] = nc.z;
    sh[15] = ua.x; sh[16] = ua.y; sh[17] = ub.x; sh[18] = ub.y; sh[19] = uc.x; sh[20] = uc.y;
    sh[21] = __int_as_float(triMaterial[t]);
    sh[22] = 0.f; sh[23] = 0.f;
}

// De-tiles [slot][3][128*128] tile buffers into three row-major planes with a saturating add (the ABI accumulates
// into the caller's planes, raytrace_opencl.c:729-740).  One thread per 8 consecutive pixels of a tile row.
__global__ __launch_bounds__(256) void rt_detile_kernel(const uint16_t *__restrict__ tileBuf, const uint32_t *__restrict__ tileIds,
                                                        uint32_t tileCount, uint32_t width, uint32_t height, uint32_t tilesX,
                                                        uint16_t *planeR, uint16_t *planeG, uint16_t *planeB)
{
    const uint32_t gid = blockIdx.x * 256 + threadIdx.x; // one per (slot, plane, row, 8-px segment)
    const uint32_t segs = RT_TILE / 8;
    const uint32_t total = tileCount * 3 * RT_TILE * segs;
    if (gid >= total) return;
    const uint32_t seg = gid % segs;
    const uint32_t row = (gid / segs) % RT_TILE;
    const uint32_t plane = (gid / (segs * RT_TILE)) % 3;
    const uint32_t slot = gid / (segs * RT_TILE * 3);
    const uint32_t tile = tileIds[slot];
    const uint32_t gy = (tile / tilesX) * RT_TILE + row; // ids beyond the image fall out at the height test
    const uint32_t gx0 = (tile % tilesX) * RT_TILE + seg * 8;
    if (gy >= height || gx0 >= width) return;
    const uint16_t *src = tileBuf + ((size_t)slot * 3 + plane) * RT_TILE_PIXELS + row * RT_TILE + seg * 8;
    uint16_t *dst = (plane == 0 ? planeR : plane == 1 ? planeG : planeB) + (size_t)gy * width + gx0;
    const uint32_t n = (width - gx0 < 8) ? width - gx0 : 8;
    for (uint32_t i = 0; i < n; ++i) {
        uint32_t v = (uint32_t)dst[i] + (uint32_t)src[i];
        dst[i] = (uint16_t)(v > 0xFFFFu ? 0xFFFFu : v);
    }
}

// ---- launch wrappers (called from rt_api.cpp; keep every <<< >>> in this translation unit) -----------------------
extern "C" hipError_t rtk_launch_trace(const RtDevScene *scene, int counted, hipStream_t stream)
{
    const uint32_t blocks = scene->tileCount * 64;
    if (blocks == 0) return hipSuccess;
    if (counted) hipLaunchKernelGGL(rt_trace_kernel<true>, dim3(blocks), dim3(256), 0, stream, *scene);
    else hipLaunchKernelGGL(rt_trace_kernel<false>, dim3(blocks), dim3(256), 0, stream, *scene);
    return hipGetLastError();
}

extern "C" hipError_t rtk_launch_prepare(uint32_t triangleCount, const void *vertex, const void *triIndex, const void *triMaterial,
                                         const void *triUv, const void *triNormal, float *triRec, float *triShade, hipStream_t stream)
{
    if (triangleCount == 0) return hipSuccess;
    hipLaunchKernelGGL(rt_prepare_triangles, dim3((triangleCount + 255) / 256), dim3(256), 0, stream, triangleCount,
                       (const float4 *)vertex, (const int4 *)triIndex, (const int *)triMaterial, (const float2 *)triUv,
                       (const float4 *)triNormal, triRec, triShade);
    return hipGetLastError();
}

extern "C" hipError_t rtk_launch_detile(const void *tileBuf, const uint32_t *tileIds, uint32_t tileCount, uint32_t width,
                                        uint32_t height, uint32_t tilesX, void *planeR, void *planeG, void *planeB, hipStream_t stream)
{
    const uint32_t total = tileCount * 3 * RT_TILE * (RT_TILE / 8);
    if (total == 0) return hipSuccess;
    hipLaunchKernelGGL(rt_detile_kernel, dim3((total + 255) / 256), dim3(256), 0, stream, (const uint16_t *)tileBuf, tileIds, tileCount,
                       width, height, tilesX, (uint16_t *)planeR, (uint16_t *)planeG, (uint16_t *)planeB);
    return hipGetLastError();
}
