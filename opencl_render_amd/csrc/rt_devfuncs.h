// rt_devfuncs.h -- device-side building blocks shared by the HIP kernels (rt_kernels.hip: single-launch megakernel,
// rt_wavefront.hip: staged wavefront pipeline).  Each function restates one piece of the reference kernel
// (source/opencl/raytrace_opencl.c) in strict fp32 with the reference's operation order; see the bit-exactness
// contract at the top of rt_kernels.hip.  Everything lives in an anonymous namespace: internal linkage per TU.
#ifndef RT_DEVFUNCS_H
#define RT_DEVFUNCS_H

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
// rotl64 (raytrace_opencl.c:1-3) by a compile-time amount: two v_alignbit_b32 -- one per half of the result, each picking 32 bits out of
// the 64 of (hi:lo) or (lo:hi) -- where the shift-shift-or of the source spelling costs four 64-bit instructions.  The generator
// makes 16 rotates per draw and the kernels that draw are bound by instruction issue (profiles/r02_primary_4k_valu.json).
__device__ __forceinline__ uint64_t rol64(uint64_t v, int n)
{
    const uint32_t lo = (uint32_t)v, hi = (uint32_t)(v >> 32);
    if (n == 0) return v;
    if (n == 32) return ((uint64_t)lo << 32) | hi;
    if (n < 32) { // result.hi = (hi:lo) >> (32 - n), result.lo = (lo:hi) >> (32 - n)
        const uint32_t rh = __builtin_amdgcn_alignbit(hi, lo, 32 - n), rl = __builtin_amdgcn_alignbit(lo, hi, 32 - n);
        return ((uint64_t)rh << 32) | rl;
    }
    const uint32_t rh = __builtin_amdgcn_alignbit(lo, hi, 64 - n), rl = __builtin_amdgcn_alignbit(hi, lo, 64 - n); // a rotate by n - 32 of the swapped halves
    return ((uint64_t)rh << 32) | rl;
}
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
    if (w == 1u && h == 1u) { // a one-texel map: u*(w-1) and v*(h-1) are 0 (or NaN, which converts to 0 here) whatever the uv
        long long idx1 = (long long)start;
        if (idx1 < 0) idx1 = 0;
        if (idx1 >= (long long)S.texelCount) idx1 = (long long)S.texelCount - 1;
        const uchar4 px1 = reinterpret_cast<const uchar4 *>(S.textures)[idx1];
        if (COUNT) cn.v[ST_TEXELS]++;
        raw = px1.x;
        return mk(sh.unit255[px1.x], sh.unit255[px1.y], sh.unit255[px1.z]);
    }
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

// A material's channel descriptors (rt_device.h, matRec) in registers: two 16-byte loads issued together.
struct MatRec { uint32_t desc[CH_COUNT]; int m; };
__device__ __forceinline__ MatRec load_mat(const RtDevScene &S, int m)
{
    const uint4 *p = reinterpret_cast<const uint4 *>(S.matRec + 8 * (size_t)m);
    const uint4 a = p[0], b = p[1];
    MatRec r;
    r.desc[0] = a.x; r.desc[1] = a.y; r.desc[2] = a.z; r.desc[3] = a.w; r.desc[4] = b.x;
    r.m = m;
    return r;
}
// Get2dTableValue3 (raytrace_opencl.c:103-122) through the descriptor: a one-texel channel is answered from the descriptor itself
// (u*(w-1) and v*(h-1) are 0 -- or NaN, which indexes texel 0 here -- whatever the uv), an image goes to the tables and the atlas.
// Call only for a present channel (descriptor != 0).
template <bool COUNT>
__device__ __forceinline__ V3 texel_rec(const RtDevScene &S, const Shared &sh, const MatRec &M, int ch, const float *uv, float l1, float l2,
                                        uint32_t &raw, Counters &cn)
{
    const uint32_t px = M.desc[ch];
    if (px & 0x80000000u) {
        if (COUNT) cn.v[ST_TEXELS]++;
        raw = px & 255u;
        return mk(sh.unit255[px & 255u], sh.unit255[(px >> 8) & 255u], sh.unit255[(px >> 16) & 255u]);
    }
    const int at = CH_COUNT * M.m + ch;
    return texel<COUNT>(S, sh, S.matStart[at], S.matSize[2 * at], S.matSize[2 * at + 1], uv, l1, l2, raw, cn);
}

// Exponent ranges in which the division (plane - o) / d needs neither operand scaling nor a fix-up (wf_trace_kernel's walk, rt_wavefront.hip): a
// plane or origin coordinate is 0 or 2^-60 <= |x| <= 2^39, so n = plane - o is 0 or 2^-84 <= |n| <= 2^40; 2^-40 <= |d| <= 2^40.
// Then exponent(n) - exponent(d) < 96, neither d, 1/d nor n/d is subnormal, and n is not tiny (the conditions of v_div_scale_f32),
// and v_div_fixup_f32 returns the quotient it is given (the sign of a zero quotient does not matter: these values are only compared).
__device__ __forceinline__ bool tame_origin(float x)
{
    const float m = __builtin_fabsf(x);
    return (m == 0.f) | ((m >= 0x1p-60f) & (m <= 0x1p39f));
}
__device__ __forceinline__ bool tame_direction(float x)
{
    const float m = __builtin_fabsf(x);
    return (m >= 0x1p-40f) & (m <= 0x1p40f);
}
// r1 of the compiler's division sequence: v_rcp_f32, then fma(fma(-d, r0, 1), r0, r0)
__device__ __forceinline__ float refined_rcp(float dd)
{
    const float r0 = __builtin_amdgcn_rcpf(dd);
    return __builtin_fmaf(__builtin_fmaf(-dd, r0, 1.f), r0, r0);
}

// The walk's quotient without the scaling and fix-up instructions: only for tame operands (above), where it is the compiler's
// own sequence minus instructions that do nothing there.  r1 = refined_rcp(d).
__device__ __forceinline__ float tame_quotient(float n, float d, float r1)
{
    const float q0 = n * r1;
    const float q1 = __builtin_fmaf(__builtin_fmaf(-d, q0, n), r1, q0);
    return __builtin_fmaf(__builtin_fmaf(-d, q1, n), r1, q1);
}

// ---- shading normal (raytrace_opencl.c:195-263) -----------------------------------------------------------------
template <bool COUNT>
__device__ V3 shading_normal(const RtDevScene &S, const Shared &sh, V3 where, V3 ray_o, V3 ray_d, uint32_t tri, float l1, float l2,
                             const float *shade, int m, Counters &cn, const MatRec *mat = nullptr, const float4 *firstVertex = nullptr)
{
    const float4 *rec = reinterpret_cast<const float4 *>(S.triRec) + 4 * (size_t)tri;
    const float4 r0 = firstVertex ? *firstVertex : rec[0];
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
        const bool oneTexel = mat && (mat->desc[CH_BUMP] & 0x80000000u); // the height map is one texel, held in the descriptor
        const uint32_t bw = oneTexel ? 1u : ((mat && mat->desc[CH_BUMP] == 0u) ? 0u : S.matSize[2 * (CH_COUNT * m + CH_BUMP)]);
        if (0 < bw) {
            const uint32_t bh = oneTexel ? 1u : S.matSize[2 * (CH_COUNT * m + CH_BUMP) + 1];
            const int bstart = oneTexel ? 0 : S.matStart[CH_COUNT * m + CH_BUMP];
            const float *uv = shade + 15;
            const V3 tb = ld3(S.tb), lr = ld3(S.lr);
            uint32_t h0, hs, he;
            float t, p1 = 0.f, p2 = 0.f;
            if (oneTexel) { // three fetches of the same texel
                h0 = hs = he = mat->desc[CH_BUMP] & 255u;
                if (COUNT) cn.v[ST_TEXELS] += 3;
            } else
            (void)texel<COUNT>(S, sh, bstart, bw, bh, uv, l1, l2, h0, cn);
            if (oneTexel) {
            } else if (bw == 1u && bh == 1u) { // one-texel height map: the neighbour samples are the same texel, wherever they land
                (void)texel<COUNT>(S, sh, bstart, bw, bh, uv, 0.f, 0.f, hs, cn);
                (void)texel<COUNT>(S, sh, bstart, bw, bh, uv, 0.f, 0.f, he, cn);
            } else {
                tri_test(S.triRec, tri, ray_o, mk(ray_d.x + tb.x, ray_d.y + tb.y, ray_d.z + tb.z), 0.f, RT_INF, t, p1, p2);
                (void)texel<COUNT>(S, sh, bstart, bw, bh, uv, p1, p2, hs, cn);
                tri_test(S.triRec, tri, ray_o, mk(ray_d.x + lr.x, ray_d.y + lr.y, ray_d.z + lr.z), 0.f, RT_INF, t, p1, p2);
                (void)texel<COUNT>(S, sh, bstart, bw, bh, uv, p1, p2, he, cn);
            }
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

// (float)pow(0.5f, x) with x = maxLen / halfAttenuationDistance (raytrace_opencl.c:631), NaN -> 1 (:632's isnan test):
// 0.5^x = 2^-x in double on the device; exp2(-0) is exactly 1, skipped for the reference's own
// scenes where halfAtt is infinite.
__device__ __forceinline__ float half_falloff(float x)
{
    const float fall = (x == 0.f) ? 1.f : __double2float_rn(exp2(-(double)x));
    return (fall == fall) ? fall : 1.f;
}

#define RT_MAX2(a, b) (((a) > (b)) ? (a) : (b)) /* raytrace.h:30 */

} // namespace

#endif // RT_DEVFUNCS_H
