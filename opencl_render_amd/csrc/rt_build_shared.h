// rt_build_shared.h -- the arithmetic of the camera candidate-list builder, shared by the host builder (rt_builders.cpp)
// and the device builder (rt_build_device.hip) so that both produce the same lists by construction.  Restates
// source/util/trianglelist.cpp:74-90 (GetCameraPosition) and :131-217 (FillRectangle) in the same fp32/fp64 operations and
// order.  The per-triangle part (RectSetup) and the per-pixel test are separate so that a big triangle's rectangle can be
// shared out among the threads of a workgroup; the tests of different pixels do not depend on each other.
#ifndef RT_BUILD_SHARED_H
#define RT_BUILD_SHARED_H

#include <math.h>
#include <stdint.h>

#if defined(__HIPCC__)
#define RT_HD __host__ __device__
#else
#define RT_HD
#endif

namespace rtbuild {

struct F2 { float x, y; };
struct F3 { float x, y, z; };

RT_HD inline float dot3(F3 a, F3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }              // raytrace.c:18-20
RT_HD inline F3 cross3(F3 a, F3 b) { return F3{ a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x }; } // :21-27

// x86-64 float/double -> unsigned conversions as MSVC/gcc emit them (cvttss2si r64 + truncate to 32 bits):
// negative values wrap, NaN/overflow give 0 in the low word.
RT_HD inline uint32_t to_u32(double v)
{
    if (!(v > -9223372036854775808.0 && v < 9223372036854775808.0)) return 0u;
    return (uint32_t)(int64_t)v;
}

struct Camera { F3 eye, topLeft, lr, tb; float pixelSizeInv; };

// trianglelist.cpp:74-90
RT_HD inline F2 camera_position(const Camera &c, F3 v)
{
    float invSq = c.pixelSizeInv * c.pixelSizeInv;
    F3 ev{ v.x - c.eye.x, v.y - c.eye.y, v.z - c.eye.z };
    F3 sn = cross3(c.lr, c.tb);
    float scale = dot3(c.topLeft, sn) / dot3(ev, sn);
    F3 tv{ scale * ev.x - c.topLeft.x, scale * ev.y - c.topLeft.y, scale * ev.z - c.topLeft.z };
    return F2{ dot3(c.lr, tv) * invSq, dot3(c.tb, tv) * invSq };
}

// Per-triangle part of trianglelist.cpp:131-217
struct RectSetup {
    F2 a, b, c, ab, bc, ca, abS, bcS, caS;
    uint32_t x0, y0, x1, y1; // bounding rectangle clipped to the image, inclusive
    uint32_t ax, ay;         // pixel of vertex a (tested separately, :160-162)
    bool aOnScreen;
};

RT_HD inline RectSetup rect_setup(uint32_t W, uint32_t H, F2 a, F2 b, F2 c)
{
    RectSetup s;
    s.a = a; s.b = b; s.c = c;
    s.ab = F2{ b.x - a.x, b.y - a.y }; s.bc = F2{ c.x - b.x, c.y - b.y }; s.ca = F2{ a.x - c.x, a.y - c.y };
    // slopes; division by zero fails the edge tests by design (:143-150)
    s.abS.x = s.ab.x / s.ab.y; s.abS.y = 1.f / s.abS.x;
    s.bcS.x = s.bc.x / s.bc.y; s.bcS.y = 1.f / s.bcS.x;
    s.caS.x = s.ca.x / s.ca.y; s.caS.y = 1.f / s.caS.x;
    // bounding rectangle clipped to the image (:153-157); fmin/fmax are the double functions on promoted floats
    s.x0 = to_u32(fmax(0.0, fmin(fmin((double)a.x, (double)b.x), fmin((double)c.x, (double)(float)(W - 1)))));
    s.y0 = to_u32(fmax(0.0, fmin(fmin((double)a.y, (double)b.y), fmin((double)c.y, (double)(float)(H - 1)))));
    s.x1 = to_u32(fmin((double)(float)(W - 1), fmax(fmax((double)a.x, (double)b.x), fmax((double)c.x, 0.0))));
    s.y1 = to_u32(fmin((double)(float)(H - 1), fmax(fmax((double)a.y, (double)b.y), fmax((double)c.y, 0.0))));
    s.ax = to_u32(floor((double)a.x)); s.ay = to_u32(floor((double)a.y));
    s.aOnScreen = (0.f <= a.x && a.x < (float)W && 0.f <= a.y && a.y < (float)H);
    return s;
}

// the pixel holding vertex a, if on screen (:160-162)
RT_HD inline uint64_t rect_a_pixel(const RectSetup &s, uint32_t W) { return (uint64_t)floor((double)s.a.x) + (uint64_t)floor((double)s.a.y) * (uint64_t)W; }

// Does pixel (x, y) of the rectangle receive the triangle?  (:164-211; the pixel of vertex a is handled by the caller)
RT_HD inline bool rect_pixel_test(const RectSetup &s, uint32_t x, uint32_t y)
{
    const F2 a = s.a, b = s.b, c = s.c;
    const float fx = (float)x, fy = (float)y;
    // where each edge crosses this pixel's row/column lines (:169-181)
    float ab0 = a.x + (fy - a.y) * s.abS.x, ab1 = a.y + (fx - a.x) * s.abS.y, ab2 = ab0 + s.abS.x, ab3 = ab1 + s.abS.y;
    float bc0 = b.x + (fy - b.y) * s.bcS.x, bc1 = b.y + (fx - b.x) * s.bcS.y, bc2 = bc0 + s.bcS.x, bc3 = bc1 + s.bcS.y;
    float ca0 = c.x + (fy - c.y) * s.caS.x, ca1 = c.y + (fx - c.x) * s.caS.y, ca2 = ca0 + s.caS.x, ca3 = ca1 + s.caS.y;
    bool edge =
        ((0.f <= (a.x - ab0) * (ab0 - b.x)) & (x == to_u32(ab0))) | ((0.f <= (a.x - ab2) * (ab2 - b.x)) & (x == to_u32(ab2))) |
        ((0.f <= (a.y - ab1) * (ab1 - b.y)) & (y == to_u32(ab1))) | ((0.f <= (a.y - ab3) * (ab3 - b.y)) & (y == to_u32(ab3))) |
        ((0.f <= (b.x - bc0) * (bc0 - c.x)) & (x == to_u32(bc0))) | ((0.f <= (b.x - bc2) * (bc2 - c.x)) & (x == to_u32(bc2))) |
        ((0.f <= (b.y - bc1) * (bc1 - c.y)) & (y == to_u32(bc1))) | ((0.f <= (b.y - bc3) * (bc3 - c.y)) & (y == to_u32(bc3))) |
        ((0.f <= (c.x - ca0) * (ca0 - a.x)) & (x == to_u32(ca0))) | ((0.f <= (c.x - ca2) * (ca2 - a.x)) & (x == to_u32(ca2))) |
        ((0.f <= (c.y - ca1) * (ca1 - a.y)) & (y == to_u32(ca1))) | ((0.f <= (c.y - ca3) * (ca3 - a.y)) & (y == to_u32(ca3)));
    if (edge) return true;
    // pixel corner inside the triangle: same-sign cross products (:197-211)
    float axx = fx - a.x, axy = fy - a.y, bxx = fx - b.x, bxy = fy - b.y, cxx = fx - c.x, cxy = fy - c.y;
    float k1 = s.ab.x * axy - s.ab.y * axx, k2 = s.bc.x * bxy - s.bc.y * bxx, k3 = s.ca.x * cxy - s.ca.y * cxx;
    return (0 <= k1 * k2) & (0 <= k2 * k3);
}

// trianglelist.cpp:131-217.  Calls emit(pixel) for every pixel whose candidate list receives the triangle.
template <class Emit> RT_HD inline void fill_rectangle(uint32_t W, uint32_t H, F2 a, F2 b, F2 c, Emit emit)
{
    const RectSetup s = rect_setup(W, H, a, b, c);
    if (s.aOnScreen) emit(rect_a_pixel(s, W));
    for (uint32_t x = s.x0; x <= s.x1; ++x)
        for (uint32_t y = s.y0; y <= s.y1; ++y) {
            if (x == s.ax && y == s.ay) continue;
            if (rect_pixel_test(s, x, y)) emit((uint64_t)x + (uint64_t)y * (uint64_t)W);
        }
}

// ---- scene grid: the membership test of SceneTriangleList::New (trianglelist.cpp:655-737) -----------------------------------
constexpr int DIV = 256; // trianglelist.h:110

// trianglelist.cpp:381-430 (Sutherland-Hodgman step with in-place insert, then removal of the outside points)
RT_HD inline bool cull(bool isMax, float limit, int dim, int *count, float poly[16][3])
{
    bool fresh[16] = { false };
    for (int i = 0; i < *count; ++i) {
        int nx = (i + 1) % *count;
        float di = limit - poly[i][dim];
        float dn = limit - poly[nx][dim];
        if (di * dn < 0.f) {
            float e0 = poly[nx][0] - poly[i][0], e1 = poly[nx][1] - poly[i][1], e2 = poly[nx][2] - poly[i][2];
            float e[3] = { e0, e1, e2 };
            float pct = di / e[dim];
            int at = i + 1;
            for (int j = (*count)++; at < j; --j) { poly[j][0] = poly[j - 1][0]; poly[j][1] = poly[j - 1][1]; poly[j][2] = poly[j - 1][2]; }
            poly[at][0] = poly[i][0] + pct * e0;
            poly[at][1] = poly[i][1] + pct * e1;
            poly[at][2] = poly[i][2] + pct * e2;
            fresh[at] = true;
            i = at;
        }
    }
    for (int i = 0; i < *count; ++i) {
        bool outside = isMax ? (limit < poly[i][dim]) : (poly[i][dim] < limit);
        if (!fresh[i] && outside) {
            int k = (*count)--;
            for (int j = i + 1; j < k; ++j) {
                poly[j - 1][0] = poly[j][0]; poly[j - 1][1] = poly[j][1]; poly[j - 1][2] = poly[j][2];
                fresh[j - 1] = fresh[j];
            }
            --i;
        }
    }
    return 0 < *count;
}

// trianglelist.cpp:433-449
RT_HD inline bool box_hits_triangle(const float lo[3], const float hi[3], F3 a, F3 b, F3 c)
{
    int n = 3;
    float poly[16][3] = { { a.x, a.y, a.z }, { b.x, b.y, b.z }, { c.x, c.y, c.z } };
    return cull(false, lo[0], 0, &n, poly) && cull(false, lo[1], 1, &n, poly) && cull(false, lo[2], 2, &n, poly) &&
           cull(true, hi[0], 0, &n, poly) && cull(true, hi[1], 1, &n, poly) && cull(true, hi[2], 2, &n, poly);
}

// raytrace_opencl.c:174-193
RT_HD inline void box_address(const float (*bm)[4], F3 p, int cell[3])
{
    int cx = 0, cy = 0, cz = 0;
    for (int div = DIV / 2; div >= 1; div /= 2) {
        if (bm[cx + div][0] < p.x) cx += div;
        if (bm[cy + div][1] < p.y) cy += div;
        if (bm[cz + div][2] < p.z) cz += div;
    }
    cell[0] = cx; cell[1] = cy; cell[2] = cz;
}

} // namespace rtbuild

#endif
