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
#include "rt_devfuncs.h"

namespace {

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
                    const float e = mag * half_falloff(x);
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

// De-tiles [slot][3][128*128] tile buffers into three row-major planes.  ACCUMULATE: saturating add into what is there (the ABI
// accumulates into the caller's planes, raytrace_opencl.c:729-740); otherwise the planes' pixels are simply written (a gather
// root whose planes would be zeroed first anyway).  One thread per 8 consecutive pixels of a tile row: 16 bytes in, 16 bytes out
// when the row segment is whole and the image width keeps it aligned, pixel by pixel otherwise.
template <bool ACCUMULATE>
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
    if (n == 8 && (width & 7u) == 0u && (reinterpret_cast<uintptr_t>(dst) & 15u) == 0u) {
        uint4 v = *reinterpret_cast<const uint4 *>(src);
        if (ACCUMULATE) {
            const uint4 o = *reinterpret_cast<const uint4 *>(dst);
            auto add2 = [](uint32_t a, uint32_t b) { // two saturating u16 adds in one word
                uint32_t lo = (a & 0xffffu) + (b & 0xffffu), hi = (a >> 16) + (b >> 16);
                lo = lo > 0xffffu ? 0xffffu : lo; hi = hi > 0xffffu ? 0xffffu : hi;
                return lo | (hi << 16);
            };
            v = make_uint4(add2(v.x, o.x), add2(v.y, o.y), add2(v.z, o.z), add2(v.w, o.w));
        }
        *reinterpret_cast<uint4 *>(dst) = v;
        return;
    }
    for (uint32_t i = 0; i < n; ++i) {
        uint32_t v = (uint32_t)src[i] + (ACCUMULATE ? (uint32_t)dst[i] : 0u);
        dst[i] = (uint16_t)(v > 0xFFFFu ? 0xFFFFu : v);
    }
}

// Builds the (cell, triangle) pair records of the dense grid view from the per-triangle records (rt_device.h).
// One thread per pair.
__global__ __launch_bounds__(256) void rt_gather_pair_records(uint32_t pairCount, const uint32_t *__restrict__ pairTri, const uint32_t *__restrict__ pairInfo,
                                                              const float4 *__restrict__ triRec, float4 *__restrict__ pairRec)
{
    const uint32_t pair = blockIdx.x * 256 + threadIdx.x;
    if (pair >= pairCount) return;
    const uint32_t tri = pairTri[pair];
    const float4 r0 = triRec[(size_t)tri * 4], r1 = triRec[(size_t)tri * 4 + 1], r2 = triRec[(size_t)tri * 4 + 2], r3 = triRec[(size_t)tri * 4 + 3];
    float4 *out = pairRec + (size_t)pair * 4;
    out[0] = make_float4(r0.x, r0.y, r0.z, __uint_as_float(tri)); // a, triangle id
    out[1] = make_float4(r2.y, r2.z, r2.w, __uint_as_float(pairInfo[pair])); // n, candidates of the cell (first records only)
    out[2] = make_float4(r0.w, r1.x, r1.y, r3.y);                  // ab, dot(ab,ac)
    out[3] = make_float4(r1.z, r1.w, r2.x, r3.w);                  // ac, 1/(abac^2 - abab*acac)
}

// ---- launch wrappers (called from rt_api.cpp; keep every <<< >>> in this translation unit) -----------------------
extern "C" hipError_t rtk_launch_gather_pairs(uint32_t pairCount, const uint32_t *pairTri, const uint32_t *pairInfo, const float *triRec, float *pairRec, hipStream_t stream)
{
    if (pairCount == 0) return hipSuccess;
    hipLaunchKernelGGL(rt_gather_pair_records, dim3((pairCount + 255) / 256), dim3(256), 0, stream, pairCount, pairTri, pairInfo,
                       (const float4 *)triRec, (float4 *)pairRec);
    return hipGetLastError();
}

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
                                        uint32_t height, uint32_t tilesX, void *planeR, void *planeG, void *planeB, int accumulate, hipStream_t stream)
{
    const uint32_t total = tileCount * 3 * RT_TILE * (RT_TILE / 8);
    if (total == 0) return hipSuccess;
    if (accumulate)
        hipLaunchKernelGGL(rt_detile_kernel<true>, dim3((total + 255) / 256), dim3(256), 0, stream, (const uint16_t *)tileBuf, tileIds, tileCount,
                           width, height, tilesX, (uint16_t *)planeR, (uint16_t *)planeG, (uint16_t *)planeB);
    else
        hipLaunchKernelGGL(rt_detile_kernel<false>, dim3((total + 255) / 256), dim3(256), 0, stream, (const uint16_t *)tileBuf, tileIds, tileCount,
                           width, height, tilesX, (uint16_t *)planeR, (uint16_t *)planeG, (uint16_t *)planeB);
    return hipGetLastError();
}
