/*
 * rt_oracle.c -- CPU oracle for the opencl_render hot path.  TEST INFRASTRUCTURE ONLY (see rt_oracle.h).
 *
 * A plain-C restatement of the reference's deterministic C path, written from its behaviour:
 *   reference kernel      source/opencl/raytrace_opencl.c:1-742
 *   reference CPU driver  source/opencl/raytrace.c:15-27 (dot/cross), :604-655 (pixel/sample loop)
 * Every function cites the lines it follows.  All arithmetic is strict fp32 with the double detours the
 * reference's C build takes (sqrt/floor/modf/sin/cos/pow/fabs are the C double functions applied to promoted
 * floats); build with -ffp-contract=off, no fast-math.  M_PI is the FLOAT 3.14159265f (raytrace.h:33).
 *
 * Deliberate deviations (each is undefined behaviour in the reference, so there is nothing to match):
 *  - GetTriangleNormal reads materialImageSize[5*m+3] before testing m>=0 (:226 vs :231): here the load is
 *    skipped for m<0 (the value is unused in that case).
 *  - the two bump probe rays (:244,:249) leave abL/acL uninitialised when the probe misses the triangle plane
 *    in (0,inf); here they start at 0.
 *  - float->int conversion of NaN / out-of-range values (:729-737) is UB in C; x86-64 cvttss2si yields INT_MIN,
 *    which is what the reference binary computes.  trunc_x86() spells that out.
 */
#include "rt_oracle.h"

#include <math.h>
#include <stddef.h>
#include <string.h>
#include <limits.h>

#define RT_PI_F 3.14159265f /* raytrace.h:33 */
#define RING 12             /* raytrace_opencl.c:404 */

enum { CH_COLOR = 0, CH_REFLECTION = 1, CH_TRANSPARENCY = 2, CH_BUMP = 3, CH_LUMINANCE = 4, CH_COUNT = 5 }; /* raytrace_opencl.h:14-22 */

typedef struct { float x, y, z; } v3;

static inline v3 ld3(const float *p) { v3 r = { p[0], p[1], p[2] }; return r; }

/* raytrace.c:18-20: (a0*b0 + a1*b1) + a2*b2 */
static inline float dot3(v3 a, v3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }

/* raytrace.c:21-27 */
static inline v3 cross3(v3 a, v3 b)
{
    v3 c;
    c.x = a.y * b.z - a.z * b.y;
    c.y = a.z * b.x - a.x * b.z;
    c.z = a.x * b.y - a.y * b.x;
    return c;
}

static inline v3 sub3(v3 a, v3 b) { v3 r = { a.x - b.x, a.y - b.y, a.z - b.z }; return r; }
/* o + t*d, one multiply and one add per component (:155-157, :351-353, :545-547) */
static inline v3 along(v3 o, float t, v3 d) { v3 r = { o.x + t * d.x, o.y + t * d.y, o.z + t * d.z }; return r; }

/* x86-64 cvttss2si: truncation toward zero; NaN and anything outside int range give INT_MIN. */
static inline int trunc_x86(float v)
{
    if (!(v > -2147483904.0f && v < 2147483648.0f)) return INT_MIN;
    return (int)v;
}

/* ---- PRNG: raytrace_opencl.c:1-23 ------------------------------------------------------------------ */
static inline uint64_t rol64(uint64_t v, int n) { return (v << n) | (v >> (64 - n)); }

static inline uint64_t xs64star(uint64_t v)
{
    v ^= v >> 12;
    v ^= v << 25;
    v ^= v >> 27;
    return v * 2685821657736338717ULL;
}

float rt_oracle_randf(uint64_t *state, float lo, float hi)
{
    /* eight rounds, odd rounds multiply by the constant, even rounds xor it (:13-20) */
    static const struct { int ra, rb; uint64_t k; } round_tab[8] = {
        { 55, 3, 0xc23f3c0ad9da6357ULL }, { 35, 3, 0xce84d6af03c16b89ULL },
        { 63, 35, 0xf097ef8bbe03ddccULL }, { 41, 12, 0x48302294fbfe30bfULL },
        { 1, 62, 0x79e7425e3f4f147dULL }, { 42, 29, 0x14d1d30856e5be9aULL },
        { 47, 45, 0x24289d47a66617c3ULL }, { 39, 6, 0x5576fb2f80a05d14ULL },
    };
    uint64_t s = *state;
    for (int r = 0; r < 8; ++r) {
        uint64_t mix = rol64(s, round_tab[r].ra) ^ rol64(s, round_tab[r].rb);
        mix = (r & 1) ? (mix ^ round_tab[r].k) : (mix * round_tab[r].k);
        s ^= xs64star(mix);
    }
    *state = s;
    /* :22 -- u64 -> double (RN), exact divide by 2^64, -> float (RN), then fp32 scale and offset */
    return lo + (hi - lo) * (float)((double)s / (double)0xffffffffffffffffULL);
}

/* raytrace_opencl.c:30-45 */
static v3 sphere_point(uint64_t *state, float radius)
{
    v3 p;
    float len, scale;
    do {
        p.x = rt_oracle_randf(state, -1.f, 1.f);
        p.y = rt_oracle_randf(state, -1.f, 1.f);
        p.z = rt_oracle_randf(state, -1.f, 1.f);
        len = (float)sqrt((double)dot3(p, p));
    } while (len <= 0.f);
    scale = (float)sqrt((double)rt_oracle_randf(state, 0.f, 1.f)) * radius / len;
    p.x = scale * p.x;
    p.y = scale * p.y;
    p.z = scale * p.z;
    return p;
}

void rt_oracle_sphere_point(uint64_t *state, float radius, float out[3])
{
    v3 p = sphere_point(state, radius);
    out[0] = p.x; out[1] = p.y; out[2] = p.z;
}

/* raytrace_opencl.c:25-28 */
float rt_oracle_positive_modf(float v)
{
    double ip;
    return (float)modf(modf((double)v, &ip) + 1., &ip);
}

/* ---- geometry helpers ------------------------------------------------------------------------------- */

/* raytrace_opencl.c:83-101 */
static float point_line_sq(v3 o, v3 e, v3 p)
{
    v3 oe = sub3(e, o);
    float oe_sq = dot3(oe, oe);
    v3 op = sub3(p, o);
    float k = dot3(op, oe) / oe_sq;
    v3 foot = along(o, k, oe);
    v3 d = sub3(foot, p);
    return dot3(d, d);
}

float rt_oracle_point_line_sq(const float o[3], const float e[3], const float p[3])
{
    return point_line_sq(ld3(o), ld3(e), ld3(p));
}

/* raytrace_opencl.c:124-172.  *t is always written; *ab_l/*ac_l only when tmin < t < tmax. */
static int ray_triangle(v3 o, v3 d, float tmin, float tmax, v3 a, v3 b, v3 c, float *t, float *ab_l, float *ac_l)
{
    int hit = 0;
    v3 ab = sub3(b, a);
    v3 ac = sub3(c, a);
    v3 ao = sub3(o, a);
    v3 n = cross3(ac, ab);
    *t = -dot3(n, ao) / dot3(n, d);
    if (tmin < *t && *t < tmax) {
        float abab = dot3(ab, ab);
        float abac = dot3(ab, ac);
        float acac = dot3(ac, ac);
        float inv = 1.f / (abac * abac - abab * acac);
        v3 ap = sub3(along(o, *t, d), a);
        float ap_ab = dot3(ap, ab);
        float ap_ac = dot3(ap, ac);
        *ab_l = (abac * ap_ac - acac * ap_ab) * inv;
        *ac_l = (abac * ap_ab - abab * ap_ac) * inv;
        hit = (0 <= *ab_l && 0 <= *ac_l && *ab_l + *ac_l <= 1.f);
    }
    return hit;
}

int rt_oracle_ray_triangle(const float o[3], const float d[3], float tmin, float tmax,
                           const float a[3], const float b[3], const float c[3],
                           float *t, float *ab_l, float *ac_l)
{
    return ray_triangle(ld3(o), ld3(d), tmin, tmax, ld3(a), ld3(b), ld3(c), t, ab_l, ac_l);
}

/* raytrace_opencl.c:174-193: per-axis binary search, strict '<' */
static void box_address(int div, const float *box_min, v3 p, int cell[3])
{
    int cx = 0, cy = 0, cz = 0;
    while (1 < div) {
        int mid;
        div /= 2;
        mid = cx + div; if (box_min[4 * mid + 0] < p.x) cx = mid;
        mid = cy + div; if (box_min[4 * mid + 1] < p.y) cy = mid;
        mid = cz + div; if (box_min[4 * mid + 2] < p.z) cz = mid;
    }
    cell[0] = cx; cell[1] = cy; cell[2] = cz;
}

void rt_oracle_box_address(int axes_div, const float *box_min, const float p[3], int out[3])
{
    box_address(axes_div, box_min, ld3(p), out);
}

/* raytrace_opencl.c:265-322: six sequential face clamps along the ray; early 'false' returns */
static int bind_in_cube(v3 *p, v3 d, v3 lo, v3 hi)
{
    float t;
    if (p->x < lo.x) { if (d.x <= 0) return 0; t = (lo.x - p->x) / d.x; p->x += t * d.x; p->y += t * d.y; p->z += t * d.z; }
    if (hi.x < p->x) { if (0 <= d.x) return 0; t = (hi.x - p->x) / d.x; p->x += t * d.x; p->y += t * d.y; p->z += t * d.z; }
    if (p->y < lo.y) { if (d.y <= 0) return 0; t = (lo.y - p->y) / d.y; p->x += t * d.x; p->y += t * d.y; p->z += t * d.z; }
    if (hi.y < p->y) { if (0 <= d.y) return 0; t = (hi.y - p->y) / d.y; p->x += t * d.x; p->y += t * d.y; p->z += t * d.z; }
    if (p->z < lo.z) { if (d.z <= 0) return 0; t = (lo.z - p->z) / d.z; p->x += t * d.x; p->y += t * d.y; p->z += t * d.z; }
    if (hi.z < p->z) { if (0 <= d.z) return 0; t = (hi.z - p->z) / d.z; p->x += t * d.x; p->y += t * d.y; p->z += t * d.z; }
    return 1;
}

int rt_oracle_bind_in_cube(float p[3], const float d[3], const float lo[3], const float hi[3])
{
    v3 q = ld3(p);
    int ok = bind_in_cube(&q, ld3(d), ld3(lo), ld3(hi));
    p[0] = q.x; p[1] = q.y; p[2] = q.z;
    return ok;
}

/* raytrace_opencl.c:103-122 */
static v3 texel(const uint8_t *table, uint32_t w, uint32_t h, const float *uv, float ab_l, float ac_l, rt_oracle_stats *st)
{
    /* uv = {aU,aV,bU,bV,cU,cV} */
    float pu = rt_oracle_positive_modf(uv[0] + (uv[2] - uv[0]) * ab_l + (uv[4] - uv[0]) * ac_l);
    float pv = rt_oracle_positive_modf(uv[1] + (uv[3] - uv[1]) * ab_l + (uv[5] - uv[1]) * ac_l);
    float lx = pu * (float)(w - 1u);
    float ly = pv * (float)(h - 1u);
    int fx = (int)floor((double)lx);
    int fy = (int)floor((double)ly);
    int at = (int)((uint32_t)fx + (uint32_t)fy * w);
    const uint8_t *px = table + 4 * (ptrdiff_t)at;
    v3 r;
    r.x = px[0] / 255.f;
    r.y = px[1] / 255.f;
    r.z = px[2] / 255.f;
    if (st) st->texel_fetches++;
    return r;
}

void rt_oracle_texel(const uint8_t *table, uint32_t w, uint32_t h, const float uv[6], float ab_l, float ac_l, float out[3])
{
    v3 r = texel(table, w, h, uv, ab_l, ac_l, 0);
    out[0] = r.x; out[1] = r.y; out[2] = r.z;
}

static inline v3 tri_vertex(const rt_oracle_scene *sc, uint32_t tri, int corner)
{
    return ld3(sc->vertex + 4 * (ptrdiff_t)sc->tri_index[4 * (ptrdiff_t)tri + corner]);
}

/* ---- secondary rays: 3-D DDA over the non-uniform grid (raytrace_opencl.c:324-401) ------------------ */
static uint32_t grid_trace(const rt_oracle_scene *sc, v3 o, v3 d, float tmin, float tmax, uint32_t excluded,
                           float *t_out, float *ab_out, float *ac_out, rt_oracle_stats *st)
{
    const int div = sc->axes_div;
    const float *bm = sc->box_min;
    const v3 lo = ld3(bm), hi = ld3(bm + 4 * (ptrdiff_t)div);
    uint32_t best = 0xffffffffu;
    int cell[3], last[3] = { -1, -1, -1 };
    v3 from = along(o, tmin, d);
    bind_in_cube(&from, d, lo, hi); /* result ignored (:354) */
    box_address(div, bm, from, cell);
    if (tmax < INFINITY) {
        v3 to = along(o, tmax, d);
        bind_in_cube(&to, d, lo, hi); /* result ignored (:360) */
        box_address(div, bm, to, last);
    }
    if (st) st->grid_rays++;
    for (;;) {
        uint32_t id = (uint32_t)(cell[0] + div * cell[1] + div * div * cell[2]);
        uint32_t i;
        *t_out = tmax; /* reset per cell (:366) */
        if (st) st->grid_cells++;
        for (i = sc->grid_start[id]; i < sc->grid_start[id + 1]; ++i) {
            uint32_t tri = sc->grid_list[i];
            if (st) st->grid_candidates++;
            if (excluded != tri) {
                float t, l1, l2;
                if (ray_triangle(o, d, tmin, *t_out, tri_vertex(sc, tri, 0), tri_vertex(sc, tri, 1), tri_vertex(sc, tri, 2), &t, &l1, &l2)) {
                    best = tri;
                    *t_out = t;
                    *ab_out = l1;
                    *ac_out = l2;
                }
            }
        }
        /* stop at the first cell that produced any hit, or at the end cell (:380-381) */
        if (best != 0xffffffffu || (cell[0] == last[0] && cell[1] == last[1] && cell[2] == last[2])) break;
        {
            /* distances measured from the ray origin, not from the clamped start (:383-385) */
            float dx = (bm[4 * (cell[0] + (0 <= d.x)) + 0] - o.x) / d.x;
            float dy = (bm[4 * (cell[1] + (0 <= d.y)) + 1] - o.y) / d.y;
            float dz = (bm[4 * (cell[2] + (0 <= d.z)) + 2] - o.z) / d.z;
            if ((dx < dy) & (dx < dz)) {
                cell[0] += (0 <= d.x) ? 1 : -1;
                if (cell[0] < 0 || div <= cell[0]) break;
            } else if (dy < dz) {
                cell[1] += (0 <= d.y) ? 1 : -1;
                if (cell[1] < 0 || div <= cell[1]) break;
            } else {
                cell[2] += (0 <= d.z) ? 1 : -1;
                if (cell[2] < 0 || div <= cell[2]) break;
            }
        }
    }
    return best;
}

uint32_t rt_oracle_grid_trace(const rt_oracle_scene *sc, const float o[3], const float d[3], float tmin, float tmax,
                              uint32_t excluded, float *t, float *ab_l, float *ac_l)
{
    return grid_trace(sc, ld3(o), ld3(d), tmin, tmax, excluded, t, ab_l, ac_l, 0);
}

/* ---- shading normal: Phong interpolation + bump (raytrace_opencl.c:195-263) ------------------------ */
static v3 shading_normal(const rt_oracle_scene *sc, v3 where, v3 ray_o, v3 ray_d, uint32_t tri, float ab_l, float ac_l,
                         rt_oracle_stats *st)
{
    const float *nrm = sc->tri_normal + 12 * (ptrdiff_t)tri;
    const v3 na = ld3(nrm), nb = ld3(nrm + 4), nc = ld3(nrm + 8);
    const int m = sc->tri_material[tri];
    const v3 a = tri_vertex(sc, tri, 0), b = tri_vertex(sc, tri, 1), c = tri_vertex(sc, tri, 2);
    const float dab = (float)sqrt((double)point_line_sq(a, b, where));
    const float dbc = (float)sqrt((double)point_line_sq(b, c, where));
    const float dca = (float)sqrt((double)point_line_sq(c, a, where));
    const float inv = 1.f / (dab + dbc + dca);
    v3 n;
    n.x = (dab * nc.x + dbc * na.x + dca * nb.x) * inv;
    n.y = (dab * nc.y + dbc * na.y + dca * nb.y) * inv;
    n.z = (dab * nc.z + dbc * na.z + dca * nb.z) * inv;

    if (0 <= m && 0 < sc->mat_size[2 * (CH_COUNT * m + CH_BUMP)]) {
        const uint32_t bw = sc->mat_size[2 * (CH_COUNT * m + CH_BUMP)], bh = sc->mat_size[2 * (CH_COUNT * m + CH_BUMP) + 1];
        const uint8_t *bump = sc->textures + 4 * (ptrdiff_t)sc->mat_start[CH_COUNT * m + CH_BUMP];
        const float *uv = sc->tri_uv + 6 * (ptrdiff_t)tri;
        const v3 tb = ld3(sc->top_to_bottom), lr = ld3(sc->left_to_right);
        float t, l1 = 0.f, l2 = 0.f;
        v3 probe, h0, hs, he;
        float xp, yp, np, li;
        h0 = texel(bump, bw, bh, uv, ab_l, ac_l, st);
        probe.x = ray_d.x + tb.x; probe.y = ray_d.y + tb.y; probe.z = ray_d.z + tb.z;
        ray_triangle(ray_o, probe, 0.f, INFINITY, a, b, c, &t, &l1, &l2);
        hs = texel(bump, bw, bh, uv, l1, l2, st);
        probe.x = ray_d.x + lr.x; probe.y = ray_d.y + lr.y; probe.z = ray_d.z + lr.z;
        ray_triangle(ray_o, probe, 0.f, INFINITY, a, b, c, &t, &l1, &l2);
        he = texel(bump, bw, bh, uv, l1, l2, st);
        xp = (float)sin((double)((he.x - h0.x) * RT_PI_F / 2.f));
        yp = (float)sin((double)((hs.x - h0.x) * RT_PI_F / 2.f));
        np = (float)cos((double)((he.x - h0.x) * RT_PI_F / 2.f)) * (float)cos((double)((hs.x - h0.x) * RT_PI_F / 2.f));
        n.x = np * n.x / sc->pixel_size_inv + xp * lr.x + yp * tb.x;
        n.y = np * n.y / sc->pixel_size_inv + xp * lr.y + yp * tb.y;
        n.z = np * n.z / sc->pixel_size_inv + xp * lr.z + yp * tb.z;
        li = 1.f / (float)sqrt((double)dot3(n, n));
        n.x *= li; n.y *= li; n.z *= li;
    }
    return n;
}

/* One queued ray (raytrace_opencl.c:461-468). */
typedef struct {
    int bounces_left;
    uint32_t excluded;
    v3 o, d, weight;
    int from_camera;
    float tmin, tmax;
} queued_ray;

#define MAX2(a, b) (((a) > (b)) ? (a) : (b)) /* raytrace.h:30 (NaN in a falls through to b) */

/* One sample of one pixel: raytrace_opencl.c:470-741, seeded as the C path does (:477-482). */
static void trace_sample(const rt_oracle_scene *sc, uint32_t pixel, uint32_t sample_id, rt_oracle_stats *st)
{
    queued_ray ring[RING];
    int head = 0, tail;
    const float px = (float)(pixel % sc->width);
    const float py = (float)(pixel / sc->width);
    uint64_t rng = (uint64_t)pixel * (uint64_t)sc->sample_count + (uint64_t)sample_id;
    const v3 lr = ld3(sc->left_to_right), tb = ld3(sc->top_to_bottom);
    v3 out = { 0.f, 0.f, 0.f };
    float k;

    /* primary ray (:490-508): LR jitter first, then TB; the direction is not normalised */
    ring[0].bounces_left = 12;
    ring[0].excluded = 0xffffffffu;
    ring[0].o = ld3(sc->eye);
    ring[0].d = ld3(sc->eye_to_top_left);
    k = px + rt_oracle_randf(&rng, 0.f, 1.f);
    ring[0].d.x += lr.x * k; ring[0].d.y += lr.y * k; ring[0].d.z += lr.z * k;
    k = py + rt_oracle_randf(&rng, 0.f, 1.f);
    ring[0].d.x += tb.x * k; ring[0].d.y += tb.y * k; ring[0].d.z += tb.z * k;
    ring[0].weight.x = ring[0].weight.y = ring[0].weight.z = 1.f;
    ring[0].from_camera = 1;
    ring[0].tmin = 0.f;
    ring[0].tmax = INFINITY;
    tail = 1;
    if (st) st->primary_samples++;

    for (; head != tail; head = (head + 1) % RING) {
        const queued_ray cur = ring[head];
        float hit_t = cur.tmax;
        uint32_t hit_tri = 0xffffffffu;
        float hit_ab = 0.f, hit_ac = 0.f;

        if (cur.from_camera) {
            /* per-pixel candidate list, running max, ties keep the earliest entry (:514-528) */
            uint32_t i;
            for (i = sc->cam_start[pixel]; i < sc->cam_end[pixel]; ++i) {
                uint32_t tri = sc->cam_list[i];
                if (st) st->primary_candidates++;
                if (cur.excluded != tri) {
                    float t, l1, l2;
                    if (ray_triangle(cur.o, cur.d, cur.tmin, hit_t, tri_vertex(sc, tri, 0), tri_vertex(sc, tri, 1), tri_vertex(sc, tri, 2), &t, &l1, &l2)) {
                        hit_t = t; hit_tri = tri; hit_ab = l1; hit_ac = l2;
                    }
                }
            }
        } else {
            hit_tri = grid_trace(sc, cur.o, cur.d, cur.tmin, cur.tmax, cur.excluded, &hit_t, &hit_ab, &hit_ac, st);
        }
        if (hit_tri == 0xffffffffu) continue;

        {
            v3 tex = { 0, 0, 0 }, transp = { 0, 0, 0 }, refl = { 0, 0, 0 }, lum = { 0, 0, 0 };
            v3 face[2] = { { 0.1f, 0.1f, 0.1f }, { 0.1f, 0.1f, 0.1f } }; /* ambient floor (:540) */
            const int m = sc->tri_material[hit_tri];
            const v3 where = along(cur.o, hit_t, cur.d);
            const v3 n = shading_normal(sc, where, cur.o, cur.d, hit_tri, hit_ab, hit_ac, st);
            uint32_t j;
            if (st) st->shaded_hits++;

            if (0 <= m) { /* :550-561 */
                const int mc = CH_COUNT * m;
                const float *uv = sc->tri_uv + 6 * (ptrdiff_t)hit_tri;
                const uint32_t *sz = sc->mat_size;
                if (0 < sz[2 * (mc + CH_COLOR)])
                    tex = texel(sc->textures + 4 * (ptrdiff_t)sc->mat_start[mc + CH_COLOR], sz[2 * (mc + CH_COLOR)], sz[2 * (mc + CH_COLOR) + 1], uv, hit_ab, hit_ac, st);
                if (0 < sz[2 * (mc + CH_TRANSPARENCY)])
                    transp = texel(sc->textures + 4 * (ptrdiff_t)sc->mat_start[mc + CH_TRANSPARENCY], sz[2 * (mc + CH_TRANSPARENCY)], sz[2 * (mc + CH_TRANSPARENCY) + 1], uv, hit_ab, hit_ac, st);
                if (0 < sz[2 * (mc + CH_REFLECTION)])
                    refl = texel(sc->textures + 4 * (ptrdiff_t)sc->mat_start[mc + CH_REFLECTION], sz[2 * (mc + CH_REFLECTION)], sz[2 * (mc + CH_REFLECTION) + 1], uv, hit_ab, hit_ac, st);
                if (0 < sz[2 * (mc + CH_LUMINANCE)])
                    lum = texel(sc->textures + 4 * (ptrdiff_t)sc->mat_start[mc + CH_LUMINANCE], sz[2 * (mc + CH_LUMINANCE)], sz[2 * (mc + CH_LUMINANCE) + 1], uv, hit_ab, hit_ac, st);
            }

            for (j = 0; j < sc->light_count; ++j) { /* :563-637 */
                v3 to_light = { 0.f, 0.f, 0.f }, atten = { 1.f, 1.f, 1.f };
                float lmin = 0.f, lmax = 0.f;
                switch (sc->light_type[j]) { /* raytrace_opencl.h:1-12 */
                case 1: case 2: case 7: case 8: case 9: { /* SPOT, SPOTRECT, TUBE, AREA, PHOTOMETRIC (:567-584) */
                    v3 r = sphere_point(&rng, sc->light_radius[j]);
                    const float *lp = sc->light_pos + 4 * (ptrdiff_t)j;
                    float inv;
                    to_light.x = r.x + lp[0] - where.x;
                    to_light.y = r.y + lp[1] - where.y;
                    to_light.z = r.z + lp[2] - where.z;
                    lmin = 0.f;
                    lmax = (float)sqrt((double)dot3(to_light, to_light));
                    inv = 1.f / lmax;
                    to_light.x *= inv; to_light.y *= inv; to_light.z *= inv;
                    break;
                }
                case 0: /* OMNI (:585-588): no shadow ray, zero direction */
                    lmin = 0.f; lmax = 0.f;
                    break;
                case 3: case 4: case 5: case 6: { /* DISTANT, PARALLEL, PARSPOT, PARSPOTRECT (:589-606) */
                    const v3 ld = ld3(sc->light_dir + 4 * (ptrdiff_t)j);
                    float spread = (float)(sin((double)((sc->light_radius[j] / 2.f) * RT_PI_F / 180.f)) * sqrt((double)dot3(ld, ld)));
                    float inv;
                    to_light = sphere_point(&rng, spread);
                    to_light.x -= ld.x; to_light.y -= ld.y; to_light.z -= ld.z;
                    inv = 1.f / (float)sqrt((double)dot3(to_light, to_light));
                    to_light.x *= inv; to_light.y *= inv; to_light.z *= inv;
                    lmin = 0.f;
                    lmax = INFINITY;
                    break;
                }
                default: break;
                }
                if (lmin < lmax) { /* shadow ray through transparent occluders (:608-627) */
                    for (;;) {
                        float t, l1, l2;
                        uint32_t occ = grid_trace(sc, where, to_light, lmin, lmax, hit_tri, &t, &l1, &l2, st);
                        v3 tr = { 0.f, 0.f, 0.f };
                        int om;
                        if (occ == 0xffffffffu) break;
                        om = sc->tri_material[occ];
                        if (0 <= om && 0 < sc->mat_size[2 * (CH_COUNT * om + CH_TRANSPARENCY)])
                            tr = texel(sc->textures + 4 * (ptrdiff_t)sc->mat_start[CH_COUNT * om + CH_TRANSPARENCY],
                                       sc->mat_size[2 * (CH_COUNT * om + CH_TRANSPARENCY)], sc->mat_size[2 * (CH_COUNT * om + CH_TRANSPARENCY) + 1],
                                       sc->tri_uv + 6 * (ptrdiff_t)occ, l1, l2, st);
                        atten.x *= tr.x; atten.y *= tr.y; atten.z *= tr.z;
                        if (!(0.f < atten.x && 0.f < atten.y && 0.f < atten.z)) break;
                        lmin = t;
                    }
                }
                { /* two-sided |N.L| with distance falloff, screen-blend into the facing side (:628-636) */
                    const float ndl = dot3(n, to_light);
                    const float mag = (float)fabs((double)ndl);
                    const int side = (int)(0.f <= ndl);
                    const float fall = (float)pow((double)0.5f, (double)(lmax / sc->light_half_att[j]));
                    const float e = mag * (fall == fall ? fall : 1.f);
                    const float *lc = sc->light_col + 4 * (ptrdiff_t)j;
                    face[side].x += (1.f - face[side].x) * atten.x * e * lc[0];
                    face[side].y += (1.f - face[side].y) * atten.y * e * lc[1];
                    face[side].z += (1.f - face[side].z) * atten.z * e * lc[2];
                }
            }

            /* emission, then lit diffuse (:639-653) */
            out.x += (1.f - out.x) * lum.x * cur.weight.x;
            out.y += (1.f - out.y) * lum.y * cur.weight.y;
            out.z += (1.f - out.z) * lum.z * cur.weight.z;
            {
                const int front = (int)(dot3(n, cur.d) <= 0.f);
                const v3 lit = face[front];
                v3 w, diffuse = { 0.f, 0.f, 0.f };
                float total;
                out.x += (1.f - out.x) * cur.weight.x * (1.f - transp.x) * tex.x * lit.x;
                out.y += (1.f - out.y) * cur.weight.y * (1.f - transp.y) * tex.y * lit.y;
                out.z += (1.f - out.z) * cur.weight.z * (1.f - transp.z) * tex.z * lit.z;

                if (cur.bounces_left <= 0) continue; /* :656 */

                total = MAX2(MAX2(refl.x + transp.x, refl.y + transp.y), refl.z + transp.z);
                if (total < 1.f) diffuse.x = diffuse.y = diffuse.z = 1.f - total;

                /* diffuse bounce (:664-683) */
                w.x = cur.weight.x * tex.x * diffuse.x;
                w.y = cur.weight.y * tex.y * diffuse.y;
                w.z = cur.weight.z * tex.z * diffuse.z;
                if (3.f / 256.f <= w.x + w.y + w.z) {
                    queued_ray *q = &ring[tail];
                    q->bounces_left = 0;
                    q->excluded = hit_tri;
                    q->o = where;
                    q->d = sphere_point(&rng, 1.f);
                    if (front != (0 <= dot3(q->d, n))) { q->d.x = -q->d.x; q->d.y = -q->d.y; q->d.z = -q->d.z; }
                    q->weight = w;
                    q->from_camera = 0;
                    q->tmin = 0.f;
                    q->tmax = INFINITY;
                    tail = (tail + 1) % RING;
                    if ((tail + 1) % RING == head) continue;
                }
                /* mirror (:686-705) */
                w.x = cur.weight.x * tex.x * refl.x;
                w.y = cur.weight.y * tex.y * refl.y;
                w.z = cur.weight.z * tex.z * refl.z;
                if (3.f / 256.f <= w.x + w.y + w.z) {
                    queued_ray *q = &ring[tail];
                    const float two = -2.f * dot3(n, cur.d);
                    q->bounces_left = cur.bounces_left - 1;
                    q->excluded = hit_tri;
                    q->o = where;
                    q->d.x = cur.d.x + two * n.x;
                    q->d.y = cur.d.y + two * n.y;
                    q->d.z = cur.d.z + two * n.z;
                    q->weight = w;
                    q->from_camera = 0;
                    q->tmin = 0.f;
                    q->tmax = INFINITY;
                    tail = (tail + 1) % RING;
                    if ((tail + 1) % RING == head) continue;
                }
                /* see-through continuation of the same ray (:707-722) */
                w.x = cur.weight.x * tex.x * transp.x;
                w.y = cur.weight.y * tex.y * transp.y;
                w.z = cur.weight.z * tex.z * transp.z;
                if (3.f / 256.f <= w.x + w.y + w.z) {
                    queued_ray *q = &ring[tail];
                    q->bounces_left = cur.bounces_left - 1;
                    q->excluded = hit_tri;
                    q->o = cur.o;
                    q->d = cur.d;
                    q->weight = w;
                    q->from_camera = cur.from_camera;
                    q->tmin = hit_t;
                    q->tmax = INFINITY;
                    tail = (tail + 1) % RING;
                    if ((tail + 1) % RING == head) continue;
                }
            }
        }
    }

    { /* saturating u16 accumulate, addend truncated per sample (:726-741) */
        const float scale = (float)(0xFFFF) / (float)sc->sample_count;
        int v;
        v = (int)sc->out_r[pixel] + trunc_x86(out.x * scale); if (v < 0) v = 0; if (0xFFFF < v) v = 0xFFFF; sc->out_r[pixel] = (uint16_t)v;
        v = (int)sc->out_g[pixel] + trunc_x86(out.y * scale); if (v < 0) v = 0; if (0xFFFF < v) v = 0xFFFF; sc->out_g[pixel] = (uint16_t)v;
        v = (int)sc->out_b[pixel] + trunc_x86(out.z * scale); if (v < 0) v = 0; if (0xFFFF < v) v = 0xFFFF; sc->out_b[pixel] = (uint16_t)v;
    }
}

static void add_stats(rt_oracle_stats *dst, const rt_oracle_stats *src)
{
    dst->primary_samples += src->primary_samples;
    dst->primary_candidates += src->primary_candidates;
    dst->grid_rays += src->grid_rays;
    dst->grid_cells += src->grid_cells;
    dst->grid_candidates += src->grid_candidates;
    dst->shaded_hits += src->shaded_hits;
    dst->texel_fetches += src->texel_fetches;
}

/* raytrace.c:612-653: for each pixel, samples 1..S in order */
int rt_oracle_render(const rt_oracle_scene *sc, uint32_t first_pixel, uint32_t pixel_count, int threads, rt_oracle_stats *stats)
{
    const int64_t n = (int64_t)pixel_count;
    if (stats) memset(stats, 0, sizeof *stats);
    if (threads < 1) threads = 1;
#pragma omp parallel num_threads(threads)
    {
        rt_oracle_stats local;
        int64_t i;
        memset(&local, 0, sizeof local);
#pragma omp for schedule(dynamic, 256)
        for (i = 0; i < n; ++i) {
            uint32_t pixel = first_pixel + (uint32_t)i, s;
            for (s = 1; s <= sc->sample_count; ++s) trace_sample(sc, pixel, s, stats ? &local : 0);
        }
        if (stats) {
#pragma omp critical
            add_stats(stats, &local);
        }
    }
    return 1;
}
