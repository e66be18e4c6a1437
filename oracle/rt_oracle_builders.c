/*
 * rt_oracle_builders.c -- CPU oracle for the two list builders (TEST INFRASTRUCTURE ONLY; never linked into the product).
 *
 * A serial, plain-C restatement of the reference's acceleration-structure builders, written from their behaviour:
 *   camera candidate lists   CameraTriangleList::New   source/util/trianglelist.cpp:520-626
 *       GetCameraPosition :74-90, FillRectangle :131-217, sort of pixel*T+triangle keys :565, CSR :569-578,
 *       neighbour de-duplication :580-613
 *   scene grid               SceneTriangleList::New    source/util/trianglelist.cpp:655-737
 *       quantile split planes :657-678, FillCube :452-503, BoxIntersectsTriangle :433-449, Cull :381-430,
 *       sort of cell*T+triangle keys :707, prefix sums :709-719
 * It follows the reference's own strategy (one 64-bit key per (pixel|cell, triangle) pair, sorted) and shares no code with
 * the product's builders (opencl_render_amd/csrc/rt_build_shared.h is deliberately NOT included), so a mis-restated edge
 * test there shows up as a list difference here.
 *
 * PARITY UNPINNED against the reference: trianglelist.cpp cannot be compiled in this container (it includes the Maxon SDK's
 * c4d.h; writing a stand-in header is not allowed) and the reference holds no fixtures for its builders.  This file is a
 * second, independent reading of the same source -- it makes the product's builder tests more than a self-comparison, it
 * does not make them reference-pinned.
 *
 * Arithmetic notes (what "the same result" depends on):
 *   * dot = (a0*b0 + a1*b1) + a2*b2 and cross per source/opencl/raytrace.c:18-27; fp32, no contraction (Makefile flags).
 *   * float/double -> unsigned conversions are x86-64's: cvttss2si/cvttsd2si into a 64-bit register, low 32 bits kept
 *     (what MSVC and gcc emit for (cl_uint)value); NaN and out-of-range give 0 in the low word.
 *   * fmin/fmax/floor are the double functions applied to promoted floats (:153-157,:160-161,:167).
 *   * the quantile index (i*(vertexCount-1))/256 is computed in 32-bit unsigned arithmetic like the reference (:669); it
 *     wraps above 2^24 vertices -- the tests stay below that.
 *   * what is NOT taken over: the 2 MiB memset of the visited set per triangle (:457; only the bits a fill set are cleared)
 *     and the 2 GiB key scratch (:522,:681; the pairs are counted first).  Neither changes a list.
 */
#include "rt_oracle_builders.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

#define GRID 256 /* AXES_DIVISION, trianglelist.h:110 */

typedef struct { float x, y, z; } p3;
typedef struct { float x, y; } p2;

static float dotp(p3 a, p3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
static p3 crossp(p3 a, p3 b)
{
    p3 r;
    r.x = a.y * b.z - a.z * b.y;
    r.y = a.z * b.x - a.x * b.z;
    r.z = a.x * b.y - a.y * b.x;
    return r;
}
static p3 vtx(const float *vertex, int32_t i) { p3 r; r.x = vertex[4 * (size_t)i]; r.y = vertex[4 * (size_t)i + 1]; r.z = vertex[4 * (size_t)i + 2]; return r; }

/* x86-64 conversion of a real to a 32-bit unsigned as the compilers emit it */
static uint32_t as_u32(double v)
{
    if (!(v > -9223372036854775808.0 && v < 9223372036854775808.0)) return 0u; /* the "integer indefinite" value: low word 0 */
    return (uint32_t)(uint64_t)(int64_t)v;
}

static int cmp_u64(const void *a, const void *b)
{
    const uint64_t x = *(const uint64_t *)a, y = *(const uint64_t *)b;
    return (x > y) - (x < y);
}
static int cmp_f32(const void *a, const void *b)
{
    const float x = *(const float *)a, y = *(const float *)b;
    return (x > y) - (x < y);
}

/* ---- camera lists ------------------------------------------------------------------------------------------- */

typedef struct { p3 eye, top_left, lr, tb; float inv; } camera;

/* :74-90 -- where the ray from the eye through v meets the image plane, in pixel units */
static p2 project(const camera *c, p3 v)
{
    const float inv_sq = c->inv * c->inv;
    p3 to_v, plane_n, on_plane;
    p2 r;
    float k;
    to_v.x = v.x - c->eye.x; to_v.y = v.y - c->eye.y; to_v.z = v.z - c->eye.z;
    plane_n = crossp(c->lr, c->tb);
    k = dotp(c->top_left, plane_n) / dotp(to_v, plane_n);
    on_plane.x = k * to_v.x - c->top_left.x;
    on_plane.y = k * to_v.y - c->top_left.y;
    on_plane.z = k * to_v.z - c->top_left.z;
    r.x = dotp(c->lr, on_plane) * inv_sq;
    r.y = dotp(c->tb, on_plane) * inv_sq;
    return r;
}

void rt_oracle_camera_position(const float eye[3], const float top_left[3], const float lr[3], const float tb[3], float pixel_size_inv,
                               const float v[3], float out[2])
{
    camera c;
    p3 p;
    p2 r;
    c.eye.x = eye[0]; c.eye.y = eye[1]; c.eye.z = eye[2];
    c.top_left.x = top_left[0]; c.top_left.y = top_left[1]; c.top_left.z = top_left[2];
    c.lr.x = lr[0]; c.lr.y = lr[1]; c.lr.z = lr[2];
    c.tb.x = tb[0]; c.tb.y = tb[1]; c.tb.z = tb[2];
    c.inv = pixel_size_inv;
    p.x = v[0]; p.y = v[1]; p.z = v[2];
    r = project(&c, p);
    out[0] = r.x; out[1] = r.y;
}

/* One edge P->Q against the pixel (x, y): does the edge cross one of the pixel's four border lines inside the pixel?
 * (:169-194: the crossing with the row line y and with the row line y+1 must fall in column x, the crossing with the column
 * line x / x+1 must fall in row y; "between P and Q" is the sign test on the product of differences.) */
static int edge_touches_pixel(p2 p, p2 q, float slope_xy, float slope_yx, uint32_t x, uint32_t y)
{
    const float at_row = p.x + ((float)y - p.y) * slope_xy;      /* x where the edge meets the line through row y */
    const float at_col = p.y + ((float)x - p.x) * slope_yx;      /* y where the edge meets the line through column x */
    const float at_next_row = at_row + slope_xy;
    const float at_next_col = at_col + slope_yx;
    int hit = 0;
    hit |= (0.f <= (p.x - at_row) * (at_row - q.x)) & (x == as_u32(at_row));
    hit |= (0.f <= (p.x - at_next_row) * (at_next_row - q.x)) & (x == as_u32(at_next_row));
    hit |= (0.f <= (p.y - at_col) * (at_col - q.y)) & (y == as_u32(at_col));
    hit |= (0.f <= (p.y - at_next_col) * (at_next_col - q.y)) & (y == as_u32(at_next_col));
    return hit;
}

/* :131-217.  Appends pixel*T + tri for every pixel that receives the triangle (keys == NULL: only counts). */
static size_t rasterise(uint32_t w, uint32_t h, uint64_t *keys, p2 a, p2 b, p2 c, uint32_t tri, uint32_t tri_count)
{
    size_t n = 0;
    p2 ab, bc, ca;
    float ab_xy, ab_yx, bc_xy, bc_yx, ca_xy, ca_yx;
    uint32_t x0, y0, x1, y1, x, y;
    const uint32_t a_col = as_u32(floor((double)a.x)), a_row = as_u32(floor((double)a.y));

    ab.x = b.x - a.x; ab.y = b.y - a.y;
    bc.x = c.x - b.x; bc.y = c.y - b.y;
    ca.x = a.x - c.x; ca.y = a.y - c.y;
    /* dx/dy of every edge and its reciprocal; an axis-parallel edge gives inf/0/NaN here and fails the tests by design (:143) */
    ab_xy = ab.x / ab.y; ab_yx = 1.f / ab_xy;
    bc_xy = bc.x / bc.y; bc_yx = 1.f / bc_xy;
    ca_xy = ca.x / ca.y; ca_yx = 1.f / ca_xy;
    /* bounding rectangle, clipped to the image (:153-157) */
    x0 = as_u32(fmax(0.f, fmin(fmin(a.x, b.x), fmin(c.x, (float)(w - 1)))));
    y0 = as_u32(fmax(0.f, fmin(fmin(a.y, b.y), fmin(c.y, (float)(h - 1)))));
    x1 = as_u32(fmin((float)(w - 1), fmax(fmax(a.x, b.x), fmax(c.x, 0.f))));
    y1 = as_u32(fmin((float)(h - 1), fmax(fmax(a.y, b.y), fmax(c.y, 0.f))));

    /* the pixel vertex a falls into always receives the triangle if it is on screen (:160-162) */
    if (0.f <= a.x && a.x < (float)w && 0.f <= a.y && a.y < (float)h) {
        if (keys) keys[n] = ((uint64_t)floor((double)a.x) + (uint64_t)floor((double)a.y) * (uint64_t)w) * (uint64_t)tri_count + (uint64_t)tri;
        ++n;
    }
    for (x = x0; x <= x1; ++x) {
        for (y = y0; y <= y1; ++y) {
            int take;
            if (x == a_col && y == a_row) continue; /* handled above (:167) */
            take = edge_touches_pixel(a, b, ab_xy, ab_yx, x, y) | edge_touches_pixel(b, c, bc_xy, bc_yx, x, y) |
                   edge_touches_pixel(c, a, ca_xy, ca_yx, x, y);
            if (!take) {
                /* no edge through the pixel: the pixel's corner lies on the same side of all three edges (:197-211) */
                const float ax = (float)x - a.x, ay = (float)y - a.y;
                const float bx = (float)x - b.x, by = (float)y - b.y;
                const float cx = (float)x - c.x, cy = (float)y - c.y;
                const float side_ab = ab.x * ay - ab.y * ax;
                const float side_bc = bc.x * by - bc.y * bx;
                const float side_ca = ca.x * cy - ca.y * cx;
                take = (0 <= side_ab * side_bc) & (0 <= side_bc * side_ca);
            }
            if (take) {
                if (keys) keys[n] = ((uint64_t)x + (uint64_t)y * (uint64_t)w) * (uint64_t)tri_count + (uint64_t)tri;
                ++n;
            }
            if (y == 0xffffffffu) break; /* y1 can only be that large through a conversion wrap; keeps the loop finite */
        }
        if (x == 0xffffffffu) break;
    }
    return n;
}

int rt_oracle_build_camera_list(uint32_t width, uint32_t height, const float eye[3], const float top_left[3], const float lr[3],
                                const float tb[3], float pixel_size_inv, uint32_t triangle_count, const float *vertex,
                                const int32_t *tri_index, uint32_t **out_start, uint32_t **out_end, uint32_t **out_list,
                                uint64_t *out_list_size)
{
    const uint64_t pixels = (uint64_t)width * height;
    camera cam;
    size_t total = 0, at = 0, i;
    uint64_t *keys;
    uint32_t *start, *end, *list, t;
    uint64_t p, squeezed = 0;

    cam.eye.x = eye[0]; cam.eye.y = eye[1]; cam.eye.z = eye[2];
    cam.top_left.x = top_left[0]; cam.top_left.y = top_left[1]; cam.top_left.z = top_left[2];
    cam.lr.x = lr[0]; cam.lr.y = lr[1]; cam.lr.z = lr[2];
    cam.tb.x = tb[0]; cam.tb.y = tb[1]; cam.tb.z = tb[2];
    cam.inv = pixel_size_inv;

    /* the reference fills a fixed scratch and re-runs on overflow (:551-563); counting first gives the same pairs */
    for (t = 0; t < triangle_count; ++t) {
        const int32_t *vi = tri_index + 4 * (size_t)t;
        total += rasterise(width, height, NULL, project(&cam, vtx(vertex, vi[0])), project(&cam, vtx(vertex, vi[1])),
                           project(&cam, vtx(vertex, vi[2])), t, triangle_count);
    }
    keys = (uint64_t *)malloc((total ? total : 1) * sizeof *keys);
    start = (uint32_t *)calloc(pixels ? pixels : 1, sizeof *start);
    end = (uint32_t *)calloc(pixels ? pixels : 1, sizeof *end);
    list = (uint32_t *)malloc((total ? total : 1) * sizeof *list);
    if (!keys || !start || !end || !list) { free(keys); free(start); free(end); free(list); return -1; }
    for (t = 0; t < triangle_count; ++t) {
        const int32_t *vi = tri_index + 4 * (size_t)t;
        at += rasterise(width, height, keys + at, project(&cam, vtx(vertex, vi[0])), project(&cam, vtx(vertex, vi[1])),
                        project(&cam, vtx(vertex, vi[2])), t, triangle_count);
    }
    qsort(keys, total, sizeof *keys, cmp_u64); /* :565 -- pixel-major, ascending triangle inside a pixel */

    /* :567-578 -- split the keys, count per pixel, running sums */
    for (i = 0; i < total; ++i) {
        list[i] = (uint32_t)(keys[i] % triangle_count);
        ++end[(uint32_t)(keys[i] / triangle_count)];
    }
    for (p = 1; p < pixels; ++p) {
        start[p] = end[p - 1];
        end[p] += start[p];
    }
    free(keys);

    /* :580-613 -- a pixel whose list equals that of its left neighbour, else of the neighbour above, shares its storage; the
     * list is closed up as the scan goes, so the neighbours' ranges are already the final ones */
    for (p = 0; p < pixels; ++p) {
        const uint32_t col = (uint32_t)(p % width), row = (uint32_t)(p / width);
        const uint32_t n = end[p] - start[p];
        int shared = 0;
        memmove(list + (start[p] - squeezed), list + start[p], (size_t)n * sizeof *list);
        start[p] -= (uint32_t)squeezed;
        end[p] -= (uint32_t)squeezed;
        if (col > 0) {
            const uint64_t q = p - 1;
            if (n == end[q] - start[q] && 0 == memcmp(list + start[q], list + start[p], (size_t)n * sizeof *list)) {
                squeezed += n;
                start[p] = start[q];
                end[p] = end[q];
                shared = 1;
            }
        }
        if (row > 0 && !shared) {
            const uint64_t q = p - width;
            if (n == end[q] - start[q] && 0 == memcmp(list + start[q], list + start[p], (size_t)n * sizeof *list)) {
                squeezed += n;
                start[p] = start[q];
                end[p] = end[q];
            }
        }
    }
    *out_start = start; *out_end = end; *out_list = list; *out_list_size = (uint64_t)total - squeezed;
    return 0;
}

/* ---- scene grid --------------------------------------------------------------------------------------------------- */

/* :381-430 -- clip a convex polygon against the plane coordinate[dim] = limit, in place: first a point is inserted on every
 * edge that crosses the plane, then every ORIGINAL point on the wrong side is dropped (inserted points always stay). */
static int clip(int keep_below, float limit, int dim, int *count, float poly[16][3])
{
    int inserted[16] = { 0 };
    int i, j;
    for (i = 0; i < *count; ++i) {
        const int next = (i + 1) % *count;
        const float here = limit - poly[i][dim];
        const float there = limit - poly[next][dim];
        if (here * there < 0.f) {
            float edge[3], part;
            const int slot = i + 1;
            edge[0] = poly[next][0] - poly[i][0];
            edge[1] = poly[next][1] - poly[i][1];
            edge[2] = poly[next][2] - poly[i][2];
            part = here / edge[dim];
            for (j = (*count)++; slot < j; --j) memcpy(poly[j], poly[j - 1], sizeof poly[0]);
            poly[slot][0] = poly[i][0] + part * edge[0];
            poly[slot][1] = poly[i][1] + part * edge[1];
            poly[slot][2] = poly[i][2] + part * edge[2];
            inserted[slot] = 1;
            i = slot; /* the loop's ++i steps over the new point */
        }
    }
    for (i = 0; i < *count; ++i) {
        const int wrong_side = keep_below ? (limit < poly[i][dim]) : (poly[i][dim] < limit);
        if (!inserted[i] && wrong_side) {
            const int last = (*count)--;
            for (j = i + 1; j < last; ++j) {
                memcpy(poly[j - 1], poly[j], sizeof poly[0]);
                inserted[j - 1] = inserted[j];
            }
            --i;
        }
    }
    return 0 < *count;
}

/* :433-449 -- lower planes x, y, z, then upper planes x, y, z; stops at the first clip that leaves nothing */
static int box_meets_triangle(const float lo[3], const float hi[3], p3 a, p3 b, p3 c)
{
    float poly[16][3];
    int n = 3;
    poly[0][0] = a.x; poly[0][1] = a.y; poly[0][2] = a.z;
    poly[1][0] = b.x; poly[1][1] = b.y; poly[1][2] = b.z;
    poly[2][0] = c.x; poly[2][1] = c.y; poly[2][2] = c.z;
    return clip(0, lo[0], 0, &n, poly) && clip(0, lo[1], 1, &n, poly) && clip(0, lo[2], 2, &n, poly) &&
           clip(1, hi[0], 0, &n, poly) && clip(1, hi[1], 1, &n, poly) && clip(1, hi[2], 2, &n, poly);
}

int rt_oracle_box_meets_triangle(const float lo[3], const float hi[3], const float a[3], const float b[3], const float c[3])
{
    p3 pa, pb, pc;
    pa.x = a[0]; pa.y = a[1]; pa.z = a[2];
    pb.x = b[0]; pb.y = b[1]; pb.z = b[2];
    pc.x = c[0]; pc.y = c[1]; pc.z = c[2];
    return box_meets_triangle(lo, hi, pa, pb, pc);
}

/* raytrace_opencl.c:174-193 -- per axis, the last plane strictly below the coordinate (8 halvings) */
static void cell_of(const float *planes /* [257][4] */, p3 p, int cell[3])
{
    int half;
    cell[0] = cell[1] = cell[2] = 0;
    for (half = GRID / 2; half >= 1; half /= 2) {
        if (planes[4 * (cell[0] + half) + 0] < p.x) cell[0] += half;
        if (planes[4 * (cell[1] + half) + 1] < p.y) cell[1] += half;
        if (planes[4 * (cell[2] + half) + 2] < p.z) cell[2] += half;
    }
}

/* :452-503 -- the face-connected set of cells whose box meets the triangle, grown from the cell of vertex a (which is taken
 * without a test).  `seen` is a GRID^3-bit set, all zero on entry and on exit; cells[] receives the ids (any order). */
static size_t flood(const float *planes, uint8_t *seen, uint32_t *cells, p3 a, p3 b, p3 c)
{
    size_t done = 0, have = 0, k;
    int cell[3];
    cell_of(planes, a, cell);
    {
        const uint32_t id = (uint32_t)cell[0] + (uint32_t)cell[1] * GRID + (uint32_t)cell[2] * GRID * GRID;
        seen[id >> 3] |= (uint8_t)(1u << (id & 7u));
        cells[have++] = id;
    }
    while (done < have) {
        const uint32_t id = cells[done++];
        float lo[3], hi[3];
        int axis, dir;
        cell[0] = (int)(id % GRID); cell[1] = (int)((id / GRID) % GRID); cell[2] = (int)(id / (GRID * GRID));
        for (axis = 0; axis < 3; ++axis) { lo[axis] = planes[4 * cell[axis] + axis]; hi[axis] = planes[4 * (cell[axis] + 1) + axis]; }
        for (axis = 0; axis < 3; ++axis) {
            for (dir = -1; dir <= 1; dir += 2) {
                const int moved = cell[axis] + dir;
                if (0 <= moved && moved < GRID) {
                    int nb[3];
                    uint32_t nid;
                    nb[0] = cell[0]; nb[1] = cell[1]; nb[2] = cell[2];
                    nb[axis] = moved;
                    nid = (uint32_t)nb[0] + (uint32_t)nb[1] * GRID + (uint32_t)nb[2] * GRID * GRID;
                    if (!(seen[nid >> 3] & (1u << (nid & 7u)))) {
                        /* the neighbour's box: this cell's, with the bounds of the moved axis replaced (:483-484) */
                        lo[axis] = planes[4 * moved + axis];
                        hi[axis] = planes[4 * (moved + 1) + axis];
                        if (box_meets_triangle(lo, hi, a, b, c)) {
                            seen[nid >> 3] |= (uint8_t)(1u << (nid & 7u));
                            cells[have++] = nid;
                        }
                    }
                }
            }
            lo[axis] = planes[4 * cell[axis] + axis];
            hi[axis] = planes[4 * (cell[axis] + 1) + axis];
        }
    }
    for (k = 0; k < have; ++k) seen[cells[k] >> 3] = 0; /* the reference clears the whole set before every triangle (:457) */
    return have;
}

int rt_oracle_build_scene_grid(uint32_t vertex_count, uint32_t triangle_count, const float *vertex, const int32_t *tri_index,
                               float *box_min, uint32_t **out_start, uint32_t **out_list, uint64_t *out_list_size)
{
    const size_t cells_total = (size_t)GRID * GRID * GRID;
    uint8_t *seen;
    uint32_t *cells, *start, *list, t;
    uint64_t *keys = NULL;
    size_t cap = 0, total = 0, i;
    int axis, k;

    memset(box_min, 0, sizeof(float) * 4 * (GRID + 1));
    /* :657-678 -- per axis: sort the coordinates, put plane i midway between the vertices around quantile i/256 */
    if (vertex_count > 0) {
        float *val = (float *)malloc((size_t)vertex_count * sizeof *val);
        if (!val) return -1;
        for (axis = 0; axis < 3; ++axis) {
            uint32_t v;
            for (v = 0; v < vertex_count; ++v) val[v] = vertex[4 * (size_t)v + axis];
            qsort(val, vertex_count, sizeof *val, cmp_f32);
            for (k = 0; k <= GRID; ++k) {
                const uint32_t index = ((uint32_t)k * (vertex_count - 1u)) / (uint32_t)GRID; /* 32-bit, as :669 */
                if (0 < index && index < vertex_count) box_min[4 * k + axis] = (val[index] + val[index - 1]) / 2.f;
                else box_min[4 * k + axis] = val[index];
            }
        }
        free(val);
    }

    seen = (uint8_t *)calloc(cells_total / 8, 1);
    cells = (uint32_t *)malloc(cells_total * sizeof *cells);
    start = (uint32_t *)calloc(cells_total + 1, sizeof *start);
    if (!seen || !cells || !start) { free(seen); free(cells); free(start); return -1; }
    for (t = 0; t < triangle_count; ++t) {
        const int32_t *vi = tri_index + 4 * (size_t)t;
        const size_t n = flood(box_min, seen, cells, vtx(vertex, vi[0]), vtx(vertex, vi[1]), vtx(vertex, vi[2]));
        if (total + n > cap) {
            uint64_t *grown;
            cap = (total + n) * 2 + 1024;
            grown = (uint64_t *)realloc(keys, cap * sizeof *keys);
            if (!grown) { free(keys); free(seen); free(cells); free(start); return -1; }
            keys = grown;
        }
        for (i = 0; i < n; ++i) keys[total + i] = (uint64_t)cells[i] * (uint64_t)triangle_count + (uint64_t)t; /* :462 */
        total += n;
    }
    free(seen);
    free(cells);
    if (total) qsort(keys, total, sizeof *keys, cmp_u64); /* :707 -- cell-major, ascending triangle inside a cell */
    list = (uint32_t *)malloc((total ? total : 1) * sizeof *list);
    if (!list) { free(keys); free(start); return -1; }
    for (i = 0; i < total; ++i) { /* :711-716 */
        list[i] = (uint32_t)(keys[i] % triangle_count);
        ++start[(size_t)(keys[i] / triangle_count) + 1];
    }
    for (i = 1; i <= cells_total; ++i) start[i] += start[i - 1]; /* :717-719 */
    free(keys);
    *out_start = start; *out_list = list; *out_list_size = total;
    return 0;
}

void rt_oracle_builders_free(void *p) { free(p); }
