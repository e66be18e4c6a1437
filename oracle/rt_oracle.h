/*
 * rt_oracle.h -- CPU oracle for the opencl_render hot path (TEST INFRASTRUCTURE ONLY).
 *
 * This is a plain-C restatement of the reference's deterministic single-thread C path
 * (reference: source/opencl/raytrace_opencl.c:1-742 driven by source/opencl/raytrace.c:604-655).
 * It exists to CHECK the HIP product; nothing under opencl_render_amd/ may include, link or call it.
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg use it.
 *
 * Pinning: the reference ships no golden vectors.  The restatement is pinned against
 * oracle/_ref/libref_kernel.so -- the reference's own kernel file compiled in place (see oracle/Makefile) --
 * and against the fixtures under tests/golden/ that were minted from it (tests/golden/make_golden.py).
 *
 * Layouts are the reference ABI's (source/3rdparty/opencl-1.2/include/CL/cl_platform.h:501,725,1025):
 * float3/int3 occupy 16 bytes (lane 3 is padding, never read), float2/uint2 8 bytes, uchar3 4 bytes.
 */
#ifndef RT_ORACLE_H
#define RT_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct rt_oracle_scene {
    /* camera (raytrace.h:61-66) */
    uint32_t width, height;
    float eye[4], eye_to_top_left[4], left_to_right[4], top_to_bottom[4];
    float pixel_size_inv;
    /* per-pixel candidate lists (raytrace.h:68-71); start/end have width*height entries */
    const uint32_t *cam_start, *cam_end, *cam_list;
    uint32_t sample_count;
    /* geometry (raytrace.h:75-82) */
    const float *vertex;        /* 4 floats per vertex */
    uint32_t triangle_count;
    const int32_t *tri_index;   /* 4 ints per triangle */
    const int32_t *tri_material;
    const float *tri_uv;        /* 2 floats x 3 per triangle */
    const float *tri_normal;    /* 4 floats x 3 per triangle */
    /* non-uniform grid (raytrace.h:84-87) */
    int32_t axes_div;
    const float *box_min;       /* 4 floats x (axes_div+1) */
    const uint32_t *grid_start; /* axes_div^3 + 1 */
    const uint32_t *grid_list;
    /* materials (raytrace.h:89-94) */
    const uint32_t *mat_size;   /* 2 uints x 5 per material */
    const int32_t *mat_start;   /* 5 per material (+1) */
    const uint8_t *textures;    /* 4 bytes per texel */
    /* lights (raytrace.h:96-102) */
    uint32_t light_count;
    const int32_t *light_type;
    const float *light_pos, *light_dir, *light_col; /* 4 floats each */
    const float *light_radius, *light_half_att;
    /* outputs (raytrace.h:104-106): accumulated in place */
    uint16_t *out_r, *out_g, *out_b;
} rt_oracle_scene;

/* Work counters for the algorithmic-byte model of SURVEY.md section 8(d). */
typedef struct rt_oracle_stats {
    uint64_t primary_samples;   /* P*S */
    uint64_t primary_candidates;/* sum of K_p over camera-list scans (incl. continuation rays) */
    uint64_t grid_rays;         /* calls of the grid traversal */
    uint64_t grid_cells;        /* cells visited */
    uint64_t grid_candidates;   /* list entries scanned */
    uint64_t shaded_hits;
    uint64_t texel_fetches;
} rt_oracle_stats;

/* Renders pixels [first_pixel, first_pixel+pixel_count) with all samples, in pixel order
 * (raytrace.c:612-653).  threads<=1: the reference's single-thread order.  threads>1 splits the pixel
 * range over OpenMP threads (pixels are independent, so the planes are identical).  stats may be NULL. */
int rt_oracle_render(const rt_oracle_scene *sc, uint32_t first_pixel, uint32_t pixel_count,
                     int threads, rt_oracle_stats *stats);

/* Function-level entry points for known-answer tests (same maths as the reference helpers). */
float rt_oracle_randf(uint64_t *state, float lo, float hi);                     /* :12-23  */
void  rt_oracle_sphere_point(uint64_t *state, float radius, float out[3]);      /* :30-45  */
float rt_oracle_positive_modf(float v);                                         /* :25-28  */
int   rt_oracle_ray_triangle(const float o[3], const float d[3], float tmin, float tmax,
                             const float a[3], const float b[3], const float c[3],
                             float *t, float *ab_l, float *ac_l);               /* :124-172 */
void  rt_oracle_box_address(int axes_div, const float *box_min, const float p[3], int out[3]); /* :174-193 */
int   rt_oracle_bind_in_cube(float p[3], const float d[3], const float lo[3], const float hi[3]); /* :265-322 */
float rt_oracle_point_line_sq(const float o[3], const float e[3], const float p[3]); /* :83-101 */
uint32_t rt_oracle_grid_trace(const rt_oracle_scene *sc, const float o[3], const float d[3],
                              float tmin, float tmax, uint32_t excluded,
                              float *t, float *ab_l, float *ac_l);              /* :324-401 */
void  rt_oracle_texel(const uint8_t *table, uint32_t w, uint32_t h, const float uv[6],
                      float ab_l, float ac_l, float out[3]);                    /* :103-122 */

#ifdef __cplusplus
}
#endif
#endif
