/*
 * rt_oracle_builders.h -- CPU oracle for the camera-list and scene-grid builders (TEST INFRASTRUCTURE ONLY).
 * See rt_oracle_builders.c: an independent serial restatement of source/util/trianglelist.cpp:74-90,131-217,381-503,
 * 520-626,655-737.  PARITY UNPINNED against the reference (that file cannot be compiled here and the reference holds no
 * builder fixtures); it exists so the product's host and device builders are checked against code they do not share.
 * Layouts: vertex = 4 floats per vertex, tri_index = 4 ints per triangle, box_min = 4 floats x 257 (the ABI's padded types).
 * Output arrays are malloc'ed; release them with rt_oracle_builders_free.
 */
#ifndef RT_ORACLE_BUILDERS_H
#define RT_ORACLE_BUILDERS_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* CameraTriangleList::New (trianglelist.cpp:520-626): start/end have width*height entries (ranges may alias), list is the
 * de-duplicated candidate array, *out_list_size its length. */
int rt_oracle_build_camera_list(uint32_t width, uint32_t height, const float eye[3], const float top_left[3], const float lr[3],
                                const float tb[3], float pixel_size_inv, uint32_t triangle_count, const float *vertex,
                                const int32_t *tri_index, uint32_t **out_start, uint32_t **out_end, uint32_t **out_list,
                                uint64_t *out_list_size);

/* SceneTriangleList::New (trianglelist.cpp:655-737): box_min receives the 257 split planes, start has 256^3+1 entries. */
int rt_oracle_build_scene_grid(uint32_t vertex_count, uint32_t triangle_count, const float *vertex, const int32_t *tri_index,
                               float *box_min, uint32_t **out_start, uint32_t **out_list, uint64_t *out_list_size);

/* function-level entry points */
void rt_oracle_camera_position(const float eye[3], const float top_left[3], const float lr[3], const float tb[3], float pixel_size_inv,
                               const float v[3], float out[2]);                                        /* :74-90 */
int  rt_oracle_box_meets_triangle(const float lo[3], const float hi[3], const float a[3], const float b[3], const float c[3]); /* :433-449 */

void rt_oracle_builders_free(void *p);

#ifdef __cplusplus
}
#endif
#endif
