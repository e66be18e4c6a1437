// rt_kat.hip -- TEST-ONLY entry point: runs the device-side building blocks of rt_devfuncs.h over caller-provided vectors,
// one item per thread, so the function-level known answers minted from the reference kernel (tests/golden/kat.npz) are
// checked on the GPU itself and not only against the CPU oracle.  Nothing here is on the render path.
//
//   op                in (per item)                                   out (per item)
//   RT_KAT_RANDF      u64 seed                                        16 x f32 draws (alternating randF(0,1), randF(-1,1)), 16 x u64 states   (:12-23)
//   RT_KAT_SPHERE     u64 seed, f32 radius, pad                       f32 x,y,z, pad, u64 state                                               (:30-45)
//   RT_KAT_PMODF      f32                                             f32                                                                     (:25-28)
//   RT_KAT_TRI        o3 d3 a3 b3 c3 tmin tmax (17 x f32)             hit (0/1), t, abL, acL (abL/acL 0 when the reference does not write them) (:124-172)
//   RT_KAT_PLINE      origin3 destination3 point3                     f32                                                                     (:83-101)
//   RT_KAT_BOX        p3 (table = split planes, 3 x 257 f32 SoA)      i32 cx, cy, cz                                                          (:174-193)
//   RT_KAT_BIND       p3 d3 lo3 hi3                                   p3 after the clamps                                                     (:265-322)
//   RT_KAT_POW        f32 x                                           (float)pow(0.5f, x), NaN -> 1                                           (:631-632)
//   RT_KAT_QUOTIENT   f32 plane, o, d                                 f32 (plane-o)/d as the compiler divides, f32 the walk's short sequence, u32 tame (:383-385)
#include "rt_devfuncs.h"
#include "raytrace_hip.h"

#include <hip/hip_runtime_api.h>

#include <vector>

namespace {

__global__ __launch_bounds__(256) void rt_kat_kernel(int op, uint32_t count, const unsigned char *__restrict__ in, uint32_t inStride,
                                                     unsigned char *__restrict__ out, uint32_t outStride, const float *__restrict__ table)
{
    __shared__ Shared sh;
    if (op == RT_KAT_BOX) {
        for (int i = threadIdx.x; i < 3 * (RT_GRID_DIV + 1); i += 256) (&sh.planes[0][0])[i] = table[i];
    }
    __syncthreads();
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i >= count) return;
    const float *fi = reinterpret_cast<const float *>(in + (size_t)i * inStride);
    float *fo = reinterpret_cast<float *>(out + (size_t)i * outStride);
    switch (op) {
    case RT_KAT_RANDF: {
        uint64_t s = *reinterpret_cast<const uint64_t *>(fi);
        uint64_t *states = reinterpret_cast<uint64_t *>(fo + 16);
        for (int j = 0; j < 16; ++j) {
            fo[j] = (j & 1) ? rand11(s) : rand01(s);
            states[j] = s;
        }
        break;
    }
    case RT_KAT_SPHERE: {
        uint64_t s = *reinterpret_cast<const uint64_t *>(fi);
        const V3 p = sphere_point(s, fi[2]);
        fo[0] = p.x; fo[1] = p.y; fo[2] = p.z; fo[3] = 0.f;
        *reinterpret_cast<uint64_t *>(fo + 4) = s;
        break;
    }
    case RT_KAT_PMODF:
        fo[0] = pos_modf(fi[0]);
        break;
    case RT_KAT_TRI: {
        // the record rt_prepare_triangles builds (same operations), then the test the kernels run on it
        const V3 o = ld3(fi), d = ld3(fi + 3), a = ld3(fi + 6), b = ld3(fi + 9), c = ld3(fi + 12);
        const V3 ab = sub3(b, a), ac = sub3(c, a);
        const V3 n = cross3(ac, ab);
        const float abab = dot3(ab, ab), abac = dot3(ab, ac), acac = dot3(ac, ac);
        const float inv = 1.f / (abac * abac - abab * acac);
        __attribute__((aligned(16))) float rec[16] = { a.x, a.y, a.z, ab.x, ab.y, ab.z, ac.x, ac.y, ac.z, n.x, n.y, n.z, abab, abac, acac, inv };
        float t = 0.f, l1 = 0.f, l2 = 0.f;
        const bool hit = tri_test(rec, 0u, o, d, fi[15], fi[16], t, l1, l2);
        fo[0] = hit ? 1.f : 0.f; fo[1] = t; fo[2] = l1; fo[3] = l2;
        // the branch-free variant the wavefront kernels use must agree wherever the reference computes the second half
        float t2, m1, m2;
        const float4 r0 = make_float4(rec[0], rec[1], rec[2], rec[3]), r1 = make_float4(rec[4], rec[5], rec[6], rec[7]);
        const float4 r2 = make_float4(rec[8], rec[9], rec[10], rec[11]), r3 = make_float4(rec[12], rec[13], rec[14], rec[15]);
        const V3 fa = mk(r0.x, r0.y, r0.z), fab = mk(r0.w, r1.x, r1.y), fac = mk(r1.z, r1.w, r2.x), fn = mk(r2.y, r2.z, r2.w);
        t2 = -dot3(fn, sub3(o, fa)) / dot3(fn, d);
        const V3 ap = sub3(along(o, t2, d), fa);
        const float ap_ab = dot3(ap, fab), ap_ac = dot3(ap, fac);
        m1 = (r3.y * ap_ac - r3.z * ap_ab) * r3.w;
        m2 = (r3.y * ap_ab - r3.x * ap_ac) * r3.w;
        const bool hit2 = (fi[15] < t2) & (t2 < fi[16]) & (0 <= m1) & (0 <= m2) & (m1 + m2 <= 1.f);
        const bool inRange = fi[15] < t && t < fi[16];
        const bool same = hit2 == hit && __float_as_uint(t2) == __float_as_uint(t) &&
                          (!inRange || (__float_as_uint(m1) == __float_as_uint(l1) && __float_as_uint(m2) == __float_as_uint(l2)));
        fo[4] = same ? 1.f : 0.f;
        break;
    }
    case RT_KAT_PLINE:
        fo[0] = point_line_sq(ld3(fi), ld3(fi + 3), ld3(fi + 6));
        break;
    case RT_KAT_BOX: {
        int cx, cy, cz;
        box_address(sh, ld3(fi), cx, cy, cz);
        int *io = reinterpret_cast<int *>(fo);
        io[0] = cx; io[1] = cy; io[2] = cz;
        break;
    }
    case RT_KAT_BIND: {
        V3 p = ld3(fi);
        bind_in_cube(p, ld3(fi + 3), ld3(fi + 6), ld3(fi + 9));
        fo[0] = p.x; fo[1] = p.y; fo[2] = p.z;
        break;
    }
    case RT_KAT_POW:
        fo[0] = half_falloff(fi[0]);
        break;
    case RT_KAT_QUOTIENT: { // both ways of wf_trace_kernel's walk; they must agree (up to the sign of zero) whenever `tame` says so
        const float plane = fi[0], o = fi[1], d = fi[2];
        fo[0] = (plane - o) / d;
        fo[1] = tame_quotient(plane - o, d, refined_rcp(d));
        reinterpret_cast<uint32_t *>(fo)[2] = (tame_origin(plane) && tame_origin(o) && tame_direction(d)) ? 1u : 0u;
        break;
    }
    default:
        break;
    }
}

} // namespace

extern "C" int rtHipDeviceKat(int device, int op, cl_uint count, const void *in, cl_uint inStride, void *out, cl_uint outStride, const float *table)
{
    static const uint32_t inNeed[RT_KAT_OPS] = { 8, 16, 4, 68, 36, 12, 48, 4, 12 }, outNeed[RT_KAT_OPS] = { 192, 24, 4, 20, 4, 12, 12, 4, 12 };
    if (op < 0 || op >= RT_KAT_OPS || !in || !out || inStride < inNeed[op] || outStride < outNeed[op] || (inStride & 3u) || (outStride & 3u)) return -1;
    if (op == RT_KAT_BOX && !table) return -1;
    if (count == 0) return 0;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || device < 0 || device >= n) return -2; // no device: no CPU stand-in
    if (hipSetDevice(device) != hipSuccess) return -2;
    unsigned char *dIn = nullptr, *dOut = nullptr;
    float *dTable = nullptr;
    int rc = -3;
    do {
        if (hipMalloc((void **)&dIn, (size_t)count * inStride) != hipSuccess) break;
        if (hipMalloc((void **)&dOut, (size_t)count * outStride) != hipSuccess) break;
        if (hipMemcpy(dIn, in, (size_t)count * inStride, hipMemcpyHostToDevice) != hipSuccess) break;
        if (hipMemset(dOut, 0, (size_t)count * outStride) != hipSuccess) break;
        if (table) {
            if (hipMalloc((void **)&dTable, sizeof(float) * 3 * (RT_GRID_DIV + 1)) != hipSuccess) break;
            if (hipMemcpy(dTable, table, sizeof(float) * 3 * (RT_GRID_DIV + 1), hipMemcpyHostToDevice) != hipSuccess) break;
        }
        hipLaunchKernelGGL(rt_kat_kernel, dim3((count + 255) / 256), dim3(256), 0, 0, op, count, dIn, inStride, dOut, outStride, dTable);
        if (hipGetLastError() != hipSuccess || hipDeviceSynchronize() != hipSuccess) break;
        if (hipMemcpy(out, dOut, (size_t)count * outStride, hipMemcpyDeviceToHost) != hipSuccess) break;
        rc = 0;
    } while (0);
    (void)hipFree(dIn); (void)hipFree(dOut); (void)hipFree(dTable);
    return rc;
}
