// rt_mathabi.cpp -- the small math helpers the plugin's other translation units link against
// (declared in the reference's raytrace.h:37-44; used at render.cpp:422,758,760,865-874,1197,
// trianglelist.cpp:54-58,105-125,278-351,455 and writebmp.cpp:81-83).  fp32, no contraction
// (-ffp-contract=off), operation order of the reference definitions (raytrace.c:18-45,
// raytrace_opencl.c:83-101,124-172,174-193).
#include "raytrace_hip.h"

#include <cmath>

extern "C" {

cl_float dot(cl_float3 a, cl_float3 b) { return a.s[0] * b.s[0] + a.s[1] * b.s[1] + a.s[2] * b.s[2]; }

cl_float3 cross(cl_float3 a, cl_float3 b)
{
    cl_float3 c;
    c.s[0] = a.s[1] * b.s[2] - a.s[2] * b.s[1];
    c.s[1] = a.s[2] * b.s[0] - a.s[0] * b.s[2];
    c.s[2] = a.s[0] * b.s[1] - a.s[1] * b.s[0];
    c.s[3] = 0.f;
    return c;
}

cl_float3 normalize(cl_float3 v)
{
    const float len = (float)std::sqrt((double)dot(v, v));
    cl_float3 r;
    r.s[0] = v.s[0] / len;
    r.s[1] = v.s[1] / len;
    r.s[2] = v.s[2] / len;
    r.s[3] = 0.f;
    return r;
}

cl_float3 vector(cl_float3 a, cl_float3 b)
{
    cl_float3 ab;
    ab.s[0] = b.s[0] - a.s[0];
    ab.s[1] = b.s[1] - a.s[1];
    ab.s[2] = b.s[2] - a.s[2];
    ab.s[3] = 0.f;
    return ab;
}

// min(max(value, a), b) with the reference's ?: macros (raytrace.h:30-31): a NaN `value` yields a, then min(a, b)
cl_float bindf(cl_float value, cl_float a, cl_float b)
{
    const float up = (value > a) ? value : a;
    return (up < b) ? up : b;
}

cl_float GetPointToLineSqLen(cl_float3 origin, cl_float3 destination, cl_float3 point)
{
    const float odx = destination.s[0] - origin.s[0], ody = destination.s[1] - origin.s[1], odz = destination.s[2] - origin.s[2];
    const float odSq = odx * odx + ody * ody + odz * odz;
    const float opx = point.s[0] - origin.s[0], opy = point.s[1] - origin.s[1], opz = point.s[2] - origin.s[2];
    const float k = (opx * odx + opy * ody + opz * odz) / odSq;
    const float dx = (origin.s[0] + k * odx) - point.s[0];
    const float dy = (origin.s[1] + k * ody) - point.s[1];
    const float dz = (origin.s[2] + k * odz) - point.s[2];
    return dx * dx + dy * dy + dz * dz;
}

cl_bool RayIntersectsTriangle(cl_float3 origin, cl_float3 ray, cl_float minDistance, cl_float maxDistance,
                              cl_float3 a, cl_float3 b, cl_float3 c, cl_float *outRayMult, cl_float *outABL, cl_float *outACL)
{
    cl_bool hit = CL_FALSE;
    const cl_float3 ab = vector(a, b), ac = vector(a, c), ao = vector(a, origin);
    const cl_float3 n = cross(ac, ab);
    *outRayMult = -dot(n, ao) / dot(n, ray);
    if (minDistance < *outRayMult && *outRayMult < maxDistance) {
        const float abab = dot(ab, ab), abac = dot(ab, ac), acac = dot(ac, ac);
        const float inv = 1.f / (abac * abac - abab * acac);
        cl_float3 ap;
        ap.s[0] = (origin.s[0] + *outRayMult * ray.s[0]) - a.s[0];
        ap.s[1] = (origin.s[1] + *outRayMult * ray.s[1]) - a.s[1];
        ap.s[2] = (origin.s[2] + *outRayMult * ray.s[2]) - a.s[2];
        ap.s[3] = 0.f;
        const float apab = dot(ap, ab), apac = dot(ap, ac);
        *outABL = (abac * apac - acac * apab) * inv;
        *outACL = (abac * apab - abab * apac) * inv;
        hit = (0 <= *outABL && 0 <= *outACL && *outABL + *outACL <= 1.f) ? CL_TRUE : CL_FALSE;
    }
    return hit;
}

cl_int3 GetBoxAddress(cl_int axesDivCount, cl_float3 *boxMin, cl_float3 position)
{
    cl_int3 cell;
    cell.s[0] = cell.s[1] = cell.s[2] = cell.s[3] = 0;
    while (1 < axesDivCount) {
        axesDivCount /= 2;
        for (int w = 0; w < 3; ++w) {
            const int mid = cell.s[w] + axesDivCount;
            if (boxMin[mid].s[w] < position.s[w]) cell.s[w] = mid;
        }
    }
    return cell;
}

} // extern "C"
