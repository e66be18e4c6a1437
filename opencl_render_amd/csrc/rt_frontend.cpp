// rt_frontend.cpp -- headless scene front-end and output sinks ("next" rows 3 and 4).
//
// The reference gets its inputs from Cinema 4D (source/render.cpp:676-1309) and shows its output in a C4D bitmap plus a
// debug BMP (render.cpp:1372-1386, source/util/writebmp.cpp:124-177).  None of the SDK-bound code can be built here; what is
// restated below is the SDK-FREE arithmetic around it, so that the library renders real meshes without Cinema 4D:
//   rtHipSetCamera        SetCamera, render.cpp:461-491 (un-normalised camera vector, tan(fov/2), the left/right naming quirk)
//   rtHipMeshCount/Fill   the SoA contract of Count/AddPolygonsRecursive, render.cpp:676-702, 707-1003: quads -> 2 triangles
//                         (a,b,c)+(a,c,d), per-corner normals normalised in double, or flat normals turned towards the camera
//                         when the mesh has none (:754-771), UVs per corner or the fallback (0,0),(0,1),(1,1) (:956-963),
//                         material -1 when unassigned (:1098)
//   rtHipLightFill        render.cpp:965-993: direction normalised with a float length, colour x brightness, radius 0.52
//                         degrees (the sun), half-attenuation distance infinite
//   rtHipBakeMaterials    the channel table rules of render.cpp:1136-1302 for channels that are bitmaps or absent (C4D
//                         shaders need the SDK): see the function
//   rtHipPlanesToRgb8 / rtHipWriteBmp / rtHipWritePpm   u16 planes -> 8 bit as render.cpp:1379-1382 (value / 256), BMP bytes
//                         laid out as writebmp3s (bottom-up rows, BGR, rows padded to 4 bytes); the reference's BMP keeps
//                         the LOW byte of every u16 (writebmp.cpp:136-141, a truncation bug) -- only behind `lowByteCompat`.
// Host code; nothing here touches the GPU.  Parity: libm tan/sqrt and the C4D-side semantics are not pinned by the reference
// (no fixtures, SDK absent): tests check these functions against an independent numpy restatement and by properties.
#include "raytrace_hip.h"

#include <cmath>
#include <cstdio>
#include <cstring>
#include <vector>

namespace {

inline float dotf(const float *a, const float *b) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]; } // raytrace.c:18-20
inline void crossf(const float *a, const float *b, float *r)                                          // raytrace.c:21-27
{
    r[0] = a[1] * b[2] - a[2] * b[1];
    r[1] = a[2] * b[0] - a[0] * b[2];
    r[2] = a[0] * b[1] - a[1] * b[0];
}

// (cl_uchar)floor(0.5 + v*255) as x86 converts it: through a 64-bit integer, low byte kept (values above 1 wrap)
inline unsigned char byte_of_unit(double v) { return (unsigned char)(long long)std::floor(0.5 + v * 255.0); }

} // namespace

extern "C" {

void rtHipSetCamera(cl_float3 *outEyeToTopLeft, cl_float3 *outLeftToRight, cl_float3 *outTopToBottom, cl_float *outPixelSizeInv,
                    const cl_float position[3], const cl_float object[3], const cl_float up[3], cl_float fov, cl_uint width, cl_uint height)
{
    // render.cpp:462-490, operation for operation (float unless the reference goes through the double libm)
    float cam[3] = { object[0] - position[0], object[1] - position[1], object[2] - position[2] };
    float side[3]; // "rightToLeft" in the reference = cross(up, cameraVector); it ends up as the LEFT-TO-RIGHT pixel vector
    crossf(up, cam, side);
    // `(float)tan(fov / 2.f)` at render.cpp:469 is C++: with MSVC's <math.h> the call most likely resolves to the float overload, i.e.
    // tanf of that runtime, where C would promote to double.  Which one -- and MSVCR120's tanf itself -- cannot be pinned here; the
    // tangent is taken in double and rounded once, which is the correctly rounded float except at a rounding tie, so the two
    // readings can differ by one ULP of midToLeft at most (parity unpinned, like the rest of the front-end).
    const float midToLeft = (float)std::sqrt((double)dotf(cam, cam)) * (float)std::tan((double)(fov / 2.f));
    const float midToTop = midToLeft * (float)height / (float)width;
    const float sideLen = (float)std::sqrt((double)dotf(side, side));
    const float upLen = (float)std::sqrt((double)dotf(up, up));
    const float sideUnit[3] = { side[0] / sideLen, side[1] / sideLen, side[2] / sideLen };
    const float upUnit[3] = { up[0] / upLen, up[1] / upLen, up[2] / upLen };
    const float inv = ((float)width) / (2.f * midToLeft);
    *outPixelSizeInv = inv;
    for (int i = 0; i < 3; ++i) {
        outEyeToTopLeft->s[i] = cam[i] - midToLeft * sideUnit[i] + midToTop * upUnit[i];
        outLeftToRight->s[i] = sideUnit[i] / inv;
        outTopToBottom->s[i] = -upUnit[i] / inv;
    }
    outEyeToTopLeft->s[3] = outLeftToRight->s[3] = outTopToBottom->s[3] = 0.f; // the padding lane is never read
}

int rtHipMeshCount(const rtHipMesh *meshes, cl_uint meshCount, cl_uint *vertexCount, cl_uint *triangleCount)
{
    if ((!meshes && meshCount) || !vertexCount || !triangleCount) return -1;
    uint64_t v = 0, t = 0;
    for (cl_uint m = 0; m < meshCount; ++m) {
        const rtHipMesh &M = meshes[m];
        if ((M.pointCount && !M.points) || (M.polygonCount && !M.polygons)) return -1;
        v += M.pointCount;
        for (cl_uint i = 0; i < M.polygonCount; ++i) {
            const cl_int *p = M.polygons + 4 * (size_t)i;
            for (int k = 0; k < 4; ++k)
                if (p[k] < 0 || (cl_uint)p[k] >= M.pointCount) return -2; // a polygon that points outside its object
            t += (p[2] != p[3]) ? 2 : 1; // render.cpp:736 (the reference ALLOCATES 2 per polygon, :683, and counts what it filled)
        }
    }
    if (v > 0xffffffffull || t > 0xffffffffull) return -3;
    *vertexCount = (cl_uint)v;
    *triangleCount = (cl_uint)t;
    return 0;
}

int rtHipMeshFill(const rtHipMesh *meshes, cl_uint meshCount, const cl_float cameraEye[3], cl_float3 *vertex, cl_int3 *triIndex,
                  cl_int *triMaterial, cl_float2 *triUv, cl_float3 *triNormal)
{
    cl_uint needV = 0, needT = 0;
    const int rc = rtHipMeshCount(meshes, meshCount, &needV, &needT);
    if (rc != 0) return rc;
    if ((needV && !vertex) || (needT && (!triIndex || !triMaterial || !triUv || !triNormal)) || !cameraEye) return -1;
    cl_uint vCursor = 0, tCursor = 0;
    for (cl_uint m = 0; m < meshCount; ++m) {
        const rtHipMesh &M = meshes[m];
        const cl_uint firstVertex = vCursor;
        for (cl_uint i = 0; i < M.pointCount; ++i, ++vCursor) { // :721-727 (points are given in world space)
            vertex[vCursor].s[0] = M.points[i].s[0]; vertex[vCursor].s[1] = M.points[i].s[1]; vertex[vCursor].s[2] = M.points[i].s[2];
            vertex[vCursor].s[3] = 0.f;
        }
        for (cl_uint i = 0; i < M.polygonCount; ++i) {
            const cl_int *p = M.polygons + 4 * (size_t)i;
            const bool quad = p[2] != p[3];
            const int corners[2][3] = { { 0, 1, 2 }, { 0, 2, 3 } }; // (a,b,c) then (a,c,d): :737-739, :781-783
            for (int half = 0; half < (quad ? 2 : 1); ++half, ++tCursor) {
                const int *c = corners[half];
                for (int k = 0; k < 3; ++k) triIndex[tCursor].s[k] = (cl_int)firstVertex + p[c[k]];
                triIndex[tCursor].s[3] = 0;
                bool haveNormals = M.cornerNormals != nullptr;
                if (haveNormals) // (a polygon with a zero corner normal -- an OBJ face without vn among faces with -- counts as one without normals)
                    for (int k = 0; k < 3; ++k) {
                        const cl_float3 &n = M.cornerNormals[4 * (size_t)i + c[k]];
                        if (n.s[0] == 0.f && n.s[1] == 0.f && n.s[2] == 0.f) haveNormals = false;
                    }
                if (haveNormals) { // :740-752 -- per-corner normals, normalised in double, then rounded to float
                    for (int k = 0; k < 3; ++k) {
                        const cl_float3 &n = M.cornerNormals[4 * (size_t)i + c[k]];
                        const double x = n.s[0], y = n.s[1], z = n.s[2];
                        const double len = std::sqrt(x * x + y * y + z * z);
                        cl_float3 &o = triNormal[3 * (size_t)tCursor + k];
                        o.s[0] = (float)(x / len); o.s[1] = (float)(y / len); o.s[2] = (float)(z / len); o.s[3] = 0.f;
                    }
                } else { // :754-771 -- no normals: the face normal, turned to face the camera
                    const float *a = vertex[triIndex[tCursor].s[0]].s, *b = vertex[triIndex[tCursor].s[1]].s, *cc = vertex[triIndex[tCursor].s[2]].s;
                    const float ab[3] = { b[0] - a[0], b[1] - a[1], b[2] - a[2] }, ac[3] = { cc[0] - a[0], cc[1] - a[1], cc[2] - a[2] };
                    float tn[3];
                    crossf(ab, ac, tn); // getNormal, :411-420
                    float lenInv = 1.f / (float)std::sqrt((double)dotf(tn, tn));
                    const float toA[3] = { a[0] - cameraEye[0], a[1] - cameraEye[1], a[2] - cameraEye[2] }; // vector(eye, a), raytrace.c:36-42
                    if (0 <= dotf(toA, tn)) lenInv = -lenInv;
                    for (int k = 0; k < 3; ++k) {
                        cl_float3 &o = triNormal[3 * (size_t)tCursor + k];
                        o.s[0] = tn[0] * lenInv; o.s[1] = tn[1] * lenInv; o.s[2] = tn[2] * lenInv; o.s[3] = 0.f;
                    }
                }
                if (M.cornerUv) { // :896-919
                    for (int k = 0; k < 3; ++k) triUv[3 * (size_t)tCursor + k] = M.cornerUv[4 * (size_t)i + c[k]];
                } else { // :956-963
                    triUv[3 * (size_t)tCursor + 0].s[0] = 0.f; triUv[3 * (size_t)tCursor + 0].s[1] = 0.f;
                    triUv[3 * (size_t)tCursor + 1].s[0] = 0.f; triUv[3 * (size_t)tCursor + 1].s[1] = 1.f;
                    triUv[3 * (size_t)tCursor + 2].s[0] = 1.f; triUv[3 * (size_t)tCursor + 2].s[1] = 1.f;
                }
                triMaterial[tCursor] = M.polygonMaterial ? M.polygonMaterial[i] : -1; // :1098 (memset to -1), :874
            }
        }
    }
    return 0;
}

void rtHipLightFill(cl_uint index, cl_int type, const cl_float position[3], const cl_float direction[3], const cl_float colour[3],
                    cl_float brightness, cl_int *lightType, cl_float3 *lightPosition, cl_float3 *lightDirection, cl_float3 *lightColour,
                    cl_float *lightRadius, cl_float *lightHalfAttenuationDistance)
{
    // render.cpp:965-993
    lightType[index] = type;
    const float len = (float)std::sqrt((double)direction[0] * direction[0] + (double)direction[1] * direction[1] + (double)direction[2] * direction[2]);
    for (int k = 0; k < 3; ++k) {
        lightPosition[index].s[k] = position[k];
        lightDirection[index].s[k] = direction[k] / len;
        lightColour[index].s[k] = colour[k] * brightness;
    }
    lightPosition[index].s[3] = lightDirection[index].s[3] = lightColour[index].s[3] = 0.f;
    lightRadius[index] = 0.52f;                  // "The Sun's angular size is 0.52 degrees" (:967), used for every light
    lightHalfAttenuationDistance[index] = INFINITY; // :980
}

int rtHipBakeMaterials(const rtHipMaterialSpec *materials, cl_uint materialCount, cl_uint2 *materialImageSize, cl_int *materialImageStart,
                       cl_uchar3 *textures, cl_uint texturesCapacity, cl_uint *texturesSize)
{
    if ((!materials && materialCount) || !materialImageSize || !materialImageStart || !texturesSize) return -1;
    uint64_t cursor = 0;
    auto put = [&](unsigned char r, unsigned char g, unsigned char b) {
        if (textures && cursor < texturesCapacity) { textures[cursor].s[0] = r; textures[cursor].s[1] = g; textures[cursor].s[2] = b; textures[cursor].s[3] = 0; }
        ++cursor;
    };
    for (cl_uint m = 0; m < materialCount; ++m) {
        const rtHipMaterialSpec &M = materials[m];
        for (int ci = 0; ci < 5; ++ci) { // colour, reflection, transparency, bump, luminance (render.cpp:1136)
            const rtHipChannelSpec &C = M.channel[ci];
            cl_uint2 &size = materialImageSize[5 * (size_t)m + ci];
            size.s[0] = size.s[1] = 0;                                 // absent or switched off: 0 x 0 (:1145-1149)
            materialImageStart[5 * (size_t)m + ci] = (cl_int)cursor;
            if (C.enabled) {
                if (C.pixels && C.width && C.height) {                  // a bitmap: copied row by row (:1165-1181)
                    size.s[0] = C.width; size.s[1] = C.height;
                    for (uint64_t i = 0; i < (uint64_t)C.width * C.height; ++i) put(C.pixels[i].s[0], C.pixels[i].s[1], C.pixels[i].s[2]);
                } else if (ci == 1) {                                   // reflection switched on without an image: 0.2 (:1220-1229)
                    size.s[0] = size.s[1] = 1;
                    const unsigned char v = (unsigned char)std::floor(0.5f + 0.2f * 255.f);
                    put(v, v, v);
                } else if (ci == 2) {                                   // transparency switched on without an image: 1.0 (:1230-1239)
                    size.s[0] = size.s[1] = 1;
                    const unsigned char v = (unsigned char)std::floor(0.5f + 1.f * 255.f);
                    put(v, v, v);
                }
            }
            if (ci != 0 && size.s[0] == 0) {                            // every non-colour channel ends up at least 1 x 1 black (:1243-1251)
                size.s[0] = size.s[1] = 1;
                materialImageStart[5 * (size_t)m + ci] = (cl_int)cursor;
                put(0, 0, 0);
            }
        }
        if (materialImageSize[5 * (size_t)m].s[0] == 0) {               // no colour image: the material colour x brightness (:1254-1275)
            materialImageSize[5 * (size_t)m].s[0] = materialImageSize[5 * (size_t)m].s[1] = 1;
            materialImageStart[5 * (size_t)m] = (cl_int)cursor;
            const float c[3] = { M.color[0] * M.brightness, M.color[1] * M.brightness, M.color[2] * M.brightness };
            put(byte_of_unit((double)c[0]), byte_of_unit((double)c[1]), byte_of_unit((double)c[2]));
        }
        // render.cpp:1276-1297 (transparency from the material's transparency colour) can never run: the loop above has
        // already given the transparency channel a 1 x 1 black image.  Kept out, like the dead code it is.
    }
    if (cursor > 0x7fffffffull) return -3;
    materialImageStart[5 * (size_t)materialCount] = (cl_int)cursor; // the total goes last (:1306; RaytraceAll's callers rely on it, raytrace.c:441)
    *texturesSize = (cl_uint)cursor;
    if (textures && cursor > texturesCapacity) return -2;
    return 0;
}

void rtHipPlanesToRgb8(cl_uint width, cl_uint height, const cl_ushort *red, const cl_ushort *green, const cl_ushort *blue,
                       cl_uchar *rgb, int lowByteCompat)
{
    const size_t n = (size_t)width * height;
    for (size_t i = 0; i < n; ++i) {
        // render.cpp:1379-1382: value / 256.  writebmp.cpp:136-141 casts the u16 to unsigned char instead (keeps the low byte).
        rgb[3 * i + 0] = lowByteCompat ? (cl_uchar)red[i] : (cl_uchar)(red[i] / 256);
        rgb[3 * i + 1] = lowByteCompat ? (cl_uchar)green[i] : (cl_uchar)(green[i] / 256);
        rgb[3 * i + 2] = lowByteCompat ? (cl_uchar)blue[i] : (cl_uchar)(blue[i] / 256);
    }
}

int rtHipWriteBmp(const char *path, cl_uint width, cl_uint height, const cl_ushort *red, const cl_ushort *green, const cl_ushort *blue,
                  int lowByteCompat)
{
    if (!path || !red || !green || !blue || width == 0 || height == 0) return -1;
    if ((uint64_t)width * height * 3 + 54 > 0x7fffffffull) return -3; // the header's 32-bit sizes (writebmp.cpp:128)
    std::vector<cl_uchar> rgb((size_t)width * height * 3);
    rtHipPlanesToRgb8(width, height, red, green, blue, rgb.data(), lowByteCompat);
    const int w = (int)width, h = (int)height;
    const int filesize = 54 + 3 * w * h; // writebmp.cpp:128 (the row padding is not counted there either)
    unsigned char fileHeader[14] = { 'B', 'M', 0, 0, 0, 0, 0, 0, 0, 0, 54, 0, 0, 0 };
    unsigned char infoHeader[40] = { 40, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 1, 0, 24, 0 };
    for (int k = 0; k < 4; ++k) {
        fileHeader[2 + k] = (unsigned char)(filesize >> (8 * k));
        infoHeader[4 + k] = (unsigned char)(w >> (8 * k));
        infoHeader[8 + k] = (unsigned char)(h >> (8 * k));
    }
    FILE *f = std::fopen(path, "wb");
    if (!f) return -4;
    const unsigned char pad[3] = { 0, 0, 0 };
    const size_t padBytes = (size_t)((4 - (w * 3) % 4) % 4);
    std::vector<unsigned char> row((size_t)w * 3);
    bool ok = std::fwrite(fileHeader, 1, 14, f) == 14 && std::fwrite(infoHeader, 1, 40, f) == 40;
    for (int i = 0; i < h && ok; ++i) { // bottom row first, BGR (writebmp.cpp:141-143,165-168)
        const cl_uchar *src = rgb.data() + (size_t)(h - i - 1) * w * 3;
        for (int x = 0; x < w; ++x) { row[3 * x + 0] = src[3 * x + 2]; row[3 * x + 1] = src[3 * x + 1]; row[3 * x + 2] = src[3 * x + 0]; }
        ok = std::fwrite(row.data(), 3, (size_t)w, f) == (size_t)w && std::fwrite(pad, 1, padBytes, f) == padBytes;
    }
    ok = (std::fclose(f) == 0) && ok;
    return ok ? 0 : -4;
}

int rtHipWritePpm(const char *path, cl_uint width, cl_uint height, const cl_ushort *red, const cl_ushort *green, const cl_ushort *blue)
{
    if (!path || !red || !green || !blue || width == 0 || height == 0) return -1;
    std::vector<cl_uchar> rgb((size_t)width * height * 3);
    rtHipPlanesToRgb8(width, height, red, green, blue, rgb.data(), 0);
    FILE *f = std::fopen(path, "wb");
    if (!f) return -4;
    bool ok = std::fprintf(f, "P6\n%u %u\n255\n", width, height) > 0;
    ok = ok && std::fwrite(rgb.data(), 1, rgb.size(), f) == rgb.size();
    ok = (std::fclose(f) == 0) && ok;
    return ok ? 0 : -4;
}

} // extern "C"
