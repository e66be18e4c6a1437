// rt_fileio.cpp -- file input of the headless front-end (include/raytrace_hip.h, section 3): Wavefront OBJ / MTL meshes and
// PPM / BMP images, and the projected UVs of the reference's ShdProjectPoint.
//
// The reference takes its geometry from Cinema 4D's object tree (source/render.cpp:707-1003) and its textures from C4D bitmaps
// (render.cpp:1136-1309); neither exists without the SDK.  What a host without Cinema 4D has is files: this reader turns an OBJ
// into the same polygon objects rtHipMeshFill consumes (points, a/b/c/d polygons with c == d marking a triangle, corner normals,
// corner UVs, one material id per polygon), an MTL into rtHipMaterialSpec-shaped channel choices, and an image file into the
// 4-byte texels rtHipBakeMaterials takes.  Host code only; no GPU involved.
#include "raytrace_hip.h"

#include <cctype>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <string>
#include <vector>

namespace {

std::string dir_of(const char *path)
{
    const std::string p = path;
    const size_t cut = p.find_last_of('/');
    return cut == std::string::npos ? std::string() : p.substr(0, cut + 1);
}

// one line of a text file, without the trailing newline / carriage return; false at the end of the file
bool read_line(FILE *f, std::string &line)
{
    line.clear();
    int c;
    bool any = false;
    while ((c = fgetc(f)) != EOF) {
        any = true;
        if (c == '\n') break;
        if (c != '\r') line.push_back((char)c);
    }
    return any;
}

struct ObjCorner { int v, vt, vn; };

// "7", "7/2", "7//3", "7/2/3"; negative indices count from the end (OBJ); returns false on a malformed token
bool parse_corner(const char *tok, size_t nv, size_t nvt, size_t nvn, ObjCorner &out)
{
    long idx[3] = { 0, 0, 0 };
    const char *p = tok;
    for (int k = 0; k < 3; ++k) {
        if (*p == '/' || *p == 0) { if (k == 0) return false; }
        else {
            char *endp = nullptr;
            idx[k] = strtol(p, &endp, 10);
            if (endp == p) return false;
            p = endp;
        }
        if (*p == '/') ++p; else break;
    }
    auto resolve = [](long i, size_t n) -> int { return i > 0 ? (int)(i - 1) : (i < 0 ? (int)((long)n + i) : -1); };
    out.v = resolve(idx[0], nv); out.vt = resolve(idx[1], nvt); out.vn = resolve(idx[2], nvn);
    return out.v >= 0 && (size_t)out.v < nv && (out.vt < 0 || (size_t)out.vt < nvt) && (out.vn < 0 || (size_t)out.vn < nvn) &&
           (idx[1] == 0 || out.vt >= 0) && (idx[2] == 0 || out.vn >= 0);
}

template <class T> T *take(const std::vector<T> &v)
{
    T *p = (T *)malloc(sizeof(T) * (v.empty() ? 1 : v.size()));
    if (p && !v.empty()) memcpy(p, v.data(), sizeof(T) * v.size());
    return p;
}

} // namespace

extern "C" {

void rtHipObjFree(rtHipObjData *d)
{
    if (!d) return;
    free(d->points); free(d->polygons); free(d->cornerNormals); free(d->cornerUv); free(d->polygonMaterial); free(d->materials);
    memset(d, 0, sizeof *d);
}

int rtHipObjRead(const char *path, rtHipObjData *out)
{
    if (!path || !out) return -1;
    memset(out, 0, sizeof *out);
    FILE *f = fopen(path, "rb");
    if (!f) return -4;
    std::vector<float> v, vt, vn;              // 3, 2, 3 floats per element
    std::vector<cl_int> polygons;              // 4 per polygon
    std::vector<float> cornerN, cornerUv;      // 4 x 4 and 4 x 2 floats per polygon
    std::vector<cl_int> polyMat;
    std::vector<rtHipObjMaterial> mats;
    std::map<std::string, int> matId;
    std::vector<std::string> mtlFiles;
    bool anyN = false, anyUv = false;
    int current = -1, rc = 0;
    std::string line;
    auto material = [&](const std::string &name) -> int {
        auto it = matId.find(name);
        if (it != matId.end()) return it->second;
        rtHipObjMaterial m;
        memset(&m, 0, sizeof m);
        snprintf(m.name, sizeof m.name, "%s", name.c_str());
        m.kd[0] = m.kd[1] = m.kd[2] = 1.f; m.dissolve = 1.f;
        mats.push_back(m);
        return matId[name] = (int)mats.size() - 1;
    };
    while (rc == 0 && read_line(f, line)) {
        const char *s = line.c_str();
        while (*s == ' ' || *s == '\t') ++s;
        if (*s == 0 || *s == '#') continue;
        if (s[0] == 'v' && (s[1] == ' ' || s[1] == '\t')) {
            float x = 0, y = 0, z = 0;
            if (sscanf(s + 2, "%f %f %f", &x, &y, &z) != 3) rc = -2;
            v.insert(v.end(), { x, y, z });
        } else if (s[0] == 'v' && s[1] == 't') {
            float a = 0, b = 0;
            if (sscanf(s + 3, "%f %f", &a, &b) < 1) rc = -2;
            vt.insert(vt.end(), { a, b });
        } else if (s[0] == 'v' && s[1] == 'n') {
            float x = 0, y = 0, z = 0;
            if (sscanf(s + 3, "%f %f %f", &x, &y, &z) != 3) rc = -2;
            vn.insert(vn.end(), { x, y, z });
        } else if (s[0] == 'f' && (s[1] == ' ' || s[1] == '\t')) {
            std::vector<ObjCorner> cs;
            const char *p = s + 2;
            while (*p) {
                while (*p == ' ' || *p == '\t') ++p;
                if (!*p) break;
                const char *q = p;
                while (*q && *q != ' ' && *q != '\t') ++q;
                ObjCorner c;
                if (!parse_corner(std::string(p, q).c_str(), v.size() / 3, vt.size() / 2, vn.size() / 3, c)) { rc = -2; break; }
                cs.push_back(c);
                p = q;
            }
            if (rc != 0) break;
            if (cs.size() < 3) { rc = -2; break; }
            // a triangle is (a,b,c,c), a quad (a,b,c,d) -- the polygon objects of render.cpp:736; a larger face is fanned into triangles
            auto emit = [&](const ObjCorner *c, int n) {
                for (int k = 0; k < 4; ++k) {
                    const ObjCorner &cc = c[k < n ? k : n - 1];
                    polygons.push_back(cc.v);
                    if (cc.vn >= 0) { anyN = true; cornerN.insert(cornerN.end(), { vn[3 * cc.vn], vn[3 * cc.vn + 1], vn[3 * cc.vn + 2], 0.f }); }
                    else cornerN.insert(cornerN.end(), { 0.f, 0.f, 0.f, 0.f });
                    if (cc.vt >= 0) { anyUv = true; cornerUv.insert(cornerUv.end(), { vt[2 * cc.vt], vt[2 * cc.vt + 1] }); }
                    else cornerUv.insert(cornerUv.end(), { 0.f, 0.f });
                }
                polyMat.push_back(current);
            };
            if (cs.size() <= 4) emit(cs.data(), (int)cs.size());
            else
                for (size_t k = 1; k + 1 < cs.size(); ++k) { const ObjCorner tri[3] = { cs[0], cs[k], cs[k + 1] }; emit(tri, 3); }
        } else if (strncmp(s, "usemtl", 6) == 0) {
            const char *p = s + 6;
            while (*p == ' ' || *p == '\t') ++p;
            current = material(p);
        } else if (strncmp(s, "mtllib", 6) == 0) {
            const char *p = s + 6;
            while (*p == ' ' || *p == '\t') ++p;
            mtlFiles.push_back(p);
        } // (o, g, s, l and the rest: ignored)
    }
    fclose(f);
    if (rc != 0) return rc;
    // MTL: Kd -> material colour, d / Tr -> transparency, Ke -> luminance, map_* -> channel images (paths relative to the OBJ)
    for (const std::string &mf : mtlFiles) {
        FILE *m = fopen((dir_of(path) + mf).c_str(), "rb");
        if (!m) continue; // (a missing library leaves its materials at their defaults)
        int at = -1;
        while (read_line(m, line)) {
            const char *s = line.c_str();
            while (*s == ' ' || *s == '\t') ++s;
            auto arg = [&](size_t skip) -> const char * { const char *p = s + skip; while (*p == ' ' || *p == '\t') ++p; return p; };
            if (strncmp(s, "newmtl", 6) == 0) at = material(arg(6));
            else if (at < 0) continue;
            else if (strncmp(s, "Kd", 2) == 0 && isspace((unsigned char)s[2])) sscanf(s + 3, "%f %f %f", &mats[at].kd[0], &mats[at].kd[1], &mats[at].kd[2]);
            else if (strncmp(s, "Ke", 2) == 0 && isspace((unsigned char)s[2])) { sscanf(s + 3, "%f %f %f", &mats[at].ke[0], &mats[at].ke[1], &mats[at].ke[2]); mats[at].hasKe = 1; }
            else if (s[0] == 'd' && isspace((unsigned char)s[1])) sscanf(s + 2, "%f", &mats[at].dissolve);
            else if (strncmp(s, "Tr", 2) == 0 && isspace((unsigned char)s[2])) { float tr = 0.f; if (sscanf(s + 3, "%f", &tr) == 1) mats[at].dissolve = 1.f - tr; }
            else if (strncmp(s, "refl", 4) == 0 && isspace((unsigned char)s[4])) { sscanf(s + 5, "%f", &mats[at].reflect); mats[at].hasReflect = 1; }
            else {
                static const struct { const char *key; int channel; } maps[] = { { "map_Kd", 0 }, { "map_refl", 1 }, { "map_d", 2 }, { "map_bump", 3 }, { "map_Bump", 3 }, { "bump", 3 }, { "map_Ke", 4 } };
                for (const auto &mp : maps) {
                    const size_t n = strlen(mp.key);
                    if (strncmp(s, mp.key, n) == 0 && isspace((unsigned char)s[n])) {
                        const std::string file = dir_of(path) + arg(n);
                        snprintf(mats[at].map[mp.channel], sizeof mats[at].map[mp.channel], "%s", file.c_str());
                    }
                }
            }
        }
        fclose(m);
    }
    out->pointCount = (cl_uint)(v.size() / 3);
    std::vector<float> pts(4 * (size_t)out->pointCount, 0.f);
    for (size_t i = 0; i < out->pointCount; ++i) { pts[4 * i] = v[3 * i]; pts[4 * i + 1] = v[3 * i + 1]; pts[4 * i + 2] = v[3 * i + 2]; }
    out->polygonCount = (cl_uint)polyMat.size();
    out->points = (cl_float3 *)take(pts);
    out->polygons = take(polygons);
    out->cornerNormals = anyN ? (cl_float3 *)take(cornerN) : nullptr;
    out->cornerUv = anyUv ? (cl_float2 *)take(cornerUv) : nullptr;
    out->polygonMaterial = take(polyMat);
    out->materialCount = (cl_uint)mats.size();
    out->materials = take(mats);
    if (!out->points || !out->polygons || !out->polygonMaterial || !out->materials || (anyN && !out->cornerNormals) || (anyUv && !out->cornerUv)) {
        rtHipObjFree(out);
        return -3;
    }
    return 0;
}

// Binary / plain PPM (P6 / P3, maxval <= 255) and uncompressed 24 / 32-bit BMP (bottom-up or top-down) -> width*height texels of 4
// bytes (r, g, b, 0), top row first: the layout rtHipChannelSpec::pixels takes.  *pixels is malloc'ed (rtHipFree).
int rtHipImageRead(const char *path, cl_uint *width, cl_uint *height, cl_uchar3 **pixels)
{
    if (!path || !width || !height || !pixels) return -1;
    *pixels = nullptr; *width = *height = 0;
    FILE *f = fopen(path, "rb");
    if (!f) return -4;
    std::vector<unsigned char> data;
    unsigned char buf[65536];
    size_t n;
    while ((n = fread(buf, 1, sizeof buf, f)) > 0) data.insert(data.end(), buf, buf + n);
    fclose(f);
    if (data.size() < 8) return -2;
    std::vector<unsigned char> px;
    cl_uint w = 0, h = 0;
    if (data[0] == 'P' && (data[1] == '6' || data[1] == '3')) {
        size_t at = 2;
        auto number = [&](long &outv) -> bool {
            for (;;) { // white space and comments
                while (at < data.size() && isspace(data[at])) ++at;
                if (at < data.size() && data[at] == '#') { while (at < data.size() && data[at] != '\n') ++at; } else break;
            }
            if (at >= data.size() || !isdigit(data[at])) return false;
            outv = 0;
            while (at < data.size() && isdigit(data[at])) { outv = outv * 10 + (data[at] - '0'); if (outv > 1000000000L) return false; ++at; }
            return true;
        };
        long lw, lh, maxv;
        if (!number(lw) || !number(lh) || !number(maxv) || lw < 1 || lh < 1 || maxv < 1 || maxv > 255 || (double)lw * lh > 1e9) return -2;
        w = (cl_uint)lw; h = (cl_uint)lh;
        px.assign((size_t)w * h * 4, 0);
        if (data[1] == '6') {
            ++at; // the single white-space byte behind maxval
            if (data.size() - at < (size_t)w * h * 3) return -2;
            for (size_t i = 0; i < (size_t)w * h; ++i)
                for (int c = 0; c < 3; ++c) px[4 * i + c] = (unsigned char)((unsigned)data[at + 3 * i + c] * 255u / (unsigned)maxv);
        } else
            for (size_t i = 0; i < (size_t)w * h * 3; ++i) {
                long vch;
                if (!number(vch) || vch > maxv) return -2;
                px[4 * (i / 3) + i % 3] = (unsigned char)(vch * 255 / maxv);
            }
    } else if (data[0] == 'B' && data[1] == 'M' && data.size() >= 54) {
        auto u32 = [&](size_t o) { return (uint32_t)data[o] | ((uint32_t)data[o + 1] << 8) | ((uint32_t)data[o + 2] << 16) | ((uint32_t)data[o + 3] << 24); };
        const uint32_t offset = u32(10), bpp = data[28] | (data[29] << 8), compression = u32(30);
        const int32_t bw = (int32_t)u32(18), bh = (int32_t)u32(22);
        if (bw < 1 || bh == 0 || (bpp != 24 && bpp != 32) || (compression != 0 && compression != 3) || (double)bw * std::abs((double)bh) > 1e9) return -2;
        w = (cl_uint)bw; h = (cl_uint)std::abs(bh);
        const size_t row = ((size_t)w * (bpp / 8) + 3) & ~(size_t)3;
        if (data.size() < offset + row * h) return -2;
        px.assign((size_t)w * h * 4, 0);
        for (cl_uint y = 0; y < h; ++y) {
            const unsigned char *src = data.data() + offset + row * (bh > 0 ? h - 1 - y : y); // (positive height: bottom row first)
            for (cl_uint x = 0; x < w; ++x) {
                const unsigned char *p = src + (size_t)x * (bpp / 8);
                px[4 * ((size_t)y * w + x)] = p[2]; px[4 * ((size_t)y * w + x) + 1] = p[1]; px[4 * ((size_t)y * w + x) + 2] = p[0];
            }
        }
    } else return -2;
    *pixels = (cl_uchar3 *)take(px);
    if (!*pixels) return -3;
    *width = w; *height = h;
    return 0;
}

// ShdProjectPoint (render.cpp:495-673): the UV of a point under a texture tag's projection, for polygons without a UVW tag
// (render.cpp:917-945).  Same operations in the same order, in double like the SDK's Float; acos / atan / sin / cos are this
// machine's libm (parity unpinned: the reference holds no fixture and cannot run here).  P_FRONTAL and P_UVW are "not handled yet"
// in the reference (:647-667, asserts): they leave uv untouched here too.  Returns what the reference returns: 1 always when the
// texture tiles, else whether uv lies in [0,1]^2.
int rtHipProjectUv(int projection, const cl_float point[3], const cl_float normal[3], cl_float offsetX, cl_float offsetY, cl_float lengthX,
                   cl_float lengthY, int tile, cl_float uv[2])
{
    if (!point || !normal || !uv) return 0;
    const double PI = 3.14159265358979323846, PI2 = 2.0 * PI;
    const double px = point[0], py = point[1], pz = point[2];
    const double ox = offsetX, oy = offsetY, lenx = lengthX, leny = lengthY;
    double lenxinv = 0.0, lenyinv = 0.0; // :510-512
    if (lenx != 0.0) lenxinv = 1.0 / lenx;
    if (leny != 0.0) lenyinv = 1.0 / leny;
    double u = uv[0], v = uv[1];
    switch (projection) {
    case RT_PROJ_VOLUMESHADER: // :514-518 (uv = p; the third component has no place in a cl_float2)
        uv[0] = (cl_float)px; uv[1] = (cl_float)py;
        return 1;
    case RT_PROJ_SHRINKWRAP: { // :546-568
        const double sq = std::sqrt(px * px + pz * pz);
        if (sq == 0.0) { u = 0.0; v = py > 0.0 ? 0.0 : 1.0; }
        else {
            u = std::acos(px / sq) / PI2;
            if (pz < 0.0) u = 1.0 - u;
            v = 0.5 - std::atan(py / sq) / PI;
        }
        const double sn = std::sin(u * PI2), cs = std::cos(u * PI2);
        u = (0.5 + 0.5 * cs * v - ox) * lenxinv;
        v = (0.5 + 0.5 * sn * v - oy) * lenyinv;
        break;
    }
    case RT_PROJ_CYLINDRICAL: { // :569-588
        const double sq = std::sqrt(px * px + pz * pz);
        if (sq == 0.0) u = 0.0;
        else {
            u = std::acos(px / sq) / PI2;
            if (pz < 0.0) u = 1.0 - u;
            u -= ox;
            if (lenx > 0.0 && u < 0.0) u += 1.0;
            else if (lenx < 0.0 && u > 0.0) u -= 1.0;
            u *= lenxinv;
        }
        v = -(py * 0.5 + oy) * lenyinv;
        break;
    }
    case RT_PROJ_FLAT: case RT_PROJ_SPATIAL: // :589-595
        u = (px * 0.5 - ox) * lenxinv;
        v = -(py * 0.5 + oy) * lenyinv;
        break;
    case RT_PROJ_CUBIC: { // :596-646
        const double nx = normal[0], ny = normal[1], nz = normal[2];
        int dir;
        if (std::fabs(nx) > std::fabs(ny)) dir = std::fabs(nx) > std::fabs(nz) ? 0 : 2;
        else dir = std::fabs(ny) > std::fabs(nz) ? 1 : 2;
        if (dir == 0) {
            u = nx < 0.0 ? (-pz * 0.5 - ox) * lenxinv : (pz * 0.5 - ox) * lenxinv;
            v = -(py * 0.5 + oy) * lenyinv;
        } else if (dir == 1) {
            v = ny < 0.0 ? (pz * 0.5 - oy) * lenyinv : (-pz * 0.5 - oy) * lenyinv;
            u = (px * 0.5 - ox) * lenxinv;
        } else {
            u = nz < 0.0 ? (px * 0.5 - ox) * lenxinv : (-px * 0.5 - ox) * lenxinv;
            v = -(py * 0.5 + oy) * lenyinv;
        }
        break;
    }
    case RT_PROJ_FRONTAL: case RT_PROJ_UVW: // :647-667: not handled by the reference
        break;
    case RT_PROJ_SPHERICAL: default: { // :519-545
        const double sq = std::sqrt(px * px + pz * pz);
        if (sq == 0.0) { u = 0.0; v = py > 0.0 ? 0.5 : -0.5; }
        else {
            u = std::acos(px / sq) / PI2;
            if (pz < 0.0) u = 1.0 - u;
            u -= ox;
            if (lenx > 0.0 && u < 0.0) u += 1.0;
            else if (lenx < 0.0 && u > 0.0) u -= 1.0;
            u *= lenxinv;
            v = 0.5 + std::atan(py / sq) / PI;
        }
        v = -(v - oy) * lenyinv;
        break;
    }
    }
    uv[0] = (cl_float)u; uv[1] = (cl_float)v; // (cl_float) casts of render.cpp:940-941
    if (tile) return 1;
    return (u >= 0.0 && u <= 1.0 && v >= 0.0 && v <= 1.0) ? 1 : 0;
}

} // extern "C"
