/*
 * raytrace_hip.h -- C ABI of libraytrace_hip.so, the MI355X (gfx950) replacement for the reference's
 * ray-trace + shade hot path (reference: source/opencl/raytrace.c + raytrace_opencl.c, boundary raytrace.h:37-106).
 *
 * Two layers, both extern "C", plain pointers and sizes only (no HIP, torch or C++ types):
 *
 *  (1) DROP-IN layer  -- exactly the symbols the untouched plugin sources (render.cpp, trianglelist.cpp,
 *      writebmp.cpp) link against today, with the reference's names, argument order and error convention.
 *      Replacing raytrace.c by this library needs no change to any caller.
 *
 *  (2) RESIDENT layer (rtHip*) -- the same path split into upload / render / read-back so a host can keep a scene
 *      in HBM across frames, render a subset of 128x128 tiles (multi-GPU partition) and hand in device pointers.
 *      The drop-in RaytraceAll() is implemented on top of it.
 *
 * Vector layouts are the reference's padded OpenCL host types
 * (source/3rdparty/opencl-1.2/include/CL/cl_platform.h:501,725,1025): float3/int3 = 16 bytes (lane 3 is padding
 * and is never read), float2/uint2 = 8 bytes, uchar3 = 4 bytes.  When <CL/cl.h> has been included first the
 * cl_* names are used as-is, so this header can stand in for raytrace.h inside the plugin.
 */
#ifndef RAYTRACE_HIP_H
#define RAYTRACE_HIP_H

#include <stddef.h>
#include <stdint.h>
#include <time.h>

#ifdef __cplusplus
extern "C" {
#endif

#ifndef __OPENCL_CL_H
typedef int32_t  cl_int;
typedef uint32_t cl_uint;
typedef uint16_t cl_ushort;
typedef uint8_t  cl_uchar;
typedef int8_t   cl_char;
typedef float    cl_float;
typedef cl_uint  cl_bool;
typedef union { cl_float s[2]; } __attribute__((aligned(8)))  cl_float2;
typedef union { cl_uint  s[2]; } __attribute__((aligned(8)))  cl_uint2;
typedef union { cl_int   s[2]; } __attribute__((aligned(8)))  cl_int2;
typedef union { cl_float s[4]; } __attribute__((aligned(16))) cl_float4;
typedef union { cl_int   s[4]; } __attribute__((aligned(16))) cl_int4;
typedef union { cl_uchar s[4]; } __attribute__((aligned(4)))  cl_uchar4;
typedef cl_float4 cl_float3;
typedef cl_int4   cl_int3;
typedef cl_uchar4 cl_uchar3;
#define CL_FALSE 0
#define CL_TRUE  1
#endif

/* ------------------------------------------------------------------------------------------------------------
 * (1) DROP-IN layer
 * ---------------------------------------------------------------------------------------------------------- */

/* Replaces RaytraceAll (reference raytrace.h:58-106, raytrace.c:230-657).
 * computationType: 0 = the reference's in-thread CPU loop -- NOT provided by this library (it is the reference's
 * own code, raytrace.c:604-655): the call fails loudly and returns CL_FALSE.
 * k in 1..N = MI355X device k-1; N+1 = all N devices, image split into 128x128 tiles (only offered when N>1).
 * Blocking.  Caller owns every array; nothing is retained.  Output planes are zeroed first, like the reference's
 * OpenCL branch (raytrace.c:476,481,486), then accumulated.  Returns non-zero on success (raytrace.c:656). */
cl_bool RaytraceAll(cl_uint computationType,
                    cl_uint2 cameraImageDimension,
                    cl_float3 cameraEye,
                    cl_float3 cameraEyeToTopLeftVector,
                    cl_float3 cameraLeftToRightPixelSizeVector,
                    cl_float3 cameraTopToBottomPixelSizeVector,
                    cl_float cameraPixelSizeInv,
                    cl_uint *cameraPixelTriangleListStart,
                    cl_uint *cameraPixelTriangleListEnd,
                    cl_uint *cameraPixelTriangleList,
                    ptrdiff_t cameraPixelTriangleListSize,
                    cl_uint sampleCount,
                    cl_uint vertexCount,
                    cl_float3 *vertex,
                    cl_uint triangleCount,
                    cl_int3 *triangleVertexIndex,
                    cl_int *triangleMaterialId,
                    cl_float2 *triangleUv,
                    cl_float3 *triangleNormal,
                    cl_int axesDivCount,
                    cl_float3 *sceneBoxMin,
                    cl_uint *scenePixelTriangleListStart,
                    cl_uint *scenePixelTriangleList,
                    cl_uint materialCount,
                    cl_uint2 *materialImageSize,
                    cl_int *materialImageStart,
                    cl_uint texturesSize,
                    cl_uchar3 *textures,
                    cl_uint lightCount,
                    cl_int *lightType,
                    cl_float3 *lightPosition,
                    cl_float3 *lightDirection,
                    cl_float3 *lightColour,
                    cl_float *lightRadius,
                    cl_float *lightHalfAttenuationDistance,
                    cl_ushort *outputRed,
                    cl_ushort *outputGreen,
                    cl_ushort *outputBlue);

/* Device enumeration (reference raytrace.h:46-50, raytrace.c:74-153).  Name 0 is the literal
 * "Local CPU single thread" (raytrace.c:138); names 1..N are "AMD HIP <device name>"; name N+1 (N>1 only) is
 * "AMD HIP all N GPUs (tiled)".  InitOpenCL publishes the table last, like raytrace.c:117-120. */
void    InitOpenCL(void);
void    ResetComputationType(void);
cl_bool GetIsComputationTypeUpdated(void);
size_t  GetComputationTypeCount(void);
cl_bool GetComputationTypeName(size_t id, size_t strLen, cl_char *str);

/* Progress / timing polled by the GUI thread (reference raytrace.h:52-56, raytrace.c:156-173).  Progress stays
 * below 1 until the caller sets it (raytrace.c:580,607,614); start time becomes non-zero when the kernel phase
 * begins; relaxed atomics inside. */
cl_float GetProgress(void);
void     SetProgress(cl_float p);
clock_t  GetStartTime(void);
clock_t  GetEndTime(void);
void     ResetTime(void);

/* Math helpers other plugin translation units link against (reference raytrace.h:37-44; definitions
 * raytrace.c:18-45 and raytrace_opencl.c:83-101,124-172,174-193).  fp32, no contraction, same operation order. */
cl_float  dot(cl_float3 a, cl_float3 b);
cl_float3 cross(cl_float3 a, cl_float3 b);
cl_float3 normalize(cl_float3 v);
cl_float3 vector(cl_float3 a, cl_float3 b);
cl_float  bindf(cl_float value, cl_float a, cl_float b);
cl_float  GetPointToLineSqLen(cl_float3 origin, cl_float3 destination, cl_float3 point);
cl_bool   RayIntersectsTriangle(cl_float3 origin, cl_float3 ray, cl_float minDistance, cl_float maxDistance,
                                cl_float3 a, cl_float3 b, cl_float3 c,
                                cl_float *outRayMult, cl_float *outABL, cl_float *outACL);
cl_int3   GetBoxAddress(cl_int axesDivCount, cl_float3 *boxMin, cl_float3 position);

/* ------------------------------------------------------------------------------------------------------------
 * (2) RESIDENT layer
 * ---------------------------------------------------------------------------------------------------------- */

#define RT_HIP_TILE 128 /* tile edge in pixels: the reference's NDRange granule (raytrace.c:507) */

/* Everything RaytraceAll receives except the device choice and the output planes (same meaning, same layouts). */
typedef struct rtHipSceneDesc {
    cl_uint   width, height;
    cl_float  eye[4], eyeToTopLeft[4], leftToRight[4], topToBottom[4];
    cl_float  pixelSizeInv;
    const cl_uint *camStart, *camEnd, *camList;
    uint64_t  camListSize;
    cl_uint   sampleCount;
    cl_uint   vertexCount;
    const cl_float3 *vertex;
    cl_uint   triangleCount;
    const cl_int3 *triIndex;
    const cl_int  *triMaterial;
    const cl_float2 *triUv;
    const cl_float3 *triNormal;
    cl_int    axesDiv;
    const cl_float3 *boxMin;
    const cl_uint *gridStart, *gridList;
    cl_uint   materialCount;
    const cl_uint2 *matSize;
    const cl_int   *matStart;
    cl_uint   texturesSize;
    const cl_uchar3 *textures;
    cl_uint   lightCount;
    const cl_int *lightType;
    const cl_float3 *lightPos, *lightDir, *lightCol;
    const cl_float *lightRadius, *lightHalfAtt;
    cl_int    arraysOnDevice;   /* 0: every array is host memory (what RaytraceAll receives).  1: every array pointer is DEVICE memory of the
                                 * GPU the scene is created on -- e.g. the tensors an RCCL broadcast left there -- and is copied device to
                                 * device; only the small tables the host has to look at (split planes, material tables, texels, lights,
                                 * one grid word) come back to it */
} rtHipSceneDesc;

typedef struct rtHipScene rtHipScene; /* opaque: a scene resident in one GPU's HBM */

/* Work counters of one render, for the algorithmic-byte model. */
typedef struct rtHipStats {
    uint64_t primarySamples, primaryCandidates, gridRays, gridCells, gridCandidates, shadedHits, texelFetches;
} rtHipStats;

/* RaytraceAll keeps the scenes of its last call resident and rebuilds only what changed (content hashes of the input arrays:
 * geometry, grid, materials | lights | camera lists | sample count).  rtHipCacheClear frees them; RT_HIP_CACHE=0 makes every
 * call build and free, like the reference (raytrace.c:330-489,594-602). */
void rtHipCacheClear(void);

/* Number of HIP devices (0 when none / no driver).  Never fails. */
int rtHipDeviceCount(void);

/* Last error text of the calling thread ("" when none). */
const char *rtHipLastError(void);

/* Uploads a scene to `device` and builds the device-side layout (pre-resolved triangle records etc.).
 * tileCount/tileIds select the 128x128 tiles this scene instance will render (row-major tile ids); NULL/0 = all.
 * Only those tiles' slices of the camera lists are uploaded.  Returns NULL on failure. */
rtHipScene *rtHipSceneCreate(int device, const rtHipSceneDesc *desc, const cl_uint *tileIds, cl_uint tileCount);
/* The same for a further instance of a scene that is already resident somewhere (`like`, built from the same description, on this or
 * another device): geometry, grid, materials and lights are copied from it device to device (hipMemcpyPeerAsync: xGMI between
 * GPUs) instead of uploaded and reshaped once more; only the instance's own tiles, camera ranges and path state are made anew. */
rtHipScene *rtHipSceneCreateLike(int device, const rtHipSceneDesc *desc, const cl_uint *tileIds, cl_uint tileCount, const rtHipScene *like);
void        rtHipSceneDestroy(rtHipScene *scene);

/* Bytes of HBM held by the scene. */
uint64_t rtHipSceneBytes(const rtHipScene *scene);

/* Renders all samples of the scene's tiles into its device-resident tile buffer
 * ([tile][plane R,G,B][128*128] u16, tiles in the order given at creation).  Asynchronous on `stream`
 * (a hipStream_t passed as void*; NULL = the scene's own stream).  Returns 0 on success. */
int rtHipRenderTiles(rtHipScene *scene, void *stream);

/* Frames after a scene's first are issued WITHOUT any host synchronisation (the first frame leaves a launch plan behind: how
 * many rounds the frame needs and how big each is; frames of one scene are deterministic).  Whether such a frame really was
 * complete is checked afterwards: rtHipSync and rtHipReadback do it themselves; a caller that consumes the tile buffer on the
 * stream (an RCCL gather enqueued behind the frame) calls rtHipFrameFinish once its own synchronisation is over.  If the plan
 * was too short the last frame is rendered again, watched, and *redone (optional) is set to 1: work that was enqueued behind
 * the incomplete frame has to be repeated.  RT_WF_BLOCKING=1 makes every frame a watched one.  Returns 0 on success. */
int rtHipFrameFinish(rtHipScene *scene, int *redone);

/* Two implementations of the same frame (identical planes):
 *   WAVEFRONT (default) staged pipeline: primary -> rounds of (per-path logic, length sort of the new ray requests, grid
 *                       trace) -> ordered accumulate.  The first frame of a scene watches its ray queue from the host; later frames
 *                       are issued without synchronisation (rtHipFrameFinish).
 *                       Tuning aids read at scene creation: RT_WF_LOOKAHEAD=0|1, RT_WF_SEG="a,b,c,d" and
 *                       RT_WF_SEG_RAYS="a,b,c" (ray segmentation by round size), RT_WF_APPEND_RAYS=n (rounds below n rays skip the length sort), RT_WF_GROUPS=n, RT_WF_STATE_MB.
 *   MEGAKERNEL          one launch, one thread per pixel (kept for A/B runs and for the work counters). */
#define RT_HIP_PIPELINE_MEGAKERNEL 0
#define RT_HIP_PIPELINE_WAVEFRONT  1
int rtHipSetPipeline(rtHipScene *scene, int pipeline);

/* Per-stage device time: enable, render frames, then read the SUM over those frames in milliseconds for
 * [0] primary, [1] logic, [2] grid trace, [3] accumulate, [4] length sort of the trace input (HIP events on the launch
 * stream; adds two event records per launch, so leave it off in timed whole-frame runs).  *rounds = logic/trace rounds
 * of the last frame. */
int rtHipStageTiming(rtHipScene *scene, int enable);
int rtHipStageTimes(rtHipScene *scene, double ms[5], uint64_t *rounds);

/* Diagnostic: copies the scene's 8 device-side debug counters (and optionally clears them).  Synchronous. */
int rtHipDebugCounters(rtHipScene *scene, unsigned long long out[8], int clear);

/* Same, with work counters (slower; never used inside a timed region).  Synchronous. */
int rtHipRenderTilesCounted(rtHipScene *scene, rtHipStats *stats);

/* Device pointer / size in bytes of the tile buffer (for RCCL gathers and peer copies). */
void    *rtHipTileBuffer(rtHipScene *scene);
uint64_t rtHipTileBufferBytes(const rtHipScene *scene);

/* De-tiles `tileCount` tiles held in a device buffer laid out like rtHipTileBuffer into three row-major
 * width x height u16 DEVICE planes (saturating add into what is there).  Used by the gather root, once per
 * source rank.  tileIdsDevice is a DEVICE array of row-major tile ids (ids >= the image's tile count are
 * skipped).  Asynchronous on `stream` (a hipStream_t as void*, NULL = the default stream); no allocation. */
int rtHipDetile(int device, const void *tileBuffer, const cl_uint *tileIdsDevice, cl_uint tileCount,
                cl_uint width, cl_uint height, void *planeR, void *planeG, void *planeB, void *stream);

/* The same without the accumulate: the planes' pixels are overwritten by the tiles' (a gather root that would zero its planes first
 * anyway -- every pixel belongs to exactly one tile of a deal).  Pixels no tile covers keep their value. */
int rtHipDetileStore(int device, const void *tileBuffer, const cl_uint *tileIdsDevice, cl_uint tileCount,
                     cl_uint width, cl_uint height, void *planeR, void *planeG, void *planeB, void *stream);

/* Plain device memory for hosts that do not include HIP headers (a gather root's planes): allocate, free, and a blocking copy
 * (toDevice != 0: host -> device, else device -> host; the device is synchronised first). */
void *rtHipDeviceAlloc(int device, uint64_t bytes);
void  rtHipDeviceFree(int device, void *p);
int   rtHipDeviceCopy(int device, void *dst, const void *src, uint64_t bytes, int toDevice);

/* Blocks until the scene's work is done, then adds its tiles into three HOST planes (width*height u16 each). */
int rtHipReadback(rtHipScene *scene, cl_ushort *outR, cl_ushort *outG, cl_ushort *outB);

/* Waits for `stream` (NULL = scene stream). */
int rtHipSync(rtHipScene *scene, void *stream);

/* Average device time in milliseconds of one rtHipRenderTiles frame (all its kernels) over the frames recorded since
 * the last call (HIP events on the launch stream), and the number of frames.  Returns 0 on success. */
int rtHipKernelTime(rtHipScene *scene, double *avgMs, uint64_t *launches);

/* ------------------------------------------------------------------------------------------------------------
 * Host-side acceleration-structure builders: the producers of the hot path's list inputs
 * (counterparts of CameraTriangleList::New, source/util/trianglelist.cpp:520-626, and SceneTriangleList::New,
 * :655-737).  Same membership tests in the same fp32 arithmetic; memory comes from malloc and is released with
 * rtHipFree.  threads<=0 = all hardware threads.
 * ---------------------------------------------------------------------------------------------------------- */
int rtHipBuildCameraList(cl_uint width, cl_uint height, const cl_float eye[4], const cl_float eyeToTopLeft[4],
                         const cl_float leftToRight[4], const cl_float topToBottom[4], cl_float pixelSizeInv,
                         cl_uint triangleCount, const cl_float3 *vertex, const cl_int3 *triIndex, int threads,
                         cl_uint **outStart, cl_uint **outEnd, cl_uint **outList, uint64_t *outListSize);

/* The same camera lists built on a HIP device (rt_build_device.hip): identical membership (the arithmetic is one shared
 * header compiled for both), every pixel's entries ascending, and the reference's neighbour de-duplication (:580-613: equal
 * neighbouring lists share storage) applied, so Start, End and the list equal rtHipBuildCameraList's.  Fails (no CPU
 * fallback) when the device is missing.
 * *deviceMs (optional) = device time of the build without the transfers. */
int rtHipBuildCameraListDevice(int device, cl_uint width, cl_uint height, const cl_float eye[4], const cl_float eyeToTopLeft[4],
                               const cl_float leftToRight[4], const cl_float topToBottom[4], cl_float pixelSizeInv,
                               cl_uint vertexCount, cl_uint triangleCount, const cl_float3 *vertex, const cl_int3 *triIndex,
                               cl_uint **outStart, cl_uint **outEnd, cl_uint **outList, uint64_t *outListSize, double *deviceMs);

int rtHipBuildSceneGrid(cl_uint vertexCount, cl_uint triangleCount, const cl_float3 *vertex, const cl_int3 *triIndex,
                        int threads, cl_float3 outBoxMin[257], cl_uint **outStart, cl_uint **outList,
                        uint64_t *outListSize);

/* The same grid built on a HIP device (rt_build_device.hip): split planes from radix-sorted coordinates, the flood fill of
 * every triangle with the shared membership test (small triangles one thread each, big ones one workgroup each), pairs
 * radix-sorted on cell << 32 | triangle.  Same planes, starts and lists as rtHipBuildSceneGrid.  No CPU fallback. */
int rtHipBuildSceneGridDevice(int device, cl_uint vertexCount, cl_uint triangleCount, const cl_float3 *vertex, const cl_int3 *triIndex,
                              cl_float3 outBoxMin[257], cl_uint **outStart, cl_uint **outList, uint64_t *outListSize, double *deviceMs);

void rtHipFree(void *p);

/* ------------------------------------------------------------------------------------------------------------
 * (3) HEADLESS FRONT-END and OUTPUT SINKS (rt_frontend.cpp; host code, no GPU involved).
 * The SDK-free arithmetic of the reference's scene extraction (source/render.cpp) and of its output path, so that a
 * host without Cinema 4D can feed RaytraceAll from plain meshes and get an image file back.
 * ---------------------------------------------------------------------------------------------------------- */

/* SetCamera (render.cpp:461-491): camera vectors from eye position, look-at point, up vector, horizontal field of view
 * (radians) and image size.  Same operations in the same order: the eye-to-top-left vector keeps the LENGTH of
 * (object - position), pixel vectors are unit vectors divided by pixelSizeInv = width / (2 |object - position| tan(fov/2)). */
void rtHipSetCamera(cl_float3 *outEyeToTopLeft, cl_float3 *outLeftToRight, cl_float3 *outTopToBottom, cl_float *outPixelSizeInv,
                    const cl_float position[3], const cl_float object[3], const cl_float up[3], cl_float fov, cl_uint width, cl_uint height);

/* One polygon object as AddPolygonsRecursive sees it (render.cpp:707-963), points already in world space. */
typedef struct rtHipMesh {
    cl_uint pointCount;
    const cl_float3 *points;
    cl_uint polygonCount;
    const cl_int *polygons;           /* 4 indices a,b,c,d per polygon; c == d marks a triangle (render.cpp:736) */
    const cl_float3 *cornerNormals;   /* optional, 4 per polygon (a,b,c,d), any length; NULL -- or a zero vector at a corner of the triangle -- =
                                       * face normal turned to the camera (:754-771) */
    const cl_float2 *cornerUv;        /* optional, 4 per polygon; NULL = (0,0),(0,1),(1,1) for every triangle (:956-963) */
    const cl_int *polygonMaterial;    /* optional, one id per polygon; NULL = -1, "no material" (:1098) */
} rtHipMesh;

/* Sizes of the arrays rtHipMeshFill writes: vertices = all points, triangles = 1 per triangle + 2 per quad.
 * Returns 0, -1 null argument, -2 a polygon index outside its object's points, -3 too many elements. */
int rtHipMeshCount(const rtHipMesh *meshes, cl_uint meshCount, cl_uint *vertexCount, cl_uint *triangleCount);

/* Fills RaytraceAll's geometry arrays (vertex[V], triangleVertexIndex[T], triangleMaterialId[T], triangleUv[3T],
 * triangleNormal[3T]) from the meshes: a quad becomes (a,b,c) and (a,c,d) with its corner normals and UVs following. */
int rtHipMeshFill(const rtHipMesh *meshes, cl_uint meshCount, const cl_float cameraEye[3], cl_float3 *vertex, cl_int3 *triIndex,
                  cl_int *triMaterial, cl_float2 *triUv, cl_float3 *triNormal);

/* One light as render.cpp:965-993 stores it: direction normalised, colour x brightness, radius 0.52 (degrees; the sun's
 * angular size, used for every light), half-attenuation distance infinite.  `type` as in raytrace_opencl.h:1-12. */
void rtHipLightFill(cl_uint index, cl_int type, const cl_float position[3], const cl_float direction[3], const cl_float colour[3],
                    cl_float brightness, cl_int *lightType, cl_float3 *lightPosition, cl_float3 *lightDirection, cl_float3 *lightColour,
                    cl_float *lightRadius, cl_float *lightHalfAttenuationDistance);

/* Material channels in the reference's order: colour, reflection, transparency, bump, luminance (render.cpp:1136). */
typedef struct rtHipChannelSpec {
    cl_int enabled;            /* the channel exists and is switched on (render.cpp:1143-1145) */
    cl_uint width, height;     /* bitmap size; 0 = the channel has no bitmap */
    const cl_uchar3 *pixels;   /* width*height texels, row-major, 4 bytes each */
} rtHipChannelSpec;
typedef struct rtHipMaterialSpec {
    rtHipChannelSpec channel[5];
    cl_float color[3];         /* MATERIAL_COLOR_COLOR, used when the colour channel has no bitmap (render.cpp:1254-1275) */
    cl_float brightness;       /* MATERIAL_COLOR_BRIGHTNESS */
} rtHipMaterialSpec;

/* The channel table rules of render.cpp:1136-1309 for bitmaps and absent channels (C4D shaders need the SDK): a channel
 * that is off is 0x0; reflection / transparency switched on without an image are 1x1 of 0.2 / 1.0; every non-colour
 * channel still 0x0 becomes 1x1 black; a colour channel without an image becomes 1x1 of color x brightness;
 * materialImageStart[5*count] receives the texel total.  Call with textures == NULL to size the atlas (*texturesSize),
 * then again with a buffer.  Returns 0, -1 null argument, -2 capacity too small, -3 atlas larger than 2^31 texels. */
int rtHipBakeMaterials(const rtHipMaterialSpec *materials, cl_uint materialCount, cl_uint2 *materialImageSize, cl_int *materialImageStart,
                       cl_uchar3 *textures, cl_uint texturesCapacity, cl_uint *texturesSize);

/* FILE INPUT (rt_fileio.cpp).  The reference walks Cinema 4D's object tree (render.cpp:707-1003) and bakes C4D bitmaps
 * (render.cpp:1136-1309); a host without Cinema 4D has files.
 *
 * rtHipObjRead: a Wavefront OBJ (v / vt / vn / f with triangles, quads and larger faces -- fanned --, negative indices, usemtl,
 * mtllib) as ONE polygon object in rtHipMesh's shape: 4-float points, polygons a,b,c,d with c == d marking a triangle, and -- when
 * any face carries them -- 4 corner normals / 4 corner UVs per polygon (zero where a face has none), one material id per polygon
 * (-1 before the first usemtl).  The materials its MTL libraries define (paths relative to the OBJ) come back as rtHipObjMaterial:
 * Kd = material colour, d / Tr = opacity, Ke = emission, refl = reflectance, and the channel image paths map_Kd (colour), map_refl
 * (reflection), map_d (transparency), map_bump / bump (bump), map_Ke (luminance).  Arrays are malloc'ed: rtHipObjFree.
 * Returns 0, -1 null argument, -2 malformed file, -3 out of memory, -4 cannot open. */
typedef struct rtHipObjMaterial {
    char name[64];
    cl_float kd[3];            /* Kd (default 1 1 1) */
    cl_float ke[3]; cl_int hasKe;
    cl_float dissolve;         /* d, or 1 - Tr (default 1 = opaque) */
    cl_float reflect; cl_int hasReflect;
    char map[5][256];          /* image path per channel in the reference's order (colour, reflection, transparency, bump, luminance); "" = none */
} rtHipObjMaterial;
typedef struct rtHipObjData {
    cl_uint pointCount; cl_float3 *points;
    cl_uint polygonCount; cl_int *polygons; cl_float3 *cornerNormals; cl_float2 *cornerUv; cl_int *polygonMaterial;
    cl_uint materialCount; rtHipObjMaterial *materials;
} rtHipObjData;
int  rtHipObjRead(const char *path, rtHipObjData *out);
void rtHipObjFree(rtHipObjData *data);

/* An image file as rtHipChannelSpec::pixels: binary or plain PPM (P6 / P3, maxval <= 255, scaled to 255) or an uncompressed 24 / 32-bit
 * BMP -> width*height texels of 4 bytes (r, g, b, 0), top row first, malloc'ed (rtHipFree).  Returns 0, -1 null argument, -2 not
 * such a file, -3 out of memory, -4 cannot open. */
int rtHipImageRead(const char *path, cl_uint *width, cl_uint *height, cl_uchar3 **pixels);

/* ShdProjectPoint (render.cpp:495-673): the UV a point gets from a texture tag's projection when its polygon has no UVW tag
 * (render.cpp:917-945).  Same operations in the same order in double (the SDK's Float), results cast to float as at :940-941;
 * RT_PROJ_FRONTAL / RT_PROJ_UVW are not handled by the reference either (uv is left as it is).  The numbering is the Cinema 4D
 * SDK's (c4d_shader.h: P_SPHERICAL ...; not in this checkout).  Returns 1 when the texture tiles or uv lies in [0,1]^2, else 0. */
enum { RT_PROJ_SPHERICAL = 0, RT_PROJ_CYLINDRICAL = 1, RT_PROJ_FLAT = 2, RT_PROJ_CUBIC = 3, RT_PROJ_FRONTAL = 4, RT_PROJ_SPATIAL = 5,
       RT_PROJ_UVW = 6, RT_PROJ_SHRINKWRAP = 7, RT_PROJ_VOLUMESHADER = 10 };
int rtHipProjectUv(int projection, const cl_float point[3], const cl_float normal[3], cl_float offsetX, cl_float offsetY, cl_float lengthX,
                   cl_float lengthY, int tile, cl_float uv[2]);

/* u16 planes -> interleaved 8-bit RGB, top row first: value / 256 (render.cpp:1379-1382).  lowByteCompat != 0 keeps the LOW
 * byte instead, which is what the reference's debug BMP does (writebmp.cpp:136-141, a truncation bug). */
void rtHipPlanesToRgb8(cl_uint width, cl_uint height, const cl_ushort *red, const cl_ushort *green, const cl_ushort *blue,
                       cl_uchar *rgb, int lowByteCompat);

/* writebmp3s (writebmp.cpp:124-177) to a path of the caller's choice: 54-byte header with file size 54 + 3wh, 24-bit BGR,
 * bottom row first, rows padded to 4 bytes.  Returns 0, -1 bad argument, -3 image too large for the header, -4 I/O error. */
int rtHipWriteBmp(const char *path, cl_uint width, cl_uint height, const cl_ushort *red, const cl_ushort *green, const cl_ushort *blue,
                  int lowByteCompat);

/* Binary PPM (P6, maxval 255) of the same 8-bit image, top row first. */
int rtHipWritePpm(const char *path, cl_uint width, cl_uint height, const cl_ushort *red, const cl_ushort *green, const cl_ushort *blue);

/* ------------------------------------------------------------------------------------------------------------
 * TEST-ONLY: device-side known-answer runner (rt_kat.hip).  Runs the kernels' own building blocks -- the restatements
 * of randF (raytrace_opencl.c:12-23), GetSpherePoint (:30-45), positive_modf (:25-28), RayIntersectsTriangle
 * (:124-172, on the pre-resolved record), GetPointToLineSqLen (:83-101), GetBoxAddress (:174-193), BindInCube
 * (:265-322) and the (float)pow(0.5f, x) of :631 -- over `count` items on a HIP device, one per thread; layouts per op
 * are documented at the top of rt_kat.hip.  `table` is only read by RT_KAT_BOX (split planes, 3 x 257 floats, one
 * array per axis).  Returns 0, -1 bad arguments, -2 no such device (there is no CPU stand-in), -3 HIP failure.
 * ---------------------------------------------------------------------------------------------------------- */
enum { RT_KAT_RANDF = 0, RT_KAT_SPHERE, RT_KAT_PMODF, RT_KAT_TRI, RT_KAT_PLINE, RT_KAT_BOX, RT_KAT_BIND, RT_KAT_POW, RT_KAT_QUOTIENT, RT_KAT_OPS };
int rtHipDeviceKat(int device, int op, cl_uint count, const void *in, cl_uint inStride, void *out, cl_uint outStride, const float *table);

/* TEST / TUNING ONLY.  The library reads no environment variables (a plugin host's environment must not be able to slow frames
 * down, make them redo themselves or fail); every tuning value and every fault injector of the tests is set here, process-wide,
 * and applies to scenes built afterwards.  Keys (rt_api.cpp, struct Tuning): "reset" (all defaults), "stage_mb", "extra_factor",
 * "state_mb", "groups", "lookahead", "seg0".."seg4", "seg_rays0".."seg_rays3", "fast_quotient", "spin_limit", "append_rays", "ordered_first", "extra_factor",
 * "slice_rays", "small_slices", "group_rays", "blocking", "batch_plan", "pipeline", "timing", "cache", and the test hooks "plan_rounds",
 * "plan_grid_tiny", "virtual_devices".  Returns 0, -1 for an unknown key. */
int rtHipTune(const char *key, double value);

/* TEST-ONLY: device addresses held by the first scene of RaytraceAll's cache -- triangle records, shading rows, the grid's pair
 * records, material descriptors, texture atlas, camera list -- so that a test can see which parts a call left in place.
 * Returns 0, -2 when nothing is cached. */
int rtHipTestCachePointers(const void *out[6]);

/* TEST-ONLY: the content hash RaytraceAll's scene cache compares per input array (rt_api.cpp, hash_chunk), on the host.  Two byte
 * strings that differ must hash differently for the cache to notice an edit; tests/test_abi.py probes the tail handling. */
uint64_t rtHipTestHashBytes(const void *bytes, uint64_t count);

#ifdef __cplusplus
}
#endif
#endif /* RAYTRACE_HIP_H */
