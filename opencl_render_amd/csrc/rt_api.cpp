// rt_api.cpp -- host side of libraytrace_hip.so: the C ABI of include/raytrace_hip.h.
//
// Replaces the reference's dispatcher (source/opencl/raytrace.c): where that file creates 35 USE_HOST_PTR buffers,
// recompiles the kernel and enqueues tiles x samples NDRanges per call (raytrace.c:330-556), this one uploads the
// scene once into HBM, reshapes it on the device (rt_prepare_triangles) and issues ONE launch per frame and GPU.
// There is no CPU fallback: without a HIP device every entry point that would compute fails with an error text.
#include "raytrace_hip.h"
#include "rt_device.h"

#include <hip/hip_runtime_api.h>

#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>
#include <thread>
#include <unordered_map>
#include <chrono>
#include <vector>

extern "C" hipError_t rtk_launch_trace(const RtDevScene *scene, int counted, hipStream_t stream);
extern "C" hipError_t rtk_launch_prepare(uint32_t triangleCount, const void *vertex, const void *triIndex, const void *triMaterial,
                                         const void *triUv, const void *triNormal, float *triRec, float *triShade, hipStream_t stream);
extern "C" hipError_t rtk_launch_gather_pairs(uint32_t pairCount, const uint32_t *pairTri, const uint32_t *pairInfo, const float *triRec, float *pairRec, hipStream_t stream);
extern "C" hipError_t rtk_launch_detile(const void *tileBuf, const uint32_t *tileIds, uint32_t tileCount, uint32_t width,
                                        uint32_t height, uint32_t tilesX, void *planeR, void *planeG, void *planeB, int accumulate, hipStream_t stream);

extern "C" hipError_t rtp_validate(uint32_t T, uint32_t V, uint32_t M, const void *triIndex, const int *triMaterial, uint64_t camListSize,
                                   const uint32_t *camList, const uint32_t *gridStart, uint64_t gridListSize, const uint32_t *gridList, uint32_t *err,
                                   hipStream_t stream);
extern "C" hipError_t rtp_camera_ranges(uint32_t W, uint32_t H, uint32_t tilesX, const uint32_t *tileIds, uint32_t tileCount, const uint32_t *camStart,
                                        const uint32_t *camEnd, uint64_t camListSize, uint32_t *outStart, uint32_t *outEnd, uint32_t *err, hipStream_t stream);
extern "C" hipError_t rtp_dense_grid(const uint32_t *gridStart, const uint32_t *gridList, unsigned long long *words, uint32_t *sparse, uint32_t *pairOrder,
                                     uint32_t *pairCount, void *scratch, size_t *scratchBytes, hipStream_t stream);

extern "C" hipError_t rtw_launch_primary(const RtDevScene *scene, const RtWavefront *wf, hipStream_t stream);
extern "C" hipError_t rtw_launch_logic(const RtDevScene *scene, const RtWavefront *wf, uint32_t round, uint32_t blocks, uint32_t slicesIn, const RtRoundMode *next, hipStream_t stream);
extern "C" hipError_t rtw_launch_scatter(const RtWavefront *wf, uint32_t round, uint32_t blocks, const RtRoundMode *mode, hipStream_t stream);
extern "C" hipError_t rtw_launch_trace(const RtDevScene *scene, const RtWavefront *wf, uint32_t round, uint32_t blocks, const RtRoundMode *mode, hipStream_t stream);
extern "C" hipError_t rtw_launch_status(const RtWavefront *wf, uint32_t round, hipStream_t stream);
extern "C" hipError_t rtw_launch_accum(const RtDevScene *scene, const RtWavefront *wf, int first, hipStream_t stream);

namespace {

thread_local std::string g_error;

int fail(const char *fmt, ...)
{
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_error = buf;
    return -1;
}

#define HIP_OK(expr)                                                                                   \
    do {                                                                                               \
        hipError_t e_ = (expr);                                                                        \
        if (e_ != hipSuccess) return fail("%s failed: %s", #expr, hipGetErrorString(e_));              \
    } while (0)

// Tuning values and test hooks.  The library reads NO environment variables: a plugin host's environment must not be able to
// make frames slower, redo themselves or fail.  Everything here is set through rtHipTune() (include/raytrace_hip.h, test / tuning
// entry point; the Python stub maps RT_* variables of ITS process onto it for the sweep scripts) and applies to scenes built
// afterwards.
struct Tuning {
    uint32_t stageMb = 32;          // size of each of the two pinned staging buffers of an upload
    uint32_t extraFactor = 6;       // region B of the entry arrays (further segments of cut rays), in units of the path capacity
    uint64_t stateMb = 0;           // path-state budget per sample batch (0 = 24 GB, never more than a third of free memory); tests force several batches
    uint32_t groups = 1;            // concurrent tile groups per instance (measured: no gain once rays are cut into segments)
    uint32_t lookAhead = 1;         // 0: one ray in flight per path
    uint32_t segLen[5] = { 4096u, 384u, 96u, 64u, 16u }; // aimed-at cell visits per segment for rounds with >= segRays[0] | [1] | [2] | [3] | fewer rays
    uint32_t segRays[4] = { 700000u, 300000u, 100000u, 30000u };
    uint32_t fastQuotient = 1;
    uint32_t spinLimit = 16384;     // a ray makes at most 766 cell visits = 154 walk phases; lowered by the test of the guard's error path
    uint32_t appendRays = 300000;   // later rounds with fewer rays are not ordered: the trace kernel plans and cuts their rays itself (RtRoundMode)
    uint32_t orderedFirst = 1;      // 0: round 1 follows the same rule (tests: the trace kernel's planning on dense rounds)
    uint32_t sliceRays = 0;         // rounds with fewer rays use smallSlices queue slices per kind instead of RT_WF_SHARDS (off: a round's appends want
                                    // many counters -- 16 slices cost the logic kernel of a 58 k-ray round 17 us -- and the trace kernel packs its pieces anyway)
    uint32_t smallSlices = 16;
    uint32_t groupRays = 0;         // rays per workgroup of the trace kernel in rounds that are not ordered (0 = by segment length)
    uint32_t blocking = 0;          // 1: every frame watches its queue (no launch plan)
    uint32_t planRounds = 0;        // test hook: planned frames issue at most this many rounds, so that the too-short-plan path runs
    uint32_t planGridTiny = 0;      // test hook: planned trace grids of one workgroup, so that the too-small-grid path runs
    uint32_t pipeline = RT_HIP_PIPELINE_WAVEFRONT;
    uint32_t timing = 0;            // 1: where the time of a scene build / a RaytraceAll call goes (stderr)
    uint32_t virtualDevices = 0;    // test hook: the all-GPUs id deals the tiles over this many instances on the devices that are there
    uint32_t cache = 1;             // 0: RaytraceAll builds and frees per call, like the reference
    uint32_t batchPlan = 1;         // 1: the sample batches after a watched frame's first are issued from that batch's launch plan
};
Tuning g_tune;
std::mutex g_tuneMutex;
Tuning tuning() { std::lock_guard<std::mutex> lock(g_tuneMutex); return g_tune; }

// Device scratch of a scene build: freed when the scope ends, whichever way it ends.
struct DevScratch {
    std::vector<void *> blocks;
    ~DevScratch() { for (void *p : blocks) (void)hipFree(p); }
    hipError_t get(void **out, size_t bytes)
    {
        const hipError_t e = hipMalloc(out, bytes ? bytes : 1);
        if (e == hipSuccess) blocks.push_back(*out);
        return e;
    }
};

// Host-to-device copies go through two pinned buffers (hipHostMalloc once per scene, RT_HIP_STAGE_MB each, default 32): the
// caller's arrays are pageable (new[] in render.cpp:1089-1123), and a pageable hipMemcpy is a synchronous bounce through the
// runtime's own small staging area.  Here the CPU fills one buffer (several threads for big pieces) while the DMA engine
// drains the other; copy() returns when the source has been read completely, so callers may free it at once.
struct Stager {
    hipStream_t stream = nullptr;
    char *buf[2] = { nullptr, nullptr };
    hipEvent_t done[2] = { nullptr, nullptr };
    bool used[2] = { false, false };
    size_t size = 0;
    int next = 0;
    // Pinning 2 x 32 MB costs 15-25 ms, more than the rest of an instance's build when its shared parts are copied from another
    // instance: the buffers are taken when the first host array needs them and go back to a process-wide pool, not to the driver.
    struct Pool {
        std::mutex lock;
        std::vector<std::pair<char *, size_t>> idle;
        char *take(size_t bytes)
        {
            {
                std::lock_guard<std::mutex> g(lock);
                for (size_t i = 0; i < idle.size(); ++i)
                    if (idle[i].second == bytes) { char *p = idle[i].first; idle.erase(idle.begin() + i); return p; }
            }
            char *p = nullptr;
            if (hipHostMalloc((void **)&p, bytes, hipHostMallocPortable) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
            return p;
        }
        void give(char *p, size_t bytes)
        {
            std::lock_guard<std::mutex> g(lock);
            if (idle.size() < 8) idle.emplace_back(p, bytes);
            else (void)hipHostFree(p);
        }
    };
    static Pool &pool() { static Pool *p = new Pool(); return *p; } // (never destroyed: buffers may come back during process exit)
    int init(hipStream_t st)
    {
        stream = st;
        size = (size_t)std::min<uint32_t>(std::max<uint32_t>(tuning().stageMb, 1u), 4096u) << 20;
        for (int i = 0; i < 2; ++i) HIP_OK(hipEventCreateWithFlags(&done[i], hipEventDisableTiming));
        return 0;
    }
    static void fill(char *dst, const char *src, size_t n)
    {
        const size_t piece = (size_t)4 << 20;
        if (n < 2 * piece) { memcpy(dst, src, n); return; }
        const size_t parts = std::min<size_t>(8, n / piece);
        std::vector<std::thread> pool;
        for (size_t t = 1; t < parts; ++t) pool.emplace_back([=] { memcpy(dst + n * t / parts, src + n * t / parts, n * (t + 1) / parts - n * t / parts); });
        memcpy(dst, src, n / parts);
        for (auto &th : pool) th.join();
    }
    // is `p` device memory (a scene description may hand over arrays that are already on the GPU)?
    static bool on_device(const void *p)
    {
        hipPointerAttribute_t at;
        memset(&at, 0, sizeof at);
        if (!p || hipPointerGetAttributes(&at, p) != hipSuccess) { (void)hipGetLastError(); return false; } // (plain host memory: "invalid value")
        return at.type == hipMemoryTypeDevice;
    }
    hipError_t copy(void *dst, const void *src, size_t bytes)
    {
        if (bytes && on_device(src)) return hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, stream); // no staging: HBM to HBM
        for (size_t off = 0; off < bytes;) {
            const size_t n = std::min(size, bytes - off);
            const int i = next;
            next ^= 1;
            if (used[i]) { const hipError_t e = hipEventSynchronize(done[i]); if (e != hipSuccess) return e; }
            if (!buf[i] && !(buf[i] = pool().take(size))) return hipErrorOutOfMemory;
            fill(buf[i], (const char *)src + off, n);
            hipError_t e = hipMemcpyAsync((char *)dst + off, buf[i], n, hipMemcpyHostToDevice, stream);
            if (e != hipSuccess) return e;
            e = hipEventRecord(done[i], stream);
            if (e != hipSuccess) return e;
            used[i] = true;
            off += n;
        }
        return hipSuccess;
    }
    hipError_t drain() // every copy so far has left its staging buffer; the buffers go back to the pool (the next instance's build takes them)
    {
        for (int i = 0; i < 2; ++i) {
            if (used[i]) { const hipError_t e = hipEventSynchronize(done[i]); if (e != hipSuccess) return e; used[i] = false; }
            if (buf[i]) { pool().give(buf[i], size); buf[i] = nullptr; }
        }
        return hipSuccess;
    }
    void destroy()
    {
        for (int i = 0; i < 2; ++i) {
            if (done[i]) { if (used[i]) (void)hipEventSynchronize(done[i]); (void)hipEventDestroy(done[i]); done[i] = nullptr; }
            if (buf[i]) { pool().give(buf[i], size); buf[i] = nullptr; }
        }
    }
};

enum { PART_FIXED = 0, PART_CAMERA, PART_GEOMETRY, PART_GRID, PART_MATERIALS, PART_LIGHTS, PART_WAVEFRONT, PART_COUNT };

} // namespace

struct rtHipScene {
    int device = -1;
    hipStream_t stream = nullptr;
    RtDevScene dev{};
    // Device allocations by PART (tiles + outputs | camera lists | geometry | grid | materials | lights | path state): the drop-in
    // layer's cache replaces the parts whose inputs changed between two RaytraceAll calls and keeps the others in HBM.
    std::vector<void *> partAllocs[PART_COUNT];
    std::vector<uint64_t> partSizes[PART_COUNT]; // bytes of every allocation (a peer instance copies the shared parts device to device)
    uint64_t partBytes[PART_COUNT] = { 0 };
    int curPart = PART_FIXED;
    Stager stager;
    uint32_t *prepErr = nullptr;   // device word: RT_PREP_ERR_* bits raised by the validation kernels
    uint64_t camListSize = 0;
    uint32_t gridListSizeHint = 0;  // scene description with device arrays: scenePixelTriangleListStart[256^3], fetched by scene_build
    bool haveGridListSize = false;
    bool wfMultiLight = false;     // what the path-state buffers were sized for
    uint64_t bytes = 0;
    std::vector<cl_uint> tileIds;
    uint32_t width = 0, height = 0, tilesX = 0;
    // kernel timing: one event pair per launch since the last rtHipKernelTime
    std::vector<std::pair<hipEvent_t, hipEvent_t>> events;
    size_t eventsUsed = 0;
    // wavefront pipeline (rt_wavefront.hip)
    int pipeline = RT_HIP_PIPELINE_WAVEFRONT;
    // The instance's tile slots are cut into contiguous GROUPS, each a view of `dev` (its own slice of tileIds, camStart/End,
    // tileBuf) with its own path state and stream.  A frame runs the groups concurrently: while one group is in a phase that
    // cannot fill the GPU (a round with few rays, a host read-back), the others' kernels do.  Per pixel nothing changes.
    struct Group {
        RtDevScene dev{};
        RtWavefront wf{};
        hipStream_t stream = nullptr;   // groups 1.. ; group 0 runs on the caller's stream
        hipEvent_t done = nullptr;
        uint32_t logicBlocks = 1, traceBlocks = 1, queueBlocks = 1;
        uint32_t *hostCount = nullptr;  // pinned: queue length read back between round chunks
        uint32_t *hostStatus = nullptr; // pinned + mapped: RT_WF_STATUS_* words the kernels write (rt_device.h)
        bool ctlClean = false;          // the batch before was a planned one: wf_status_kernel left the control words zeroed
        uint32_t rounds = 0;
        uint32_t slot0 = 0, slot1 = 0;  // this group's range of the instance's tile slots
        // launch plan (render_wavefront): what the last discovery frame needed
        uint32_t roundsNeeded = 0;
        // per round: rays (entries in region A), the longest queue slice, entries in region B -- the maximum over the watched batches
        // rays of the round, and -- an ordered round -- the further segments of its cut rays under the cut it was logged with
        struct RoundPlan { uint32_t rays = 0, extra = 0, extraSegLen = 0; };
        uint4 *hostLog = nullptr;       // pinned + mapped: RtWavefront::roundLog, written by the kernels, read by the host after a sync
        RoundPlan plan[RT_WF_ROUND_LOG], planNext[RT_WF_ROUND_LOG];
        std::vector<RtRoundMode> modes; // how the rounds of the batch being issued are laid out (modes[r] is decided when logic(r-1) is launched)
        uint64_t guessRays = 0;         // watched batches: what the next round is assumed to hold
    };
    std::vector<Group> groups;
    hipEvent_t forkEvent = nullptr;
    uint32_t samplesPerBatch = 1;
    uint32_t planRounds = 0;   // rounds a planned frame issues per batch; 0 = no plan yet (the next frame is a discovery frame)
    bool blocking = false;     // every frame watches the queue (no plan)
    Tuning tune;               // the tuning values this scene was built with (rtHipTune)
    bool unverified = false;   // planned frames were issued since the last frame_finish()
    hipStream_t lastStream = nullptr; // where the last frame was issued
    std::atomic<float> *progress = nullptr; // drop-in layer: where finished sample batches are reported (GetProgress, raytrace.c:566-587)
    float progressBase = 0.f, progressSpan = 0.f;
    // per-stage device time of the frames since the last query: [primary, logic, trace, accum, sort]
    struct StageEvent { int stage; hipEvent_t a, b; };
    std::vector<StageEvent> stageEvents;
    size_t stageEventsUsed = 0;
    bool stageTiming = false;
    uint64_t roundsLast = 0;

    template <class T> int upload(const T *src, uint64_t count, const T **dst, const char *what)
    {
        void *p = nullptr;
        const uint64_t n = count ? count : 1;
        HIP_OK(hipMalloc(&p, n * sizeof(T)));
        partAllocs[curPart].push_back(p);
        partSizes[curPart].push_back(n * sizeof(T));
        partBytes[curPart] += n * sizeof(T);
        bytes += n * sizeof(T);
        if (count) {
            if (!src) return fail("%s: null pointer with %llu elements", what, (unsigned long long)count);
            HIP_OK(stager.copy(p, src, count * sizeof(T)));
        }
        *dst = (const T *)p;
        return 0;
    }
    template <class T> int alloc(uint64_t count, T **dst)
    {
        void *p = nullptr;
        const uint64_t n = count ? count : 1;
        HIP_OK(hipMalloc(&p, n * sizeof(T)));
        partAllocs[curPart].push_back(p);
        partSizes[curPart].push_back(n * sizeof(T));
        partBytes[curPart] += n * sizeof(T);
        bytes += n * sizeof(T);
        *dst = (T *)p;
        return 0;
    }
    void release_part(int part)
    {
        if (partAllocs[part].empty()) return;
        if (stream) (void)hipStreamSynchronize(stream);
        for (void *p : partAllocs[part]) (void)hipFree(p);
        partAllocs[part].clear();
        partSizes[part].clear();
        bytes -= partBytes[part];
        partBytes[part] = 0;
    }
    // looks at the device-side validation word (after the caller's stream synchronisation)
    int check_prep()
    {
        uint32_t err = 0;
        HIP_OK(hipMemcpy(&err, prepErr, 4, hipMemcpyDeviceToHost));
        if (!err) return 0;
        HIP_OK(hipMemset(prepErr, 0, 4));
        return fail("scene rejected (0x%x):%s%s%s%s%s%s", err,
                    (err & RT_PREP_ERR_TRI_INDEX) ? " a triangle references a vertex that does not exist;" : "",
                    (err & RT_PREP_ERR_TRI_MATERIAL) ? " a triangle uses a material >= materialCount;" : "",
                    (err & RT_PREP_ERR_CAM_ENTRY) ? " a camera list entry is not a triangle;" : "",
                    (err & RT_PREP_ERR_CAM_RANGE) ? " a camera list range exceeds the list size;" : "",
                    (err & RT_PREP_ERR_GRID_MONOTONE) ? " scenePixelTriangleListStart is not monotone;" : "",
                    (err & RT_PREP_ERR_GRID_ENTRY) ? " a grid list entry is not a triangle;" : "");
    }
};

namespace {

// ---- scene parts -------------------------------------------------------------------------------------------------------
// A resident scene is built in PARTS with their own device allocations, so that the drop-in layer's cache (RaytraceAll
// below) can replace what changed between two calls -- typically the camera -- and keep the rest in HBM.
int build_fixed(rtHipScene *sc, const rtHipSceneDesc *d, const cl_uint *tileIds, cl_uint tileCount)
{
    sc->curPart = PART_FIXED;
    const uint32_t tilesX = (d->width + RT_TILE - 1) / RT_TILE, tilesY = (d->height + RT_TILE - 1) / RT_TILE;
    sc->width = d->width; sc->height = d->height; sc->tilesX = tilesX;
    const bool allTiles = (tileIds == nullptr || tileCount == 0);
    if (allTiles) {
        sc->tileIds.resize((size_t)tilesX * tilesY);
        for (uint32_t i = 0; i < tilesX * tilesY; ++i) sc->tileIds[i] = i;
    } else {
        sc->tileIds.assign(tileIds, tileIds + tileCount);
        for (cl_uint t : sc->tileIds)
            if (t >= tilesX * tilesY) return fail("tile id %u out of range (%u tiles)", t, tilesX * tilesY);
    }
    const uint32_t nt = (uint32_t)sc->tileIds.size();
    RtDevScene &D = sc->dev;
    D.width = d->width; D.height = d->height;
    D.tileCount = nt; D.tilesX = tilesX;
    const cl_uint *ids = nullptr;
    if (sc->upload(sc->tileIds.data(), nt, &ids, "tileIds")) return -1;
    D.tileIds = ids;
    uint16_t *buf = nullptr;
    if (sc->alloc<uint16_t>((uint64_t)nt * 3 * RT_TILE_PIXELS, &buf)) return -1;
    HIP_OK(hipMemsetAsync(buf, 0, (uint64_t)nt * 3 * RT_TILE_PIXELS * 2, sc->stream));
    D.tileBuf = buf;
    unsigned long long *st = nullptr;
    if (sc->alloc<unsigned long long>(8, &st)) return -1;
    HIP_OK(hipMemsetAsync(st, 0, 64, sc->stream));
    D.stats = st;
    if (sc->alloc<uint32_t>(1, &sc->prepErr)) return -1;
    HIP_OK(hipMemsetAsync(sc->prepErr, 0, 4, sc->stream));
    // bump: xPart/yPart = (float)sin(dh*PI_F/2.f), normalPart factors = (float)cos(...) (raytrace_opencl.c:251-253) with
    // dh = hE/255.f - h0/255.f.  Only 256x256 byte pairs exist: tabulate with the HOST libm -- the library the reference's
    // C path calls -- so the device result is that library's, bit for bit.
    std::vector<float> tsin(65536), tcos(65536);
    for (int e = 0; e < 256; ++e)
        for (int h = 0; h < 256; ++h) {
            const float fe = (float)e / 255.f, fh = (float)h / 255.f;
            const float arg = (fe - fh) * 3.14159265f / 2.f;
            tsin[(e << 8) | h] = (float)std::sin((double)arg);
            tcos[(e << 8) | h] = (float)std::cos((double)arg);
        }
    if (sc->upload(tsin.data(), tsin.size(), &D.bumpSin, "bumpSin")) return -1;
    if (sc->upload(tcos.data(), tcos.size(), &D.bumpCos, "bumpCos")) return -1;
    HIP_OK(sc->stager.drain());
    return 0;
}

// camera vectors + the per-pixel candidate lists: uploaded as they are, the tile-major view of this instance's tiles is made
// on the device (rt_scene_prep.hip)
int build_camera(rtHipScene *sc, const rtHipSceneDesc *d)
{
    sc->release_part(PART_CAMERA);
    sc->curPart = PART_CAMERA;
    RtDevScene &D = sc->dev;
    for (int i = 0; i < 3; ++i) { D.eye[i] = d->eye[i]; D.topLeft[i] = d->eyeToTopLeft[i]; D.lr[i] = d->leftToRight[i]; D.tb[i] = d->topToBottom[i]; }
    D.pixelSizeInv = d->pixelSizeInv;
    const uint64_t P = (uint64_t)d->width * d->height;
    if (!d->camStart || !d->camEnd) return fail("null cameraPixelTriangleListStart/End");
    if (d->camListSize && !d->camList) return fail("null cameraPixelTriangleList");
    if (d->camListSize > 0xffffffffull) return fail("camera list too large");
    const cl_uint *srcStart = nullptr, *srcEnd = nullptr;
    DevScratch scratch; // the row-major ranges are only needed until the tile-major ones exist
    void *p0 = nullptr, *p1 = nullptr;
    HIP_OK(scratch.get(&p0, P * 4)); HIP_OK(scratch.get(&p1, P * 4));
    HIP_OK(sc->stager.copy(p0, d->camStart, P * 4));
    HIP_OK(sc->stager.copy(p1, d->camEnd, P * 4));
    srcStart = (const cl_uint *)p0; srcEnd = (const cl_uint *)p1;
    if (sc->upload(d->camList, d->camListSize, &D.camList, "camList")) return -1;
    sc->camListSize = d->camListSize;
    uint32_t *ts = nullptr, *te = nullptr;
    const uint64_t n = (uint64_t)D.tileCount * RT_TILE_PIXELS;
    if (sc->alloc<uint32_t>(n, &ts) || sc->alloc<uint32_t>(n, &te)) return -1;
    HIP_OK(rtp_camera_ranges(d->width, d->height, D.tilesX, D.tileIds, D.tileCount, srcStart, srcEnd, d->camListSize, ts, te, sc->prepErr, sc->stream));
    D.camStart = ts; D.camEnd = te;
    HIP_OK(hipStreamSynchronize(sc->stream)); // scratch dies here
    return 0;
}

// geometry: upload the ABI arrays, reshape on the device (rt_prepare_triangles), drop the originals
int build_geometry(rtHipScene *sc, const rtHipSceneDesc *d)
{
    sc->release_part(PART_GEOMETRY);
    sc->curPart = PART_GEOMETRY;
    RtDevScene &D = sc->dev;
    D.triangleCount = d->triangleCount;
    if (d->triangleCount && (!d->triIndex || !d->triMaterial || !d->triUv || !d->triNormal)) return fail("null triangle array");
    if (d->vertexCount && !d->vertex) return fail("null vertex array");
    void *dv = nullptr, *di = nullptr, *dm = nullptr, *du = nullptr, *dn = nullptr;
    const uint64_t T = d->triangleCount, V = d->vertexCount;
    DevScratch scratch;
    HIP_OK(scratch.get(&dv, V * 16)); HIP_OK(scratch.get(&di, T * 16)); HIP_OK(scratch.get(&dm, T * 4));
    HIP_OK(scratch.get(&du, T * 24)); HIP_OK(scratch.get(&dn, T * 48));
    HIP_OK(sc->stager.copy(dv, d->vertex, V * 16));
    HIP_OK(sc->stager.copy(di, d->triIndex, T * 16));
    HIP_OK(sc->stager.copy(dm, d->triMaterial, T * 4));
    HIP_OK(sc->stager.copy(du, d->triUv, T * 24));
    HIP_OK(sc->stager.copy(dn, d->triNormal, T * 48));
    float *rec = nullptr, *shade = nullptr;
    if (sc->alloc<float>(T * 16, &rec) || sc->alloc<float>(T * 24, &shade)) return -1;
    // ids are checked on the device before anything gathers through them; a scene with a bad id fails at the end of the build
    HIP_OK(rtp_validate(d->triangleCount, d->vertexCount, d->materialCount, di, (const int *)dm, 0, nullptr, nullptr, 0, nullptr, sc->prepErr, sc->stream));
    HIP_OK(hipStreamSynchronize(sc->stream));
    if (sc->check_prep() != 0) return -1; // rt_prepare_triangles would gather out of bounds
    HIP_OK(rtk_launch_prepare(d->triangleCount, dv, di, dm, du, dn, rec, shade, sc->stream));
    D.triRec = rec; D.triShade = shade;
    HIP_OK(hipStreamSynchronize(sc->stream));
    return 0;
}

// the grid as the ABI hands it over, plus its dense view for the wavefront trace kernel, built on the device
int build_grid(rtHipScene *sc, const rtHipSceneDesc *d)
{
    sc->release_part(PART_GRID);
    sc->curPart = PART_GRID;
    RtDevScene &D = sc->dev;
    if (!d->boxMin) return fail("null sceneBoxMin");
    std::vector<float> planes(3 * (RT_GRID_DIV + 1));
    for (int w = 0; w < 3; ++w)
        for (int i = 0; i <= RT_GRID_DIV; ++i) planes[w * (RT_GRID_DIV + 1) + i] = d->boxMin[i].s[w];
    if (sc->upload(planes.data(), planes.size(), &D.boxMin, "boxMin")) return -1;
    { // the cell estimate table of rt_device.h (cellLut)
        std::vector<uint8_t> lut(3 * 256);
        for (int w = 0; w < 3; ++w) {
            const float *pw = planes.data() + w * (RT_GRID_DIV + 1);
            const float lo = pw[0], step = (pw[RT_GRID_DIV] - pw[0]) / 256.f;
            int c = 0;
            for (int i = 0; i < 256; ++i) {
                const float x = lo + ((float)i + 0.5f) * step;
                while (c < RT_GRID_DIV - 1 && pw[c + 1] < x) ++c; // (planes ascend: the cell index only ever grows with i)
                lut[w * 256 + i] = (uint8_t)c;
            }
        }
        if (sc->upload(lut.data(), lut.size(), &D.cellLut, "cellLut")) return -1;
    }
    D.planesTame = 1u;
    for (float pl : planes) {
        const float m = std::fabs(pl);
        if (!(m == 0.f || (m >= 0x1p-60f && m <= 0x1p39f))) D.planesTame = 0u;
    }
    const uint64_t cells = (uint64_t)RT_GRID_DIV * RT_GRID_DIV * RT_GRID_DIV;
    if (!d->gridStart) return fail("null scenePixelTriangleListStart");
    const uint64_t listSize = sc->haveGridListSize ? sc->gridListSizeHint : d->gridStart[cells]; // (the last start = the list's length; read from the device by scene_build when the array lives there)
    if (listSize && !d->gridList) return fail("null scenePixelTriangleList");
    if (sc->upload(d->gridStart, cells + 1, &D.gridStart, "gridStart")) return -1;
    if (sc->upload(d->gridList, listSize, &D.gridList, "gridList")) return -1;
    // (the triangle count the entries are checked against is this scene's: geometry is built before the grid)
    HIP_OK(rtp_validate(D.triangleCount, 0, 0, nullptr, nullptr, 0, nullptr, D.gridStart, listSize, D.gridList, sc->prepErr, sc->stream));
    HIP_OK(hipStreamSynchronize(sc->stream));
    if (sc->check_prep() != 0) return -1; // the dense view walks the lists through the starts
    const size_t blocks = (size_t)(RT_GRID_DIV / 4) * (RT_GRID_DIV / 4) * (RT_GRID_DIV / 4);
    unsigned long long *words = nullptr;
    uint32_t *sparse = nullptr;
    const size_t sparseWords = (size_t)3 * ((63u << 16 | 63u << 8 | 63u) + 1u);
    if (listSize >= RT_PAIR_LIMIT) return fail("scenePixelTriangleList has 2^28 entries or more: pair indices would not fit the trace kernel's records");
    if (sc->alloc<unsigned long long>(blocks, &words) || sc->alloc<uint32_t>(sparseWords, &sparse)) return -1;
    HIP_OK(hipMemsetAsync(sparse, 0, sparseWords * 4, sc->stream));
    DevScratch scratch;
    uint32_t *pairOrder = nullptr, *pairCount = nullptr;
    void *tmp = nullptr;
    size_t tmpBytes = 0;
    HIP_OK(rtp_dense_grid(nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, &tmpBytes, sc->stream));
    HIP_OK(scratch.get((void **)&pairOrder, (size_t)listSize * 4)); HIP_OK(scratch.get((void **)&pairCount, (size_t)listSize * 4));
    HIP_OK(scratch.get(&tmp, tmpBytes));
    HIP_OK(rtp_dense_grid(D.gridStart, D.gridList, words, sparse, pairOrder, pairCount, tmp, &tmpBytes, sc->stream));
    float *pairRec = nullptr;
    if (sc->alloc<float>((uint64_t)listSize * 16, &pairRec)) return -1;
    HIP_OK(rtk_launch_gather_pairs((uint32_t)listSize, pairOrder, pairCount, D.triRec, pairRec, sc->stream));
    D.gridBits = words; D.gridBlockSparse = sparse; D.pairRec = pairRec;
    D.cellCount = 0; // informational; the kernels find a cell's records through the block table
    HIP_OK(hipStreamSynchronize(sc->stream));
    return 0;
}

int build_materials(rtHipScene *sc, const rtHipSceneDesc *d)
{
    sc->release_part(PART_MATERIALS);
    sc->curPart = PART_MATERIALS;
    RtDevScene &D = sc->dev;
    D.materialCount = d->materialCount;
    D.texelCount = d->texturesSize ? d->texturesSize : 1;
    if (sc->upload((const uint32_t *)d->matSize, (uint64_t)d->materialCount * 10, &D.matSize, "materialImageSize")) return -1;
    if (sc->upload((const int32_t *)d->matStart, (uint64_t)d->materialCount * 5, &D.matStart, "materialImageStart")) return -1;
    if (sc->upload((const uint8_t *)d->textures, (uint64_t)d->texturesSize * 4, &D.textures, "textures")) return -1;
    for (uint32_t m = 0; m < d->materialCount * 5; ++m) {
        const uint64_t w = d->matSize[m].s[0], h = d->matSize[m].s[1];
        if (w && ((int64_t)d->matStart[m] < 0 || (uint64_t)d->matStart[m] + w * h > d->texturesSize))
            return fail("material channel %u: %llux%llu texels at %d exceed the %u-texel atlas", m, (unsigned long long)w, (unsigned long long)h, d->matStart[m], d->texturesSize);
    }
    { // the per-material channel descriptors of rt_device.h
        std::vector<uint32_t> rec((size_t)d->materialCount * 8, 0u);
        for (uint32_t m = 0; m < d->materialCount; ++m)
            for (int c = 0; c < 5; ++c) {
                const uint32_t w = d->matSize[5 * m + c].s[0], h = d->matSize[5 * m + c].s[1];
                uint32_t desc = 0u;
                if (w == 1u && h == 1u) {
                    const cl_uchar *px = d->textures[d->matStart[5 * m + c]].s; // (start was checked above)
                    desc = 0x80000000u | (uint32_t)px[0] | ((uint32_t)px[1] << 8) | ((uint32_t)px[2] << 16);
                } else if (w) desc = 0x40000000u; // (w > 0, h == 0: an image without rows -- the general path indexes it as the reference would)
                rec[8 * (size_t)m + c] = desc;
            }
        if (sc->upload(rec.data(), rec.size(), &D.matRec, "matRec")) return -1;
    }
    HIP_OK(sc->stager.drain());
    return 0;
}

int build_lights(rtHipScene *sc, const rtHipSceneDesc *d)
{
    sc->release_part(PART_LIGHTS);
    sc->curPart = PART_LIGHTS;
    RtDevScene &D = sc->dev;
    if (d->lightCount >= 65536u) return fail("lightCount %u too large", d->lightCount);
    D.lightCount = d->lightCount;
    if (d->lightCount && (!d->lightType || !d->lightPos || !d->lightDir || !d->lightCol || !d->lightRadius || !d->lightHalfAtt))
        return fail("null light array with lightCount %u", d->lightCount); // (the spread below reads two of them before upload() would notice)
    std::vector<float> spread(d->lightCount ? d->lightCount : 1, 0.f);
    for (uint32_t j = 0; j < d->lightCount; ++j) {
        const float *ld = d->lightDir[j].s;
        const float dd = ld[0] * ld[0] + ld[1] * ld[1] + ld[2] * ld[2];
        // raytrace_opencl.c:594 (double sin * double sqrt, then one rounding to float)
        spread[j] = (float)(std::sin((double)((d->lightRadius[j] / 2.f) * 3.14159265f / 180.f)) * std::sqrt((double)dd));
    }
    if (sc->upload(d->lightType, d->lightCount, &D.lightType, "lightType")) return -1;
    if (sc->upload((const float *)d->lightPos, (uint64_t)d->lightCount * 4, &D.lightPos, "lightPosition")) return -1;
    if (sc->upload((const float *)d->lightDir, (uint64_t)d->lightCount * 4, &D.lightDir, "lightDirection")) return -1;
    if (sc->upload((const float *)d->lightCol, (uint64_t)d->lightCount * 4, &D.lightCol, "lightColour")) return -1;
    if (sc->upload(d->lightRadius, d->lightCount, &D.lightRadius, "lightRadius")) return -1;
    if (sc->upload(d->lightHalfAtt, d->lightCount, &D.lightHalfAtt, "lightHalfAttenuationDistance")) return -1;
    if (sc->upload(spread.data(), d->lightCount, &D.lightSpread, "lightSpread")) return -1;
    HIP_OK(sc->stager.drain());
    return 0;
}

// wavefront pipeline buffers: worst case every pixel of every sample in a batch becomes a path
int build_wavefront(rtHipScene *sc, uint32_t sampleCount)
{
    for (auto &G : sc->groups) {
        if (G.hostCount) (void)hipHostFree(G.hostCount);
        if (G.hostStatus) (void)hipHostFree(G.hostStatus);
        if (G.hostLog) (void)hipHostFree(G.hostLog);
        if (G.stream) { (void)hipStreamSynchronize(G.stream); (void)hipStreamDestroy(G.stream); }
        if (G.done) (void)hipEventDestroy(G.done);
    }
    sc->groups.clear();
    sc->release_part(PART_WAVEFRONT);
    sc->curPart = PART_WAVEFRONT;
    RtDevScene &D = sc->dev;
    D.sampleCount = sampleCount;
    sc->planRounds = 0; // the next frame watches its queue again
    sc->unverified = false;
    const Tuning &T = sc->tune;
    const uint32_t nt = D.tileCount;
    hipDeviceProp_t prop;
    HIP_OK(hipGetDeviceProperties(&prop, sc->device));
    const uint32_t cus = prop.multiProcessorCount > 0 ? (uint32_t)prop.multiProcessorCount : 256u;
    const uint64_t pix = (uint64_t)nt * RT_TILE_PIXELS;
    const uint32_t extraFactor = std::min<uint32_t>(std::max<uint32_t>(T.extraFactor, 1u), 16u);
    // per path: rng, meta, outc, ring, shadow-wait state, look-ahead answer + slot, primary hit, finished colour; per queue entry (two per
    // path): path id and answer per round parity; per entry (2 + extraFactor per path): the 64-byte entry per round parity, rank, class,
    // sorted position
    const uint64_t perPath = 8 + 16 + 16 + (uint64_t)RT_RING * 48 + 3 * 16 + 8 + 4 + 16 + 16 + 2 * 2 * (4 + 8) + (uint64_t)(2 + extraFactor) * (2 * 64 + 4 + 2 + 4);
    // bytes of path state per sample batch: more samples per batch = fewer, fuller rounds (S=4 at 1080p: 5.0 ms with one
    // sample per batch, 4.5 ms with all four); 24 GB of the 288 GB, and never more than a third of what is free
    uint64_t budget = 24ull << 30;
    {
        size_t freeB = 0, totalB = 0;
        if (hipMemGetInfo(&freeB, &totalB) == hipSuccess && freeB / 3 < budget) budget = freeB / 3;
    }
    if (T.stateMb) budget = T.stateMb << 20;
    uint64_t sb = budget / (perPath * (pix ? pix : 1));
    if (sb < 1) sb = 1;
    if (sb > sampleCount) sb = sampleCount;
    if (sb > 65535) sb = 65535; // the primary kernel's gridDim.y
    sc->samplesPerBatch = (uint32_t)sb;
    uint32_t groupCount = std::min<uint32_t>(std::max<uint32_t>(T.groups, 1u), 16u);
    if (groupCount > nt) groupCount = nt ? (uint32_t)nt : 1u;
    const bool multiLight = D.lightCount > 1;
    sc->wfMultiLight = multiLight;
    if (!sc->forkEvent) HIP_OK(hipEventCreateWithFlags(&sc->forkEvent, hipEventDisableTiming));
    sc->groups.resize(groupCount);
    for (uint32_t g = 0; g < groupCount; ++g) {
        rtHipScene::Group &G = sc->groups[g];
        const uint32_t slot0 = (uint32_t)((uint64_t)nt * g / groupCount), slot1 = (uint32_t)((uint64_t)nt * (g + 1) / groupCount);
        G.slot0 = slot0; G.slot1 = slot1;
        if (g > 0) HIP_OK(hipStreamCreateWithFlags(&G.stream, hipStreamNonBlocking));
        HIP_OK(hipEventCreateWithFlags(&G.done, hipEventDisableTiming));
        // queue slices: the primary kernel's workgroups are dealt to the shards round-robin, 256 paths each at most
        const uint64_t gpix = (uint64_t)(slot1 - slot0) * RT_TILE_PIXELS;
        const uint64_t primaryBlocks = (uint64_t)(slot1 - slot0) * 64 * sb;
        const uint64_t shardCap = ((primaryBlocks + RT_WF_SHARDS - 1) / RT_WF_SHARDS) * 256;
        const uint64_t cap = shardCap * RT_WF_SHARDS;
        if (cap > 0x7ffffff0ull) return fail("tile set too large for one batch");
        RtWavefront &Wf = G.wf;
        Wf.capacity = (uint32_t)cap;
        Wf.shardCap = (uint32_t)shardCap;
        Wf.lookAhead = T.lookAhead ? 1u : 0u;
        const uint64_t qcap = 2 * cap; // queue entries: up to two rays in flight per path
        const uint64_t extraCap = (uint64_t)extraFactor * cap; // room for the further segments of cut rays (a wave that finds it full leaves its rays whole)
        const uint64_t ecap = qcap + extraCap;
        if (ecap > 0xfffffff0ull) return fail("tile set too large for one batch");
        Wf.extraCap = (uint32_t)extraCap;
        Wf.sampleBase = 0; Wf.samplesInBatch = (uint32_t)sb;
        if (sc->alloc<unsigned long long>(cap, &Wf.rng) || sc->alloc<uint4>(cap, &Wf.meta) || sc->alloc<float4>(cap, &Wf.outc) ||
            sc->alloc<float4>(cap * RT_RING * 3, &Wf.ring) || sc->alloc<float4>(cap, &Wf.shP) || sc->alloc<float4>(cap, &Wf.shFace) ||
            sc->alloc<float4>(cap, &Wf.shAtt) || sc->alloc<float4>(multiLight ? cap : 1, &Wf.shN) ||
            sc->alloc<unsigned long long>(multiLight ? cap : 1, &Wf.rngL) || sc->alloc<unsigned long long>(cap, &Wf.laKey) ||
            sc->alloc<uint4>(ecap * 4, &Wf.ent[0]) || sc->alloc<uint4>(ecap * 4, &Wf.ent[1]) || sc->alloc<uint2>(cap, &Wf.pathOf[0]) ||
            sc->alloc<uint2>(cap, &Wf.pathOf[1]) || sc->alloc<unsigned long long>(qcap, &Wf.hitKey[0]) || sc->alloc<unsigned long long>(qcap, &Wf.hitKey[1]) ||
            sc->alloc<uint4>(cap, &Wf.res) || sc->alloc<float4>(gpix * sb, &Wf.sampleOut) ||
            sc->alloc<uint32_t>(ecap, &Wf.sortRank) || sc->alloc<uint16_t>(ecap, &Wf.sortTag) || sc->alloc<uint32_t>(ecap, &Wf.sortedIdx) ||
            sc->alloc<uint32_t>((uint64_t)3 * RT_WF_CTL_WORDS, &Wf.ctl))
            return -1;
        HIP_OK(hipMemsetAsync(Wf.ctl, 0, sizeof(uint32_t) * 3 * RT_WF_CTL_WORDS, sc->stream));
        HIP_OK(hipHostMalloc((void **)&G.hostLog, sizeof(uint4) * RT_WF_ROUND_LOG, hipHostMallocMapped));
        memset(G.hostLog, 0, sizeof(uint4) * RT_WF_ROUND_LOG);
        HIP_OK(hipHostGetDevicePointer((void **)&Wf.roundLog, G.hostLog, 0));
        HIP_OK(hipHostMalloc((void **)&G.hostCount, sizeof(uint32_t) * RT_WF_SHARDS, hipHostMallocDefault));
        HIP_OK(hipHostMalloc((void **)&G.hostStatus, sizeof(uint32_t) * RT_WF_STATUS_WORDS, hipHostMallocMapped));
        memset(G.hostStatus, 0, sizeof(uint32_t) * RT_WF_STATUS_WORDS);
        HIP_OK(hipHostGetDevicePointer((void **)&Wf.hostStatus, G.hostStatus, 0));
        Wf.spinLimit = T.spinLimit ? T.spinLimit : 16384u;
        Wf.fastQuotient = T.fastQuotient ? 1u : 0u;
        // fixed grids: the kernels stride over the work that is really there (queues are sized for the worst case)
        G.queueBlocks = std::min<uint32_t>(cus * 16, (uint32_t)(qcap / 256)); // scatter: two generations of 8 resident workgroups per CU
        G.traceBlocks = (uint32_t)(ecap / 256); // trace of an ordered round, worst case: one workgroup per 256 entries; surplus groups exit at once
        // (a whole number of waves per queue slice: wf_logic_kernel keeps a wave in one slice)
        G.logicBlocks = std::min<uint32_t>(cus * 8, (uint32_t)((cap + 255) / 256));
        G.logicBlocks = std::max<uint32_t>(RT_WF_SHARDS / 4, (G.logicBlocks + RT_WF_SHARDS / 4 - 1) / (RT_WF_SHARDS / 4) * (RT_WF_SHARDS / 4));
    }
    HIP_OK(hipStreamSynchronize(sc->stream));
    sc->blocking = T.blocking != 0;
    return 0;
}

// the groups' views of the scene (same scene, a contiguous range of this instance's tile slots): refreshed whenever a part
// was rebuilt
void refresh_views(rtHipScene *sc)
{
    const RtDevScene &D = sc->dev;
    for (auto &G : sc->groups) {
        G.dev = D;
        G.dev.tileIds = D.tileIds + G.slot0;
        G.dev.tileCount = G.slot1 - G.slot0;
        G.dev.camStart = D.camStart + (size_t)G.slot0 * RT_TILE_PIXELS;
        G.dev.camEnd = D.camEnd + (size_t)G.slot0 * RT_TILE_PIXELS;
        G.dev.tileBuf = D.tileBuf + (size_t)G.slot0 * 3 * RT_TILE_PIXELS;
    }
}

// One of the parts every instance of a scene holds alike (geometry, grid, materials, lights), copied from an instance that has it --
// device to device, over xGMI between GPUs -- instead of uploaded and reshaped once more: the "all GPUs" mode builds the scene once
// ("upload once via root then broadcast").  The source's work on the part must be complete (its builders synchronise).
int clone_part(rtHipScene *dst, const rtHipScene *src, int part)
{
    dst->release_part(part);
    dst->curPart = part;
    std::vector<std::pair<const char *, char *>> moved; // (source block, copy)
    for (size_t i = 0; i < src->partAllocs[part].size(); ++i) {
        char *p = nullptr;
        const uint64_t n = src->partSizes[part][i];
        if (dst->alloc<char>(n, &p)) return -1;
        HIP_OK(hipMemcpyPeerAsync(p, dst->device, src->partAllocs[part][i], src->device, n, dst->stream));
        moved.emplace_back((const char *)src->partAllocs[part][i], p);
    }
    auto at = [&](const void *old) -> const void * { // the copy of the block `old` points to (the builders hand out block starts only)
        for (auto &m : moved) if (m.first == (const char *)old) return m.second;
        return nullptr;
    };
    const RtDevScene &S = src->dev;
    RtDevScene &D = dst->dev;
#define RT_MOVE(field) D.field = (decltype(D.field))at(S.field)
    if (part == PART_GEOMETRY) { D.triangleCount = S.triangleCount; RT_MOVE(triRec); RT_MOVE(triShade); }
    if (part == PART_GRID) {
        D.planesTame = S.planesTame; D.cellCount = S.cellCount;
        RT_MOVE(boxMin); RT_MOVE(cellLut); RT_MOVE(gridStart); RT_MOVE(gridList); RT_MOVE(gridBits); RT_MOVE(gridBlockSparse); RT_MOVE(pairRec);
    }
    if (part == PART_MATERIALS) { D.materialCount = S.materialCount; D.texelCount = S.texelCount; RT_MOVE(matSize); RT_MOVE(matStart); RT_MOVE(textures); RT_MOVE(matRec); }
    if (part == PART_LIGHTS) {
        D.lightCount = S.lightCount;
        RT_MOVE(lightType); RT_MOVE(lightPos); RT_MOVE(lightDir); RT_MOVE(lightCol); RT_MOVE(lightRadius); RT_MOVE(lightHalfAtt); RT_MOVE(lightSpread);
    }
#undef RT_MOVE
    HIP_OK(hipStreamSynchronize(dst->stream));
    return 0;
}

// `like`: an instance of the same scene (same inputs) on this or another device whose shared parts are copied instead of built
int scene_build(rtHipScene *sc, const rtHipSceneDesc *d, const cl_uint *tileIds, cl_uint tileCount, const rtHipScene *like = nullptr)
{
    if (!d) return fail("null scene description");
    if (d->width == 0 || d->height == 0) return fail("empty image %ux%u", d->width, d->height);
    if (d->sampleCount == 0) return fail("sampleCount must be >= 1");
    if (d->axesDiv != RT_GRID_DIV) return fail("axesDivCount %d unsupported (the reference builds %d, trianglelist.h:110)", d->axesDiv, RT_GRID_DIV);
    if ((uint64_t)d->width * d->height > 0xffffffffull) return fail("image too large");
    HIP_OK(hipSetDevice(sc->device));
    HIP_OK(hipStreamCreateWithFlags(&sc->stream, hipStreamNonBlocking));
    sc->tune = tuning();
    sc->pipeline = sc->tune.pipeline == RT_HIP_PIPELINE_MEGAKERNEL ? RT_HIP_PIPELINE_MEGAKERNEL : RT_HIP_PIPELINE_WAVEFRONT;
    if (sc->stager.init(sc->stream) != 0) return -1;
    // Tuning::timing: where the time of a scene upload goes (stderr)
    const bool timing = sc->tune.timing != 0;
    auto tLast = std::chrono::steady_clock::now();
    auto mark = [&](const char *what) {
        if (!timing) return;
        (void)hipStreamSynchronize(sc->stream);
        const auto now = std::chrono::steady_clock::now();
        fprintf(stderr, "libraytrace_hip: scene build: %-28s %8.1f ms\n", what, std::chrono::duration<double, std::milli>(now - tLast).count());
        tLast = now;
    };
    // arrays that are already on the device: the few tables the host looks at itself come back to it, the rest is copied HBM to HBM
    rtHipSceneDesc shadowed;
    std::vector<char> hostCopies[10];
    cl_uint gridEnd[1] = { 0 };
    if (d->arraysOnDevice) {
        shadowed = *d;
        auto fetch = [&](int slot, const void *&field, size_t bytes) -> int {
            if (!field || !bytes) return 0;
            hostCopies[slot].resize(bytes);
            HIP_OK(hipMemcpy(hostCopies[slot].data(), field, bytes, hipMemcpyDeviceToHost));
            field = hostCopies[slot].data();
            return 0;
        };
        const uint64_t cells = (uint64_t)RT_GRID_DIV * RT_GRID_DIV * RT_GRID_DIV;
        if (d->gridStart) { HIP_OK(hipMemcpy(gridEnd, d->gridStart + cells, 4, hipMemcpyDeviceToHost)); sc->gridListSizeHint = gridEnd[0]; sc->haveGridListSize = true; }
        if (fetch(0, (const void *&)shadowed.boxMin, (size_t)(RT_GRID_DIV + 1) * 16) || fetch(1, (const void *&)shadowed.matSize, (size_t)d->materialCount * 40) ||
            fetch(2, (const void *&)shadowed.matStart, (size_t)d->materialCount * 20) || fetch(3, (const void *&)shadowed.textures, (size_t)d->texturesSize * 4) ||
            fetch(4, (const void *&)shadowed.lightType, (size_t)d->lightCount * 4) || fetch(5, (const void *&)shadowed.lightPos, (size_t)d->lightCount * 16) ||
            fetch(6, (const void *&)shadowed.lightDir, (size_t)d->lightCount * 16) || fetch(7, (const void *&)shadowed.lightCol, (size_t)d->lightCount * 16) ||
            fetch(8, (const void *&)shadowed.lightRadius, (size_t)d->lightCount * 4) || fetch(9, (const void *&)shadowed.lightHalfAtt, (size_t)d->lightCount * 4))
            return -1;
        d = &shadowed;
    }
    if (build_fixed(sc, d, tileIds, tileCount) != 0) return -1;
    mark("tiles, outputs, bump tables");
    if (like) {
        for (int part : { PART_GEOMETRY, PART_GRID, PART_MATERIALS, PART_LIGHTS })
            if (clone_part(sc, like, part) != 0) return -1;
        mark("geometry, grid, materials, lights (copied from another instance)");
        if (build_camera(sc, d) != 0) return -1;
        mark("camera lists");
    } else {
        if (build_geometry(sc, d) != 0) return -1;
        mark("geometry");
        if (build_grid(sc, d) != 0) return -1;
        mark("grid + dense view");
        if (build_camera(sc, d) != 0) return -1;
        mark("camera lists");
        if (build_materials(sc, d) != 0) return -1;
        if (build_lights(sc, d) != 0) return -1;
        mark("materials + lights");
    }
    // what only a kernel can tell about the inputs (ids inside the lists): one look at the error word
    HIP_OK(rtp_validate(sc->dev.triangleCount, 0, 0, nullptr, nullptr, sc->camListSize, sc->dev.camList, nullptr, 0, nullptr, sc->prepErr, sc->stream));
    HIP_OK(hipStreamSynchronize(sc->stream));
    HIP_OK(sc->stager.drain()); // (the stream is idle: the staging buffers go back to the pool)
    if (sc->check_prep() != 0) return -1;
    if (build_wavefront(sc, d->sampleCount) != 0) return -1;
    refresh_views(sc);
    mark("path state buffers");
    return 0;
}

// How a round with `rays` rays is laid out (RtRoundMode, rt_device.h): big rounds are ordered by predicted walk length and spread
// their appends over all queue slices, small ones are cut into segments and keep their entries dense.
RtRoundMode mode_for(const Tuning &T, uint32_t round, uint64_t rays, uint32_t slicesBefore)
{
    RtRoundMode m;
    // planned where the rays are made (and ordered) when most paths make one: the round behind the primary hits, and any big round
    m.ordered = (round <= 1u || rays >= T.appendRays) ? 1u : 0u;
    if (T.orderedFirst == 0u && rays < T.appendRays) m.ordered = 0u;
    m.segLen = rays >= T.segRays[0] ? T.segLen[0] : (rays >= T.segRays[1] ? T.segLen[1] : (rays >= T.segRays[2] ? T.segLen[2] : (rays >= T.segRays[3] ? T.segLen[3] : T.segLen[4])));
    if (m.segLen < 1u) m.segLen = 1u;
    uint32_t small = T.smallSlices;
    if (small < 1u || small > RT_WF_SHARDS || (small & (small - 1u))) small = 16u;
    m.slices = rays >= T.sliceRays ? (uint32_t)RT_WF_SHARDS : small;
    if (m.slices > slicesBefore) m.slices = slicesBefore; // a slice holds at most the paths of its shards: slices only ever merge
    // rays per workgroup of the trace kernel: about 256 lanes / the segments a ray of this round is expected to be cut into ...
    // ... and few enough workgroups for all of them to run at once (5 per CU): the round lasts as long as its slowest workgroup
    // (at most 64: with 128 -- two segments per ray on average fill the 256 lanes -- every second workgroup has to cut coarser, and its
    // longer segments are the round's duration: 4K, 232 k rays, 403 us against 273)
    m.groupRays = (uint32_t)std::min<uint64_t>(64, std::max<uint64_t>(16, ((rays + 1099) / 1100 + 15) / 16 * 16));
    if (T.groupRays >= 1u && T.groupRays <= 128u) m.groupRays = T.groupRays;
    return m;
}

// One frame through the staged pipeline: per sample batch and tile group -- primary, then rounds of (scatter,) trace, logic until
// no path is waiting for the grid, then the ordered accumulate.  The round count is data dependent.
//
// WATCHED batch (the first batch of a scene's first frame, or Tuning::blocking): rounds are issued in chunks and the queue length
// is read back after every chunk (one small pinned copy + stream sync per chunk and group).  It leaves a PLAN behind: the number of
// rounds the batch needed and, per round, its rays, its longest queue slice and its further segments (RtWavefront::roundLog).
// PLANNED batches (every later frame, and the further batches of a watched frame): the plan's rounds are issued back to back with
// the layouts and grid sizes the plan implies and NO host synchronisation -- whatever the caller enqueues behind the frame (the tile
// gather) follows immediately.  The last kernel of a batch (wf_status_kernel) adds the number of paths still waiting to a mapped
// host word; frame_finish() looks at it after the caller's own synchronisation.  A non-zero count means the plan was too short for
// this frame (the frames of a scene are deterministic, so that only happens when something about the frame changed, or when a later
// sample batch needs more than the first one did): the frame is rendered again, every batch watched, and the caller is told, so
// that work enqueued behind the incomplete frame can be redone.
// How a round's entries are laid out (ordered or not, cut how finely, how many queue slices) is decided HERE, before the round exists,
// from the plan -- or, in a watched batch, from a guess: the layout changes when cells are visited, never what is found.
// Groups run concurrently on their own streams, forked from and joined to `st`; with stage timing on they run one after the
// other on `st`, so that a kernel's measured duration is its own.
int render_wavefront(rtHipScene *sc, hipStream_t st, bool forceDiscovery)
{
    const Tuning &T = sc->tune;
    auto stage = [&](int which, hipStream_t on, auto &&launch) -> hipError_t {
        if (!sc->stageTiming) return launch();
        if (sc->stageEventsUsed == sc->stageEvents.size()) {
            rtHipScene::StageEvent e{};
            hipError_t er = hipEventCreate(&e.a);
            if (er != hipSuccess) return er;
            er = hipEventCreate(&e.b);
            if (er != hipSuccess) return er;
            sc->stageEvents.push_back(e);
        }
        rtHipScene::StageEvent &e = sc->stageEvents[sc->stageEventsUsed++];
        e.stage = which;
        hipError_t er = hipEventRecord(e.a, on);
        if (er != hipSuccess) return er;
        er = launch();
        if (er != hipSuccess) return er;
        return hipEventRecord(e.b, on);
    };
    const bool serial = sc->stageTiming || sc->groups.size() == 1;
    auto streamOf = [&](size_t g) { return (serial || g == 0) ? st : sc->groups[g].stream; };
    if (!serial) {
        HIP_OK(hipEventRecord(sc->forkEvent, st));
        for (size_t g = 1; g < sc->groups.size(); ++g) HIP_OK(hipStreamWaitEvent(sc->groups[g].stream, sc->forkEvent, 0));
    }
    const bool watchedFrame = forceDiscovery || sc->blocking || sc->planRounds == 0;
    bool planned = !watchedFrame; // per batch: a watched frame's further batches follow its first batch's plan (Tuning::batchPlan)
    uint32_t planRounds = sc->planRounds;
    // the layout of round r: from the plan, or (watched batch) from what the round is assumed to hold
    auto round_mode = [&](rtHipScene::Group &G, uint32_t r) -> RtRoundMode {
        if (G.modes.size() <= r) {
            const uint32_t before = G.modes.empty() ? (uint32_t)RT_WF_SHARDS : G.modes.back().slices;
            uint64_t rays = G.guessRays;
            if (planned) rays = r < RT_WF_ROUND_LOG ? G.plan[r].rays : 0;
            else G.guessRays = std::max<uint64_t>(G.guessRays / 4, 1); // (watched: a round is assumed to hold a quarter of the one before)
            G.modes.resize(r + 1, mode_for(T, r, rays, before));
        }
        return G.modes[r];
    };
    auto trace_blocks = [&](rtHipScene::Group &G, uint32_t r, const RtRoundMode &m) -> uint32_t {
        // worst case: every queue slot in use -- an ordered round takes 256 entries per workgroup, any other mode.groupRays rays of
        // one queue slice (every slice's last piece may be a partial one)
        const uint64_t worst = m.ordered ? G.traceBlocks : 2ull * G.wf.capacity / m.groupRays + 2ull * m.slices;
        if (!planned || r >= RT_WF_ROUND_LOG) return (uint32_t)std::min<uint64_t>(worst, 0x7fffffffu);
        if (T.planGridTiny) return 1;
        // the same frame gave this many rays last time: a tenth more plus a few, never more than the worst case
        // (an ordered round's further segments: as many as last time if the cut is the same -- a watched frame guesses the round's size
        // and with it the cut -- and at most RT_WF_MAXSEG - 1 per ray whatever the cut)
        const uint64_t rays = G.plan[r].rays;
        const uint64_t extra = std::min<uint64_t>(G.wf.extraCap, G.plan[r].extraSegLen == m.segLen ? G.plan[r].extra : (m.segLen >= 4096u ? 0 : rays * 11));
        const uint64_t want = m.ordered ? ((rays + extra) * 11 / 10 + 255) / 256 + 8 : (rays * 11 / 10 + m.groupRays - 1) / m.groupRays + 2ull * m.slices + 8;
        return (uint32_t)std::max<uint64_t>(std::min<uint64_t>(want, worst), 1);
    };
    auto issue_round = [&](rtHipScene::Group &G, hipStream_t on) -> int {
        const uint32_t r = G.rounds;
        const RtRoundMode mode = r > 0 ? round_mode(G, r) : RtRoundMode{ 0u, 4096u, 64u, (uint32_t)RT_WF_SHARDS };
        if (r > 0) { // the entries appended by logic(r-1): order them if the round is an ordered one, then walk the grid
            if (mode.ordered) HIP_OK(stage(4, on, [&] { return rtw_launch_scatter(&G.wf, r, G.queueBlocks, &mode, on); }));
            HIP_OK(stage(2, on, [&] { return rtw_launch_trace(&G.dev, &G.wf, r, trace_blocks(G, r, mode), &mode, on); }));
        }
        const RtRoundMode next = round_mode(G, r + 1);
        HIP_OK(stage(1, on, [&] { return rtw_launch_logic(&G.dev, &G.wf, r, G.logicBlocks, mode.slices, &next, on); }));
        ++G.rounds;
        return 0;
    };
    // Watched: rounds are issued in chunks without looking at the queue.  A one-bounce scene needs exactly three logic rounds
    // (shade the primary hits | consume shadow + bounce answers, shade the bounce hits | consume their shadow answers) with a
    // trace before the last two, so a chunk is 3 rounds.
    auto issue_chunk = [&](rtHipScene::Group &G, hipStream_t on) -> int {
        for (uint32_t k = 0; k < 3 && G.rounds < RT_WF_MAX_ROUNDS; ++k)
            if (issue_round(G, on) != 0) return -1;
        HIP_OK(hipMemcpyAsync(G.hostCount, G.wf.ctl + (G.rounds % 3) * RT_WF_CTL_WORDS + RT_WF_CTL_COUNTS, sizeof(uint32_t) * RT_WF_SHARDS, hipMemcpyDeviceToHost, on)); // main slices
        return 0;
    };
    auto device_error = [&](rtHipScene::Group &G) -> int {
        if (const uint32_t err = G.hostStatus[RT_WF_STATUS_ERROR]) {
            G.hostStatus[RT_WF_STATUS_ERROR] = 0u;
            return fail("wavefront pipeline: device error 0x%x%s -- the frame is invalid", err,
                        (err & RT_WF_ERR_SPIN) ? " (wf_trace_kernel's walk guard tripped: rays were abandoned)" : "");
        }
        return 0;
    };
    uint64_t rounds = 0;
    bool anyPlannedBatch = false;
    const uint32_t sampleCount = sc->dev.sampleCount;
    for (uint32_t base = 0; base < sampleCount; base += sc->samplesPerBatch) {
        for (size_t g = 0; g < sc->groups.size(); ++g) {
            rtHipScene::Group &G = sc->groups[g];
            const hipStream_t on = streamOf(g);
            G.wf.sampleBase = base;
            G.wf.samplesInBatch = std::min<uint32_t>(sc->samplesPerBatch, sampleCount - base);
            G.rounds = 0;
            G.modes.clear();
            G.guessRays = (uint64_t)(G.slot1 - G.slot0) * RT_TILE_PIXELS * G.wf.samplesInBatch; // round 1 of a watched batch: as if every pixel were a path
            if (!G.ctlClean) HIP_OK(hipMemsetAsync(G.wf.ctl, 0, sizeof(uint32_t) * (size_t)3 * RT_WF_CTL_WORDS, on));
            G.ctlClean = false;
            if (!planned) memset(G.hostLog, 0, sizeof(uint4) * RT_WF_ROUND_LOG); // (nothing of this group is in flight: the batch before was waited for)
            HIP_OK(stage(0, on, [&] { return rtw_launch_primary(&G.dev, &G.wf, on); }));
            if (planned) {
                while (G.rounds < planRounds)
                    if (issue_round(G, on) != 0) return -1;
                HIP_OK(rtw_launch_status(&G.wf, G.rounds, on));
                G.ctlClean = true;
            } else if (issue_chunk(G, on) != 0) return -1;
        }
        for (size_t g = 0; g < sc->groups.size(); ++g) {
            rtHipScene::Group &G = sc->groups[g];
            const hipStream_t on = streamOf(g);
            if (!planned) {
                for (;;) {
                    HIP_OK(hipStreamSynchronize(on));
                    if (device_error(G) != 0) return -1;
                    uint64_t waiting = 0;
                    for (int i = 0; i < RT_WF_SHARDS; ++i) waiting += G.hostCount[i];
                    if (waiting == 0) break;
                    if (G.rounds >= RT_WF_MAX_ROUNDS) return fail("wavefront pipeline: more than %d rounds", RT_WF_MAX_ROUNDS);
                    G.guessRays = 2 * waiting; // the next round holds at most two rays per waiting path
                    if (issue_chunk(G, on) != 0) return -1;
                }
                // the plan for the batches and frames to come: the rounds that had anything to trace (+ the logic round that consumed
                // the last answers), and every round's sizes
                const uint4 *log = G.hostLog; // (mapped host memory; the stream was synchronised in the loop above)
                uint32_t needed = 1; // logic(0) always runs
                for (uint32_t r = 1; r < std::min<uint32_t>(G.rounds, RT_WF_ROUND_LOG); ++r)
                    if (log[r].x) needed = r + 1;
                if (G.rounds > RT_WF_ROUND_LOG) needed = G.rounds;
                G.roundsNeeded = std::max(G.roundsNeeded, needed);
                for (uint32_t r = 0; r < RT_WF_ROUND_LOG; ++r) {
                    rtHipScene::Group::RoundPlan &N = G.planNext[r]; // the maximum over the watched batches
                    N.rays = std::max(N.rays, log[r].x);
                    const uint32_t seg = r < G.modes.size() ? G.modes[r].segLen : 4096u;
                    if (N.extraSegLen != seg && N.extraSegLen != 0u) N.extraSegLen = 0xffffffffu; // (batches cut differently: no figure)
                    else { N.extraSegLen = seg; N.extra = std::max(N.extra, log[r].z); }
                }
            } else anyPlannedBatch = true;
            rounds = std::max<uint64_t>(rounds, G.rounds);
            if (sampleCount > 1) // a one-sample frame's pixels were written by the kernels that finished them
                HIP_OK(stage(3, on, [&] { return rtw_launch_accum(&G.dev, &G.wf, base == 0 ? 1 : 0, on); }));
        }
        // a watched batch knows here that its rounds are over: tell whoever polls GetProgress (raytrace.c:566-587)
        if (!planned && sc->progress)
            sc->progress->store(sc->progressBase + sc->progressSpan * (float)std::min<uint64_t>(sampleCount, (uint64_t)base + sc->samplesPerBatch) / (float)sampleCount,
                                std::memory_order_relaxed);
        // A watched frame's further batches: the same pixels with other sample ids -- statistically the same rounds.  They are issued from
        // the plan the batches so far left (a fifth more than the largest of them saw, see trace_blocks) and verified like any planned
        // frame; should one of them need more, frame_finish renders the whole frame again, every batch watched.
        if (!planned && !forceDiscovery && !sc->blocking && T.batchPlan && base + sc->samplesPerBatch < sampleCount) {
            uint32_t need = 1;
            for (auto &G : sc->groups) {
                need = std::max(need, G.roundsNeeded);
                for (uint32_t r = 0; r < RT_WF_ROUND_LOG; ++r) {
                    G.plan[r] = G.planNext[r];
                    G.plan[r].rays += G.plan[r].rays / 10; G.plan[r].extra += G.plan[r].extra / 10;
                }
            }
            planRounds = need + 1; // (one spare round: a later batch's deepest path may go one bounce further)
            planned = true;
        }
    }
    if (watchedFrame) { // adopt what this frame's watched batches needed (the maximum over batches and groups)
        uint32_t need = 1;
        for (auto &G : sc->groups) {
            need = std::max(need, G.roundsNeeded);
            for (uint32_t r = 0; r < RT_WF_ROUND_LOG; ++r) { G.plan[r] = G.planNext[r]; G.planNext[r] = rtHipScene::Group::RoundPlan(); }
            G.roundsNeeded = 0;
        }
        if (anyPlannedBatch && sampleCount > sc->samplesPerBatch) need += 1; // (the spare round of the further batches, kept for the frames to come)
        sc->planRounds = T.planRounds ? std::min(need, T.planRounds) : need;
    }
    if (anyPlannedBatch) sc->unverified = true;
    if (!serial)
        for (size_t g = 1; g < sc->groups.size(); ++g) {
            HIP_OK(hipEventRecord(sc->groups[g].done, sc->groups[g].stream));
            HIP_OK(hipStreamWaitEvent(st, sc->groups[g].done, 0));
        }
    sc->roundsLast = rounds;
    return 0;
}

// After the caller's synchronisation of `st`: were the planned frames since the last call complete?  If not, the LAST frame
// is rendered again with the queue watched (earlier incomplete frames were overwritten by it anyway).  *redone (optional)
// = 1 when that happened: whatever was enqueued behind the incomplete frame saw unfinished tiles.
int frame_finish(rtHipScene *sc, hipStream_t st, int *redone)
{
    if (redone) *redone = 0;
    if (!sc->unverified) return 0;
    HIP_OK(hipStreamSynchronize(st));
    for (auto &G : sc->groups)
        if (G.stream) HIP_OK(hipStreamSynchronize(G.stream));
    sc->unverified = false;
    uint64_t waiting = 0;
    for (auto &G : sc->groups) {
        if (const uint32_t err = G.hostStatus[RT_WF_STATUS_ERROR]) {
            G.hostStatus[RT_WF_STATUS_ERROR] = 0u;
            if (err & ~RT_WF_ERR_GRID)
                return fail("wavefront pipeline: device error 0x%x%s -- the frame is invalid", err,
                            (err & RT_WF_ERR_SPIN) ? " (wf_trace_kernel's walk guard tripped: rays were abandoned)" : "");
            waiting += 1; // a planned trace grid was too small: same remedy as a plan with too few rounds
        }
        waiting += G.hostStatus[RT_WF_STATUS_WAITING];
        G.hostStatus[RT_WF_STATUS_WAITING] = 0u;
    }
    if (waiting == 0) {
        return 0;
    }
    if (redone) *redone = 1;
    if (render_wavefront(sc, st, true) != 0) return -1;
    HIP_OK(hipStreamSynchronize(st));
    return 0;
}

} // namespace

extern "C" {

int rtHipDeviceCount(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

const char *rtHipLastError(void) { return g_error.c_str(); }

rtHipScene *rtHipSceneCreate(int device, const rtHipSceneDesc *desc, const cl_uint *tileIds, cl_uint tileCount)
{
    return rtHipSceneCreateLike(device, desc, tileIds, tileCount, nullptr);
}

rtHipScene *rtHipSceneCreateLike(int device, const rtHipSceneDesc *desc, const cl_uint *tileIds, cl_uint tileCount, const rtHipScene *like)
{
    g_error.clear();
    const int n = rtHipDeviceCount();
    if (n <= 0) { fail("no HIP device available: libraytrace_hip has no CPU fallback"); return nullptr; }
    if (device < 0 || device >= n) { fail("device %d out of range (%d HIP devices)", device, n); return nullptr; }
    rtHipScene *sc = new rtHipScene();
    sc->device = device;
    if (scene_build(sc, desc, tileIds, tileCount, like) != 0) {
        std::string keep = g_error;
        rtHipSceneDestroy(sc);
        g_error = keep;
        return nullptr;
    }
    return sc;
}

void rtHipSceneDestroy(rtHipScene *sc)
{
    if (!sc) return;
    if (sc->device >= 0) (void)hipSetDevice(sc->device);
    if (sc->stream) (void)hipStreamSynchronize(sc->stream);
    for (auto &e : sc->events) { (void)hipEventDestroy(e.first); (void)hipEventDestroy(e.second); }
    for (auto &e : sc->stageEvents) { (void)hipEventDestroy(e.a); (void)hipEventDestroy(e.b); }
    for (auto &G : sc->groups) {
        if (G.hostCount) (void)hipHostFree(G.hostCount);
        if (G.hostStatus) (void)hipHostFree(G.hostStatus);
        if (G.hostLog) (void)hipHostFree(G.hostLog);
        if (G.stream) { (void)hipStreamSynchronize(G.stream); (void)hipStreamDestroy(G.stream); }
        if (G.done) (void)hipEventDestroy(G.done);
    }
    if (sc->forkEvent) (void)hipEventDestroy(sc->forkEvent);
    for (int part = 0; part < PART_COUNT; ++part) sc->release_part(part);
    sc->stager.destroy();
    if (sc->stream) (void)hipStreamDestroy(sc->stream);
    delete sc;
}

uint64_t rtHipSceneBytes(const rtHipScene *sc) { return sc ? sc->bytes : 0; }

int rtHipRenderTiles(rtHipScene *sc, void *stream)
{
    if (!sc) return fail("null scene");
    HIP_OK(hipSetDevice(sc->device));
    hipStream_t st = stream ? (hipStream_t)stream : sc->stream;
    sc->lastStream = st;
    if (sc->eventsUsed == sc->events.size()) {
        hipEvent_t a, b;
        HIP_OK(hipEventCreate(&a)); HIP_OK(hipEventCreate(&b));
        sc->events.emplace_back(a, b);
    }
    auto &ev = sc->events[sc->eventsUsed++];
    HIP_OK(hipEventRecord(ev.first, st));
    if (sc->pipeline == RT_HIP_PIPELINE_WAVEFRONT) {
        if (render_wavefront(sc, st, false) != 0) return -1;
    } else {
        HIP_OK(rtk_launch_trace(&sc->dev, 0, st));
    }
    HIP_OK(hipEventRecord(ev.second, st));
    return 0;
}

int rtHipSetPipeline(rtHipScene *sc, int pipeline)
{
    if (!sc) return fail("null scene");
    if (pipeline != RT_HIP_PIPELINE_MEGAKERNEL && pipeline != RT_HIP_PIPELINE_WAVEFRONT) return fail("unknown pipeline %d", pipeline);
    sc->pipeline = pipeline;
    return 0;
}

// Diagnostic: raw copy of the 8 device-side debug counters (work counters of the counted kernel, or the cycle sums of
// an RT_DIAG_STAMPS build).  clear != 0 zeroes them afterwards.
int rtHipDebugCounters(rtHipScene *sc, unsigned long long out[8], int clear)
{
    if (!sc || !out) return fail("null argument");
    HIP_OK(hipSetDevice(sc->device));
    HIP_OK(hipDeviceSynchronize());
    HIP_OK(hipMemcpy(out, sc->dev.stats, 64, hipMemcpyDeviceToHost));
    if (clear) HIP_OK(hipMemset(sc->dev.stats, 0, 64));
    return 0;
}

int rtHipStageTiming(rtHipScene *sc, int enable)
{
    if (!sc) return fail("null scene");
    sc->stageTiming = enable != 0;
    sc->stageEventsUsed = 0;
    return 0;
}

int rtHipStageTimes(rtHipScene *sc, double ms[5], uint64_t *rounds)
{
    if (!sc || !ms) return fail("null argument");
    HIP_OK(hipSetDevice(sc->device));
    for (int i = 0; i < 5; ++i) ms[i] = 0.0;
    for (size_t i = 0; i < sc->stageEventsUsed; ++i) {
        HIP_OK(hipEventSynchronize(sc->stageEvents[i].b));
        float t = 0.f;
        HIP_OK(hipEventElapsedTime(&t, sc->stageEvents[i].a, sc->stageEvents[i].b));
        ms[sc->stageEvents[i].stage] += t;
    }
    sc->stageEventsUsed = 0;
    if (rounds) *rounds = sc->roundsLast;
    return 0;
}

int rtHipRenderTilesCounted(rtHipScene *sc, rtHipStats *stats)
{
    if (!sc || !stats) return fail("null argument");
    HIP_OK(hipSetDevice(sc->device));
    HIP_OK(hipMemsetAsync(sc->dev.stats, 0, 64, sc->stream));
    HIP_OK(rtk_launch_trace(&sc->dev, 1, sc->stream));
    unsigned long long host[8] = { 0 };
    HIP_OK(hipMemcpyAsync(host, sc->dev.stats, 64, hipMemcpyDeviceToHost, sc->stream));
    HIP_OK(hipStreamSynchronize(sc->stream));
    stats->primarySamples = host[0]; stats->primaryCandidates = host[1]; stats->gridRays = host[2]; stats->gridCells = host[3];
    stats->gridCandidates = host[4]; stats->shadedHits = host[5]; stats->texelFetches = host[6];
    return 0;
}

void *rtHipTileBuffer(rtHipScene *sc) { return sc ? (void *)sc->dev.tileBuf : nullptr; }
uint64_t rtHipTileBufferBytes(const rtHipScene *sc) { return sc ? (uint64_t)sc->tileIds.size() * 3 * RT_TILE_PIXELS * 2 : 0; }

int rtHipDetile(int device, const void *tileBuffer, const cl_uint *tileIdsDevice, cl_uint tileCount, cl_uint width, cl_uint height,
                void *planeR, void *planeG, void *planeB, void *stream)
{
    if (!tileBuffer || !tileIdsDevice || !planeR || !planeG || !planeB) return fail("null argument");
    if (width == 0 || height == 0) return fail("empty image");
    HIP_OK(hipSetDevice(device));
    const uint32_t tilesX = (width + RT_TILE - 1) / RT_TILE;
    HIP_OK(rtk_launch_detile(tileBuffer, tileIdsDevice, tileCount, width, height, tilesX, planeR, planeG, planeB, 1, (hipStream_t)stream));
    return 0;
}

int rtHipDetileStore(int device, const void *tileBuffer, const cl_uint *tileIdsDevice, cl_uint tileCount, cl_uint width, cl_uint height,
                     void *planeR, void *planeG, void *planeB, void *stream)
{
    if (!tileBuffer || !tileIdsDevice || !planeR || !planeG || !planeB) return fail("null argument");
    if (width == 0 || height == 0) return fail("empty image");
    HIP_OK(hipSetDevice(device));
    const uint32_t tilesX = (width + RT_TILE - 1) / RT_TILE;
    HIP_OK(rtk_launch_detile(tileBuffer, tileIdsDevice, tileCount, width, height, tilesX, planeR, planeG, planeB, 0, (hipStream_t)stream));
    return 0;
}

// Plain device memory for hosts that do not include HIP headers (a gather root's planes, test buffers).
void *rtHipDeviceAlloc(int device, uint64_t bytes)
{
    void *p = nullptr;
    if (hipSetDevice(device) != hipSuccess || hipMalloc(&p, bytes ? bytes : 1) != hipSuccess) { fail("rtHipDeviceAlloc(%d, %llu) failed", device, (unsigned long long)bytes); return nullptr; }
    return p;
}
void rtHipDeviceFree(int device, void *p)
{
    if (p && hipSetDevice(device) == hipSuccess) (void)hipFree(p);
}
int rtHipDeviceCopy(int device, void *dst, const void *src, uint64_t bytes, int toDevice)
{
    HIP_OK(hipSetDevice(device));
    HIP_OK(hipDeviceSynchronize());
    HIP_OK(hipMemcpy(dst, src, bytes, toDevice ? hipMemcpyHostToDevice : hipMemcpyDeviceToHost));
    return 0;
}

int rtHipSync(rtHipScene *sc, void *stream)
{
    if (!sc) return fail("null scene");
    HIP_OK(hipSetDevice(sc->device));
    HIP_OK(hipStreamSynchronize(stream ? (hipStream_t)stream : sc->stream));
    return frame_finish(sc, sc->lastStream ? sc->lastStream : sc->stream, nullptr);
}

int rtHipFrameFinish(rtHipScene *sc, int *redone)
{
    if (!sc) return fail("null scene");
    HIP_OK(hipSetDevice(sc->device));
    return frame_finish(sc, sc->lastStream ? sc->lastStream : sc->stream, redone);
}

int rtHipReadback(rtHipScene *sc, cl_ushort *outR, cl_ushort *outG, cl_ushort *outB)
{
    if (!sc || !outR || !outG || !outB) return fail("null argument");
    HIP_OK(hipSetDevice(sc->device));
    HIP_OK(hipDeviceSynchronize());
    if (frame_finish(sc, sc->lastStream ? sc->lastStream : sc->stream, nullptr) != 0) return -1;
    const size_t nt = sc->tileIds.size();
    std::vector<uint16_t> host(nt * 3 * RT_TILE_PIXELS);
    HIP_OK(hipMemcpy(host.data(), sc->dev.tileBuf, host.size() * 2, hipMemcpyDeviceToHost));
    cl_ushort *planes[3] = { outR, outG, outB };
    for (size_t s = 0; s < nt; ++s) {
        const uint32_t tx = sc->tileIds[s] % sc->tilesX, ty = sc->tileIds[s] / sc->tilesX;
        for (int c = 0; c < 3; ++c)
            for (uint32_t ly = 0; ly < RT_TILE; ++ly) {
                const uint32_t gy = ty * RT_TILE + ly;
                if (gy >= sc->height) break;
                const uint16_t *src = host.data() + (s * 3 + c) * RT_TILE_PIXELS + ly * RT_TILE;
                cl_ushort *dst = planes[c] + (size_t)gy * sc->width + tx * RT_TILE;
                const uint32_t n = std::min<uint32_t>(RT_TILE, sc->width - tx * RT_TILE);
                for (uint32_t i = 0; i < n; ++i) {
                    const uint32_t v = (uint32_t)dst[i] + src[i]; // saturating accumulate (raytrace_opencl.c:729-740)
                    dst[i] = (cl_ushort)(v > 0xFFFFu ? 0xFFFFu : v);
                }
            }
    }
    return 0;
}

int rtHipKernelTime(rtHipScene *sc, double *avgMs, uint64_t *launches)
{
    if (!sc || !avgMs || !launches) return fail("null argument");
    HIP_OK(hipSetDevice(sc->device));
    double total = 0.0;
    for (size_t i = 0; i < sc->eventsUsed; ++i) {
        HIP_OK(hipEventSynchronize(sc->events[i].second));
        float ms = 0.f;
        HIP_OK(hipEventElapsedTime(&ms, sc->events[i].first, sc->events[i].second));
        total += ms;
    }
    *launches = sc->eventsUsed;
    *avgMs = sc->eventsUsed ? total / (double)sc->eventsUsed : 0.0;
    sc->eventsUsed = 0;
    return 0;
}

// ---- drop-in layer ---------------------------------------------------------------------------------------------

// device table, published last like raytrace.c:117-120
static std::atomic<cl_bool> g_tableReady{ CL_FALSE };
static std::atomic<int> g_deviceCount{ 0 };
static char g_deviceName[256][256];
static std::mutex g_tableMutex;

void InitOpenCL(void)
{
    std::lock_guard<std::mutex> lock(g_tableMutex);
    char names[256][256];
    int n = rtHipDeviceCount();
    if (n > 254) n = 254;
    for (int i = 0; i < n; ++i) {
        hipDeviceProp_t prop;
        memset(&prop, 0, sizeof prop);
        if (hipGetDeviceProperties(&prop, i) != hipSuccess) strcpy(prop.name, "unknown");
        snprintf(names[i], 256, "AMD HIP %.200s #%d", prop.name, i);
    }
    int entries = n;
    if (n > 1) snprintf(names[entries++], 256, "AMD HIP all %d GPUs (tiled)", n);
    memcpy(g_deviceName, names, sizeof(names[0]) * (size_t)entries);
    g_deviceCount.store(entries, std::memory_order_relaxed);
    g_tableReady.store(CL_TRUE, std::memory_order_release);
}

void ResetComputationType(void)
{
    if (g_tableReady.load(std::memory_order_acquire)) {
        g_tableReady.store(CL_FALSE, std::memory_order_relaxed);
        g_deviceCount.store(0, std::memory_order_relaxed);
    }
}

cl_bool GetIsComputationTypeUpdated(void) { return g_tableReady.load(std::memory_order_acquire); }

size_t GetComputationTypeCount(void) { return 1 + (size_t)g_deviceCount.load(std::memory_order_relaxed); }

cl_bool GetComputationTypeName(size_t id, size_t strLen, cl_char *str)
{
    if (!str) return CL_FALSE;
    if (0 == id--) {
        static const char cpu[] = "Local CPU single thread"; // raytrace.c:138
        if (strlen(cpu) <= strLen) { strcpy((char *)str, cpu); return CL_TRUE; }
    } else if (id < (size_t)g_deviceCount.load(std::memory_order_relaxed)) {
        if (strlen(g_deviceName[id]) <= strLen) {
            memcpy(str, g_deviceName[id], std::min<size_t>(256, strLen)); // raytrace.c:147
            return CL_TRUE;
        }
    }
    return CL_FALSE;
}

static std::atomic<float> g_progress{ 0.f };
static std::atomic<long> g_startTime{ 0 }, g_endTime{ 0 };

cl_float GetProgress(void) { return g_progress.load(std::memory_order_relaxed); }
void SetProgress(cl_float p) { g_progress.store(p, std::memory_order_relaxed); }
clock_t GetStartTime(void) { return (clock_t)g_startTime.load(std::memory_order_relaxed); }
clock_t GetEndTime(void) { return (clock_t)g_endTime.load(std::memory_order_relaxed); }
void ResetTime(void) { g_startTime.store(0, std::memory_order_relaxed); g_endTime.store(0, std::memory_order_relaxed); }

} // extern "C"

// ---- the drop-in layer's scene cache ("next" row 2) ---------------------------------------------------------
// The reference rebuilds everything on every call: program, 35 buffers, tiles x samples launches (raytrace.c:330-489).  A second
// Render click usually changes the camera, sometimes the sample count, rarely the scene.  The scenes of the last call stay in
// HBM; the next call hashes its input arrays (all host threads, a few ms for a 1 M-triangle scene) and rebuilds only the parts
// whose hash changed: geometry + grid + materials are the expensive ones (upload, triangle records, dense grid view), the
// camera lists, the lights and the path-state buffers are cheap.  RT_HIP_CACHE=0 switches the cache off (every call builds and
// frees, like the reference); rtHipCacheClear() frees what is held.
namespace {

inline uint64_t mix64(uint64_t v)
{
    v ^= v >> 32; v *= 0xd6e8feb86659fd93ull; v ^= v >> 32; v *= 0xd6e8feb86659fd93ull; v ^= v >> 32;
    return v;
}

// content hash of one chunk: four independent multiply-rotate lanes over 32-byte blocks, then the tail
uint64_t hash_chunk(const unsigned char *p, size_t n)
{
    uint64_t a = 0x9e3779b97f4a7c15ull, b = 0xc2b2ae3d27d4eb4full, c = 0x165667b19e3779f9ull, d = 0x27d4eb2f165667c5ull;
    size_t i = 0;
    for (; i + 32 <= n; i += 32) {
        uint64_t w[4];
        memcpy(w, p + i, 32);
        a = (a ^ w[0]) * 0x9fb21c651e98df25ull; a = (a << 29) | (a >> 35);
        b = (b ^ w[1]) * 0x9fb21c651e98df25ull; b = (b << 29) | (b >> 35);
        c = (c ^ w[2]) * 0x9fb21c651e98df25ull; c = (c << 29) | (c >> 35);
        d = (d ^ w[3]) * 0x9fb21c651e98df25ull; d = (d << 29) | (d >> 35);
    }
    // the tail (< 32 bytes): whole 8-byte words folded one by one, then the last partial word -- each assembled in a zeroed word
    // of its own, so that no byte is ever ORed over another one's bits
    for (; i + 8 <= n; i += 8) {
        uint64_t w;
        memcpy(&w, p + i, 8);
        a = mix64(a ^ w);
    }
    uint64_t tail = 0;
    if (i < n) memcpy(&tail, p + i, n - i);
    return mix64(a ^ mix64(b ^ mix64(c ^ mix64(d ^ tail ^ (uint64_t)n))));
}

struct HashJob { const void *ptr; size_t bytes; int group; };

// hashes of the five input groups (geometry, grid, materials, lights, camera), computed chunk-parallel
void hash_inputs(const std::vector<HashJob> &jobs, uint64_t out[5])
{
    struct Chunk { const unsigned char *p; size_t n; uint64_t h; int group; };
    std::vector<Chunk> chunks;
    const size_t piece = (size_t)2 << 20;
    for (const HashJob &j : jobs) {
        const unsigned char *p = (const unsigned char *)j.ptr;
        size_t left = p ? j.bytes : 0;
        chunks.push_back(Chunk{ nullptr, j.bytes, 0, j.group }); // the length always counts, also for an empty array
        while (left) { const size_t n = std::min(piece, left); chunks.push_back(Chunk{ p, n, 0, j.group }); p += n; left -= n; }
    }
    unsigned threads = std::thread::hardware_concurrency();
    threads = std::max(1u, std::min(threads ? threads : 1u, 16u));
    std::atomic<size_t> next{ 0 };
    auto work = [&] {
        for (size_t i; (i = next.fetch_add(1)) < chunks.size();)
            chunks[i].h = chunks[i].p ? hash_chunk(chunks[i].p, chunks[i].n) : mix64(chunks[i].n + 0x51ull);
    };
    std::vector<std::thread> pool;
    for (unsigned t = 1; t < threads; ++t) pool.emplace_back(work);
    work();
    for (auto &t : pool) t.join();
    for (int g = 0; g < 5; ++g) out[g] = 0x12345678u + g;
    for (const Chunk &c : chunks) out[c.group] = mix64(out[c.group] * 0x100000001b3ull ^ c.h);
}

struct SceneCache {
    std::vector<rtHipScene *> scenes;
    std::vector<std::vector<cl_uint>> tiles; // per scene: its tile ids (empty = all)
    int first = 0, count = 0, devices = 0;
    Tuning tune;               // the tuning values the scenes were built with: other values, other scenes
    uint32_t width = 0, height = 0, sampleCount = 0;
    uint64_t hash[5] = { 0 };
    bool valid = false;
    // gather root (device `first`): row-major planes and, for several scenes, the peers' tile buffers
    uint16_t *planes = nullptr;
    uint16_t *gather = nullptr;
    uint32_t *gatherIds = nullptr;
    size_t gatherTiles = 0;
    void clear()
    {
        for (rtHipScene *s : scenes) rtHipSceneDestroy(s);
        scenes.clear(); tiles.clear();
        if (valid || planes || gather || gatherIds) {
            if (devices > 0) (void)hipSetDevice(first % devices);
            if (planes) (void)hipFree(planes);
            if (gather) (void)hipFree(gather);
            if (gatherIds) (void)hipFree(gatherIds);
        }
        planes = nullptr; gather = nullptr; gatherIds = nullptr; gatherTiles = 0;
        valid = false;
    }
};
SceneCache g_cache;
std::mutex g_cacheMutex;

} // namespace

extern "C" {

int rtHipTune(const char *key, double value)
{
    if (!key) return fail("rtHipTune: null key");
    std::lock_guard<std::mutex> lock(g_tuneMutex);
    Tuning &T = g_tune;
    const std::string k = key;
    const uint32_t u = value < 0 ? 0u : (value > 4294967295.0 ? 0xffffffffu : (uint32_t)value);
    if (k == "reset") { T = Tuning(); return 0; }
    struct { const char *name; uint32_t *field; } table[] = {
        { "stage_mb", &T.stageMb }, { "extra_factor", &T.extraFactor }, { "groups", &T.groups }, { "lookahead", &T.lookAhead },
        { "seg0", &T.segLen[0] }, { "seg1", &T.segLen[1] }, { "seg2", &T.segLen[2] }, { "seg3", &T.segLen[3] }, { "seg4", &T.segLen[4] },
        { "seg_rays0", &T.segRays[0] }, { "seg_rays1", &T.segRays[1] }, { "seg_rays2", &T.segRays[2] }, { "seg_rays3", &T.segRays[3] },
        { "fast_quotient", &T.fastQuotient }, { "spin_limit", &T.spinLimit }, { "append_rays", &T.appendRays }, { "ordered_first", &T.orderedFirst }, { "slice_rays", &T.sliceRays },
        { "small_slices", &T.smallSlices }, { "group_rays", &T.groupRays }, { "blocking", &T.blocking }, { "plan_rounds", &T.planRounds }, { "plan_grid_tiny", &T.planGridTiny },
        { "pipeline", &T.pipeline }, { "timing", &T.timing }, { "virtual_devices", &T.virtualDevices }, { "cache", &T.cache }, { "batch_plan", &T.batchPlan },
    };
    if (k == "state_mb") { T.stateMb = (uint64_t)(value < 0 ? 0 : value); return 0; }
    for (auto &e : table)
        if (k == e.name) { *e.field = u; return 0; }
    return fail("rtHipTune: unknown key '%s'", key);
}

int rtHipTestCachePointers(const void *out[6])
{
    if (!out) return -1;
    std::lock_guard<std::mutex> lock(g_cacheMutex);
    if (!g_cache.valid || g_cache.scenes.empty() || !g_cache.scenes[0]) return -2;
    const RtDevScene &D = g_cache.scenes[0]->dev;
    out[0] = D.triRec; out[1] = D.triShade; out[2] = D.pairRec; out[3] = D.matRec; out[4] = D.textures; out[5] = D.camList;
    return 0;
}

uint64_t rtHipTestHashBytes(const void *bytes, uint64_t count) { return hash_chunk((const unsigned char *)bytes, (size_t)count); }

void rtHipCacheClear(void)
{
    std::lock_guard<std::mutex> lock(g_cacheMutex);
    const std::string keep = g_error;
    g_cache.clear();
    g_error = keep;
}

cl_bool RaytraceAll(cl_uint computationType, cl_uint2 cameraImageDimension, cl_float3 cameraEye, cl_float3 cameraEyeToTopLeftVector,
                    cl_float3 cameraLeftToRightPixelSizeVector, cl_float3 cameraTopToBottomPixelSizeVector, cl_float cameraPixelSizeInv,
                    cl_uint *cameraPixelTriangleListStart, cl_uint *cameraPixelTriangleListEnd, cl_uint *cameraPixelTriangleList,
                    ptrdiff_t cameraPixelTriangleListSize, cl_uint sampleCount, cl_uint vertexCount, cl_float3 *vertex,
                    cl_uint triangleCount, cl_int3 *triangleVertexIndex, cl_int *triangleMaterialId, cl_float2 *triangleUv,
                    cl_float3 *triangleNormal, cl_int axesDivCount, cl_float3 *sceneBoxMin, cl_uint *scenePixelTriangleListStart,
                    cl_uint *scenePixelTriangleList, cl_uint materialCount, cl_uint2 *materialImageSize, cl_int *materialImageStart,
                    cl_uint texturesSize, cl_uchar3 *textures, cl_uint lightCount, cl_int *lightType, cl_float3 *lightPosition,
                    cl_float3 *lightDirection, cl_float3 *lightColour, cl_float *lightRadius, cl_float *lightHalfAttenuationDistance,
                    cl_ushort *outputRed, cl_ushort *outputGreen, cl_ushort *outputBlue)
{
    g_error.clear();
    auto refuse = [&](const char *why) -> cl_bool {
        fprintf(stderr, "libraytrace_hip: %s\n", why);
        return CL_FALSE;
    };
    if (computationType == 0) {
        // The reference's id 0 is its own in-thread C loop (raytrace.c:604-655).  This library is the device path
        // only; falling back to a CPU here would hide a missing GPU.  Fail loudly.
        fail("RaytraceAll: computationType 0 (\"Local CPU single thread\") is the reference's own C path and is not "
             "provided by libraytrace_hip; pick a HIP device (computationType >= 1)");
        return refuse(g_error.c_str());
    }
    const int n = rtHipDeviceCount();
    // "All GPUs": one scene per device with the tiles dealt round-robin, one host thread per device while the scenes are built
    // and the first frame watches its ray queue.  RT_HIP_VIRTUAL_DEVICES=k (test hook): the all-GPUs id deals the tiles over k
    // instances that share the real devices, so the path runs on a one-GPU box.
    const Tuning tune = tuning();
    const int virt = (int)tune.virtualDevices;
    const bool all = (n > 1 && computationType == (cl_uint)n + 1);
    const bool allVirtual = n > 0 && virt > 1 && computationType == (cl_uint)n + 1;
    if (n <= 0 || (!all && !allVirtual && computationType > (cl_uint)n)) {
        fail("RaytraceAll: computationType %u but %d HIP device(s) present", computationType, n);
        return refuse(g_error.c_str());
    }
    if (!outputRed || !outputGreen || !outputBlue) { fail("RaytraceAll: null output plane"); return CL_FALSE; }

    rtHipSceneDesc d;
    memset(&d, 0, sizeof d);
    d.width = cameraImageDimension.s[0]; d.height = cameraImageDimension.s[1];
    for (int i = 0; i < 3; ++i) {
        d.eye[i] = cameraEye.s[i]; d.eyeToTopLeft[i] = cameraEyeToTopLeftVector.s[i];
        d.leftToRight[i] = cameraLeftToRightPixelSizeVector.s[i]; d.topToBottom[i] = cameraTopToBottomPixelSizeVector.s[i];
    }
    d.pixelSizeInv = cameraPixelSizeInv;
    d.camStart = cameraPixelTriangleListStart; d.camEnd = cameraPixelTriangleListEnd; d.camList = cameraPixelTriangleList;
    d.camListSize = cameraPixelTriangleListSize < 0 ? 0 : (uint64_t)cameraPixelTriangleListSize;
    d.sampleCount = sampleCount;
    d.vertexCount = vertexCount; d.vertex = vertex;
    d.triangleCount = triangleCount; d.triIndex = triangleVertexIndex; d.triMaterial = triangleMaterialId;
    d.triUv = triangleUv; d.triNormal = triangleNormal;
    d.axesDiv = axesDivCount; d.boxMin = sceneBoxMin; d.gridStart = scenePixelTriangleListStart; d.gridList = scenePixelTriangleList;
    d.materialCount = materialCount; d.matSize = materialImageSize; d.matStart = materialImageStart;
    d.texturesSize = texturesSize; d.textures = textures;
    d.lightCount = lightCount; d.lightType = lightType; d.lightPos = lightPosition; d.lightDir = lightDirection;
    d.lightCol = lightColour; d.lightRadius = lightRadius; d.lightHalfAtt = lightHalfAttenuationDistance;
    if (d.width == 0 || d.height == 0 || (uint64_t)d.width * d.height > 0xffffffffull) { fail("RaytraceAll: bad image size %ux%u", d.width, d.height); return refuse(g_error.c_str()); }
    if (d.axesDiv != RT_GRID_DIV || !d.gridStart) { fail("RaytraceAll: axesDivCount %d / null grid (the reference builds %d, trianglelist.h:110)", d.axesDiv, RT_GRID_DIV); return refuse(g_error.c_str()); }

    const size_t P = (size_t)d.width * d.height;
    const bool every = all || allVirtual;
    const int first = every ? 0 : (int)computationType - 1, count = allVirtual ? virt : (all ? n : 1);
    const uint32_t tilesTotal = ((d.width + RT_TILE - 1) / RT_TILE) * ((d.height + RT_TILE - 1) / RT_TILE);
    g_progress.store(0.f, std::memory_order_relaxed);
    const long t0 = (long)clock();
    g_startTime.store(t0 ? t0 : 1, std::memory_order_relaxed); // must read non-zero once the kernel phase begins
    g_endTime.store(t0 ? t0 : 1, std::memory_order_relaxed);

    // ---- what changed since the last call? ---------------------------------------------------------------------------------
    const uint64_t cells = (uint64_t)RT_GRID_DIV * RT_GRID_DIV * RT_GRID_DIV;
    const uint64_t gridListSize = d.gridStart[cells];
    uint64_t h[5];
    const auto tCall = std::chrono::steady_clock::now();
    {
        const uint32_t scalars[4] = { d.vertexCount, d.triangleCount, d.materialCount, d.texturesSize }; // (light count and image size have their own groups / keys)
        std::vector<HashJob> jobs = {
            { d.vertex, (size_t)d.vertexCount * 16, 0 }, { d.triIndex, (size_t)d.triangleCount * 16, 0 }, { d.triMaterial, (size_t)d.triangleCount * 4, 0 },
            { d.triUv, (size_t)d.triangleCount * 24, 0 }, { d.triNormal, (size_t)d.triangleCount * 48, 0 }, { scalars, sizeof scalars, 0 },
            { d.boxMin, (size_t)(RT_GRID_DIV + 1) * 16, 1 }, { d.gridStart, (size_t)(cells + 1) * 4, 1 }, { d.gridList, (size_t)gridListSize * 4, 1 },
            { d.matSize, (size_t)d.materialCount * 40, 2 }, { d.matStart, (size_t)d.materialCount * 20, 2 }, { d.textures, (size_t)d.texturesSize * 4, 2 },
            { d.lightType, (size_t)d.lightCount * 4, 3 }, { d.lightPos, (size_t)d.lightCount * 16, 3 }, { d.lightDir, (size_t)d.lightCount * 16, 3 },
            { d.lightCol, (size_t)d.lightCount * 16, 3 }, { d.lightRadius, (size_t)d.lightCount * 4, 3 }, { d.lightHalfAtt, (size_t)d.lightCount * 4, 3 },
            { d.camStart, P * 4, 4 }, { d.camEnd, P * 4, 4 }, { d.camList, (size_t)d.camListSize * 4, 4 }, { d.eye, 4 * 17, 4 },
        };
        hash_inputs(jobs, h);
    }
    const bool timing = tune.timing != 0;
    const auto tHash = std::chrono::steady_clock::now();

    std::lock_guard<std::mutex> lock(g_cacheMutex);
    SceneCache &C = g_cache;
    const bool useCache = tune.cache != 0;
    const bool sameSet = useCache && C.valid && C.first == first && C.count == count && C.devices == n && C.width == d.width && C.height == d.height &&
                         (int)C.scenes.size() == count && memcmp(&C.tune, &tune, sizeof tune) == 0;
    // What is rebuilt.  Geometry (triangle records), grid (its dense view holds copies of triangle records: it follows the geometry),
    // materials, lights, camera lists and the path-state buffers are separate parts of a resident scene, each behind its own gate: a
    // changed texel re-bakes the materials and leaves the 260 MB of geometry and grid of a 1 M-triangle scene where they are.
    const bool geometryChanged = !sameSet || C.hash[0] != h[0];
    const bool gridChanged = geometryChanged || C.hash[1] != h[1];
    const bool materialsChanged = !sameSet || C.hash[2] != h[2];
    const bool lightsChanged = !sameSet || C.hash[3] != h[3];
    const bool cameraChanged = !sameSet || C.hash[4] != h[4];
    const bool samplesChanged = !sameSet || C.sampleCount != d.sampleCount;
    const bool reuse = sameSet;
    if (!sameSet) {
        C.clear();
        C.first = first; C.count = count; C.devices = n; C.width = d.width; C.height = d.height; C.tune = tune;
        C.scenes.assign((size_t)count, nullptr);
        C.tiles.assign((size_t)count, std::vector<cl_uint>());
        if (count > 1)
            for (int g = 0; g < count; ++g)
                for (uint32_t t = (uint32_t)g; t < tilesTotal; t += (uint32_t)count) C.tiles[g].push_back(t); // round-robin tile deal
    }
    std::vector<std::string> errors((size_t)count);
    std::vector<char> failed((size_t)count, 0);
    // build or update one device's scene; errors are thread-local, so they are carried out by hand.  Instance 0 goes first: the others
    // copy the parts all instances hold alike from it, device to device (clone_part), instead of uploading and reshaping them again
    auto build = [&](int g) -> bool {
        if (count > 1 && C.tiles[g].empty()) return true; // more instances than tiles
        bool ok = true;
        const rtHipScene *root = g > 0 ? C.scenes[0] : nullptr;
        if (!C.scenes[g]) {
            C.scenes[g] = rtHipSceneCreateLike((first + g) % n, &d, C.tiles[g].empty() ? nullptr : C.tiles[g].data(), (cl_uint)C.tiles[g].size(), root);
            ok = C.scenes[g] != nullptr;
        } else {
            rtHipScene *sc = C.scenes[g];
            ok = hipSetDevice(sc->device) == hipSuccess;
            if (ok && geometryChanged) ok = (root ? clone_part(sc, root, PART_GEOMETRY) : build_geometry(sc, &d)) == 0;
            if (ok && gridChanged) ok = (root ? clone_part(sc, root, PART_GRID) : build_grid(sc, &d)) == 0;
            if (ok && materialsChanged) ok = (root ? clone_part(sc, root, PART_MATERIALS) : build_materials(sc, &d)) == 0;
            if (ok && lightsChanged) ok = (root ? clone_part(sc, root, PART_LIGHTS) : build_lights(sc, &d)) == 0;
            if (ok && cameraChanged) ok = build_camera(sc, &d) == 0;
            if (ok && (cameraChanged || geometryChanged)) // ids inside the list against the triangles there are now: one look at the validation word
                ok = rtp_validate(sc->dev.triangleCount, 0, 0, nullptr, nullptr, sc->camListSize, sc->dev.camList, nullptr, 0, nullptr, sc->prepErr, sc->stream) == hipSuccess &&
                     hipStreamSynchronize(sc->stream) == hipSuccess && sc->check_prep() == 0;
            if (ok && (samplesChanged || (sc->dev.lightCount > 1) != sc->wfMultiLight)) ok = build_wavefront(sc, d.sampleCount) == 0;
            if (ok && (lightsChanged || cameraChanged || gridChanged || materialsChanged)) sc->planRounds = 0; // other rays: the next frame watches its queue again
            if (ok) refresh_views(sc);
        }
        if (!ok) { failed[g] = 1; errors[g] = g_error; }
        return ok;
    };
    auto work = [&](int g, bool built) { // (instance 0 is built before the threads start), then render the instance's share
        if (count > 1 && C.tiles[g].empty()) return;
        bool ok = built || build(g);
        if (failed[g]) return;
        if (ok) {
            rtHipScene *sc = C.scenes[g];
            sc->progress = &g_progress;
            sc->progressBase = 0.999f * (float)g / (float)count;
            sc->progressSpan = 0.999f / (float)count;
            sc->eventsUsed = 0; // one event pair per call: nobody asks the drop-in layer for kernel times, and a cached scene lives on
            ok = rtHipRenderTiles(sc, nullptr) == 0;
            if (ok && sc->unverified) { // a planned frame runs without the host: follow the batch counter its kernels bump
                const uint32_t batches = (d.sampleCount + sc->samplesPerBatch - 1) / sc->samplesPerBatch * (uint32_t)sc->groups.size();
                uint32_t before = 0;
                for (auto &G : sc->groups) before += G.hostStatus[RT_WF_STATUS_BATCHES];
                while (hipStreamQuery(sc->stream) == hipErrorNotReady) {
                    uint32_t now = 0;
                    for (auto &G : sc->groups) now += G.hostStatus[RT_WF_STATUS_BATCHES];
                    g_progress.store(sc->progressBase + sc->progressSpan * (float)std::min(now - before, batches) / (float)batches, std::memory_order_relaxed);
                    std::this_thread::sleep_for(std::chrono::microseconds(200));
                }
            }
            ok = ok && rtHipSync(sc, nullptr) == 0;
            sc->progress = nullptr;
        }
        if (!ok) { failed[g] = 1; errors[g] = g_error; }
    };
    const bool rootBuilt = build(0);
    if (count == 1) { if (rootBuilt) work(0, true); }
    else if (rootBuilt) {
        std::vector<std::thread> pool;
        for (int g = 0; g < count; ++g) pool.emplace_back(work, g, g == 0);
        for (auto &t : pool) t.join();
    }
    bool ok = true;
    for (int g = 0; g < count; ++g)
        if (failed[g]) { ok = false; g_error = errors[g]; }
    const auto tRender = std::chrono::steady_clock::now();

    // ---- gather on the root device: peers' tile buffers over the fabric, one de-tiling launch, one copy per plane to the caller -----
    if (ok) {
        auto gather = [&]() -> int {
            rtHipScene *root = nullptr;
            for (rtHipScene *s : C.scenes) if (s) { root = s; break; }
            if (!root) return fail("RaytraceAll: nothing to render");
            HIP_OK(hipSetDevice(root->device));
            if (!C.planes) HIP_OK(hipMalloc((void **)&C.planes, 3 * P * sizeof(uint16_t)));
            // (the ABI's planes are zeroed first, raytrace.c:476,481,486, and every pixel belongs to exactly one tile of the deal:
            // the de-tiling launch simply writes them)
            const void *tileBuf = root->dev.tileBuf;
            const cl_uint *ids = root->dev.tileIds;
            size_t tiles = root->tileIds.size();
            if (count > 1) {
                // [scene][slot] tile buffers next to each other on the root, ids alongside; a peer's buffer travels device to device
                size_t total = 0;
                for (rtHipScene *s : C.scenes) if (s) total += s->tileIds.size();
                if (C.gatherTiles != total) {
                    if (C.gather) HIP_OK(hipFree(C.gather));
                    if (C.gatherIds) HIP_OK(hipFree(C.gatherIds));
                    C.gather = nullptr; C.gatherIds = nullptr;
                    HIP_OK(hipMalloc((void **)&C.gather, total * 3 * RT_TILE_PIXELS * sizeof(uint16_t)));
                    HIP_OK(hipMalloc((void **)&C.gatherIds, total * sizeof(uint32_t)));
                    std::vector<uint32_t> allIds;
                    for (rtHipScene *s : C.scenes) if (s) allIds.insert(allIds.end(), s->tileIds.begin(), s->tileIds.end());
                    HIP_OK(hipMemcpy(C.gatherIds, allIds.data(), total * sizeof(uint32_t), hipMemcpyHostToDevice));
                    C.gatherTiles = total;
                }
                size_t at = 0;
                for (rtHipScene *s : C.scenes) {
                    if (!s) continue;
                    const size_t bytes = s->tileIds.size() * 3 * RT_TILE_PIXELS * sizeof(uint16_t);
                    HIP_OK(hipMemcpyPeerAsync((char *)C.gather + at, root->device, s->dev.tileBuf, s->device, bytes, root->stream)); // (the scenes were synchronised above)
                    at += bytes;
                }
                tileBuf = C.gather; ids = C.gatherIds; tiles = total;
            }
            HIP_OK(rtk_launch_detile(tileBuf, ids, (uint32_t)tiles, d.width, d.height, root->tilesX, C.planes, C.planes + P, C.planes + 2 * P, 0, root->stream));
            HIP_OK(hipStreamSynchronize(root->stream));
            HIP_OK(hipMemcpy(outputRed, C.planes, P * sizeof(uint16_t), hipMemcpyDeviceToHost));
            HIP_OK(hipMemcpy(outputGreen, C.planes + P, P * sizeof(uint16_t), hipMemcpyDeviceToHost));
            HIP_OK(hipMemcpy(outputBlue, C.planes + 2 * P, P * sizeof(uint16_t), hipMemcpyDeviceToHost));
            return 0;
        };
        ok = gather() == 0;
    }
    if (ok) {
        for (int i = 0; i < 5; ++i) C.hash[i] = h[i];
        C.sampleCount = d.sampleCount;
        C.valid = true;
        g_progress.store(0.999f, std::memory_order_relaxed); // capped like raytrace.c:580; the caller sets 1.0 (render.cpp:1397)
    }
    if (timing) {
        const auto tEnd = std::chrono::steady_clock::now();
        auto ms = [](auto a, auto b) { return std::chrono::duration<double, std::milli>(b - a).count(); };
        fprintf(stderr, "libraytrace_hip: RaytraceAll: %s; hashing %.1f ms, build/update + render %.1f ms, gather + copy out %.1f ms\n",
                reuse ? (cameraChanged || lightsChanged || samplesChanged || gridChanged || materialsChanged ? "scene reused, parts rebuilt" : "scene reused as it is") : "scene built",
                ms(tCall, tHash), ms(tHash, tRender), ms(tRender, tEnd));
    }
    if (!ok || !useCache) {
        const std::string keep = g_error;
        C.clear();
        g_error = keep;
    }
    g_endTime.store((long)clock(), std::memory_order_relaxed);
    if (!ok) fprintf(stderr, "libraytrace_hip: RaytraceAll failed: %s\n", g_error.c_str());
    return ok ? CL_TRUE : CL_FALSE;
}

} // extern "C"

