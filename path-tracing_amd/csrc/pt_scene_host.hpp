// pt_scene_host.hpp -- host-side state of a device-resident scene, shared by the
// translation units that implement the C ABI.
#pragma once

#include <hip/hip_runtime.h>

#include <mutex>
#include <string>
#include <vector>

#include "pt_amd.h"
#include "pt_device.hpp"
#include "pt_internal.hpp"

struct RenderArgs {
    pt_camera cam;
    pt_render_params p;
    int x0, y0, tw, th; // tile origin and size
    int s0, s1;         // sample range
    ptd::PathParams pp;
};


#define HIP_TRY(expr)                                                                                  \
    do {                                                                                               \
        hipError_t e_ = (expr);                                                                        \
        if (e_ != hipSuccess)                                                                          \
            return ptint::fail(PT_ERR_DEVICE, std::string(#expr) + ": " + hipGetErrorString(e_));             \
    } while (0)

struct DeviceGuard {
    int prev = -1;
    bool ok = false;
    explicit DeviceGuard(int dev)
    {
        if (hipGetDevice(&prev) != hipSuccess)
            prev = -1;
        ok = hipSetDevice(dev) == hipSuccess;
    }
    ~DeviceGuard()
    {
        if (prev >= 0)
            (void)hipSetDevice(prev);
    }
};

template <typename T>
int upload(const T* host, size_t count, T** dev, std::vector<void*>& owned)
{
    *dev = nullptr;
    size_t bytes = std::max<size_t>(count, 1) * sizeof(T);
    void* p = nullptr;
    HIP_TRY(hipMalloc(&p, bytes));
    owned.push_back(p);
    if (count > 0)
        HIP_TRY(hipMemcpy(p, host, count * sizeof(T), hipMemcpyHostToDevice));
    *dev = static_cast<T*>(p);
    return PT_OK;
}


// Work areas of the wavefront renderer, allocated on first use and kept.
struct WavefrontBuffers {
    void* pool = nullptr;       // path-state arrays, queues, retry list
    size_t poolCapacity = 0;    // bytes
    float* samples = nullptr;   // per-sample radiance of one batch
    uint64_t sampleCapacity = 0; // in samples (3 floats each)
    float* accum = nullptr;     // running per-pixel sums across batches
    uint64_t accumCapacity = 0; // in pixels
    uint32_t* counters = nullptr;
    uint32_t* hostCounters = nullptr; // pinned
    unsigned long long* walkStats = nullptr; // sums of the instrumented tracing kernels (see pt_walk_counters)
    bool fixed = false;      // the areas are parts of a fixed work area: never grown, never freed
    uint32_t fixedSlots = 0; // records per pool that the fixed pool area holds
};

// Per-kernel device time of the wavefront renderer, measured with pairs of events on the
// stream a kernel is launched on.  begin(k) / end() bracket one launch; collect() (once the
// streams have been synchronised) adds the elapsed times to the sums.
struct KernelTimes {
    enum Kind { kGenerate = 0, kTrace, kShade, kResolve, kClassify, kTraceOrdered, kTraceRetry, kFinish, kKinds };
    struct Span {
        size_t start, stop; // indices into `events`
        int kind;
    };
    bool enabled = false;
    std::vector<hipEvent_t> events;
    std::vector<Span> spans; // one per launch since the last collect()
    size_t used = 0;         // events in use
    bool chained = false;    // the last thing put on the stream was the stop event of a launch
    double ms[kKinds] = {};
    uint64_t launches[kKinds] = {};

    void reset()
    {
        used = 0;
        spans.clear();
        chained = false;
        for (int k = 0; k < kKinds; k++) {
            ms[k] = 0;
            launches[k] = 0;
        }
    }
    void record(hipStream_t stream)
    {
        if (used == events.size()) {
            hipEvent_t e = nullptr;
            if (hipEventCreate(&e) != hipSuccess)
                return;
            events.push_back(e);
        }
        (void)hipEventRecord(events[used], stream);
        used++;
    }
    // Launches that follow one another on the stream share an event: the stop of one is the start of the next (an
    // event between two kernels costs a few microseconds of the stream's time; two per launch were 1.9 % of a
    // Cornell render).  Whoever puts anything else on the stream in between calls unchain().
    void begin(int kind, hipStream_t stream)
    {
        if (!enabled)
            return;
        if (!chained || used == 0)
            record(stream);
        if (used == 0)
            return; // (no event could be created)
        spans.push_back(Span{ used - 1, used - 1, kind });
        chained = false;
    }
    void end(hipStream_t stream)
    {
        if (!enabled || spans.empty())
            return;
        const size_t before = used;
        record(stream);
        if (used > before) {
            spans.back().stop = used - 1;
            chained = true;
        }
    }
    void unchain() { chained = false; }
    // (called once the stream of the launches recorded so far has been waited for; if a span's events have not
    // completed all the same -- launches of an earlier call on another stream -- everything is kept for the next
    // collect rather than dropped)
    void collect()
    {
        for (const Span& sp : spans)
            if (sp.stop != sp.start && hipEventQuery(events[sp.stop]) == hipErrorNotReady)
                return;
        for (const Span& sp : spans) {
            float t = 0.0f;
            if (sp.stop != sp.start && hipEventElapsedTime(&t, events[sp.start], events[sp.stop]) == hipSuccess) {
                ms[sp.kind] += double(t);
                launches[sp.kind]++;
            }
        }
        spans.clear();
        used = 0;
        chained = false;
    }
    void release()
    {
        for (hipEvent_t e : events)
            (void)hipEventDestroy(e);
        events.clear();
        spans.clear();
        used = 0;
        chained = false;
    }
};

// What pt_render_multi keeps with a handle between calls.  As a peer: the image it renders into and the event it
// records behind its last kernel.  As the first handle: the result buffer of the host-output call, per peer a staging
// buffer, a copy stream and an event, and the figures of the last combine.
struct MultiPeer {
    float* staging = nullptr;
    size_t stagingCapacity = 0; // floats
    hipStream_t stream = nullptr;
    hipEvent_t copied = nullptr;
};
struct MultiState {
    std::mutex mu; // one pt_render_multi call at a time with this handle as the first
    float* image = nullptr;
    size_t imageCapacity = 0;
    hipEvent_t rendered = nullptr;
    float* result = nullptr;
    size_t resultCapacity = 0;
    std::vector<MultiPeer> peers;
    hipEvent_t combineStart = nullptr, combineStop = nullptr;
    bool timed = false;
    int lastKernels = 0, lastCopies = 0, lastDirectCopies = 0;
};

struct pt_scene {
    int device = 0;
    int computeUnits = 0; // of that GPU: the tracing launches are sized to what it holds at once
    ptd::DevScene dev;
    std::vector<void*> owned; // every device allocation of the scene
    float4* nodes = nullptr;  // also in owned; kept for pt_scene_update_bvh
    uint4* wide = nullptr;    // likewise (room for n_surfaces - 1 rows)
    uint4* quad = nullptr;    // rows of the four-wide walk, if made (room for n_surfaces - 1 rows of 4 x uint4)
    uint2* walkLinks = nullptr; // likewise, allocated when first needed: link table of the walk from LDS (room for any tree)
    std::vector<float4> hostSurf; // the surface table as uploaded: which leaves have the box of their vertices
    void* buildArea = nullptr; // work area of pt_scene_build_bvh_device, kept between calls (not in owned)
    size_t buildCapacity = 0;
    void* quadArea = nullptr;  // work area of the four-wide rows' builder, likewise
    size_t quadCapacity = 0;
    uint32_t nQuadRows = 0;    // rows of `quad` in use
    uint64_t n_surfaces = 0;
    uint32_t nMaterials = 0, nTextures = 0, nMeshes = 0; // entries of the material-side tables (see shadeHitsKernel)
    hipEvent_t evStart = nullptr, evStop = nullptr;
    bool timed = false;
    bool walkStatsOn = false; // launch the instrumented tracing kernels (pt_scene_count_walk)
    bool hasDynamic = false; // some surface or sphere has a time-dependent (two-key) pose
    WavefrontBuffers wf;
    KernelTimes kt;
    // The work areas above, the tree and the timing state belong to one call at a time: every entry point that
    // touches them holds `mu` while it enqueues, and makes its stream wait for `evDone`, which the previous such
    // call recorded behind its last kernel (calls may come from different threads and on different streams).
    std::mutex mu;
    hipEvent_t evDone = nullptr;
    bool pending = false;
    MultiState multi;
};

// One call on a scene: takes the scene's lock, selects its GPU, and orders `stream` behind the previous call's work.
struct SceneCall {
    std::lock_guard<std::mutex> lock;
    DeviceGuard guard;
    pt_scene* scene;
    hipError_t status = hipSuccess;
    SceneCall(pt_scene* sc, hipStream_t stream) : lock(sc->mu), guard(sc->device), scene(sc)
    {
        if (!guard.ok)
            status = hipErrorInvalidDevice;
        else if (sc->pending)
            status = hipStreamWaitEvent(stream, sc->evDone, 0);
    }
    // everything this call enqueued on `stream` is what the next call has to wait for
    hipError_t done(hipStream_t stream)
    {
        hipError_t e = hipEventRecord(scene->evDone, stream);
        if (e == hipSuccess)
            scene->pending = true;
        return e;
    }
};


// pt_multi.hip
void ptMultiRelease(pt_scene* scene);

// pt_wavefront.hip
int ptWavefrontRender(pt_scene* scene, const RenderArgs& a, float* d_out, hipStream_t stream);
int ptWavefrontSamples(pt_scene* scene, const RenderArgs& a, float* h_out_pixel_major);
int ptWavefrontTraceRays(pt_scene* scene, uint64_t n, const float* d_origins, const float* d_dirs, const float* d_times,
                         pt_hit* d_out, hipStream_t stream);
void ptWavefrontRelease(pt_scene* scene);
// the work area in the library's image on `device` (which must be current): looked up once per device
int ptWavefrontWalkStats(pt_scene* scene, int enable);
int ptWavefrontReadWalkStats(pt_scene* scene, pt_walk_counters* out);
int ptWavefrontStaticSetup(int device);
// (re)make DevScene::walkLinks for the tree the scene has now; the caller holds the scene's lock
int ptWavefrontBuildWalkLinks(pt_scene* scene, float time, hipStream_t stream);
// the rows of the four-wide walk for the tree the scene holds now (dev.nodes, dev.n_nodes, the grid of dev.wide)
int ptBuildQuadRows(pt_scene* scene, hipStream_t stream);
int ptBuildQuadRowsDevice(pt_scene* scene, hipStream_t stream);
int ptWavefrontRenderStatic(pt_scene* scene, const RenderArgs& a, float* d_out, hipStream_t stream);
