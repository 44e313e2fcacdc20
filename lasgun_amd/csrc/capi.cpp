// lasgun_amd/csrc/capi.cpp -- the C ABI of include/lasgun_hip.h: host objects, device upload,
// kernel launches.  No torch types, no C++ exceptions across the boundary, no CPU render path.
#include <hip/hip_runtime_api.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <cstring>
#include <functional>
#include <map>
#include <memory>
#include <mutex>
#include <set>
#include <string>
#include <thread>
#include <vector>

#include "../../include/lasgun_hip.h"
#include "host.h"
#include "tune.h"

namespace lg {
// k_mega.hip, k_wavefront.hip, k_queue.hip, k_probe.hip
hipError_t launch_trace(const DParams &P, bool stats, bool fast, uint32_t blocks, uint32_t stack_depth, hipStream_t stream);
hipError_t trace_occupancy(uint32_t stack_depth, bool fast, size_t extra_lds, int *blocks_per_cu);
hipError_t launch_wf_trace(const DParams &P, bool fast, bool shadow, uint32_t blocks, uint32_t stack_depth, hipStream_t stream);
hipError_t launch_wf_shade(const DParams &P, uint32_t blocks, hipStream_t stream);
hipError_t launch_wf_combine(const DParams &P, uint32_t blocks, hipStream_t stream);
hipError_t launch_wf_resolve(const DParams &P, uint32_t blocks, hipStream_t stream);
hipError_t wf_trace_occupancy(uint32_t stack_depth, bool fast, size_t extra_lds, int *blocks_per_cu);
hipError_t launch_queue(const DParams &P, uint32_t blocks, hipStream_t stream);
hipError_t queue_occupancy(uint32_t stack_depth, size_t extra_lds, int *blocks_per_cu);
hipError_t queue_set_lds_limit(size_t bytes, bool ldss);
hipError_t mega_set_lds_limit(size_t bytes, bool ldss);
hipError_t wf_set_lds_limit(size_t bytes, bool ldss);
hipError_t launch_kat(int kind, const double *params, const float *vpos, const uint32_t *tri_v, uint32_t ntri, V3 o, V3 d, double *out,
                      hipStream_t stream);
hipError_t launch_kat_si(V3 o, V3 d, double t, V3 dpdu, V3 dpdv, double *out, hipStream_t stream);
hipError_t launch_math(int op, size_t n, const double *a, const double *b, double *out, hipStream_t stream);
hipError_t launch_trace_pixel(const DParams &P, bool fast, uint32_t stack_depth, uint32_t x, uint32_t y, double *out, hipStream_t stream);
hipError_t launch_probe_copy(const void *src, void *dst, size_t bytes, hipStream_t stream);
hipError_t launch_probe_lds(uint32_t blocks, uint32_t iters, uint32_t *sink, hipStream_t stream);
} // namespace lg

using namespace lg;

static thread_local std::string tl_error;
static int g_device = 0;
static bool g_device_chosen = false; // lg_set_device was called: single-device captures stay on that device
static std::vector<int> g_devices; // lg_set_devices: the devices a host-film lg_capture is split over (empty = g_device)

static int fail(const std::string &msg) {
    tl_error = msg;
    return 1;
}
#define HIP_TRY(expr)                                                                                                   \
    do {                                                                                                                \
        hipError_t _e = (expr);                                                                                         \
        if (_e != hipSuccess) throw Error(std::string(#expr) + ": " + hipGetErrorString(_e));                           \
    } while (0)

static void use_device(int dev) {
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0)
        throw Error("no HIP device available: liblasgun_hip has no CPU fallback (hipGetDeviceCount: " +
                    std::string(e == hipSuccess ? "0 devices" : hipGetErrorString(e)) + ")");
    if (dev < 0 || dev >= n) throw Error("device index out of range");
    HIP_TRY(hipSetDevice(dev));
}
static void use_device() { use_device(g_device); }

// Device allocations of 1 KiB and more are recycled through a small per-process pool (at most 4 GiB parked per
// device): capture() builds and drops an accel -- film staging, per-pixel state -- for every frame, like the
// reference, and hipMalloc / hipFree of those buffers would otherwise cost about a millisecond of each frame.
// Nothing relies on the contents of a fresh buffer: every buffer is written (kernel, memset or copy) before it is read.
namespace {
struct DevPool {
    std::mutex mtx;
    struct Block { int device; size_t bytes; void *p; };
    std::vector<Block> parked;
    size_t parked_bytes[64] = {0};
    static constexpr size_t MIN_BYTES = 1u << 10, CAP = 4ull << 30, MAX_BLOCKS = 256; // (from 1 KiB: the table arena, tile counters and queue counts of a frame are recycled too)
    // a parked block of at least `bytes` (and at most 1.25x that); *capacity receives its real size
    void *take(int device, size_t bytes, size_t *capacity) {
        std::lock_guard<std::mutex> g(mtx);
        size_t best = parked.size();
        for (size_t i = 0; i < parked.size(); ++i)
            if (parked[i].device == device && parked[i].bytes >= bytes && parked[i].bytes <= bytes + bytes / 4 &&
                (best == parked.size() || parked[i].bytes < parked[best].bytes)) best = i;
        if (best == parked.size()) return nullptr;
        void *p = parked[best].p;
        *capacity = parked[best].bytes;
        parked_bytes[device & 63] -= parked[best].bytes;
        parked.erase(parked.begin() + (long)best);
        return p;
    }
    bool park(int device, size_t bytes, void *p) {
        std::lock_guard<std::mutex> g(mtx);
        if (bytes < MIN_BYTES || parked_bytes[device & 63] + bytes > CAP || parked.size() >= MAX_BLOCKS) return false;
        parked.push_back(Block{device, bytes, p});
        parked_bytes[device & 63] += bytes;
        return true;
    }
    // give the parked blocks of `device` (or of every device: -1) back to the driver; returns the bytes freed.
    // The caller has made sure nothing on the device still uses them (blocks are parked only after a synchronise).
    size_t trim(int device) {
        std::vector<Block> drop;
        {
            std::lock_guard<std::mutex> g(mtx);
            for (size_t i = 0; i < parked.size();)
                if (device < 0 || parked[i].device == device) {
                    drop.push_back(parked[i]);
                    parked_bytes[parked[i].device & 63] -= parked[i].bytes;
                    parked.erase(parked.begin() + (long)i);
                } else ++i;
        }
        size_t freed = 0;
        int cur = 0;
        (void)hipGetDevice(&cur);
        for (const Block &b : drop) {
            if (hipSetDevice(b.device) == hipSuccess && hipFree(b.p) == hipSuccess) freed += b.bytes;
        }
        (void)hipSetDevice(cur);
        return freed;
    }
};
// never destroyed: buffers released at interpreter exit, after static destructors have begun, still find it
DevPool &g_pool = *new DevPool();
} // namespace

template <class T> struct DevBuf {
    T *p = nullptr;
    size_t n = 0;
    size_t bytes_ = 0; // capacity in bytes (what the pool is told)
    int device_ = 0;
    bool borrowed_ = false; // a view into another DevBuf's allocation (TableStage): nothing to free
    DevBuf() = default;
    DevBuf(const DevBuf &) = delete;
    DevBuf &operator=(const DevBuf &) = delete;
    friend void swap(DevBuf &a, DevBuf &b) { std::swap(a.p, b.p); std::swap(a.n, b.n); std::swap(a.bytes_, b.bytes_); std::swap(a.device_, b.device_); std::swap(a.borrowed_, b.borrowed_); }
    void view(T *ptr, size_t count) { release(); p = ptr; n = count; borrowed_ = true; }
    void obtain(size_t bytes) {
        release();
        HIP_TRY(hipGetDevice(&device_));
        bytes_ = bytes;
        void *q = bytes >= DevPool::MIN_BYTES ? g_pool.take(device_, bytes, &bytes_) : nullptr; // bytes_: the block's real capacity
        if (!q) {
            hipError_t e = hipMalloc(&q, bytes);
            if (e == hipErrorOutOfMemory) { // the pool may be sitting on the memory: hand it back and try once more
                (void)hipGetLastError();
                g_pool.trim(device_);
                e = hipMalloc(&q, bytes);
            }
            if (e != hipSuccess) throw Error(std::string("hipMalloc(") + std::to_string(bytes) + " bytes): " + hipGetErrorString(e));
        }
        p = (T *)q;
    }
    void upload(const std::vector<T> &v) {
        obtain((v.size() ? v.size() : 1) * sizeof(T));
        n = v.size();
        if (n) HIP_TRY(hipMemcpy(p, v.data(), n * sizeof(T), hipMemcpyHostToDevice));
    }
    void alloc(size_t count) {
        obtain((count ? count : 1) * sizeof(T));
        n = count;
    }
    void release() {
        if (p && !borrowed_ && !g_pool.park(device_, bytes_, p)) (void)hipFree(p);
        p = nullptr; n = 0; bytes_ = 0; borrowed_ = false;
    }
    ~DevBuf() { release(); }
};

// The scene tables of an accel go to the device in ONE allocation and ONE copy: capture() builds an accel for every frame like the
// reference (lib.rs:64), and two dozen hipMalloc + hipMemcpy pairs of a few kilobytes each were a third of lg_accel_from's millisecond
// on the headline scene.  Tables of 256 KiB and more keep an allocation and a copy of their own (staging them would cost a host
// memcpy of megabytes); the rest are staged here, 256-byte aligned, and become views into `arena` at commit().
struct TableStage {
    std::vector<uint8_t> host;
    std::vector<std::function<void(uint8_t *)>> fix;
    template <class T> void add(DevBuf<T> &buf, const std::vector<T> &v) {
        const size_t bytes = v.size() * sizeof(T);
        if (bytes >= (256u << 10)) { buf.upload(v); return; }
        const size_t off = (host.size() + 255) & ~(size_t)255;
        host.resize(off + (bytes ? bytes : 1));
        if (bytes) std::memcpy(host.data() + off, v.data(), bytes);
        const size_t count = v.size();
        DevBuf<T> *b = &buf;
        fix.push_back([b, off, count](uint8_t *base) { b->view(reinterpret_cast<T *>(base + off), count); });
    }
    void commit(DevBuf<uint8_t> &arena) {
        arena.alloc(host.size() ? host.size() : 1);
        if (!host.empty()) HIP_TRY(hipMemcpy(arena.p, host.data(), host.size(), hipMemcpyHostToDevice));
        for (auto &f : fix) f(arena.p);
    }
};

// Streams are recycled per device as well: capture() makes an accel per frame, and creating its stream (and the copy stream of a
// banded capture) cost a tenth of a millisecond each.  A stream goes back when its accel dies; whatever it may still hold is ahead
// of the next owner's work in stream order.
namespace {
struct StreamPool {
    std::mutex mtx;
    std::vector<std::pair<int, hipStream_t>> spare;
    hipStream_t take(int device) {
        {
            std::lock_guard<std::mutex> g(mtx);
            for (size_t i = 0; i < spare.size(); ++i)
                if (spare[i].first == device) { hipStream_t s = spare[i].second; spare.erase(spare.begin() + (long)i); return s; }
        }
        hipStream_t s = nullptr;
        HIP_TRY(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
        return s;
    }
    void give(int device, hipStream_t s) {
        if (!s) return;
        {
            std::lock_guard<std::mutex> g(mtx);
            if (spare.size() < 64) { spare.emplace_back(device, s); return; }
        }
        (void)hipStreamDestroy(s);
    }
};
StreamPool &g_streams = *new StreamPool(); // never destroyed (see g_pool)
} // namespace

// Pinned host staging for the small tables a *_device entry point uploads (a batch's k table, a lattice row table): the copy is
// enqueued on the caller's stream from memory that stays put until the copy is through, so the call only enqueues (round 5 used a blocking
// hipMemcpy, and a device-wide synchronise when a row table changed: ADVICE r5).  Blocks are powers of two, recycled per process.
namespace {
struct PinnedPool {
    std::mutex mtx;
    std::vector<std::pair<size_t, void *>> spare;
    void *take(size_t bytes, size_t *capacity) {
        size_t cap = 4096;
        while (cap < bytes) cap <<= 1;
        *capacity = cap;
        {
            std::lock_guard<std::mutex> g(mtx);
            for (size_t i = 0; i < spare.size(); ++i)
                if (spare[i].first == cap) { void *q = spare[i].second; spare.erase(spare.begin() + (long)i); return q; }
        }
        void *q = nullptr;
        hipError_t e = hipHostMalloc(&q, cap, hipHostMallocPortable);
        if (e != hipSuccess) throw Error(std::string("hipHostMalloc(staging): ") + hipGetErrorString(e));
        return q;
    }
    void give(size_t cap, void *q) {
        if (!q) return;
        {
            std::lock_guard<std::mutex> g(mtx);
            if (spare.size() < 64) { spare.emplace_back(cap, q); return; }
        }
        (void)hipHostFree(q);
    }
};
PinnedPool &g_pinned = *new PinnedPool(); // never destroyed (see g_pool)
struct PinnedBuf {
    void *p = nullptr;
    size_t cap = 0;
    PinnedBuf() = default;
    PinnedBuf(const PinnedBuf &) = delete;
    PinnedBuf &operator=(const PinnedBuf &) = delete;
    void need(size_t bytes) { if (cap < bytes) { g_pinned.give(cap, p); p = nullptr; cap = 0; p = g_pinned.take(bytes, &cap); } }
    ~PinnedBuf() { g_pinned.give(cap, p); }
};
} // namespace

// Sticky error words of the queue organisation (k_queue.hip: a wave that gave up waiting for work).  They live in PINNED HOST memory
// that every device writes straight into (system-scope store), one word per accel, handed out from pages of 1024: the host reads
// a word without a HIP call -- after any synchronise, at the head of every enqueue, in lg_accel_synchronize -- and only the host
// ever clears it, after it has reported it.  (Round 4 kept the word among the per-launch control words: the memset before the
// next chunk or supersample erased it, and a launch on a caller's stream was looked at before it had finished -- ADVICE r4.)
namespace {
struct ErrWords {
    std::mutex mtx;
    std::vector<uint32_t *> pages;
    std::vector<uint32_t *> spare;
    uint32_t *take() {
        std::lock_guard<std::mutex> g(mtx);
        if (spare.empty()) {
            void *q = nullptr;
            hipError_t e = hipHostMalloc(&q, 4096, hipHostMallocPortable | hipHostMallocMapped);
            if (e != hipSuccess) throw Error(std::string("hipHostMalloc(error words): ") + hipGetErrorString(e));
            std::memset(q, 0, 4096);
            pages.push_back((uint32_t *)q);
            for (int i = 1023; i >= 0; --i) spare.push_back((uint32_t *)q + i);
        }
        uint32_t *w = spare.back();
        spare.pop_back();
        *(volatile uint32_t *)w = 0u;
        return w;
    }
    void give(uint32_t *w) {
        if (!w) return;
        std::lock_guard<std::mutex> g(mtx);
        spare.push_back(w);
    }
};
ErrWords &g_err_words = *new ErrWords(); // never destroyed (see g_pool)
} // namespace

struct lg_scene {
    Scene s;
};
struct lg_aggregate {
    Aggregate a;
};
struct lg_film {
    uint32_t w = 0, h = 0;
    std::vector<uint8_t> owned;
    uint8_t *px = nullptr;
};

struct lg_accel {
    const Scene *scene = nullptr;
    int device = 0; // the HIP device this accel's tables and launches live on
    FlatScene flat;
    DevBuf<uint8_t> arena; // the small tables live here (TableStage); the DevBufs below are views into it or allocations of their own
    DevBuf<DNode> nodes;
    DevBuf<DNode4> nodes4;
    DevBuf<uint32_t> primref;
    DevBuf<DSphere> spheres;
    DevBuf<int32_t> sphere_mat;
    DevBuf<DCuboid> cuboids;
    DevBuf<int32_t> cuboid_mat;
    DevBuf<uint32_t> tri_v, tri_n, tri_t;
    DevBuf<float> vpos, vnorm, vtex;
    DevBuf<DLeafRec> leaf_soup;
    DevBuf<DChunk> chunks;
    DevBuf<DStrip> strips;
    DevBuf<uint32_t> sphere_ref_leaf, cuboid_ref_leaf, tri_ref_leaf, accel_ref_leaf;
    DevBuf<DAccel> accels;
    DevBuf<DMaterial> materials;
    DevBuf<DLight> lights;
    // launch resources (mutable: a `const lg_accel*` render call still enqueues work).  Everything a launch
    // scribbles on lives in a per-STREAM context, so launches of one accel on different streams (frame k+1's
    // primary pass filling the tail of frame k's shadow pass) do not share tile counters or per-pixel state.
    struct LaunchCtx {
        hipStream_t key = nullptr;
        unsigned long long last_use = 0;
        DevBuf<uint32_t> tile_counter;                         // [0] next tile; one head per XCD band behind it (TILE_COUNTER_WORDS)
        DevBuf<double> frames, stash;                          // megakernel: Whitted frame stack, parked shading frame
        DevBuf<uint8_t> wf_mem;                                // wavefront pipeline: every per-level array of a chunk, carved from one allocation
        DevBuf<uint32_t> wf_counters;                          // its queue counts and per-launch tile counters
        // strided subsets by lattice column (shade.h, modes 4 / 5): (floor(y*w / n), (y*w) mod n) per film row -- one table per (w, h, n), the
        // last MAX_ROW_TABLES of them kept (a caller that alternates periods or films on one stream finds each again), each uploaded from
        // pinned staging of its own on the context's stream (`up`: that copy is through; the staging may be rewritten)
        struct RowTable { uint32_t w = 0, h = 0; unsigned long long n = 0, last_use = 0; DevBuf<DRowTab> buf; PinnedBuf stage; hipEvent_t up = nullptr; };
        std::vector<std::unique_ptr<RowTable>> rowtabs;
        unsigned long long rowtab_clock = 0;
        // lg_capture_subsets: the k tables of the batches in flight on this stream (addressing mode 3), each with the event that says
        // its launch is through -- a table is copied (from pinned staging of its own, on the stream) into a buffer of its own before its
        // launch is enqueued, so neither a later batch on the stream nor the caller's freed array can reach it
        struct KsTable { DevBuf<unsigned long long> buf; PinnedBuf stage; hipEvent_t done = nullptr; };
        std::vector<std::unique_ptr<KsTable>> ks_live;
        // the level-by-level chain of a SMALL frame as a HIP graph (enqueue_wavefront): what the chain was captured for (a hash of its
        // parameters), and the chain the context saw last -- a chain is captured when it comes a second time in a row, so a one-frame
        // program never pays for a capture
        hipGraphExec_t wf_graph = nullptr;
        uint64_t wf_graph_sig = 0, wf_last_sig = 0;
        unsigned wf_graph_captures = 0;
        ~LaunchCtx() {
            for (auto &k : ks_live) if (k->done) (void)hipEventDestroy(k->done);
            for (auto &r : rowtabs) if (r->up) (void)hipEventDestroy(r->up);
            if (wf_graph) (void)hipGraphExecDestroy(wf_graph);
        }
    };
    mutable std::vector<std::unique_ptr<LaunchCtx>> ctxs;
    // wavefront pipeline, big launches: the frame is cut into bands rendered on internal streams (each with a launch context
    // of its own), so one band's closest pass fills the tails of another band's shadow and shade passes and the sparse deeper
    // levels of a recursive scene run beside other bands' level 0 (enqueue_wavefront)
    mutable std::vector<hipStream_t> aux_streams;
    mutable std::vector<hipEvent_t> aux_done;
    mutable hipEvent_t aux_fork = nullptr;
    mutable unsigned wf_split = 0;               // lg_accel_set_wf_split: bands of a big wavefront launch (0 = LASGUN_WF_SPLIT, default 1)
    mutable unsigned long long ctx_clock = 0;
    mutable bool streaming = true; // use the streaming pipeline when the scene allows it
    mutable bool streaming_forced = false; // lg_accel_set_streaming(2): ignore the two criteria below (tests)
    // the pipeline pays for its per-pixel state traffic only where node / sphere / box traversal dominates a
    // ray's cost (tools/threshold_sweep.py): set from the scene by lg_accel_from
    bool streaming_pays = false;
    unsigned long long streaming_min_items = 1ull << 20;
    unsigned long long specular_small_items = 1ull << 20; // a glass / mirror scene resident in LDS: frames up to this many pixels go level by level
    bool mega_narrow = false;        // the LDS-resident megakernel in 768-lane workgroups: scenes of fewer than 512 spheres / boxes (measured, k_mega.hip)
    uint32_t wf_blocks = 1, wf_blocks_fast = 1;   // grids of the wavefront pipeline's 256-lane traversal kernels
    uint32_t queue_blocks = 1;                    // grid of the queue organisation's persistent kernel (256-lane form)
    mutable uint32_t *q_err = nullptr;            // the queue organisation's sticky error word (pinned host memory, g_err_words): taken at its first launch
    mutable int queue = -1;                       // lg_accel_set_streaming(3) forces the queue organisation, (0..2) rule it out; -1 = queue_default
    mutable int last_org = -1;                    // what the last launch ran as: 0 megakernel, 1 level by level, 2 queue, + 16 with its tiles claimed bottom-up (lg_accel_last_organisation)
    mutable int tile_parts = -1;                  // lg_accel_set_tile_parts: the megakernel hands a tile out whole (1) or in 2 / 4 / 8 parts; -1 = whole unless the measured choice says quarters
    mutable int sample_order = -1;                // lg_accel_set_sample_order: 0 a pixel's samples side by side, 1 one after the other, -1 = side by side (megakernel: rule / measured)
    mutable int tile_order = -1;                  // lg_accel_set_tile_order: 0 top-down, 1 bottom-up, 2 from the middle row outwards, -1 = middle-out unless the measured choice says otherwise
    bool queue_default = false;                   // glass / mirror over a big mesh: long uneven walks, sparse deep levels (k_queue.hip)
    mutable size_t queue_budget = 0;              // bytes one launch context may hold for it (0 = from the free memory at first use)
    unsigned long long queue_min_items = 1ull << 16; // launches below this many pixels stay with the megakernel
    mutable size_t wf_budget = 0;                 // bytes one launch context may hold for it (0 = from the free memory at first use)
    // LDS-resident scene (reference tree only): the tables in their LDS layout, when they fit beside the stacks
    DevBuf<uint32_t> lds_image;
    DevBuf<uint32_t> accel_image;     // scenes in L2: the accel records alone, in their LDS layout (DParams::accel_image); empty: too many accels
    uint32_t accel_image_n16 = 0;
    uint32_t lds_image_n16 = 0, lds_node_off = 0, lds_prim_off = 0, lds_soup_off = 0, lds_accel_off = 0;
    uint32_t ldss_blocks = 0;         // one 1024-lane workgroup per CU; 0 = variant unavailable for this scene
    uint32_t cus = 1;                 // compute units of the accel's device
    mutable bool lds_scene = true;    // lg_accel_set_lds_scene
    mutable DevBuf<DStats> stats;
    mutable DevBuf<uint8_t> staging;    // device film for host-film captures
    mutable DevBuf<double> staging_rad;
    mutable std::mutex mtx;
    hipStream_t stream = nullptr;
    uint32_t stack_depth = 1;      // reference traversal
    uint32_t stack_depth_fast1 = 1; // fast traversal (one word per pending child; also deep enough for its reference re-trace)
    uint32_t max_blocks = 1;
    uint32_t max_blocks_fast = 1;
    uint64_t device_bytes = 0;
    mutable bool profiling = false;
    mutable bool fast = false; // opt-in fast traversal mode (lg_accel_set_mode)
    mutable int prune = -1;    // lg_accel_set_prune: -1 = prune_default
    bool prune_default = false; // the scene carries a mesh with fat leaves
    bool fast_available = true;
    std::string fast_refusal = "fast mode unavailable: its tree is too deep for the LDS stack";
    mutable std::vector<std::pair<hipEvent_t, hipEvent_t>> events;
    // streaming pipeline, per kernel kind: 0 primary trace, 1 frame, 2 shadow trace, 3 shade; 4 = megakernel
    mutable std::vector<std::pair<hipEvent_t, hipEvent_t>> kind_events[5];
    ~lg_accel() {
        for (auto &e : events) { (void)hipEventDestroy(e.first); (void)hipEventDestroy(e.second); }
        for (auto &v : kind_events) for (auto &e : v) { (void)hipEventDestroy(e.first); (void)hipEventDestroy(e.second); }
        for (auto e : aux_done) (void)hipEventDestroy(e);
        if (aux_fork) (void)hipEventDestroy(aux_fork);
        for (auto st : aux_streams) g_streams.give(device, st);
        g_streams.give(device, stream);
        g_err_words.give(q_err);
    }
};

constexpr size_t MAX_LAUNCH_CTXS = 8;
constexpr unsigned MAX_WF_BANDS = 4; // bands of a big wavefront launch on internal streams (lg_accel_set_wf_split)
// The launch context of `stream` (at most MAX_LAUNCH_CTXS are kept; the least recently used one is recycled after a
// device-wide synchronise).  Caller holds a.mtx and has made the accel's device current.
static lg_accel::LaunchCtx &ctx_for(const lg_accel &a, hipStream_t stream) {
    for (auto &c : a.ctxs)
        if (c->key == stream) { c->last_use = ++a.ctx_clock; return *c; }
    lg_accel::LaunchCtx *c = nullptr;
    if (a.ctxs.size() < MAX_LAUNCH_CTXS) {
        a.ctxs.emplace_back(new lg_accel::LaunchCtx());
        c = a.ctxs.back().get();
        c->tile_counter.alloc(TILE_COUNTER_WORDS);
        HIP_TRY(hipMemset(c->tile_counter.p, 0, TILE_COUNTER_WORDS * sizeof(uint32_t)));
    } else {
        c = a.ctxs[0].get();
        for (auto &x : a.ctxs) if (x->last_use < c->last_use) c = x.get();
        HIP_TRY(hipDeviceSynchronize()); // nothing may still be using the recycled buffers
    }
    c->key = stream;
    c->last_use = ++a.ctx_clock;
    return *c;
}

// The queue organisation's error word: a wave that gave up waiting for work that never came (a scheduler bug) must fail a call, not
// leave a half-rendered film behind.  The word is sticky -- the device only ever sets it -- and is cleared here, once reported.
// Looked at after every synchronise of the accel's stream, at the head of every enqueue and in lg_accel_synchronize: a launch on a
// CALLER's stream that stalled is reported by the first of those that follows its end.  Caller holds a.mtx.
static void check_queue_error(const lg_accel &a) {
    if (!a.q_err) return;
    volatile uint32_t *w = a.q_err;
    if (*w == 0u) return;
    *w = 0u;
    throw Error("queue organisation: a wave gave up waiting for work (scheduler stalled); the film is incomplete");
}
static void sync_checked(const lg_accel &a) {
    HIP_TRY(hipStreamSynchronize(a.stream));
    check_queue_error(a);
}

// ------------------------------------------------------------------------------------------
static DParams base_params(const lg_accel &a, uint32_t w, uint32_t h) {
    const Scene &s = *a.scene;
    DParams P{};
    P.nodes = a.nodes.p; P.nodes4 = a.nodes4.p; P.primref = a.primref.p; P.spheres = a.spheres.p; P.sphere_mat = a.sphere_mat.p;
    P.cuboids = a.cuboids.p; P.cuboid_mat = a.cuboid_mat.p; P.tri_v = a.tri_v.p; P.tri_n = a.tri_n.p; P.tri_t = a.tri_t.p;
    P.vpos = a.vpos.p; P.vnorm = a.vnorm.p; P.vtex = a.vtex.p; P.leaf_soup = a.leaf_soup.p; P.chunks = a.chunks.p; P.strips = a.strips.p; P.sphere_ref_leaf = a.sphere_ref_leaf.p; P.cuboid_ref_leaf = a.cuboid_ref_leaf.p;
    P.tri_ref_leaf = a.tri_ref_leaf.p; P.accel_ref_leaf = a.accel_ref_leaf.p; P.accels = a.accels.p; P.materials = a.materials.p;
    P.lights = a.lights.p;
    P.nlights = (uint32_t)a.flat.lights.size();
    P.recursion = s.recursion;
    P.default_material = a.flat.default_material;
    P.stack_depth = a.stack_depth;
    {   // LASGUN_ACCEL_LDS=0 (A/B): the accel records from the DAccel table in L2
        static const bool accel_lds = [] { const char *e = std::getenv("LASGUN_ACCEL_LDS"); return !(e && e[0] == '0'); }();
        P.accel_image = a.accel_image_n16 && accel_lds ? a.accel_image.p : nullptr; P.accel_image_n16 = a.accel_image_n16;
    }
    {   // LASGUN_PRUNE=0|1 replaces the scene-dependent DEFAULT (test suites run whole under either); lg_accel_set_prune still wins
        static const int env_default = [] { const char *e = std::getenv("LASGUN_PRUNE"); return e && (e[0] == '0' || e[0] == '1') ? e[0] - '0' : -1; }();
        const bool dflt = env_default < 0 ? a.prune_default : env_default != 0;
        P.prune = !a.fast && (a.prune < 0 ? dflt : a.prune != 0) ? 1u : 0u;
    }
    P.cam_origin = s.camera.origin; P.cam_view = s.camera.view; P.cam_up = s.camera.up; P.cam_aux = s.camera.aux;
    P.image_plane_height = s.camera.image_plane_height;
    P.pixel_separation = s.camera.pixel_separation;
    P.ss_distance = s.camera.ss_distance;
    P.ss_root = s.camera.ss_root;
    {   // LASGUN_SLAB_SIGNS=0: the reference's slab formula as written in every node step (A/B, tests)
        static const bool signs = [] { const char *e = std::getenv("LASGUN_SLAB_SIGNS"); return !(e && e[0] == '0'); }();
        P.boxes_finite = a.flat.boxes_finite && signs ? 1u : 0u;
    }
    P.bg_inner = s.bg_inner; P.bg_outer = s.bg_outer; P.bg_scale = s.bg_scale;
    P.ambient = s.ambient;
    P.w = w; P.h = h;
    P.winv = 1. / (double)w; P.hinv = 1. / (double)h; P.aspect = (double)w / (double)h; // film.rs:40-42
    return P;
}

// the accel's internal streams (bands of a big wavefront launch, bands of a whole-film capture).  Caller holds a.mtx.
static void ensure_aux_streams(const lg_accel &a, unsigned n) {
    while (a.aux_streams.size() < n) {
        hipStream_t st = g_streams.take(a.device); hipEvent_t ev = nullptr;
        HIP_TRY(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
        a.aux_streams.push_back(st); a.aux_done.push_back(ev);
    }
    if (!a.aux_fork) HIP_TRY(hipEventCreateWithFlags(&a.aux_fork, hipEventDisableTiming));
}

// The wavefront pipeline (k_wavefront.hip): per chunk of the film and per supersample, levels 0 .. L-1 top-down (closest,
// shadow, shade), then the combine passes bottom-up.  Queue capacities are worst case (level d holds at most 2^d rays per
// pixel of the chunk), so the chunk is sized to the memory budget of the launch context: nothing can overflow.
constexpr size_t WF_FULL_MIN_HOST = 48; // == WF_FULL_MIN of k_wavefront.hip
constexpr uint32_t MEGA_SPLIT = 4;      // the parts a small launch's tiles are handed out in where the measured choice found that faster (enqueue_mega, enqueue_queue)
static void enqueue_wavefront(const lg_accel &a, DParams &P0, lg_accel::LaunchCtx &c, hipStream_t stream) {
    const uint32_t levels = (a.flat.has_specular && P0.recursion > 0) ? P0.recursion + 1u : 1u;
    const uint32_t nsamples = P0.ss_root * P0.ss_root;
    // A supersampled launch runs its samples SIDE BY SIDE (DParams::ss_par): level 0 holds pixels x samples work items, one chain of
    // launches per chunk instead of one per sample, and a resolve pass sums each pixel's samples in their order.  The levels of a 9-sample
    // frame are nine times as wide -- a 512^2 film of glass fills the machine at its deep levels, which one sample at a time does not.
    // lg_accel_set_sample_order(1) / LASGUN_SS_SERIAL=1 (A/B): one chain per sample, summed as they come.
    static const bool ss_serial = [] { const char *e = std::getenv("LASGUN_SS_SERIAL"); return e && e[0] == '1'; }();
    const uint32_t S = nsamples > 1 && !ss_serial && a.sample_order != 1 ? nsamples : 1u; // level-0 work items per pixel
    // bytes per level-0 work item of a chunk
    auto level_bytes = [&](uint32_t d) -> size_t {
        size_t b = 0;
        if (d >= 1) b += 6 * 8;                       // ray queue
        if (levels > 1) b += 3 * 8;                   // output / li
        if (d + 1 < levels) b += 8 * 8 + 2 * 4;       // children's weights and indices
        return b;
    };
    size_t per_item = (nsamples > 1 ? 3 * 8 : 0);
    for (uint32_t d = 0; d < levels; ++d) per_item += level_bytes(d) << d;
    per_item += ((size_t)(4 + STASH_DOUBLES * 8 + 4) << (levels - 1)) * 7 / 4; // hit queue, frame, visibility of the widest level: dense part + appended part
    const size_t per_pixel = per_item * S;
    if (a.wf_budget == 0) {
        size_t free_b = 0, total_b = 0;
        HIP_TRY(hipMemGetInfo(&free_b, &total_b));
        // per launch CONTEXT, and an accel keeps up to MAX_LAUNCH_CTXS of them plus the band contexts (a caller with
        // four frames in flight uses five): a sixteenth of the free memory, at most 8 GiB each (the headline frame
        // needs 3.3 GB and stays one chunk; an allocation that fails anyway halves the chunk below)
        size_t budget = free_b / 16;
        const char *env = std::getenv("LASGUN_WF_BUDGET_MB");
        if (env && std::atoll(env) > 0) budget = (size_t)std::atoll(env) << 20;
        else if (budget > (8ull << 30)) budget = 8ull << 30;
        a.wf_budget = budget < (64ull << 20) ? (64ull << 20) : budget;
    }
    unsigned long long chunk_tiles = a.wf_budget / (per_pixel * 64);
    const unsigned long long cap_limit = (0xFFFFFFF0ull >> (levels - 1)) / (64ull * S); // ray indices are 32-bit
    if (chunk_tiles > cap_limit) chunk_tiles = cap_limit;
    if (chunk_tiles < 1) chunk_tiles = 1;
    if (chunk_tiles > P0.ntiles) chunk_tiles = P0.ntiles;
    // Bands on internal streams: opt-in (lg_accel_set_wf_split, or LASGUN_WF_SPLIT=n as the default), launches of 2 Mpixel and
    // more.  Measured: one headline frame at a time 7.79 -> 7.50 ms with 4 bands, but 7.20 -> 7.46 ms
    // when the caller already keeps four frames in flight -- which is why it is not the default.
    static const unsigned split_env = [] { const char *e = std::getenv("LASGUN_WF_SPLIT"); return e && std::atoi(e) > 0 ? (unsigned)std::atoi(e) : 1u; }();
    const unsigned want = a.wf_split ? a.wf_split : split_env;
    // (at most MAX_WF_BANDS bands: their contexts and the callers' streams share the accel's MAX_LAUNCH_CTXS slots, and a
    // context that has to be recycled costs a device-wide synchronise)
    const unsigned split = (unsigned long long)P0.ntiles * 64ull >= (1ull << 21) ? std::min(want, MAX_WF_BANDS) : 1u;
    if (split > 1) chunk_tiles = std::min<unsigned long long>(chunk_tiles, (P0.ntiles + split - 1) / split);
    unsigned long long nchunks = (P0.ntiles + chunk_tiles - 1) / chunk_tiles;
    const unsigned nstreams = split > 1 && nchunks > 1 ? (unsigned)std::min<unsigned long long>(split, nchunks) : 0u; // 0: everything on the caller's stream
    if (nstreams) ensure_aux_streams(a, nstreams);
    unsigned long long n0 = 0;
    size_t need = 0, hit_cap = 0, hit_len = 0;
    const uint32_t nlaunch = 4 * levels;
    const uint32_t CL = TILE_COUNTER_WORDS; // the queue counts (3 per level) in the first block, then a block of tile heads per launch (one head per XCD, each on a line of its own)
    auto size_chunk = [&] {
        n0 = chunk_tiles * 64ull * S;
        need = (size_t)n0 * per_item + 4096 * (3 * levels + 4);
        hit_cap = (size_t)n0 << (levels - 1);
        hit_len = hit_cap + hit_cap / 64 * (WF_FULL_MIN_HOST - 1); // appended part: fewer than WF_FULL_MIN hits per block of 64 rays
        nchunks = (P0.ntiles + chunk_tiles - 1) / chunk_tiles;
    };
    size_chunk();
    struct Carved {
        std::vector<double *> q, out, spec;
        std::vector<uint32_t *> child;
        uint32_t *hq = nullptr, *vis = nullptr, *counters = nullptr;
        double *frame = nullptr, *accum = nullptr;
    };
    auto carve = [&](lg_accel::LaunchCtx &cx) { // this context's arrays for one chunk (256-byte aligned)
        if (cx.wf_mem.n < need) { HIP_TRY(hipDeviceSynchronize()); cx.wf_mem.alloc(need); }
        if (cx.wf_counters.n < CL * (1 + nlaunch)) { HIP_TRY(hipDeviceSynchronize()); cx.wf_counters.alloc(CL * (1 + nlaunch)); }
        Carved k;
        k.q.assign(levels, nullptr); k.out.assign(levels, nullptr); k.spec.assign(levels, nullptr); k.child.assign(levels, nullptr);
        uint8_t *cur = cx.wf_mem.p;
        auto take = [&](size_t bytes) { uint8_t *p = cur; cur += (bytes + 255) & ~(size_t)255; return p; };
        for (uint32_t d = 0; d < levels; ++d) {
            const size_t cap = (size_t)n0 << d;
            if (d >= 1) k.q[d] = (double *)take(cap * 6 * 8);
            if (levels > 1) k.out[d] = (double *)take(cap * 3 * 8);
            if (d + 1 < levels) { k.spec[d] = (double *)take(cap * 8 * 8); k.child[d] = (uint32_t *)take(cap * 2 * 4); }
        }
        k.hq = (uint32_t *)take(hit_len * 4);
        k.frame = (double *)take(hit_len * STASH_DOUBLES * 8);
        k.vis = (uint32_t *)take(hit_len * 4);
        k.accum = nsamples > 1 ? (double *)take((size_t)n0 * 3 * 8) : nullptr;
        k.counters = cx.wf_counters.p;
        return k;
    };
    std::vector<Carved> carved;
    std::vector<hipStream_t> lanes;
    for (;;) { // memory that is not there (other contexts, other accels, other processes): halve the chunk and carve again
        try {
            carved.clear(); lanes.clear();
            if (nstreams) for (unsigned j = 0; j < nstreams; ++j) { lanes.push_back(a.aux_streams[j]); carved.push_back(carve(ctx_for(a, a.aux_streams[j]))); }
            else { lanes.push_back(stream); carved.push_back(carve(c)); }
            break;
        } catch (const Error &e) {
            if (std::string(e.what()).find("hipMalloc") == std::string::npos || chunk_tiles <= 1) throw;
            (void)hipGetLastError(); // (the failed allocation's error must not be what the next launch reports)
            chunk_tiles = (chunk_tiles + 1) / 2;
            a.wf_budget = std::max<size_t>(a.wf_budget / 2, 64ull << 20); // (later launches start from what fitted)
            size_chunk();
            if (std::getenv("LASGUN_DEBUG")) std::fprintf(stderr, "[lasgun] wavefront: %s -- chunks of %llu tiles instead\n", e.what(), chunk_tiles);
        }
    }
    if (nstreams) {
        HIP_TRY(hipEventRecord(a.aux_fork, stream)); // the bands start after whatever the caller's stream holds (a film clear, the previous frame's copy)
        for (unsigned j = 0; j < nstreams; ++j) HIP_TRY(hipStreamWaitEvent(a.aux_streams[j], a.aux_fork, 0));
    }

    const bool ldss = !a.fast && a.lds_scene && a.ldss_blocks;
    const uint32_t depth = a.fast ? a.stack_depth_fast1 : a.stack_depth;
    const uint32_t trace_cap = ldss ? a.ldss_blocks : (a.fast ? a.wf_blocks_fast : a.wf_blocks);
    const uint32_t flat_cap = a.cus * 16u;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    if (a.profiling) { HIP_TRY(hipEventCreate(&e0)); HIP_TRY(hipEventCreate(&e1)); HIP_TRY(hipEventRecord(e0, stream)); }
    hipStream_t ls = stream; // the stream of the chunk being enqueued
    auto timed = [&](int kind, auto &&launch) { // HIP events around ONE kernel on its launch stream
        hipEvent_t k0 = nullptr, k1 = nullptr;
        if (a.profiling) { HIP_TRY(hipEventCreate(&k0)); HIP_TRY(hipEventCreate(&k1)); HIP_TRY(hipEventRecord(k0, ls)); }
        HIP_TRY(launch());
        if (a.profiling) { HIP_TRY(hipEventRecord(k1, ls)); a.kind_events[kind].emplace_back(k0, k1); }
    };
    if (std::getenv("LASGUN_DEBUG"))
        std::fprintf(stderr, "[lasgun] wavefront: levels %u, %llu tiles in chunks of %llu on %u stream(s) (%.1f MiB per context), trace grid %u x %u, stack %u, max_blocks %u\n", levels,
                     (unsigned long long)P0.ntiles, chunk_tiles, nstreams ? nstreams : 1u, need / 1048576.0, trace_cap, ldss ? 1024u : 256u, depth, a.max_blocks);
    // A SMALL frame's chain is a dozen dependent launches of a few microseconds each (Cornell glass 512^2: 16 launches for 0.4 ms), and
    // what separates them on a stream is the runtime's launch path per kernel.  The chain has no host decision in it -- fixed grids, counts
    // on the device -- so it is captured ONCE into a HIP graph and replayed: frames of <= 2^20 work items, one chunk, the caller's own
    // stream (not the null stream), not profiling; captured when the same chain (a hash of every parameter: scene tables, camera, film
    // pointer, carved arrays, grids) comes a second time in a row on the context, so a program that renders one frame never pays
    // for a capture, and re-captured at most MAX_GRAPH_CAPTURES times per context (a caller that changes the film every frame gains nothing
    // and stops paying).  The bytes are the same launches' bytes.
    // MEASURED, and OFF unless LASGUN_GRAPH=1 (profiles/r06_small_frames.jsonl, tools/ab_small_frames.sh, variants in turn on one box): the
    // replay is SLOWER where it was meant to pay -- the README sphere at 512^2 0.086 -> 0.093 ms alone and 0.072 -> 0.081 back to back (4 nodes),
    // Cornell plastic 0.119 -> 0.129 / 0.104 -> 0.116 -- and within +-2 % on every longer chain (simple.rs 9 spp, Cornell glass at 256^2 / 512^2,
    // spooky.rs, playground.rs, simplecows.rs: 16 - 28 nodes).  On this runtime (ROCm 7.2) a graph launch costs more than the stream launches it
    // replaces and the gaps between dependent kernels do not shrink.
    static const bool graphs_on = [] { const char *e = std::getenv("LASGUN_GRAPH"); return e && e[0] == '1'; }();
    constexpr unsigned MAX_GRAPH_CAPTURES = 8;
    bool capturing = false;
    if (graphs_on && stream != nullptr && nstreams == 0 && nchunks == 1 && !a.profiling && (unsigned long long)P0.ntiles * 64ull * S <= (1ull << 20)) {
        hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
        if (hipStreamIsCapturing(stream, &cs) != hipSuccess) { (void)hipGetLastError(); cs = hipStreamCaptureStatusActive; }
        if (cs == hipStreamCaptureStatusNone) {
            uint64_t sig = 1469598103934665603ull;
            auto mix = [&sig](const void *q, size_t n) { const uint8_t *b = (const uint8_t *)q; for (size_t i = 0; i < n; ++i) { sig ^= b[i]; sig *= 1099511628211ull; } };
            mix(&P0, sizeof P0);
            const Carved &K0 = carved[0];
            mix(&K0.hq, sizeof K0.hq); mix(&K0.counters, sizeof K0.counters); mix(&K0.frame, sizeof K0.frame); mix(&K0.accum, sizeof K0.accum);
            const uint64_t shape[8] = {levels, S, n0, chunk_tiles, trace_cap, depth, (uint64_t)a.fast | ((uint64_t)ldss << 1), (uint64_t)(uintptr_t)a.lds_image.p};
            mix(shape, sizeof shape);
            if (sig == 0) sig = 1;
            if (c.wf_graph && c.wf_graph_sig == sig) {
                const hipError_t ge = hipGraphLaunch(c.wf_graph, stream);
                if (ge == hipSuccess) return;
                (void)hipGetLastError(); // a replay that is refused: the plain chain below
                (void)hipGraphExecDestroy(c.wf_graph); c.wf_graph = nullptr; c.wf_graph_sig = 0; c.wf_graph_captures = MAX_GRAPH_CAPTURES;
            } else if (c.wf_last_sig == sig && c.wf_graph_captures < MAX_GRAPH_CAPTURES) {
                if (hipStreamBeginCapture(stream, hipStreamCaptureModeThreadLocal) == hipSuccess) { capturing = true; c.wf_graph_captures++; }
                else (void)hipGetLastError();
            }
            c.wf_last_sig = sig;
        }
    }
    const uint64_t chain_sig = c.wf_last_sig;
    auto run_chain = [&] {
    unsigned long long chunk_no = 0;
    for (unsigned long long t0 = 0; t0 < P0.ntiles; t0 += chunk_tiles, ++chunk_no) {
        const Carved &K = carved[chunk_no % carved.size()];
        ls = lanes[chunk_no % lanes.size()];
        const std::vector<double *> &q = K.q, &out = K.out, &spec = K.spec;
        const std::vector<uint32_t *> &child = K.child;
        uint32_t *const hq = K.hq, *const vis = K.vis;
        double *const frame = K.frame, *const accum = K.accum;
        DParams P = P0;
        P.tile0 = (uint32_t)t0;
        const uint32_t pixel_tiles = (uint32_t)std::min<unsigned long long>(chunk_tiles, P0.ntiles - t0);
        P.ntiles = pixel_tiles * S; // level 0's work tiles
        P.ss_par = S;
        P.n_items = n0; // stride of the sample accumulator
        P.accum = accum;
        P.wf_levels = levels;
        P.wf_counts = K.counters;
        P.wf_hit_cap = hit_cap; P.wf_hit_stride = hit_len; P.wf_hq = hq; P.frame = frame; P.vis = vis;
#ifdef LG_STAMPS
        P.stats = a.stats.p;
        P.stamp_counts = reinterpret_cast<unsigned long long *>(a.stats.p + 1);
#endif
        if (ldss) {
            P.lds_image = a.lds_image.p; P.lds_image_n16 = a.lds_image_n16;
            P.lds_node_off = a.lds_node_off; P.lds_prim_off = a.lds_prim_off; P.lds_soup_off = a.lds_soup_off; P.lds_accel_off = a.lds_accel_off;
        }
        const uint32_t tiles_needed = (P.ntiles + 3u) / 4u;
        const uint32_t trace_blocks0 = ldss ? trace_cap : std::min(trace_cap, tiles_needed);
        const uint32_t flat_blocks0 = (uint32_t)(((unsigned long long)P.ntiles * 64ull + 255ull) / 256ull); // level 0: one thread per pixel
        // level-0 shade: one wave per dense tile and per tile the appended hits can fill (< WF_FULL_MIN of every 64 rays)
        const uint32_t shade_blocks0 = (uint32_t)(((unsigned long long)P.ntiles + ((unsigned long long)P.ntiles * (WF_FULL_MIN_HOST - 1) + 63ull) / 64ull + 3ull) / 4ull);
        for (uint32_t sidx = 0; sidx < nsamples / S; ++sidx) {
            P.sample_index = sidx;
            HIP_TRY(hipMemsetAsync(K.counters, 0, CL * (1 + nlaunch) * sizeof(uint32_t), ls));
            uint32_t launch_no = 0;
            auto level_params = [&](uint32_t d) {
                P.wf_level = d;
                P.wf_cap = (unsigned long long)n0 << d; P.wf_cap_next = (unsigned long long)n0 << (d + 1);
                P.wf_q = q[d]; P.wf_out = out[d]; P.wf_spec = spec[d]; P.wf_child = child[d];
                P.wf_q_next = d + 1 < levels ? q[d + 1] : nullptr;
                P.wf_out_next = d + 1 < levels ? out[d + 1] : nullptr;
                P.tile_counter = K.counters + CL * (1 + launch_no++);
            };
            for (uint32_t d = 0; d < levels; ++d) {
                // (deeper levels: the number of rays is only known on the device; grids are sized for a full level 0, which
                // every deeper level may exceed only in waves, never in work per wave)
                const uint32_t tb = d == 0 ? trace_blocks0 : trace_cap, fb = d == 0 ? shade_blocks0 : flat_cap;
                level_params(d);
                timed(0, [&] { return launch_wf_trace(P, a.fast, false, tb, depth, ls); });
                if (P.nlights > 0) {
                    level_params(d);
                    timed(2, [&] { return launch_wf_trace(P, a.fast, true, tb, depth, ls); });
                }
                level_params(d);
                timed(3, [&] { return launch_wf_shade(P, fb, ls); });
            }
            for (uint32_t d = levels - 1; d-- > 0;) {
                level_params(d);
                timed(1, [&] { return launch_wf_combine(P, d == 0 ? flat_blocks0 : flat_cap, ls); });
            }
        }
        if (S > 1) {
            DParams R = P;
            R.ntiles = pixel_tiles;
            timed(1, [&] { return launch_wf_resolve(R, (uint32_t)(((unsigned long long)pixel_tiles * 64ull + 255ull) / 256ull), ls); });
        }
    }
    };
    if (capturing) { // record the chain, replay it; a capture that was begun is always ended (a stream left in capture mode is lost to its owner)
        hipGraph_t g = nullptr;
        try { run_chain(); } catch (...) { (void)hipStreamEndCapture(stream, &g); if (g) (void)hipGraphDestroy(g); (void)hipGetLastError(); throw; }
        bool launched = false;
        if (hipStreamEndCapture(stream, &g) == hipSuccess && g) {
            if (c.wf_graph) { (void)hipGraphExecDestroy(c.wf_graph); c.wf_graph = nullptr; c.wf_graph_sig = 0; }
            hipGraphExec_t x = nullptr;
            if (hipGraphInstantiate(&x, g, nullptr, nullptr, 0) == hipSuccess && x) {
                if (hipGraphLaunch(x, stream) == hipSuccess) { c.wf_graph = x; c.wf_graph_sig = chain_sig; launched = true; }
                else (void)hipGraphExecDestroy(x);
            }
        }
        if (g) (void)hipGraphDestroy(g);
        (void)hipGetLastError();
        if (launched) return; // (no bands, no profiling on this path)
        c.wf_graph_captures = MAX_GRAPH_CAPTURES; // captured but not launched: the frame still has to be rendered, plainly, and this context stops trying
    }
    run_chain();
    for (unsigned j = 0; j < nstreams; ++j) { // join: the caller's stream continues when every band is done
        HIP_TRY(hipEventRecord(a.aux_done[j], a.aux_streams[j]));
        HIP_TRY(hipStreamWaitEvent(stream, a.aux_done[j], 0));
    }
    if (a.profiling) { HIP_TRY(hipEventRecord(e1, stream)); a.events.emplace_back(e0, e1); }
}

// The queue organisation (k_queue.hip): per chunk of the film and per supersample ONE persistent launch that runs every recursion
// level -- its waves pull 64-ray packets from per-level queues, deepest level first -- then the combine passes bottom-up, shared
// with the level-by-level pipeline.  Queue capacities are worst case (level d: 2^d rays per pixel of the chunk), so nothing can
// overflow; a recursive scene may take a large share of the HBM for it (a 4096^2 frame at recursion 3: 26 GB of 288) and keeps ONE
// chunk in flight per launch context.
static void enqueue_queue(const lg_accel &a, DParams &P0, lg_accel::LaunchCtx &c, bool split, hipStream_t stream) {
    const uint32_t levels = (a.flat.has_specular && P0.recursion > 0) ? P0.recursion + 1u : 1u;
    const uint32_t nsamples = P0.ss_root * P0.ss_root;
    static const bool ss_serial = [] { const char *e = std::getenv("LASGUN_SS_SERIAL"); return e && e[0] == '1'; }();
    // level 0's tiles in parts (enqueue_mega, DParams::split_shift): the children's packets are then as narrow as their parents -- a small
    // launch's recursion chains are walked by four times the waves, 16 lanes each
    const uint32_t parts = a.tile_parts >= 1 ? (uint32_t)a.tile_parts : split ? MEGA_SPLIT : 1u, split_shift = parts == 8u ? 3u : parts == 4u ? 2u : parts == 2u ? 1u : 0u;
    const uint32_t S = (nsamples > 1 && !ss_serial && a.sample_order != 1 ? nsamples : 1u) * parts; // samples side by side (enqueue_wavefront) x parts: level-0 tiles per pixel tile
    auto level_bytes = [&](uint32_t d) -> size_t { // per ray of level d
        size_t b = 0;
        if (d >= 1) b += 6 * 8;                       // ray queue
        if (levels > 1) b += 3 * 8;                   // output / li
        if (d + 1 < levels) b += 8 * 8 + 2 * 4;       // children's weights and indices
        return b;
    };
    size_t per_item = (nsamples > 1 ? 3 * 8 : 0) + 1;
    for (uint32_t d = 0; d < levels; ++d) per_item += level_bytes(d) << d;
    const size_t per_pixel = per_item * S;
    if (a.queue_budget == 0) {
        size_t free_b = 0, total_b = 0;
        HIP_TRY(hipMemGetInfo(&free_b, &total_b));
        // a recursive scene: up to a quarter of the free memory (at most 48 GiB) per launch context, so that a 4096^2 frame is one
        // launch; others need a few bytes per pixel and take the wavefront pipeline's share (an allocation that fails halves the chunk)
        size_t budget = levels > 1 ? free_b / 4 : free_b / 16;
        const size_t cap = levels > 1 ? (48ull << 30) : (8ull << 30);
        const char *env = std::getenv("LASGUN_QUEUE_BUDGET_MB");
        const bool from_env = env && std::atoll(env) > 0;
        if (from_env) budget = (size_t)std::atoll(env) << 20; // (as given: tests cut small films into many chunks with it)
        else if (budget > cap) budget = cap;
        a.queue_budget = !from_env && budget < (64ull << 20) ? (64ull << 20) : budget;
    }
    unsigned long long chunk_tiles = a.queue_budget / (per_pixel * 64);
    const unsigned long long cap_limit = (0xFFFFFF00ull >> (levels - 1)) / (64ull * S); // ray indices are 32-bit
    if (chunk_tiles > cap_limit) chunk_tiles = cap_limit;
    if (chunk_tiles < 1) chunk_tiles = 1;
    if (chunk_tiles > P0.ntiles) chunk_tiles = P0.ntiles;
    if (chunk_tiles < P0.ntiles && P0.mode == 0u && P0.tiles_x != 0u && chunk_tiles >= (unsigned long long)P0.tiles_x * 32ull)
        chunk_tiles -= chunk_tiles % ((unsigned long long)P0.tiles_x * 32ull); // whole rows of 32 x 32-tile blocks: the block order applies to every chunk
    // level 0's work items: units of consecutive 8x8 tiles whose specular children the wave compacts into packets of its own.  Default
    // 1 (measured, config 4 / 4m in ms: 1 tile 39.0 / 16.4, 2: 40.0 / 17.0, 4: 41.0 / 18.5, 8: 44.6 / 22.2, 16: 51.7 / 31.6 -- a mesh tile is a
    // millisecond of work, so longer units lengthen the launch's tail by more than fuller packets save); LASGUN_QUEUE_UNIT: A/B
    static const uint32_t unit_tiles = [] { const char *e = std::getenv("LASGUN_QUEUE_UNIT"); const int v = e ? std::atoi(e) : 0; return v >= 1 && v <= 64 ? (uint32_t)v : 1u; }();
    // LASGUN_QUEUE_ORDER=1 (A/B): 32 x 32-tile blocks in Morton order, claimed XCD by XCD -- measured no better than row order with one
    // claim counter (config 4 / 4m / 5: 39.4 / 16.7 / 63.7 against 38.4 / 16.0 / 64.8 ms): which tiles are in flight together does not
    // move these kernels, as round 3 found for the other organisations
    static const bool order_blocks = [] { const char *e = std::getenv("LASGUN_QUEUE_ORDER"); return e && e[0] == '1'; }();
    const bool ldss = a.lds_scene && a.ldss_blocks;
    const uint32_t blocks_cap = ldss ? a.ldss_blocks : a.queue_blocks;
    const unsigned long long threads = (unsigned long long)blocks_cap * (ldss ? 1024ull : 256ull);
    unsigned long long n0 = 0;
    size_t need = 0, nready = 0;
    struct Carved {
        std::vector<double *> q, out, spec;
        std::vector<uint32_t *> child;
        double *accum = nullptr;
    } K;
    for (;;) { // memory that is not there: halve the chunk and carve again
        try {
            n0 = chunk_tiles * 64ull * S;
            need = (size_t)n0 * per_item + 4096 * (4 * levels + 4);
            nready = (size_t)chunk_tiles * S * ((1ull << levels) - 2ull) + (size_t)levels * QR_SLACK; // one word per packet of the levels >= 1, + slack per level
            if (c.wf_mem.n < need) { HIP_TRY(hipDeviceSynchronize()); c.wf_mem.alloc(need); }
            if (c.wf_counters.n < QC_WORDS + nready) { HIP_TRY(hipDeviceSynchronize()); c.wf_counters.alloc(QC_WORDS + nready); }
            if (P0.nlights > 0 && c.stash.n < (size_t)threads * STASH_DOUBLES) { HIP_TRY(hipDeviceSynchronize()); c.stash.alloc((size_t)threads * STASH_DOUBLES); }
            break;
        } catch (const Error &e) {
            if (std::string(e.what()).find("hipMalloc") == std::string::npos || chunk_tiles <= 1) throw;
            (void)hipGetLastError();
            chunk_tiles = (chunk_tiles + 1) / 2;
            a.queue_budget = std::max<size_t>(a.queue_budget / 2, 64ull << 20);
            if (std::getenv("LASGUN_DEBUG")) std::fprintf(stderr, "[lasgun] queue: %s -- chunks of %llu tiles instead\n", e.what(), chunk_tiles);
        }
    }
    {
        K.q.assign(levels, nullptr); K.out.assign(levels, nullptr); K.spec.assign(levels, nullptr); K.child.assign(levels, nullptr);
        uint8_t *cur = c.wf_mem.p;
        auto take = [&](size_t bytes) { uint8_t *p = cur; cur += (bytes + 255) & ~(size_t)255; return p; };
        for (uint32_t d = 0; d < levels; ++d) {
            const size_t cap = (size_t)n0 << d;
            if (d >= 1) K.q[d] = (double *)take(cap * 6 * 8);
            if (levels > 1) K.out[d] = (double *)take(cap * 3 * 8);
            if (d + 1 < levels) { K.spec[d] = (double *)take(cap * 8 * 8); K.child[d] = (uint32_t *)take(cap * 2 * 4); }
        }
        K.accum = nsamples > 1 ? (double *)take((size_t)n0 * 3 * 8) : nullptr;
    }
    if (!a.q_err) a.q_err = g_err_words.take();
    hipEvent_t e0 = nullptr, e1 = nullptr;
    if (a.profiling) { HIP_TRY(hipEventCreate(&e0)); HIP_TRY(hipEventCreate(&e1)); HIP_TRY(hipEventRecord(e0, stream)); }
    auto timed = [&](int kind, auto &&launch) { // HIP events around ONE kernel on its launch stream
        hipEvent_t k0 = nullptr, k1 = nullptr;
        if (a.profiling) { HIP_TRY(hipEventCreate(&k0)); HIP_TRY(hipEventCreate(&k1)); HIP_TRY(hipEventRecord(k0, stream)); }
        HIP_TRY(launch());
        if (a.profiling) { HIP_TRY(hipEventRecord(k1, stream)); a.kind_events[kind].emplace_back(k0, k1); }
    };
    if (std::getenv("LASGUN_DEBUG"))
        std::fprintf(stderr, "[lasgun] queue: levels %u, %llu tiles in chunks of %llu (%.1f MiB), grid %u x %u, stack %u\n", levels,
                     (unsigned long long)P0.ntiles, chunk_tiles, need / 1048576.0, blocks_cap, ldss ? 1024u : 256u, a.stack_depth);
    const uint32_t flat_cap = a.cus * 16u;
    for (unsigned long long t0 = 0; t0 < P0.ntiles; t0 += chunk_tiles) {
        DParams P = P0;
        P.tile0 = (uint32_t)t0;
        const uint32_t pixel_tiles = (uint32_t)std::min<unsigned long long>(chunk_tiles, P0.ntiles - t0);
        P.ntiles = pixel_tiles * S; // level 0's tiles
        P.ss_par = S / parts; P.split_shift = split_shift;
        P.n_items = n0; // SoA stride of level 0's arrays and of the sample accumulator
        P.accum = K.accum;
        P.wf_levels = levels;
        P.q_ctl = c.wf_counters.p; P.q_ready = c.wf_counters.p + QC_WORDS; P.q_err = a.q_err;
        P.q_unit_tiles = levels > 1 ? unit_tiles : 1u;
        // the tile sequence: rectangles whose chunk is whole tile rows go block by block, XCD by XCD (k_queue.hip, q_seq_tile)
        P.q_order = (order_blocks && !ldss && S == 1u && P.mode == 0u && P.tiles_x != 0u && t0 % P.tiles_x == 0u && P.ntiles % P.tiles_x == 0u) ? 1u : 0u;
        P.q_tiles_y = P.q_order ? P.ntiles / P.tiles_x : 0u;
        P.q_blocks_x = P.q_order ? (P.tiles_x + 31u) / 32u : 0u;
        P.q_seq_len = P.q_order ? P.q_blocks_x * ((P.q_tiles_y + 31u) / 32u) * 1024u : P.ntiles;
        P.q_units = (P.q_seq_len + P.q_unit_tiles - 1u) / P.q_unit_tiles;
        for (uint32_t d = 0; d < levels; ++d) { P.q_rays[d] = K.q[d]; P.q_out[d] = K.out[d]; P.q_spec[d] = K.spec[d]; P.q_child[d] = K.child[d]; }
        P.stash = c.stash.p; P.frame_threads = threads;
        if (ldss) {
            P.lds_image = a.lds_image.p; P.lds_image_n16 = a.lds_image_n16;
            P.lds_node_off = a.lds_node_off; P.lds_prim_off = a.lds_prim_off; P.lds_soup_off = a.lds_soup_off; P.lds_accel_off = a.lds_accel_off;
        }
        const uint32_t blocks = ldss ? blocks_cap : std::min(blocks_cap, (P.q_units + 3u) / 4u);
        const size_t nready_now = nready;
        for (uint32_t sidx = 0; sidx < nsamples / (S / parts); ++sidx) {
            P.sample_index = sidx;
            HIP_TRY(hipMemsetAsync(c.wf_counters.p, 0, (QC_WORDS + (levels > 1 ? nready_now : 0)) * sizeof(uint32_t), stream));
            timed(4, [&] { return launch_queue(P, blocks, stream); });
            for (uint32_t d = levels - 1; d-- > 0;) { // bottom-up: li of level d's rays from their children's (integrate.rs:79, 103, 129)
                P.wf_level = d;
                P.wf_cap = (unsigned long long)n0 << d; P.wf_cap_next = (unsigned long long)n0 << (d + 1);
                P.wf_out = K.out[d]; P.wf_spec = K.spec[d]; P.wf_child = K.child[d]; P.wf_out_next = K.out[d + 1];
                const uint32_t flat_blocks0 = (uint32_t)(((unsigned long long)P.ntiles * 64ull + 255ull) / 256ull);
                timed(1, [&] { return launch_wf_combine(P, d == 0 ? flat_blocks0 : flat_cap, stream); });
            }
        }
        if (S / parts > 1) { // a pixel's samples summed in their order (k_wavefront.hip, wf_resolve_kernel)
            DParams R = P;
            R.ntiles = pixel_tiles;
            timed(1, [&] { return launch_wf_resolve(R, (uint32_t)(((unsigned long long)pixel_tiles * 64ull + 255ull) / 256ull), stream); });
        }
    }
    if (a.profiling) { HIP_TRY(hipEventRecord(e1, stream)); a.events.emplace_back(e0, e1); }
}

// ---- which organisation renders a launch -----------------------------------------------------------------
enum Org : int { ORG_MEGA = 0, ORG_WAVEFRONT = 1, ORG_QUEUE = 2 };
static uint32_t levels_of(const lg_accel &a, const DParams &P) { return (a.flat.has_specular && P.recursion > 0) ? P.recursion + 1u : 1u; }
// what each organisation can take: the queue organisation the reference traversal with <= 32 lights and <= 8 recursion levels, the
// level-by-level pipeline any scene with <= 32 lights; neither the counting variant
static bool org_possible(const lg_accel &a, const DParams &P, bool stats, Org org) {
    if (org == ORG_QUEUE) return !stats && !a.fast && P.nlights <= 32 && levels_of(a, P) <= QC_MAX_LEVELS;
    if (org == ORG_WAVEFRONT) return !stats && P.nlights <= 32 && P.recursion < 20;
    return true;
}
// The FITTED rule of rounds 2-4 (primitive count, glass / mirror, pixels per launch, samples per pixel): what a launch gets when
// nothing has been measured for its kind -- LASGUN_AUTOTUNE=0, or the first candidate the measurement below starts from.
//   * queue organisation: glass / mirror over a big mesh (long uneven walks, sparse deep levels), launches of 2^16 pixels and more;
//   * level by level, for a scene resident in LDS (round 4, once a launch no longer ended in 75-90 us of failed tile claims): glass /
//     mirror frames up to 2^20 pixels (Cornell glass 0.41 against 0.81 ms at 512^2, 0.99 / 1.18 at 1024^2, 1.81 / 1.64 at 1536^2), and
//     frames from 2^18 pixels of few primitives at one sample per pixel (README sphere 2.2 / 3.5 ms at 4096^2); and wherever node and
//     sphere tests dominate (>= 512 spheres / boxes) from 2^21 pixels;
//   * the megakernel otherwise (supersampled frames of small scenes: its 768-lane form is ahead at every size).
static Org org_by_rule(const lg_accel &a, const DParams &P, bool stats) {
    const unsigned long long items = (unsigned long long)P.ntiles * 64ull;
    if (org_possible(a, P, stats, ORG_QUEUE) && a.streaming && a.queue_default && items >= a.queue_min_items) return ORG_QUEUE;
    const bool lds_resident = !a.fast && a.lds_scene && a.ldss_blocks, specular = a.flat.has_specular && P.recursion > 0;
    // (level 0's work items: with a pixel's samples side by side -- round 5 -- a 9-sample frame is nine times as wide as its film;
    // profiles/r05_ss_par.jsonl: Cornell glass at 9 spp goes level by level at 256^2 and in the megakernel from 512^2, like its
    // one-sample frames of nine times the pixels; simple.rs at 16 spp level by level at every size)
    const unsigned long long work = items * (a.sample_order != 1 ? (unsigned long long)P.ss_root * P.ss_root : 1ull);
    const bool small_specular = lds_resident && specular && work <= a.specular_small_items;
    const bool light_scene = lds_resident && !specular && !a.streaming_pays && (P.ss_root == 1u || a.sample_order != 1) && work >= (1ull << 18);
    if (a.streaming && org_possible(a, P, stats, ORG_WAVEFRONT) && (small_specular || light_scene || (a.streaming_pays && work >= a.streaming_min_items)))
        return ORG_WAVEFRONT;
    return ORG_MEGA;
}

// the megakernel (k_mega.hip): the whole of li() per lane
// The megakernel with a pixel's samples SIDE BY SIDE (DParams::ss_par, enqueue_wavefront): the launch hands out (tile, sample) pairs,
// so that a wave's share is 1 / samples of what it was and the launch's tail with it; the samples are parked (24 bytes each) and summed
// in their order by the resolve pass.  Measured (tools/ss_probe.py, profiles/r05_ss_par.jsonl): 4- and 9-sample frames of 256^2 .. 1024^2
// 1.2 - 10 x faster (a 512^2 film is one tile per wave of the grid: nine samples in a row on each, or nine times the tiles); frames of
// 1024^2 and more of a cheap scene 30-50 % SLOWER (nine times the claims on one head word, 8 ns each).  So: possible while the parked
// samples fit 1 GiB, the rule below where nothing is measured, and one more thing the measured choice times.
constexpr size_t PRUNE_MIN_TRIS = 4096; // the pruned walk (and its leaf records) by default: from this many triangles in one mesh
constexpr unsigned long long SS_PAR_WAVES = 8; // the rule: side by side below this many pixel tiles per wave of the grid (9 of 12 scenes faster at 1024^2, none at 2048^2)
static bool mega_par_possible(const DParams &P, bool stats) {
    const unsigned long long nsamples = (unsigned long long)P.ss_root * P.ss_root;
    return nsamples > 1 && !stats && (unsigned long long)P.ntiles * 64ull * nsamples * 24ull <= (1ull << 30);
}
static bool mega_par_by_rule(const lg_accel &a, const DParams &P, bool stats) {
    static const int ss_mega = [] { const char *e = std::getenv("LASGUN_SS_MEGA"); return e ? std::atoi(e) : -1; }(); // A/B: 0 never, 1 always
    if (!mega_par_possible(P, stats) || a.sample_order == 1) return false;
    if (a.sample_order == 0) return true;
    const bool lds_form = !a.fast && a.lds_scene && a.ldss_blocks;
    const unsigned long long grid_waves = lds_form ? (unsigned long long)a.ldss_blocks * (a.mega_narrow ? 12u : 16u) : (unsigned long long)(a.fast ? a.max_blocks_fast : a.max_blocks) * 4ull;
    return ss_mega >= 0 ? ss_mega == 1 : P.ntiles < SS_PAR_WAVES * grid_waves;
}
// A SMALL launch may hand its tiles out in QUARTERS (DParams::split_shift: 16 lanes of a tile per wave, four times the waves at work): a frame of
// fewer tiles than the grid has waves is as slow as its slowest tile's recursion tree, and a quarter of a tile is a shorter tree walked by
// fewer diverging lanes.  Measured (tools/split_probe.py, profiles/r05_ab_split.jsonl): the kitchen sink at 512^2 2.40 -> 1.73 ms, the
// 100k-triangle metal torus at 256^2 2.23 -> 1.82; cheap scenes and frames of 1024^2 and more lose (idle lanes are then lost throughput).
// One more candidate of the measured choice; never by rule.
static bool mega_split_possible(const lg_accel &a, const DParams &P, bool stats) {
    const bool lds_form = !a.fast && a.lds_scene && a.ldss_blocks;
    const unsigned long long grid_waves = lds_form ? (unsigned long long)a.ldss_blocks * (a.mega_narrow ? 12u : 16u) : (unsigned long long)(a.fast ? a.max_blocks_fast : a.max_blocks) * 4ull;
    const unsigned long long work = (unsigned long long)P.ntiles * (mega_par_by_rule(a, P, stats) ? (unsigned long long)P.ss_root * P.ss_root : 1ull);
    return !stats && P.ntiles >= 2u && work <= 4ull * grid_waves;
}
static void enqueue_mega(const lg_accel &a, DParams &P, lg_accel::LaunchCtx &c, bool par, bool split, bool stats, hipStream_t stream) {
    const uint32_t nsamples = P.ss_root * P.ss_root;
    par = par && mega_par_possible(P, stats);
    if (par) {
        const size_t n_items = (size_t)P.ntiles * 64ull * nsamples, need = n_items * 3 * 8;
        if (c.wf_mem.n < need) { HIP_TRY(hipDeviceSynchronize()); c.wf_mem.alloc(need); }
        P.accum = reinterpret_cast<double *>(c.wf_mem.p); P.n_items = n_items; P.ss_par = nsamples; P.ntiles *= nsamples;
    }
    { // tiles handed out in parts: the measured choice's candidate, or LASGUN_MEGA_SPLIT=2|4|8 (A/B)
        static const uint32_t split_env = [] { const char *e = std::getenv("LASGUN_MEGA_SPLIT"); const int v = e ? std::atoi(e) : 0; return v == 2 || v == 4 || v == 8 ? (uint32_t)v : 1u; }();
        const uint32_t parts = a.tile_parts >= 1 ? (uint32_t)a.tile_parts : split_env > 1u ? split_env : split ? MEGA_SPLIT : 1u;
        if (parts > 1u && !stats && (unsigned long long)P.ntiles * parts < (1ull << 31)) { P.split_shift = parts == 2u ? 1u : parts == 4u ? 2u : 3u; P.ntiles *= parts; }
    }
    uint32_t cap = a.fast ? a.max_blocks_fast : a.max_blocks;
    uint32_t blocks = (P.ntiles + 3u) / 4u;
    if (blocks > cap) blocks = cap;
    uint32_t maxb = a.max_blocks > a.max_blocks_fast ? a.max_blocks : a.max_blocks_fast;
    if (!stats && !a.fast && a.lds_scene && a.ldss_blocks) { // scene tables resident in LDS: one 1024-lane workgroup per CU
        P.lds_image = a.lds_image.p; P.lds_image_n16 = a.lds_image_n16;
        P.lds_node_off = a.lds_node_off; P.lds_prim_off = a.lds_prim_off; P.lds_soup_off = a.lds_soup_off; P.lds_accel_off = a.lds_accel_off;
        P.mega_lanes = a.mega_narrow ? 768u : 1024u; // (k_mega.hip: three waves per SIMD and 168 registers where shading weighs more than walking)
        blocks = a.ldss_blocks;
    }
    if (maxb < a.ldss_blocks * 4u) maxb = a.ldss_blocks * 4u; // per-lane slots below: 1024 lanes per LDS-scene workgroup
    // Whitted frames: one slot per resident lane and recursion level, only for glass / mirror scenes
    if (a.flat.has_specular && P.recursion > 0) {
        unsigned long long threads = (unsigned long long)maxb * 256ull;
        size_t need = (size_t)threads * P.recursion * FRAME_DOUBLES;
        if (c.frames.n < need) {
            HIP_TRY(hipDeviceSynchronize()); // (re)allocation: nothing may still use the old buffer
            c.frames.alloc(need);
        }
        P.frames = c.frames.p;
        P.frame_threads = threads;
    }
    if (P.nlights > 0) { // shading frame parked across the shadow traversals
        unsigned long long threads = (unsigned long long)maxb * 256ull;
        size_t need = (size_t)threads * STASH_DOUBLES;
        if (c.stash.n < need) {
            HIP_TRY(hipDeviceSynchronize()); // (re)allocation: nothing may still use the old buffer
            c.stash.alloc(need);
        }
        P.stash = c.stash.p;
        P.frame_threads = threads;
    }
    if (stats) {
        P.stats = a.stats.p;
        HIP_TRY(hipMemsetAsync(a.stats.p, 0, sizeof(DStats), stream));
    }
    HIP_TRY(hipMemsetAsync(c.tile_counter.p, 0, TILE_COUNTER_WORDS * sizeof(uint32_t), stream));
    hipEvent_t e0 = nullptr, e1 = nullptr;
    if (a.profiling) {
        HIP_TRY(hipEventCreate(&e0)); HIP_TRY(hipEventCreate(&e1));
        HIP_TRY(hipEventRecord(e0, stream));
    }
#ifdef LG_QIDLE // diagnostic build: the waves' start / exit times of this launch (k_mega.hip), read by lg_debug_stats
    if (!stats) { P.stats = a.stats.p; HIP_TRY(hipMemsetAsync(a.stats.p, 0, sizeof(DStats), stream)); }
#endif
    HIP_TRY(launch_trace(P, stats, a.fast, blocks, a.fast ? a.stack_depth_fast1 : a.stack_depth, stream));
    if (par) {
        DParams R = P;
        R.ntiles = (P.ntiles >> P.split_shift) / nsamples; R.tile_rev = 0u; R.split_shift = 0u;
        HIP_TRY(launch_wf_resolve(R, (uint32_t)(((unsigned long long)R.ntiles * 64ull + 255ull) / 256ull), stream));
    }
    if (a.profiling) {
        HIP_TRY(hipEventRecord(e1, stream));
        a.events.emplace_back(e0, e1);
    }
}
static void enqueue_org(const lg_accel &a, DParams P, lg_accel::LaunchCtx &c, Org org, int dir, bool ss_serial, bool split, bool stats, hipStream_t stream) { // (P by value: an organisation fills in its own fields)
    P.tile_counter = c.tile_counter.p;
    P.tile_rev = org != ORG_WAVEFRONT ? (uint32_t)dir : 0u; // 0 top-down, 1 bottom-up, 2 from the middle outwards (the level-by-level passes are short and alike: one direction)
    if (org == ORG_QUEUE) enqueue_queue(a, P, c, split, stream);
    else if (org == ORG_WAVEFRONT) enqueue_wavefront(a, P, c, stream);
    else enqueue_mega(a, P, c, !ss_serial, split, stats, stream);
}

// The MEASURED choice (round 5; the rule above was a fit to eight scenes and wrong by 6-22 % on the first scene that was not among
// them; round 6: the table and the race live in tune.cpp, this file supplies the kind, the candidates and how one is launched).
// Every organisation renders the same bytes, so which one runs is a question of time alone, and the answer is taken from the clock: the
// second API call that launches a KIND in the process (the first gets the rule's choice at no cost) -- the scene's shape (table sizes,
// materials, lights, recursion, samples per pixel, traversal mode, LDS residency), the device, the launch's size class (log2 of its pixels)
// and addressing mode -- renders the launch with every CANDIDATE that can take it (a warm-up pass, then three timed passes over the
// candidates in turn, HIP events on the caller's stream, the HOST WAITING -- the one place where a *_device entry point blocks; never on a
// stream that is being captured), keeps the fastest (the rule's own choice unless another beats it by 1 %) and remembers it for
// the process: capture() rebuilds its accel for every frame (lib.rs:64), so the memory is keyed by the scene's shape, not by the accel.
// A candidate that cannot run (no memory for its buffers) drops out of the race instead of failing the caller's render.
// A candidate is an organisation and, for the megakernel and the queue organisation, the DIRECTION the launch's tiles are claimed in:
// a launch ends with the recursion trees of its last tiles, and whether the film's top or its bottom should come last is the scene's
// and the camera's business -- simple.rs at 9 spp and the metal torus gain 6-9 % from the bottom up, the glass torus loses 3 %
// (profiles/r05_ab_tile_order.jsonl); which tile is rendered when never changes a pixel.  (The kind does not know the camera: a
// direction measured for one view is kept for the next.)  The launch itself is then enqueued as usual; what the measurement rendered
// into the caller's film are the same pixels.  Overridden by lg_accel_set_streaming(0 / 2 / 3) and lg_accel_set_tile_order
// (lg_accel_last_organisation says what a launch ran as); LASGUN_AUTOTUNE=0 keeps the rule and the middle-out direction;
// lg_tune_export / lg_tune_import / lg_tune_clear read, pin and forget choices.
namespace {
using TuneKey = lg::tune::Key;
constexpr int TUNE_REV = 16;    // a remembered choice: organisation | TUNE_REV when the tiles go bottom-up
constexpr int TUNE_MID = 64;    //   | TUNE_MID when they go from the middle row outwards
static int dir_bits(int dir) { return dir == 1 ? TUNE_REV : dir == 2 ? TUNE_MID : 0; }
static int dir_of(int choice) { return (choice & TUNE_REV) ? 1 : (choice & TUNE_MID) ? 2 : 0; }
// The direction a launch's tiles are claimed in when nothing is forced or measured: from the middle row outwards.  What a frame shows
// tends to sit in its middle, and a launch should END on cheap tiles: config 4 in the megakernel 36.2 -> 32.8 ms, 4m 13.1 -> 12.7,
// simple.rs 0.55 -> 0.53, nothing slower among the configs (profiles/r05_ab_tile_middle.jsonl).
constexpr int DIR_DEFAULT = 2;
static int dir_unmeasured(const lg_accel &a, Org org) { return org == ORG_WAVEFRONT ? 0 : a.tile_order >= 0 ? a.tile_order : DIR_DEFAULT; }
constexpr int TUNE_SPLIT = 128; //   | TUNE_SPLIT when the megakernel hands a small launch's tiles out in quarters (enqueue_mega)
constexpr int TUNE_SERIAL = 32; //   | TUNE_SERIAL when the megakernel takes a pixel's samples one after the other (enqueue_mega)
// LASGUN_AUTOTUNE (tune.cpp: mode()): 0 = never measure (the fitted rule), 1 (default) = measure a kind at the second API CALL that
// launches it, 2 = at the first.  A program that renders one frame and exits (every example of the reference) gets the rule's choice at no
// cost -- timing seven candidates three times over costs 30-50 frames' worth; whatever renders a kind twice (an animation, the progressive
// front end's hundred subsets, a benchmark) is measured from then on.  What counts is the CALL, not the launch: one lg_capture of a big
// film launches its kind four times (row bands), lg_multi_* once per share of a device (round 5 counted launches and measured inside
// the first frame: ADVICE r5).
bool autotune_enabled() { return lg::tune::mode() != 0; }
// Which API call is running: bumped when a call enters the library from outside (CallScope in guarded(), lg_capture, lg_multi_*:
// calls nested in it, on this thread or on the threads it starts, belong to it).
std::atomic<uint64_t> g_call_serial{1};
std::atomic<int> g_call_depth{0};
struct CallScope {
    CallScope() { if (g_call_depth.fetch_add(1) == 0) g_call_serial.fetch_add(1); }
    ~CallScope() { g_call_depth.fetch_sub(1); }
};
} // namespace
extern "C" void lg_internal_call_scope(int enter) { // (multi.cpp: one lg_multi_capture* is one call, whatever its shares launch)
    if (enter) { if (g_call_depth.fetch_add(1) == 0) g_call_serial.fetch_add(1); }
    else g_call_depth.fetch_sub(1);
}
static TuneKey tune_key(const lg_accel &a, const DParams &P) {
    const FlatScene &f = a.flat;
    const unsigned long long items = (unsigned long long)P.ntiles * 64ull;
    uint64_t cls = 0;
    while ((items >> cls) > 1ull) ++cls;
    TuneKey k{};
    k.v[0] = f.nodes.size(); k.v[1] = f.primref.size(); k.v[2] = f.spheres.size(); k.v[3] = f.cuboids.size(); k.v[4] = f.tri_v.size();
    k.v[5] = f.accels.size(); k.v[6] = ((uint64_t)f.max_stack << 32) | (uint64_t)f.lights.size();
    k.v[7] = ((uint64_t)P.recursion << 32) | ((uint64_t)P.ss_root << 8) | (f.has_specular ? 1u : 0u);
    k.v[8] = ((uint64_t)P.prune << 2) | (a.fast ? 2u : 0u) | (a.lds_scene && a.ldss_blocks ? 1u : 0u);
    {   // what the primitives are made of decides how many rays have children: two scenes of one shape (config 4's glass torus, 4m's metal one) are two kinds
        uint64_t hsh = 1469598103934665603ull;
        auto mix = [&hsh](const void *p, size_t n) { const uint8_t *b = (const uint8_t *)p; for (size_t i = 0; i < n; ++i) { hsh ^= b[i]; hsh *= 1099511628211ull; } };
        for (const DMaterial &m : f.materials) mix(&m.kind, sizeof m.kind);
        for (const DAccel &A : f.accels) { mix(&A.material, sizeof A.material); mix(&A.flags, sizeof A.flags); }
        if (!f.sphere_mat.empty()) mix(f.sphere_mat.data(), f.sphere_mat.size() * sizeof f.sphere_mat[0]);
        if (!f.cuboid_mat.empty()) mix(f.cuboid_mat.data(), f.cuboid_mat.size() * sizeof f.cuboid_mat[0]);
        k.v[9] = hsh ^ ((uint64_t)a.device << 56);
    }
    k.v[10] = cls;
    k.v[11] = (P.mode == 0u ? 0u : 1u) | (a.tile_order >= 0 ? 2u + (uint64_t)a.tile_order : 0u) | ((uint64_t)(a.sample_order + 1) << 4) | ((uint64_t)(a.tile_parts + 1) << 8); // (a forced direction is a kind of its own: only the organisations race)
    return k;
}
// a remembered choice (measured here, or pinned by lg_tune_import for a kind this build may see differently) that the launch cannot take
// falls back to the rule's
static int rule_choice(const lg_accel &a, const DParams &P) {
    const Org rule = org_by_rule(a, P, false);
    return (int)rule | dir_bits(P.ntiles < 2u ? 0 : dir_unmeasured(a, rule)) | (rule == ORG_MEGA && !mega_par_by_rule(a, P, false) ? TUNE_SERIAL : 0);
}
static bool choice_possible(const lg_accel &a, const DParams &P, int choice) {
    const int org = choice & (TUNE_REV - 1);
    if (org < 0 || org > (int)ORG_QUEUE || !org_possible(a, P, false, (Org)org)) return false;
    if ((choice & TUNE_SPLIT) && !mega_split_possible(a, P, false)) return false;
    if (org == (int)ORG_MEGA && !(choice & TUNE_SERIAL) && P.ss_root > 1 && !mega_par_possible(P, false)) return false;
    return true;
}
static int tuned_choice(const lg_accel &a, const DParams &P, lg_accel::LaunchCtx &c, hipStream_t stream) {
    const Org rule = org_by_rule(a, P, false);
    const TuneKey key = tune_key(a, P);
    int known;
    if (lg::tune::lookup(key, &known)) return choice_possible(a, P, known) ? known : rule_choice(a, P);
    if (lg::tune::mode() == 1 && lg::tune::first_call_of_kind(key, g_call_serial.load())) return rule_choice(a, P); // the first call that launches the kind: the rule's choice, at no cost
    {   // a stream that is being captured into a graph cannot be waited on: no race there (the next plain launch of the kind measures)
        hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
        if (hipStreamIsCapturing(stream, &cs) != hipSuccess) (void)hipGetLastError();
        else if (cs != hipStreamCaptureStatusNone) return rule_choice(a, P);
    }
    // candidates: [organisation][samples side by side, one after the other (megakernel only)][top-down, bottom-up, middle-out]
    constexpr int NC = 20, K_SPLIT = 18, K_QSPLIT = 19; // (+ the megakernel / the queue organisation with their tiles in quarters: sample order by the rule, middle-out)
    lg::tune::Candidate cand[NC];
    const unsigned long long items = (unsigned long long)P.ntiles * 64ull;
    for (int k = 0; k < NC; ++k) {
        if (k == K_SPLIT) {
            cand[k].choice = (int)ORG_MEGA | dir_bits(DIR_DEFAULT) | (!mega_par_by_rule(a, P, false) ? TUNE_SERIAL : 0) | TUNE_SPLIT;
            cand[k].in_race = mega_split_possible(a, P, false) && a.tile_parts < 0 && (a.tile_order < 0 || a.tile_order == DIR_DEFAULT);
            continue;
        }
        if (k == K_QSPLIT) {
            cand[k].choice = (int)ORG_QUEUE | dir_bits(DIR_DEFAULT) | TUNE_SPLIT;
            cand[k].in_race = org_possible(a, P, false, ORG_QUEUE) && items >= 4096ull && mega_split_possible(a, P, false) && a.tile_parts < 0 && (a.tile_order < 0 || a.tile_order == DIR_DEFAULT);
            continue;
        }
        const int org = k / 6, ser = (k / 3) & 1, dir = k % 3;
        cand[k].choice = org | dir_bits(dir) | (ser ? TUNE_SERIAL : 0);
        cand[k].in_race = org_possible(a, P, false, (Org)org) &&
                     (ser ? org == ORG_MEGA : (org != ORG_MEGA || mega_par_possible(P, false))) &&
                     !(org == ORG_MEGA && mega_par_possible(P, false) && a.sample_order >= 0 && ser != a.sample_order) && // (lg_accel_set_sample_order) // (one form of the megakernel for a frame of one sample per pixel: the serial one)
                     !(org == ORG_QUEUE && items < 4096ull && rule != ORG_QUEUE) && // (a persistent scheduler for a handful of tiles: never ahead)
                     !(dir != 0 && (org == ORG_WAVEFRONT || P.ntiles < 2u)) &&      // (one direction for the level-by-level passes and for a single tile)
                     (a.tile_order < 0 || org == ORG_WAVEFRONT || dir == a.tile_order); // (lg_accel_set_tile_order: only the organisations race)
    }
    const int rule_k = (int)rule * 6 + (rule == ORG_MEGA && !mega_par_by_rule(a, P, false) ? 3 : 0) + (P.ntiles < 2u ? 0 : dir_unmeasured(a, rule));
    float best_ms[NC];
    for (float &m : best_ms) m = INFINITY;
    const bool was_profiling = a.profiling;
    a.profiling = false; // (the measurement's launches are not the caller's: lg_profile_read must not count them)
    struct Restore { const lg_accel &a; bool was; ~Restore() { a.profiling = was; } } restore{a, was_profiling};
    const int choice = lg::tune::race(key, cand, NC, rule_k, stream, [&](int k) {
        if (k == K_SPLIT) enqueue_org(a, P, c, ORG_MEGA, DIR_DEFAULT, !mega_par_by_rule(a, P, false), true, false, stream);
        else if (k == K_QSPLIT) enqueue_org(a, P, c, ORG_QUEUE, DIR_DEFAULT, false, true, false, stream);
        else enqueue_org(a, P, c, (Org)(k / 6), k % 3, ((k / 3) & 1) != 0, false, false, stream);
    }, best_ms);
    check_queue_error(a);
    if (std::getenv("LASGUN_DEBUG")) {
        std::fprintf(stderr, "[lasgun] measured for %llu pixels (top-down / bottom-up / middle-out): megakernel %.3f / %.3f / %.3f ms (samples in a row: %.3f / %.3f / %.3f), level by level %.3f ms, queue %.3f / %.3f / %.3f ms -> choice %d (rule: %d)\n",
                     items, best_ms[0], best_ms[1], best_ms[2], best_ms[3], best_ms[4], best_ms[5], best_ms[6], best_ms[12], best_ms[13], best_ms[14], // (one sample per pixel: "in a row" is the megakernel)
                     choice, (int)rule);
        if (std::isfinite(best_ms[K_SPLIT]) || std::isfinite(best_ms[K_QSPLIT])) std::fprintf(stderr, "[lasgun]   (tiles in quarters: megakernel %.3f ms, queue %.3f ms)\n", best_ms[K_SPLIT], best_ms[K_QSPLIT]);
    }
    return choice_possible(a, P, choice) ? choice : rule_choice(a, P);
}

// Enqueue one render on `stream`.  Caller holds a.mtx.
static void enqueue(const lg_accel &a, DParams &P, bool stats, hipStream_t stream) {
    if (P.ntiles == 0) return;
    check_queue_error(a); // (an earlier launch on a caller's stream that stalled: reported here at the latest)
    lg_accel::LaunchCtx &c = ctx_for(a, stream);
    Org org;
    int dir = -1; // lg_accel_set_tile_order; -1: from the middle outwards unless measured otherwise (dir_unmeasured)
    bool ss_serial = !mega_par_by_rule(a, P, stats); // the megakernel's samples: by the rule unless measured
    bool split = false;                              // its tiles in quarters: only as measured
    if (stats) { org = ORG_MEGA; dir = 0; }                                                      // the counting variant
    else if (a.queue == 1) org = org_possible(a, P, stats, ORG_QUEUE) ? ORG_QUEUE : org_by_rule(a, P, stats); // lg_accel_set_streaming(3)
    else if (!a.streaming) org = ORG_MEGA;                                                       // lg_accel_set_streaming(0)
    else if (a.streaming_forced) org = org_possible(a, P, stats, ORG_WAVEFRONT) ? ORG_WAVEFRONT : ORG_MEGA; // lg_accel_set_streaming(2)
    else if (a.queue == 0 || !autotune_enabled()) {                                              // the fitted rule (queue ruled out by set_streaming(0 .. 2))
        org = org_by_rule(a, P, stats);
        if (a.queue == 0 && org == ORG_QUEUE) org = ORG_MEGA;
    } else {
        const int choice = tuned_choice(a, P, c, stream);
        org = (Org)(choice & (TUNE_REV - 1));
        dir = dir_of(choice);
        ss_serial = (choice & TUNE_SERIAL) != 0;
        split = (choice & TUNE_SPLIT) != 0;
    }
    if (dir < 0) dir = P.ntiles < 2u ? 0 : dir_unmeasured(a, org);
    if (org == ORG_WAVEFRONT) dir = 0;
    a.last_org = (int)org | dir_bits(dir) | (org == ORG_MEGA && ss_serial && P.ss_root > 1 ? TUNE_SERIAL : 0) | (org != ORG_WAVEFRONT && (split || a.tile_parts > 1) ? TUNE_SPLIT : 0);
    enqueue_org(a, P, c, org, dir, ss_serial, split, stats, stream);
}

static void set_rect(DParams &P, uint32_t x0, uint32_t y0, uint32_t x1, uint32_t y1) {
    P.mode = 0; P.x0 = x0; P.y0 = y0; P.x1 = x1; P.y1 = y1;
    P.tiles_x = (x1 - x0 + 7u) / 8u;
    uint32_t tiles_y = (y1 - y0 + 7u) / 8u;
    P.ntiles = P.tiles_x * tiles_y;
    P.ilv_n = 1; P.ilv_r = 0; P.ilv_b = 1;
    P.out_x0 = 0; P.out_pitch = P.w;
}
// the row table of the lattice addressing for (w, h, n): made once per launch context and kept while the caller stays with that film and period
// (the progressive front end's hundred calls share it)
constexpr size_t MAX_ROW_TABLES = 4;
static const DRowTab *lattice_rows(const lg_accel &a, hipStream_t stream, uint32_t w, uint32_t h, unsigned long long n) {
    lg_accel::LaunchCtx &c = ctx_for(a, stream);
    using RowTable = lg_accel::LaunchCtx::RowTable;
    for (auto &r : c.rowtabs)
        if (r->w == w && r->h == h && r->n == n) { r->last_use = ++c.rowtab_clock; return r->buf.p; }
    RowTable *r = nullptr;
    if (c.rowtabs.size() < MAX_ROW_TABLES) {
        c.rowtabs.emplace_back(new RowTable());
        r = c.rowtabs.back().get();
    } else { // the least recently used table makes room: launches that read it are ahead of the new copy in stream order, unless its buffer must grow
        r = c.rowtabs[0].get();
        for (auto &x : c.rowtabs) if (x->last_use < r->last_use) r = x.get();
        if (r->up) HIP_TRY(hipEventSynchronize(r->up)); // (its staging is rewritten below)
        if (r->buf.n < h) HIP_TRY(hipStreamSynchronize(stream)); // (a buffer goes back to the pool only when nothing can still read it)
    }
    r->w = 0; r->h = 0; r->n = 0; // (not a table of anything until the copy below is enqueued)
    if (r->buf.n < h) r->buf.alloc(h);
    r->stage.need((size_t)h * sizeof(DRowTab));
    DRowTab *t = static_cast<DRowTab *>(r->stage.p);
    for (uint32_t y = 0; y < h; ++y) { const unsigned long long o = (unsigned long long)y * w; t[y] = DRowTab{(uint32_t)(o / n), (uint32_t)(o % n)}; }
    HIP_TRY(hipMemcpyAsync(r->buf.p, t, (size_t)h * sizeof(DRowTab), hipMemcpyHostToDevice, stream));
    if (!r->up) HIP_TRY(hipEventCreateWithFlags(&r->up, hipEventDisableTiming));
    HIP_TRY(hipEventRecord(r->up, stream));
    r->w = w; r->h = h; r->n = n; r->last_use = ++c.rowtab_clock;
    return r->buf.p;
}
// the pixels of the subset {k + i*n} of an `area`-pixel film (k < area, n > 0), without forming area - k + n - 1 (which wraps for n near SIZE_MAX)
static unsigned long long subset_count(unsigned long long area, unsigned long long k, unsigned long long n) { return k < area ? 1ull + (area - 1ull - k) / n : 0ull; }
static void set_subset(const lg_accel &a, hipStream_t stream, DParams &P, size_t k, size_t n, uint32_t w, uint32_t h) {
    unsigned long long area = (unsigned long long)w * h;
    P.mode = 1; P.sub_k = k; P.sub_n = n;
    P.sub_count = subset_count(area, k, n);
    P.ntiles = (uint32_t)((P.sub_count + 63ull) / 64ull);
    // The subset tile by lattice column (mode 4, shade.h: 64 rows x <= n pixels per tile instead of 64 consecutive i) where that is the denser
    // window: a period shorter than the film's width and longer than a tile's 64 pixels in a row would be.  LASGUN_SUBSET_LATTICE=0: never (A/B).
    static const bool lattice = [] { const char *e = std::getenv("LASGUN_SUBSET_LATTICE"); return !(e && e[0] == '0'); }();
    const unsigned long long cols = (w + n - 1) / n, tiles4 = ((unsigned long long)h + 63ull) / 64ull * cols;
    if (lattice && P.sub_count != 0 && n >= 8 && n <= w && h >= 16 && area < (1ull << 32) && tiles4 < (1ull << 31) && tiles4 <= 2ull * P.ntiles + 8ull) {
        P.mode = 4; P.sub_cols = (uint32_t)cols; P.sub_rows = 64u; P.ntiles = (uint32_t)tiles4;
        P.sub_kk = (uint32_t)(k % n); P.sub_kdiv = (uint32_t)(k / n);
        P.sub_rowtab = lattice_rows(a, stream, w, h, n);
    }
}

// Several subsets {k_j + i*n} of one n as ONE render (lg_capture_subsets): the k values sorted and without repeats or empty subsets,
// the periods the longest subset has, and whether the batch is every pixel of the film (every k of 0 .. n-1: the frame itself).
struct SubsetBatch {
    std::vector<unsigned long long> ks;
    unsigned long long n = 1, periods = 0, items = 0;
    bool whole = false;
};
static SubsetBatch make_batch(const size_t *ks, size_t count, size_t n, uint32_t w, uint32_t h) {
    if (n == 0) throw Error("n must be > 0");
    if (count != 0 && !ks) throw Error("ks is NULL");
    const unsigned long long area = (unsigned long long)w * h;
    SubsetBatch b;
    b.n = n;
    for (size_t j = 0; j < count; ++j) if (ks[j] < area) b.ks.push_back(ks[j]); // (a subset that starts behind the film has no pixel: lib.rs:152)
    std::sort(b.ks.begin(), b.ks.end());
    b.ks.erase(std::unique(b.ks.begin(), b.ks.end()), b.ks.end());
    for (unsigned long long k : b.ks) b.periods = std::max(b.periods, subset_count(area, k, n));
    b.items = (unsigned long long)b.ks.size() * b.periods;
    if (b.items >= 0xFFFFFFFFull) throw Error("too many pixels for one batch of subsets (2^32 work items)");
    b.whole = b.ks.size() == n;
    for (size_t j = 0; b.whole && j < b.ks.size(); ++j) b.whole = b.ks[j] == j;
    return b;
}
// the batch as addressing mode 3 on `stream` (its k table lives in the stream's launch context).  Caller holds a.mtx.
static void set_subsets(const lg_accel &a, DParams &P, const SubsetBatch &b, hipStream_t stream) {
    lg_accel::LaunchCtx &c = ctx_for(a, stream);
    for (size_t i = 0; i < c.ks_live.size();) { // tables whose launch is through go back to the pool
        if (c.ks_live[i]->done && hipEventQuery(c.ks_live[i]->done) == hipSuccess) {
            (void)hipEventDestroy(c.ks_live[i]->done);
            c.ks_live.erase(c.ks_live.begin() + (long)i);
        } else ++i;
    }
    (void)hipGetLastError(); // (hipEventQuery's "not ready" is not an error of this call)
    // (a table whose launch was never enqueued -- the call failed between set_subsets and subsets_enqueued -- has no event to wait for:
    // subsets_abandoned() below takes it out again on that path)
    c.ks_live.emplace_back(new lg_accel::LaunchCtx::KsTable());
    lg_accel::LaunchCtx::KsTable &t = *c.ks_live.back();
    const size_t m_ = b.ks.size();
    t.stage.need(2 * m_ * sizeof(unsigned long long));
    unsigned long long *tab = static_cast<unsigned long long *>(t.stage.p); // the m values of k, then (k mod n) | (k / n) << 32 of each (the lattice form, shade.h mode 5)
    for (size_t j = 0; j < m_; ++j) { tab[j] = b.ks[j]; tab[m_ + j] = (b.ks[j] % b.n) | ((b.ks[j] / b.n) << 32); }
    t.buf.alloc(std::max<size_t>(2 * m_, 128));
    HIP_TRY(hipMemcpyAsync(t.buf.p, tab, 2 * m_ * sizeof(unsigned long long), hipMemcpyHostToDevice, stream));
    P.mode = 3; P.pixel_list = t.buf.p; P.sub_m = (uint32_t)b.ks.size(); P.sub_n = b.n; P.sub_k = 0; P.sub_count = b.items;
    P.ntiles = (uint32_t)((b.items + 63ull) / 64ull);
    // the batch tile by lattice column (mode 5, shade.h: 64 / m rows x <= n pixels per tile instead of 64 consecutive work items -- 64 / m
    // periods of one row) where that wastes few lanes; LASGUN_SUBSET_LATTICE=0: never (A/B)
    static const bool lattice = [] { const char *e = std::getenv("LASGUN_SUBSET_LATTICE"); return !(e && e[0] == '0'); }();
    const unsigned long long m = b.ks.size(), rows = m != 0 && m <= 64 ? 64ull / m : 0ull, cols = (P.w + b.n - 1) / b.n;
    const unsigned long long tiles5 = rows ? ((unsigned long long)P.h + rows - 1ull) / rows * cols : ~0ull;
    if (lattice && rows >= 2 && b.n >= 8 && b.n <= P.w && (unsigned long long)P.w * P.h < (1ull << 32) && tiles5 < (1ull << 31) && tiles5 * 3ull <= (unsigned long long)P.ntiles * 4ull + 24ull) {
        P.mode = 5; P.sub_cols = (uint32_t)cols; P.sub_rows = (uint32_t)rows; P.ntiles = (uint32_t)tiles5;
        P.sub_rowtab = lattice_rows(a, stream, P.w, P.h, b.n);
    }
}
// ... and once the batch's launch is enqueued: the event that releases its table
static void subsets_enqueued(const lg_accel &a, hipStream_t stream) {
    lg_accel::LaunchCtx &c = ctx_for(a, stream);
    if (c.ks_live.empty() || c.ks_live.back()->done) return;
    HIP_TRY(hipEventCreateWithFlags(&c.ks_live.back()->done, hipEventDisableTiming));
    HIP_TRY(hipEventRecord(c.ks_live.back()->done, stream));
}

// ... and when the call fails before its launch is enqueued: the table set_subsets made has no launch that reads it and no event that
// would ever release it (the copy into it may still be in flight: the stream is drained first)
static void subsets_abandoned(const lg_accel &a, hipStream_t stream) {
    for (auto &c : a.ctxs)
        if (c->key == stream && !c->ks_live.empty() && !c->ks_live.back()->done) {
            (void)hipStreamSynchronize(stream);
            (void)hipGetLastError();
            c->ks_live.pop_back();
        }
}

template <class F> static int guarded(F f) {
    try {
        CallScope call; // (which API call a launch belongs to: the measured choice counts calls, tune.h)
        f();
        return 0;
    } catch (const std::exception &e) {
        return fail(e.what());
    }
}

// ============================================================================================
extern "C" {

const char *lg_last_error(void) { return tl_error.c_str(); }

// ---- the measured choice's table from outside (tune.h): a caller that knows its workload pins the choice and never pays for a race;
// a test runs with a fixed table.  An entry is the twelve words of a kind and the remembered choice, opaque to the caller.
size_t lg_tune_export(lg_tune_entry *out, size_t capacity) {
    const size_t n = lg::tune::snapshot(nullptr, nullptr, 0);
    if (!out || capacity == 0) return n;
    std::vector<lg::tune::Key> keys(capacity);
    std::vector<int> choices(capacity);
    const size_t m = lg::tune::snapshot(keys.data(), choices.data(), capacity);
    for (size_t i = 0; i < std::min(m, capacity); ++i) {
        std::memcpy(out[i].key, keys[i].v, sizeof out[i].key);
        out[i].choice = choices[i];
        out[i].reserved = 0;
    }
    return m;
}
int lg_tune_import(const lg_tune_entry *entries, size_t count) {
    if (count != 0 && !entries) return fail("lg_tune_import: entries is NULL");
    for (size_t i = 0; i < count; ++i) { // (a choice is checked against the launch when it is used: one the launch cannot take falls back to the rule's)
        if (entries[i].choice < 0 || entries[i].choice >= 256 || (entries[i].choice & 15) > (int)ORG_QUEUE) return fail("lg_tune_import: entry " + std::to_string(i) + " holds no choice this library makes");
    }
    for (size_t i = 0; i < count; ++i) {
        lg::tune::Key k;
        std::memcpy(k.v, entries[i].key, sizeof k.v);
        lg::tune::remember(k, entries[i].choice);
    }
    return 0;
}
void lg_tune_clear(void) { lg::tune::clear(); }
void lg_set_last_error(const char *msg) { tl_error = msg ? msg : ""; } // (multi.cpp reports through the same thread-local message)

static lg_material pack(const Material &m) { lg_material r; r.kind = m.kind; std::memcpy(r.p, m.p, sizeof r.p); return r; }
static Material unpack(const lg_material *m) { Material r; r.kind = m->kind; std::memcpy(r.p, m->p, sizeof r.p); return r; }
static lg_material mat2(int kind, const double a[3], const double b[3], double s0, double s1) {
    lg_material m{}; m.kind = kind;
    for (int i = 0; i < 3; ++i) { m.p[i] = a[i]; if (b) m.p[3 + i] = b[i]; }
    m.p[6] = s0; m.p[7] = s1;
    return m;
}
lg_material lg_material_default(void) { return pack(material_default()); }
lg_material lg_material_matte(const double kd[3], double sigma) { return pack(material_matte(kd, sigma)); }
lg_material lg_material_plastic(const double kd[3], const double ks[3], double roughness) { return mat2(MAT_PLASTIC, kd, ks, roughness, 0.0); }
lg_material lg_material_metal(const double eta[3], const double k[3], double u, double v) { return mat2(MAT_METAL, eta, k, u, v); }
lg_material lg_material_glass(const double kr[3], const double kt[3], double eta) { return mat2(MAT_GLASS, kr, kt, eta, 0.0); }
lg_material lg_material_mirror(const double kr[3]) { return mat2(MAT_MIRROR, kr, nullptr, 0.0, 0.0); }

lg_scene *lg_scene_new(void) { return new lg_scene(); }
void lg_scene_free(lg_scene *s) { delete s; }
void lg_scene_set_perspective_camera(lg_scene *s, double fov) { s->s.camera.init(true, fov); }
void lg_scene_set_orthographic_camera(lg_scene *s, double scale) { s->s.camera.init(false, scale); }
void lg_camera_look_at(lg_scene *s, const double o[3], const double l[3], const double u[3]) {
    s->s.camera.look_at(V3{o[0], o[1], o[2]}, V3{l[0], l[1], l[2]}, V3{u[0], u[1], u[2]});
}
void lg_camera_set_supersampling(lg_scene *s, uint8_t base) { s->s.camera.set_supersampling(base); }
void lg_camera_set_aperture_radius(lg_scene *s, double r) { s->s.camera.aperture_radius = r; }
void lg_scene_set_solid_background(lg_scene *s, const double c[3]) {
    s->s.bg_inner = V3{c[0], c[1], c[2]}; s->s.bg_outer = s->s.bg_inner; s->s.bg_scale = 1.0; // background.rs:18-20
}
void lg_scene_set_radial_background(lg_scene *s, const double i[3], const double o[3], double scale) {
    s->s.bg_inner = V3{i[0], i[1], i[2]}; s->s.bg_outer = V3{o[0], o[1], o[2]}; s->s.bg_scale = scale;
}
void lg_scene_set_ambient_light(lg_scene *s, const double c[3]) { s->s.ambient = V3{c[0], c[1], c[2]}; }
void lg_scene_set_mesh_smoothing(lg_scene *s, int e) { s->s.smoothing = e != 0; }
void lg_scene_set_max_recursion_depth(lg_scene *s, uint32_t d) { s->s.recursion = d; }
void lg_scene_set_threads(lg_scene *s, size_t t) { s->s.threads = t; }
void lg_scene_add_point_light(lg_scene *s, const double p[3], const double i[3], const double f[3]) {
    Light l;
    std::memcpy(l.pos, p, sizeof l.pos); std::memcpy(l.intensity, i, sizeof l.intensity); std::memcpy(l.falloff, f, sizeof l.falloff);
    s->s.lights.push_back(l);
}
int lg_scene_parse_obj(lg_scene *s, const char *text, size_t len, uint32_t *out_ref) { // scene.rs:109-123
    return guarded([&] {
        std::unique_ptr<Obj> obj(new Obj());
        parse_obj_text(text, len, *obj);
        if (!s->s.smoothing) obj->normal.clear();
        *out_ref = (uint32_t)s->s.meshes.size();
        s->s.meshes.push_back(std::move(obj));
    });
}
int lg_scene_load_obj(lg_scene *s, const char *path, uint32_t *out_ref) {
    FILE *f = std::fopen(path, "rb");
    if (!f) return fail(std::string("cannot open ") + path);
    std::string buf;
    char tmp[65536];
    size_t n;
    while ((n = std::fread(tmp, 1, sizeof tmp, f)) > 0) buf.append(tmp, n);
    std::fclose(f);
    return lg_scene_parse_obj(s, buf.data(), buf.size(), out_ref);
}
lg_aggregate *lg_scene_root(lg_scene *s) { return reinterpret_cast<lg_aggregate *>(s->s.root.get()); }
void lg_scene_set_root(lg_scene *s, lg_aggregate *moved) {
    s->s.root.reset(new Aggregate(std::move(moved->a)));
    delete moved;
}

// lg_aggregate is layout-compatible with its only member, so a borrowed `Aggregate*` (scene
// root) can be handed out as lg_aggregate*.
static_assert(sizeof(lg_aggregate) == sizeof(Aggregate), "lg_aggregate must wrap Aggregate exactly");
lg_aggregate *lg_aggregate_new(void) { return new lg_aggregate(); }
void lg_aggregate_free(lg_aggregate *a) { delete a; }
static SceneNode node_of(SceneNode::Kind k) { SceneNode n; n.kind = k; n.mat = material_default(); return n; }
void lg_aggregate_add_group(lg_aggregate *a, lg_aggregate *moved) {
    SceneNode n = node_of(SceneNode::GROUP);
    n.group.reset(new Aggregate(std::move(moved->a)));
    delete moved;
    a->a.contents.push_back(std::move(n));
}
void lg_aggregate_add_sphere(lg_aggregate *a, const double c[3], double r, const lg_material *m) {
    SceneNode n = node_of(SceneNode::SPHERE);
    std::memcpy(n.a, c, sizeof n.a); n.b[0] = r; n.mat = unpack(m); n.has_mat = true;
    a->a.contents.push_back(std::move(n));
}
void lg_aggregate_add_cube(lg_aggregate *a, const double o[3], double dim, const lg_material *m) {
    SceneNode n = node_of(SceneNode::CUBE);
    std::memcpy(n.a, o, sizeof n.a); n.b[0] = dim; n.mat = unpack(m); n.has_mat = true;
    a->a.contents.push_back(std::move(n));
}
void lg_aggregate_add_box(lg_aggregate *a, const double mn[3], const double mx[3], const lg_material *m) {
    SceneNode n = node_of(SceneNode::CUBOID);
    std::memcpy(n.a, mn, sizeof n.a); std::memcpy(n.b, mx, sizeof n.b); n.mat = unpack(m); n.has_mat = true;
    a->a.contents.push_back(std::move(n));
}
void lg_aggregate_add_obj(lg_aggregate *a, uint32_t mesh) {
    SceneNode n = node_of(SceneNode::MESH); n.obj = mesh;
    a->a.contents.push_back(std::move(n));
}
void lg_aggregate_add_obj_of(lg_aggregate *a, uint32_t mesh, const lg_material *m) {
    SceneNode n = node_of(SceneNode::MESH); n.obj = mesh; n.mat = unpack(m); n.has_mat = true;
    a->a.contents.push_back(std::move(n));
}
void lg_aggregate_swap_backface(lg_aggregate *a) { a->a.swap_backface = !a->a.swap_backface; }
void lg_aggregate_translate(lg_aggregate *a, const double d[3]) { transform_concat_self(a->a.transform, transform_translate(d)); }
void lg_aggregate_scale(lg_aggregate *a, double x, double y, double z) { transform_concat_self(a->a.transform, transform_scale(x, y, z)); }
void lg_aggregate_rotate_x(lg_aggregate *a, double t) { transform_concat_self(a->a.transform, transform_rotate_x(t)); }
void lg_aggregate_rotate_y(lg_aggregate *a, double t) { transform_concat_self(a->a.transform, transform_rotate_y(t)); }
void lg_aggregate_rotate_z(lg_aggregate *a, double t) { transform_concat_self(a->a.transform, transform_rotate_z(t)); }
void lg_aggregate_rotate(lg_aggregate *a, double t, const double axis[3]) { transform_concat_self(a->a.transform, transform_rotate(t, axis)); }
void lg_aggregate_get_transform(lg_aggregate *a, double m[16], double minv[16]) {
    for (int c = 0; c < 4; ++c)
        for (int r = 0; r < 4; ++r) { m[4 * c + r] = a->a.transform.m.m[c][r]; minv[4 * c + r] = a->a.transform.minv.m[c][r]; }
}

lg_film *lg_film_new(uint32_t w, uint32_t h) {
    lg_film *f = new lg_film();
    f->w = w; f->h = h;
    f->owned.assign((size_t)w * h * 4, 0);
    f->px = f->owned.data();
    return f;
}
lg_film *lg_film_wrap(uint32_t w, uint32_t h, uint8_t *rgba) {
    lg_film *f = new lg_film();
    f->w = w; f->h = h; f->px = rgba;
    return f;
}
uint8_t *lg_film_pixels(lg_film *f) { return f->px; }
uint32_t lg_film_width(lg_film *f) { return f->w; }
uint32_t lg_film_height(lg_film *f) { return f->h; }
void lg_film_free(lg_film *f) { delete f; }

int lg_set_device(int device) {
    g_device = device;
    g_device_chosen = true;
    return guarded([] { use_device(); });
}
int lg_set_devices(const int *ids, int count) {
    return guarded([&] {
        if (count < 0 || (count > 0 && !ids)) throw Error("bad device list");
        int n = 0;
        if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) throw Error("no HIP device available: liblasgun_hip has no CPU fallback");
        std::vector<int> v;
        if (count == 0) for (int i = 0; i < n; ++i) v.push_back(i); // 0 ids = every visible device
        for (int i = 0; i < count; ++i) {
            if (ids[i] < 0 || ids[i] >= n) throw Error("device index out of range");
            v.push_back(ids[i]); // an index may repeat: its shares then run concurrently on that device
        }
        g_devices = v;
    });
}
uint64_t lg_trim_pool(int device) { return (uint64_t)g_pool.trim(device); }
int lg_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

// Host build (reference trees; with `with_fast` also the fast mode's), upload, and everything derived from the tables.
// Called at creation without the fast trees -- they cost 5-10x the reference build and only mode 1 walks them -- and
// once more, with them, by the first lg_accel_set_mode(accel, 1).
static void build_and_upload(lg_accel *a, bool with_fast) {
    a->ldss_blocks = 0; a->lds_image_n16 = 0; a->fast_available = true;
        static const bool times = std::getenv("LASGUN_DEBUG_TIMES") != nullptr; // (where lg_accel_from's time goes: flatten / upload / derived)
        const auto t_begin = std::chrono::steady_clock::now();
        // The culling records and strips of the pruned walk's mesh leaves are built when that walk will run: by default from PRUNE_MIN_TRIS
        // triangles in a mesh (below: on), when LASGUN_PRUNE=1 or lg_accel_set_prune(1) ask for it (rebuild_tables).
        size_t big_mesh_tris = 0;
        for (const auto &m : a->scene->meshes) if (m && m->tri.size() / 3 > big_mesh_tris) big_mesh_tris = m->tri.size() / 3;
        static const int prune_env = [] { const char *e = std::getenv("LASGUN_PRUNE"); return e && (e[0] == '0' || e[0] == '1') ? e[0] - '0' : -1; }();
        const bool with_records = a->prune == 1 || (a->prune < 0 && (prune_env == 1 || (prune_env < 0 && big_mesh_tris >= PRUNE_MIN_TRIS)));
        flatten_scene(*a->scene, a->flat, with_fast, with_records); // host HLBVH build + flatten (throws on what the reference would panic on)
        const auto t_flat = std::chrono::steady_clock::now();
        use_device(a->device);
        const FlatScene &f = a->flat;
        TableStage stage; // (committed at the end: one allocation, one copy)
        stage.add(a->nodes, f.nodes); stage.add(a->nodes4, f.nodes4); stage.add(a->primref, f.primref); stage.add(a->spheres, f.spheres); stage.add(a->sphere_mat, f.sphere_mat);
        stage.add(a->cuboids, f.cuboids); stage.add(a->cuboid_mat, f.cuboid_mat); stage.add(a->tri_v, f.tri_v); stage.add(a->tri_n, f.tri_n);
        stage.add(a->tri_t, f.tri_t); stage.add(a->leaf_soup, f.leaf_soup); stage.add(a->chunks, f.chunks); stage.add(a->strips, f.strips); stage.add(a->sphere_ref_leaf, f.sphere_ref_leaf); stage.add(a->cuboid_ref_leaf, f.cuboid_ref_leaf);
        stage.add(a->tri_ref_leaf, f.tri_ref_leaf); stage.add(a->accel_ref_leaf, f.accel_ref_leaf); stage.add(a->vpos, f.vpos); stage.add(a->vnorm, f.vnorm); stage.add(a->vtex, f.vtex);
        stage.add(a->materials, f.materials); stage.add(a->lights, f.lights); // (the accel records: below, once their compact bases are known)
        {   // the counters' two records, zeroed (the second: iteration counters of the diagnostic build)
            static const std::vector<DStats> zero(2);
            stage.add(a->stats, zero);
        }
        const auto t_up = std::chrono::steady_clock::now();
        a->device_bytes = f.nodes.size() * sizeof(DNode) + f.nodes4.size() * sizeof(DNode4) + f.primref.size() * 4 + f.spheres.size() * sizeof(DSphere) +
                          f.cuboids.size() * sizeof(DCuboid) + f.tri_v.size() * 12 + f.vpos.size() * 4 + f.vnorm.size() * 4 + f.leaf_soup.size() * sizeof(DLeafRec) +
                          f.strips.size() * sizeof(DStrip) + f.chunks.size() * sizeof(DChunk) +
                          f.accels.size() * sizeof(DAccel) + f.materials.size() * sizeof(DMaterial);
        if (!a->stream) a->stream = g_streams.take(a->device);
        // per-lane LDS stack: worst case of this scene graph, +2 guard entries
        a->stack_depth = f.max_stack + 2;
        // the fast kernel falls back to the reference traversal on exact ties, so its stack must hold either
        a->stack_depth_fast1 = (f.max_stack > f.max_stack_fast1 ? f.max_stack : f.max_stack_fast1) + 2;
        const size_t LDS_MAX = 160 * 1024;
        if ((size_t)a->stack_depth * 256 * 4 > LDS_MAX)
            throw Error("BVH too deep for the LDS traversal stack (" + std::to_string(a->stack_depth) + " entries per lane; the reference panics beyond 64 per level, bvh.rs:497)");
        a->fast_available = (size_t)a->stack_depth_fast1 * 256 * 4 <= LDS_MAX;
        // The fast tree's tight boxes are only meaningful if every accel's `minv` (which moves the rays) really is the
        // inverse of its `m` (which moved the boxes).  Transform3::rotate(theta, axis) takes the transpose for the inverse
        // without normalising the axis (transform.rs:144-148), so a non-unit axis gives a pair that is not: the reference
        // still renders *something* through its fat, overlapping leaves, and the reference traversal reproduces that
        // bit for bit, but the fast mode is refused for such a scene.
        for (const DAccel &A : f.accels) {
            double worst = 0.0;
            for (int r = 0; r < 3; ++r)
                for (int c = 0; c < 4; ++c) {
                    double v = (c == 3 ? A.m.c[3][r] : 0.0);
                    for (int k = 0; k < 3; ++k) v += A.m.c[k][r] * (c == 3 ? A.minv.c[3][k] : A.minv.c[c][k]);
                    const double want = (c < 3 && r == c) ? 1.0 : 0.0;
                    if (!(std::fabs(v - want) <= worst)) worst = std::fabs(v - want);
                }
            if (!(worst <= 1e-11)) { // two orders below the 1e-9 the fast tree's boxes are pushed out by
                a->fast_available = false;
                a->fast_refusal = "fast mode unavailable: an aggregate's transform and inverse do not match (rotate() about a non-unit axis?)";
            }
        }
        // The fast mode cannot be made exact for meshes in principle: a ray that lies within rounding of a FAR triangle's
        // plane is accepted by the reference wherever it passes (its fat leaves test every triangle), and a tight tree never visits that
        // triangle.  The band in which that happens is ~ 64 u R^2 / edge wide: negligible for a tessellated surface seen from nearby,
        // not for a mesh whose coordinates dwarf its small triangles (round 4's progression_soup_scene: triangles at 1e9 beside
        // triangles of 0.05 -- 5 wrong pixels in 4,100 scenes).  Such a mesh is refused, like a transform that does not invert.
        for (const auto &m : a->scene->meshes) { // (a question about the fast mode: asked when its trees are built -- every lg_accel_set_mode(1) goes through such a build first)
            if (!m || !with_fast) continue;
            double max_abs = 0.0, min_edge = INFINITY;
            for (float v : m->position) if (std::isfinite(v)) max_abs = std::fmax(max_abs, std::fabs((double)v));
            for (size_t t = 0; t + 2 < m->tri.size(); t += 3) {
                double longest = 0.0;
                for (int e = 0; e < 3; ++e) {
                    const size_t i = (size_t)m->tri[t + e].v, j = (size_t)m->tri[t + (e + 1) % 3].v;
                    double d2 = 0.0;
                    for (int c = 0; c < 3; ++c) { const double d = (double)m->position[3 * i + c] - (double)m->position[3 * j + c]; d2 += d * d; }
                    longest = std::fmax(longest, std::sqrt(d2));
                }
                if (longest > 0.0 && std::isfinite(longest)) min_edge = std::fmin(min_edge, longest);
            }
            if (std::isfinite(min_edge) && max_abs > min_edge * 0x1p20) {
                a->fast_available = false;
                a->fast_refusal = "fast mode unavailable: a mesh whose coordinates exceed 2^20 times its smallest triangle (a ray in a far triangle's plane is accepted by the reference wherever it passes)";
            }
        }
        if (!a->fast_available) a->stack_depth_fast1 = a->stack_depth;
        // Scenes whose tables stay in L2: the accel records (13 x 16 bytes each) go into LDS behind the stacks of the 256-lane
        // kernels when that keeps four workgroups on a CU -- entering and leaving nested accels is a chain of dependent fetches of
        // these records (37 % of the walk's cycles on config 4m when they come from L2)
        a->accel_image_n16 = 0;
        {
            const size_t img = f.accels.size() * LDS_ACCEL_UNITS * 16;
            if (f.accels.size() <= 64 && ((size_t)a->stack_depth * 256 * 4 + img) * 4 <= LDS_MAX) a->accel_image_n16 = (uint32_t)(f.accels.size() * LDS_ACCEL_UNITS);
        }
        const size_t extra_lds = (size_t)a->accel_image_n16 * 16;
        size_t lds = (size_t)std::max(a->stack_depth, a->stack_depth_fast1) * 256 * 4 + extra_lds;
        if (lds > 64 * 1024) { HIP_TRY(mega_set_lds_limit(lds, false)); HIP_TRY(wf_set_lds_limit(lds, false)); HIP_TRY(queue_set_lds_limit(lds, false)); }
        int per_cu = 0, cus = 0;
        HIP_TRY(trace_occupancy(a->stack_depth, false, extra_lds, &per_cu));
        int per_cu_fast = 0;
        HIP_TRY(trace_occupancy(a->stack_depth_fast1, true, 0, &per_cu_fast));
        if (per_cu_fast < 1) per_cu_fast = 1;
        a->max_blocks_fast = (uint32_t)per_cu_fast;
        HIP_TRY(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, a->device));
        if (per_cu < 1) per_cu = 1;
        a->max_blocks = (uint32_t)(per_cu * cus);
        a->max_blocks_fast *= (uint32_t)cus;
        int wb = 0, wbf = 0;
        HIP_TRY(wf_trace_occupancy(a->stack_depth, false, extra_lds, &wb));
        HIP_TRY(wf_trace_occupancy(a->stack_depth_fast1, true, 0, &wbf));
        a->wf_blocks = (uint32_t)((wb < 1 ? 1 : wb) * cus);
        a->wf_blocks_fast = (uint32_t)((wbf < 1 ? 1 : wbf) * cus);
        int qb = 0;
        HIP_TRY(queue_occupancy(a->stack_depth, extra_lds, &qb));
        if (const char *e = std::getenv("LASGUN_QUEUE_BLOCKS_PER_CU")) { const int v = std::atoi(e); if (v >= 1 && v < qb) qb = v; } // (diagnostic: how much the kernel gains from each resident workgroup)
        a->queue_blocks = (uint32_t)((qb < 1 ? 1 : qb) * cus);
        a->cus = (uint32_t)cus;
        // LDS-resident scene: the REFERENCE tree's nodes (56 of 64 bytes, padded to 80 when that fits),
        // its primrefs, the spheres (padded to 48 when that fits) and cuboids, behind 1024 per-lane
        // stacks, all within one CU's LDS.  The flat tables interleave reference and fast trees per
        // accel, so the image renumbers the reference trees compactly (DAccel::lnode_base / lprim_base).
        {
            FlatScene &fm = a->flat;
            std::vector<uint32_t> nb, pb; // start offsets of every tree / primref run, both kinds
            for (const DAccel &A : fm.accels) { nb.push_back(A.node_base); nb.push_back(A.fnode_base); pb.push_back(A.prim_base); pb.push_back(A.fprim_base); }
            std::sort(nb.begin(), nb.end()); nb.erase(std::unique(nb.begin(), nb.end()), nb.end());
            std::sort(pb.begin(), pb.end()); pb.erase(std::unique(pb.begin(), pb.end()), pb.end());
            auto extent = [](const std::vector<uint32_t> &starts, uint32_t b, size_t total) {
                auto it = std::upper_bound(starts.begin(), starts.end(), b);
                return (uint32_t)((it == starts.end() ? total : (size_t)*it) - b);
            };
            std::vector<std::pair<uint32_t, uint32_t>> nruns, pruns; // (global base, compact base) of each reference tree, once
            uint32_t nn = 0, np = 0;
            // compact numbering: the non-mesh accels first (their slots get a leaf record in the image), then the meshes
            uint32_t np_soup = 0; // slots of the non-mesh accels
            for (int pass = 0; pass < 2; ++pass) {
              if (pass == 1) np_soup = np;
              for (DAccel &A : fm.accels) {
                if (((A.flags & AF_MESH) != 0u) != (pass == 1)) continue;
                auto fn = std::find_if(nruns.begin(), nruns.end(), [&](auto &r) { return r.first == A.node_base; });
                if (fn == nruns.end()) { nruns.emplace_back(A.node_base, nn); A.lnode_base = nn; nn += extent(nb, A.node_base, fm.nodes.size()); }
                else A.lnode_base = fn->second;
                auto fp = std::find_if(pruns.begin(), pruns.end(), [&](auto &r) { return r.first == A.prim_base; });
                if (fp == pruns.end()) { pruns.emplace_back(A.prim_base, np); A.lprim_base = np; np += extent(pb, A.prim_base, fm.primref.size()); }
                else A.lprim_base = fp->second;
              }
            }
            const size_t stack_bytes = (size_t)a->stack_depth * 1024 * 4; // per-lane stacks of the private walks
            const size_t prim16 = ((size_t)np + 3) / 4;
            // image: [nodes, LDS_NODE_STRIDE units each][primrefs][leaf records, 3 units per slot][accel records]
            const size_t accel16 = fm.accels.size() * LDS_ACCEL_UNITS;
            const size_t n16 = (size_t)nn * LDS_NODE_STRIDE + prim16 + (size_t)np_soup * 3 + accel16;
            if (stack_bytes + n16 * 16 <= LDS_MAX) {
                std::vector<uint32_t> img(n16 * 4, 0u);
                a->lds_node_off = 0;
                for (auto &r : nruns)
                    for (uint32_t i = 0, e = extent(nb, r.first, fm.nodes.size()); i < e; ++i) {
                        uint32_t *rec = &img[((size_t)(r.second + i) * LDS_NODE_STRIDE) * 4];
                        std::memcpy(rec, &fm.nodes[r.first + i], 56);
                    }
                a->lds_prim_off = nn * LDS_NODE_STRIDE;
                a->lds_soup_off = a->lds_prim_off + (uint32_t)prim16;
                for (auto &r : pruns)
                    for (uint32_t i = 0, e = extent(pb, r.first, fm.primref.size()); i < e; ++i) {
                        img[(size_t)a->lds_prim_off * 4 + r.second + i] = fm.primref[r.first + i];
                        if (r.second + i < np_soup)
                            std::memcpy(&img[((size_t)a->lds_soup_off + (size_t)(r.second + i) * 3) * 4], &fm.leaf_soup[r.first + i], 48);
                    }
                // walk words of every record (words 16..19; walk.h, traverse_ref): the second formulation of the reference walk
                // addresses nodes by their byte offset in the image and takes a leaf's slot range ready-made
                for (const DAccel &A : fm.accels) {
                    const uint32_t tree0 = a->lds_node_off * 16u + A.lnode_base * LDS_NODE_STRIDE * 16u;
                    for (uint32_t i = 0, e = extent(nb, A.node_base, fm.nodes.size()); i < e; ++i) {
                        uint32_t *rec = &img[((size_t)(A.lnode_base + i) * LDS_NODE_STRIDE) * 4];
                        const DNode &nd = fm.nodes[A.node_base + i];
                        if (nd.meta & NODE_LEAF) { rec[16] = A.lprim_base + nd.link; rec[17] = NODE_LEAF; rec[18] = rec[16] + (nd.meta & 0xFFFFu); rec[19] = nd.pad; }
                        else { rec[16] = tree0 + nd.link * LDS_NODE_STRIDE * 16u; rec[17] = 1u << (nd.meta & 3u); rec[18] = 0u; }
                        rec[17] |= nd.meta & NODE_NOPRUNE;
                    }
                }
                a->lds_accel_off = a->lds_soup_off + np_soup * 3u;
                for (size_t i = 0; i < fm.accels.size(); ++i) {
                    const DAccel &A = fm.accels[i];
                    uint32_t *rec = &img[((size_t)a->lds_accel_off + i * LDS_ACCEL_UNITS) * 4];
                    std::memcpy(rec, &A.minv, 96);
                    rec[24] = a->lds_node_off * 16u + A.lnode_base * LDS_NODE_STRIDE * 16u;
                    rec[25] = A.lprim_base; rec[26] = A.prim_base - A.lprim_base; rec[27] = A.flags;
                    rec[28] = (uint32_t)A.parent; rec[29] = A.nchain;
                    for (int k = 0; k < MAX_CHAIN; ++k) rec[32 + k] = A.chain[k];
                    std::memcpy(rec + 40, A.prune, sizeof A.prune);
                }
                stage.add(a->lds_image, img);
                a->lds_image_n16 = (uint32_t)n16;
                HIP_TRY(mega_set_lds_limit(LDS_MAX, true)); HIP_TRY(wf_set_lds_limit(LDS_MAX, true)); HIP_TRY(queue_set_lds_limit(LDS_MAX, true));
                a->ldss_blocks = (uint32_t)cus;
            }
            stage.add(a->accels, fm.accels); // with the compact bases
            if (a->accel_image_n16) { // the accel records alone, global bases in unit [6] (the LDS-resident image carries compact ones)
                std::vector<uint32_t> img((size_t)a->accel_image_n16 * 4, 0u);
                for (size_t i = 0; i < fm.accels.size(); ++i) {
                    const DAccel &A = fm.accels[i];
                    uint32_t *rec = &img[i * LDS_ACCEL_UNITS * 4];
                    std::memcpy(rec, &A.minv, 96);
                    rec[24] = A.node_base; rec[25] = A.prim_base; rec[26] = 0u; rec[27] = A.flags;
                    rec[28] = (uint32_t)A.parent; rec[29] = A.nchain;
                    for (int k = 0; k < MAX_CHAIN; ++k) rec[32 + k] = A.chain[k];
                    std::memcpy(rec + 40, A.prune, sizeof A.prune);
                }
                stage.add(a->accel_image, img);
            }
        }
        stage.commit(a->arena);
        // Which organisation is the default (measured, tools/threshold_sweep.py): the megakernel unless the scene has
        // so many spheres / boxes that BVH-node and sphere tests dominate a ray (>= 512: with the scene tables in LDS
        // the megakernel keeps up to ~50 node + primitive tests per ray; beyond that the traversal kernels' lower
        // register pressure outweighs the per-pixel state traffic); scenes that also carry a big mesh have long,
        // uneven tiles and need more of them per wave to balance.
        {
            size_t big_mesh = 0;
            for (const auto &m : a->scene->meshes) if (m && m->tri.size() / 3 > big_mesh) big_mesh = m->tri.size() / 3;
            // Scenes with glass / mirror run level by level in the wavefront pipeline under the same criterion (tools/bench_configs.py
            // --org=...): where node and sphere tests dominate.  Small specular scenes are a wash (Cornell glass
            // 512^2: 0.81 ms level by level, 0.80 ms in the megakernel since both walk with traverse_ref), and with a big mesh the deeper
            // levels are few, long, incoherent walks through 254-triangle leaves whose slowest wave sets each launch's length
            // (100k-triangle glass torus: 226 against 136 ms): those stay in the megakernel, where other tiles fill the gaps.
            // the reference's mesh leaves hold up to 254 triangles (bvh.rs:187,289): skipping one pays for many node steps -- from PRUNE_MIN_TRIS
            // triangles (tools/prune_threshold_probe.py, profiles/r05_prune_threshold.jsonl: below that the pruned walk is 5-30 % SLOWER on
            // 1024^2 frames and its records are a third to a half of the accel build; rounds 3-5 had 256)
            a->prune_default = big_mesh >= PRUNE_MIN_TRIS;
            a->streaming_pays = f.spheres.size() + f.cuboids.size() >= 512 && !(f.has_specular && big_mesh >= 4096);
            a->mega_narrow = f.spheres.size() + f.cuboids.size() < 512;
            if (const char *e = std::getenv("LASGUN_MEGA_LANES")) a->mega_narrow = std::atoi(e) == 768; // (A/B)
            // a big mesh of glass / mirror: the queue organisation (round 4; config 4: 38.7 against the megakernel's 40.8 ms and the
            // level-by-level pipeline's 80; a metal mesh beside a small mirror -- config 4m -- stays in the megakernel: 14.7 / 16.4)
            bool specular_mesh = false; // a mesh of >= 4096 triangles that is itself glass / mirror: every hit on it spawns secondary rays
            for (const DAccel &A : f.accels)
                if ((A.flags & AF_MESH) && A.material >= 0) {
                    const int kind = f.materials[(size_t)A.material].kind;
                    size_t tris = 0;
                    for (const auto &m : a->scene->meshes) if (m && m->tri.size() / 3 > tris) tris = m->tri.size() / 3; // (an upper bound: the largest mesh)
                    specular_mesh = specular_mesh || ((kind == MAT_GLASS || kind == MAT_MIRROR) && tris >= 4096);
                }
            a->queue_default = f.has_specular && big_mesh >= 4096 && specular_mesh;
            a->streaming_min_items = big_mesh >= 4096 ? (1ull << 23) : (1ull << 21); // (config 3's scene at 1024^2: 0.84 ms in the megakernel, 0.99 level by level; at 2048^2: 2.20 / 2.11)
        }
        if (times) {
            const auto t_end = std::chrono::steady_clock::now();
            auto ms = [](auto a0, auto a1) { return std::chrono::duration<double, std::milli>(a1 - a0).count(); };
            std::fprintf(stderr, "[lasgun] accel build: flatten %.3f ms, table uploads %.3f ms, streams / occupancy / LDS images %.3f ms\n", ms(t_begin, t_flat), ms(t_flat, t_up), ms(t_up, t_end));
        }
}

// First request for the fast mode: its trees are built and every table is uploaded into a SECOND accel; only when all of
// that has succeeded are the tables swapped in (a failure -- a HIP error, out of memory -- leaves the accel as it was).
// The reference trees must come out as they did at lg_accel_from: a scene modified since is an error, not a silent
// change of the parity tables.  Caller holds a->mtx.
static void swap_tables(lg_accel &x, lg_accel &y) {
    using std::swap;
    swap(x.flat, y.flat);
    swap(x.arena, y.arena); swap(x.stats, y.stats); // (the counters' record is a view into the arena like the small tables)
    swap(x.nodes, y.nodes); swap(x.nodes4, y.nodes4); swap(x.primref, y.primref); swap(x.spheres, y.spheres); swap(x.sphere_mat, y.sphere_mat);
    swap(x.cuboids, y.cuboids); swap(x.cuboid_mat, y.cuboid_mat); swap(x.tri_v, y.tri_v); swap(x.tri_n, y.tri_n); swap(x.tri_t, y.tri_t);
    swap(x.vpos, y.vpos); swap(x.vnorm, y.vnorm); swap(x.vtex, y.vtex); swap(x.leaf_soup, y.leaf_soup); swap(x.chunks, y.chunks); swap(x.strips, y.strips);
    swap(x.sphere_ref_leaf, y.sphere_ref_leaf); swap(x.cuboid_ref_leaf, y.cuboid_ref_leaf); swap(x.tri_ref_leaf, y.tri_ref_leaf); swap(x.accel_ref_leaf, y.accel_ref_leaf);
    swap(x.accels, y.accels); swap(x.materials, y.materials); swap(x.lights, y.lights);
    swap(x.lds_image, y.lds_image); swap(x.accel_image, y.accel_image); swap(x.accel_image_n16, y.accel_image_n16);
    swap(x.lds_image_n16, y.lds_image_n16); swap(x.lds_node_off, y.lds_node_off); swap(x.lds_prim_off, y.lds_prim_off);
    swap(x.lds_soup_off, y.lds_soup_off); swap(x.lds_accel_off, y.lds_accel_off);
    swap(x.ldss_blocks, y.ldss_blocks); swap(x.cus, y.cus);
    swap(x.stack_depth, y.stack_depth); swap(x.stack_depth_fast1, y.stack_depth_fast1); swap(x.max_blocks, y.max_blocks); swap(x.max_blocks_fast, y.max_blocks_fast);
    swap(x.wf_blocks, y.wf_blocks); swap(x.wf_blocks_fast, y.wf_blocks_fast); swap(x.queue_blocks, y.queue_blocks);
    swap(x.queue_default, y.queue_default); swap(x.prune_default, y.prune_default); swap(x.queue_min_items, y.queue_min_items); swap(x.specular_small_items, y.specular_small_items);
    swap(x.device_bytes, y.device_bytes); swap(x.fast_available, y.fast_available); swap(x.fast_refusal, y.fast_refusal);
    swap(x.streaming_pays, y.streaming_pays); swap(x.streaming_min_items, y.streaming_min_items); swap(x.mega_narrow, y.mega_narrow);
}
// the tables once more, with what was left out of them: the fast mode's trees (lg_accel_set_mode(1)), the pruned walk's leaf records (lg_accel_set_prune(1), lg_audit_prune)
static void rebuild_tables(const lg_accel *ca, bool fast) {
    if (fast && ca->flat.has_fast) return;
    lg_accel *a = const_cast<lg_accel *>(ca);
    use_device(a->device);
    std::unique_ptr<lg_accel> next(new lg_accel());
    next->scene = a->scene;
    next->device = a->device;
    next->prune = a->prune == 1 || a->flat.has_records ? 1 : a->prune; // (what the tables hold stays in them)
    build_and_upload(next.get(), fast || a->flat.has_fast); // throws: `a` is untouched
    next->prune = a->prune;
    // (bit patterns, not values: a NaN bound of a degenerate scene equals itself here)
    if (next->flat.dump_f.size() != a->flat.dump_f.size() ||
        (!a->flat.dump_f.empty() && std::memcmp(next->flat.dump_f.data(), a->flat.dump_f.data(), a->flat.dump_f.size() * sizeof(a->flat.dump_f[0])) != 0) ||
        next->flat.dump_i != a->flat.dump_i)
        throw Error("the scene was modified after lg_accel_from: the accel's reference trees no longer match it (build a new accel)");
    HIP_TRY(hipDeviceSynchronize()); // nothing may still be reading the tables that are about to be replaced
    swap_tables(*a, *next);
    std::swap(a->flat.dump_f, next->flat.dump_f); // same contents; keeps the storage lg_accel_dump's callers point into
    std::swap(a->flat.dump_i, next->flat.dump_i);
    // `next` (the old tables) is released here; its stream was never created for launches
}

static lg_accel *accel_from_on(const lg_scene *s, int device) {
    lg_accel *a = nullptr;
    int rc = guarded([&] {
        a = new lg_accel();
        a->scene = &s->s;
        a->device = device;
        build_and_upload(a, false);
    });
    if (rc) { delete a; return nullptr; }
    return a;
}
lg_accel *lg_accel_from(const lg_scene *s) { return accel_from_on(s, g_device); }
lg_accel *lg_accel_from_on(const lg_scene *s, int device) { return accel_from_on(s, device); }
void lg_accel_free(lg_accel *a) {
    if (!a) return;
    // its buffers go back to the pool and may be handed out again at once: nothing on any stream may still use them
    // (hipFree used to imply the same wait)
    if (hipSetDevice(a->device) == hipSuccess) (void)hipDeviceSynchronize();
    delete a;
}
void *lg_accel_stream(const lg_accel *a) { return (void *)a->stream; }
int lg_accel_synchronize(const lg_accel *a) {
    return guarded([&] { std::lock_guard<std::mutex> g(a->mtx); use_device(a->device); sync_checked(*a); });
}

int lg_capture_rows_device(const lg_accel *a, uint32_t w, uint32_t h, uint32_t y0, uint32_t y1, uint32_t row0, void *dev_rgba, void *hip_stream) {
    return guarded([&] {
        if (y1 > h || y0 > y1 || row0 > y0) throw Error("bad row range");
        std::lock_guard<std::mutex> g(a->mtx);
        use_device(a->device);
        DParams P = base_params(*a, w, h);
        set_rect(P, 0, y0, w, y1);
        P.out_row0 = row0;
        P.out_rgba = (uint8_t *)dev_rgba;
        enqueue(*a, P, false, (hipStream_t)hip_stream);
    });
}
int lg_capture_interleaved_device(const lg_accel *a, uint32_t w, uint32_t h, uint32_t block_rows, uint32_t n, uint32_t r, void *dev_rgba, void *hip_stream) {
    return guarded([&] {
        if (n == 0 || r >= n || block_rows == 0 || h % (block_rows * n) != 0) throw Error("height must be a multiple of block_rows * n");
        std::lock_guard<std::mutex> g(a->mtx);
        use_device(a->device);
        DParams P = base_params(*a, w, h);
        set_rect(P, 0, 0, w, h / n); // virtual rows of the compact tile
        P.ilv_n = n; P.ilv_r = r; P.ilv_b = block_rows;
        P.out_row0 = 0;
        P.out_rgba = (uint8_t *)dev_rgba;
        enqueue(*a, P, false, (hipStream_t)hip_stream);
    });
}
int lg_capture_subset_device(size_t k, size_t n, const lg_accel *a, uint32_t w, uint32_t h, void *dev_rgba, void *hip_stream) {
    return guarded([&] {
        if (n == 0) throw Error("n must be > 0");
        std::lock_guard<std::mutex> g(a->mtx);
        use_device(a->device);
        DParams P = base_params(*a, w, h);
        if (n == 1 && k == 0) set_rect(P, 0, 0, w, h);
        else set_subset(*a, (hipStream_t)hip_stream, P, k, n, w, h);
        P.out_row0 = 0;
        P.out_rgba = (uint8_t *)dev_rgba;
        enqueue(*a, P, false, (hipStream_t)hip_stream);
    });
}

// capture_subset into a HOST film (lib.rs:110-162).  The reference calls this from n threads at once on one
// Accel and one film (lib.rs:67-103: write-disjoint pixel sets), so concurrent calls must not disturb each other:
// every call renders into a device buffer of its own -- the whole film for (0, 1), otherwise a COMPACT buffer of
// just the subset's pixels, one word per work item -- and only the owned pixels {k + i*n} of the host film are
// written.  Nothing is uploaded and no other pixel of the film is touched (lib.rs:152).
int lg_capture_subset(size_t k, size_t n, const lg_accel *a, lg_film *film) {
    return guarded([&] {
        if (n == 0) throw Error("n must be > 0");
        const uint32_t w = film->w, h = film->h;
        const unsigned long long area = (unsigned long long)w * h;
        const bool whole = n == 1 && k == 0;
        const unsigned long long count = whole ? area : subset_count(area, k, n);
        if (count == 0) return;
        DevBuf<uint32_t> buf; // this call's own output (returned to the pool when the call ends)
        // A whole film of 2^22 pixels and more goes out in ROW BANDS: the bands are rendered one after the other on the accel's stream, and
        // each is copied home on a second stream as soon as it is done -- band j's 16 MiB cross PCIe while band j + 1 renders; round 4
        // issued ONE copy of the whole film after the last kernel (config 3: 1.2 of capture()'s 9.96 ms).  Same pixels, same bytes: a
        // pixel's value does not depend on the launch it is rendered in.  (Bands side by side on four streams were measured first: their
        // persistent grids share the chip, all four finish together and the copies still come last -- 8.57 against 8.29 ms.)
        static const unsigned bands_env = [] { const char *e = std::getenv("LASGUN_CAPTURE_BANDS"); return e ? (unsigned)std::atoi(e) : 4u; }();
        const unsigned nbands = whole && area >= (1ull << 22) && bands_env >= 2 && h >= 64 ? std::min(bands_env, 16u) : 1u;
        const uint32_t rows8 = (h + 7u) / 8u; // bands are whole rows of 8x8 tiles
        auto band_rows = [&](unsigned j, uint32_t &y0, uint32_t &y1) {
            y0 = (uint32_t)((unsigned long long)rows8 * j / nbands) * 8u;
            y1 = std::min(h, (uint32_t)((unsigned long long)rows8 * (j + 1) / nbands) * 8u);
        };
        hipStream_t copy_stream = nullptr;
        struct Events { // (destroyed on every way out: an enqueue that throws must not leak the bands' events)
            std::vector<hipEvent_t> v;
            ~Events() { for (hipEvent_t e : v) (void)hipEventDestroy(e); }
        } rendered_events;
        std::vector<hipEvent_t> &rendered = rendered_events.v;
        {
            std::lock_guard<std::mutex> g(a->mtx);
            use_device(a->device);
            buf.alloc((size_t)count);
            DParams P = base_params(*a, w, h);
            P.out_row0 = 0;
            P.out_rgba = (uint8_t *)buf.p;
            if (nbands > 1) {
                ensure_aux_streams(*a, 1);
                copy_stream = a->aux_streams[0];
                try {
                    for (unsigned j = 0; j < nbands; ++j) {
                        uint32_t y0, y1;
                        band_rows(j, y0, y1);
                        hipEvent_t ev = nullptr;
                        HIP_TRY(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
                        rendered.push_back(ev);
                        if (y1 > y0) {
                            DParams B = P;
                            set_rect(B, 0, y0, w, y1);
                            enqueue(*a, B, false, a->stream);
                        }
                        HIP_TRY(hipEventRecord(ev, a->stream));
                    }
                } catch (...) {
                    (void)hipStreamSynchronize(a->stream); // (the bands already enqueued still write `buf`, which goes back to the pool on the way out)
                    throw;
                }
            } else {
                if (whole) set_rect(P, 0, 0, w, h);
                else { set_subset(*a, a->stream, P, k, n, w, h); P.out_compact = 1; }
                enqueue(*a, P, false, a->stream);
                if (whole) HIP_TRY(hipMemcpyAsync(film->px, buf.p, (size_t)area * 4, hipMemcpyDeviceToHost, a->stream));
            }
        }
        if (nbands > 1) { // the copies, in band order, every render already enqueued (a copy into pageable memory holds this thread until its band is home)
            use_device(a->device);
            hipError_t err = hipSuccess;
            for (unsigned j = 0; j < nbands && err == hipSuccess; ++j) {
                uint32_t y0, y1;
                band_rows(j, y0, y1);
                if (y1 <= y0) continue;
                const size_t off = (size_t)y0 * w * 4, bytes = (size_t)(y1 - y0) * w * 4;
                err = hipStreamWaitEvent(copy_stream, rendered[j], 0);
                if (err == hipSuccess) err = hipMemcpyAsync(film->px + off, (const uint8_t *)buf.p + off, bytes, hipMemcpyDeviceToHost, copy_stream);
            }
            if (err == hipSuccess) err = hipStreamSynchronize(copy_stream);
            const hipError_t err2 = hipStreamSynchronize(a->stream); // (nothing may still write `buf` when it goes back to the pool)
            if (err != hipSuccess || err2 != hipSuccess) throw Error(std::string("banded capture: ") + hipGetErrorString(err != hipSuccess ? err : err2));
            std::lock_guard<std::mutex> g(a->mtx);
            check_queue_error(*a);
            return;
        }
        if (whole) { use_device(a->device); sync_checked(*a); return; }
        std::vector<uint32_t> host((size_t)count);
        use_device(a->device);
        HIP_TRY(hipMemcpyAsync(host.data(), buf.p, (size_t)count * 4, hipMemcpyDeviceToHost, a->stream));
        sync_checked(*a);
        uint8_t *px = film->px;
        for (unsigned long long i = 0; i < count; ++i) std::memcpy(px + 4 * (k + i * n), &host[(size_t)i], 4);
    });
}
// Several subsets of one n in ONE render (no counterpart in the reference, whose progressive caller -- www/renderer.ts:103-120 -- makes a
// hundred capture_subset calls one after the other; each is then a launch chain that fills a fraction of the GPU).  The pixels written
// are exactly those of the `count` calls lg_capture_subset(ks[j], n, ...), every other pixel is left alone; a batch that lists every
// k of 0 .. n-1 is the frame and is rendered as one (8x8 tiles).
int lg_capture_subsets_device(const size_t *ks, size_t count, size_t n, const lg_accel *a, uint32_t w, uint32_t h, void *dev_rgba, void *hip_stream) {
    return guarded([&] {
        const SubsetBatch b = make_batch(ks, count, n, w, h);
        if (b.items == 0) return;
        std::lock_guard<std::mutex> g(a->mtx);
        use_device(a->device);
        DParams P = base_params(*a, w, h);
        try {
            if (b.whole) set_rect(P, 0, 0, w, h);
            else set_subsets(*a, P, b, (hipStream_t)hip_stream);
            P.out_row0 = 0;
            P.out_rgba = (uint8_t *)dev_rgba;
            enqueue(*a, P, false, (hipStream_t)hip_stream);
            if (!b.whole) subsets_enqueued(*a, (hipStream_t)hip_stream);
        } catch (...) { subsets_abandoned(*a, (hipStream_t)hip_stream); throw; }
    });
}
int lg_capture_subsets(const size_t *ks, size_t count, size_t n, const lg_accel *a, lg_film *film) {
    return guarded([&] {
        const uint32_t w = film->w, h = film->h;
        const SubsetBatch b = make_batch(ks, count, n, w, h);
        if (b.items == 0) return;
        if (b.whole) { if (lg_capture_subset(0, 1, a, film)) throw Error(tl_error); return; }
        // as lg_capture_subset: a compact buffer of this call's own (work item i -> word i), one D2H copy for the batch, and only
        // the owned pixels of the host film are written (concurrent callers on one film own disjoint pixels)
        DevBuf<uint32_t> buf;
        std::vector<uint32_t> host((size_t)b.items);
        {
            std::lock_guard<std::mutex> g(a->mtx);
            use_device(a->device);
            buf.alloc((size_t)b.items);
            DParams P = base_params(*a, w, h);
            try {
                set_subsets(*a, P, b, a->stream);
                P.out_compact = 1;
                P.out_row0 = 0;
                P.out_rgba = (uint8_t *)buf.p;
                enqueue(*a, P, false, a->stream);
                subsets_enqueued(*a, a->stream);
            } catch (...) { subsets_abandoned(*a, a->stream); throw; }
            HIP_TRY(hipMemcpyAsync(host.data(), buf.p, (size_t)b.items * 4, hipMemcpyDeviceToHost, a->stream));
        }
        use_device(a->device);
        sync_checked(*a);
        const unsigned long long area = (unsigned long long)w * h, m = b.ks.size();
        uint8_t *px = film->px;
        auto scatter = [&](unsigned long long q0, unsigned long long q1) { // periods [q0, q1): ascending addresses
            for (unsigned long long q = q0; q < q1; ++q)
                for (unsigned long long j = 0; j < m; ++j) {
                    const unsigned long long off = b.ks[(size_t)j] + q * b.n;
                    if (off < area) std::memcpy(px + 4 * off, &host[(size_t)(q * m + j)], 4);
                }
        };
        const unsigned nthreads = b.items >= (1ull << 21) ? std::min(8u, std::max(1u, std::thread::hardware_concurrency())) : 1u;
        if (nthreads <= 1) scatter(0, b.periods);
        else {
            std::vector<std::thread> pool;
            for (unsigned t = 0; t < nthreads; ++t) pool.emplace_back(scatter, b.periods * t / nthreads, b.periods * (t + 1) / nthreads);
            for (auto &th : pool) th.join();
        }
    });
}
// Any list of pixels of a width x height film (offset = y * width + x), results compact and in list order:
// rgba_out[4*i ..] and/or rgb_out[3*i ..] (f64 radiance before quantisation) for offsets[i].  Test / tooling hook:
// samples and crops of films too large to move whole.
int lg_capture_pixels(const lg_accel *a, uint32_t w, uint32_t h, const uint64_t *offsets, size_t count, uint8_t *rgba_out, double *rgb_out) {
    return guarded([&] {
        if (count == 0) return;
        if (!offsets) throw Error("offsets is NULL");
        const unsigned long long area = (unsigned long long)w * h;
        for (size_t i = 0; i < count; ++i) if (offsets[i] >= area) throw Error("pixel offset outside the film");
        if (count > 0xFFFFFFFFull * 64ull) throw Error("too many pixels");
        DevBuf<unsigned long long> list;
        DevBuf<uint32_t> rgba;
        DevBuf<double> rad;
        std::lock_guard<std::mutex> g(a->mtx);
        use_device(a->device);
        list.alloc(count);
        HIP_TRY(hipMemcpyAsync(list.p, offsets, count * 8, hipMemcpyHostToDevice, a->stream));
        DParams P = base_params(*a, w, h);
        P.mode = 2; P.pixel_list = list.p; P.sub_count = count; P.out_compact = 1;
        P.ntiles = (uint32_t)((count + 63) / 64);
        if (rgba_out) { rgba.alloc(count); P.out_rgba = (uint8_t *)rgba.p; }
        if (rgb_out) { rad.alloc(count * 3); P.out_radiance = rad.p; }
        enqueue(*a, P, false, a->stream);
        if (rgba_out) HIP_TRY(hipMemcpyAsync(rgba_out, rgba.p, count * 4, hipMemcpyDeviceToHost, a->stream));
        if (rgb_out) HIP_TRY(hipMemcpyAsync(rgb_out, rad.p, count * 24, hipMemcpyDeviceToHost, a->stream));
        sync_checked(*a);
    });
}
// The crop [x0, x1) x [y0, y1) of a width x height film, compact and row-major: rgba_out (x1-x0)*(y1-y0)*4 bytes and/or
// rgb_out (x1-x0)*(y1-y0)*3 doubles on the HOST.
int lg_capture_rect(const lg_accel *a, uint32_t w, uint32_t h, uint32_t x0, uint32_t y0, uint32_t x1, uint32_t y1, uint8_t *rgba_out, double *rgb_out) {
    return guarded([&] {
        if (x1 > w || y1 > h || x0 > x1 || y0 > y1) throw Error("bad rectangle");
        const size_t count = (size_t)(x1 - x0) * (y1 - y0);
        if (count == 0) return;
        DevBuf<uint32_t> rgba;
        DevBuf<double> rad;
        std::lock_guard<std::mutex> g(a->mtx);
        use_device(a->device);
        DParams P = base_params(*a, w, h);
        set_rect(P, x0, y0, x1, y1);
        P.out_row0 = y0; P.out_x0 = x0; P.out_pitch = x1 - x0;
        if (rgba_out) { rgba.alloc(count); P.out_rgba = (uint8_t *)rgba.p; }
        if (rgb_out) { rad.alloc(count * 3); P.out_radiance = rad.p; }
        enqueue(*a, P, false, a->stream);
        if (rgba_out) HIP_TRY(hipMemcpyAsync(rgba_out, rgba.p, count * 4, hipMemcpyDeviceToHost, a->stream));
        if (rgb_out) HIP_TRY(hipMemcpyAsync(rgb_out, rad.p, count * 24, hipMemcpyDeviceToHost, a->stream));
        sync_checked(*a);
    });
}
// One device's share of a multi-device capture: rows of `film` rendered on `device` and copied home.
// With `block_rows` > 0 the share is the 64-row blocks {r, r+n, ...} (even load, one D2H copy per block);
// otherwise the contiguous row tile r of n.
static int capture_share(const lg_scene *s, lg_film *film, int device, uint32_t n, uint32_t r, uint32_t block_rows) {
    lg_accel *a = accel_from_on(s, device);
    if (!a) return 1;
    const uint32_t w = film->w, h = film->h;
    int rc = guarded([&] {
        use_device(device);
        const size_t row_bytes = (size_t)w * 4;
        if (block_rows) {
            const uint32_t rows = h / n;
            a->staging.alloc((size_t)rows * row_bytes);
            if (lg_capture_interleaved_device(a, w, h, block_rows, n, r, a->staging.p, (void *)a->stream)) throw Error(tl_error);
            for (uint32_t g = 0; g < rows / block_rows; ++g) { // block g of the compact tile is image block g*n + r
                const size_t src = (size_t)g * block_rows * row_bytes, dst = ((size_t)g * n + r) * block_rows * row_bytes;
                HIP_TRY(hipMemcpyAsync(film->px + dst, a->staging.p + src, (size_t)block_rows * row_bytes, hipMemcpyDeviceToHost, a->stream));
            }
        } else {
            const uint32_t base = h / n, rem = h % n;
            const uint32_t y0 = r * base + (r < rem ? r : rem), y1 = y0 + base + (r < rem ? 1u : 0u);
            if (y1 > y0) {
                a->staging.alloc((size_t)(y1 - y0) * row_bytes);
                if (lg_capture_rows_device(a, w, h, y0, y1, y0, a->staging.p, (void *)a->stream)) throw Error(tl_error);
                HIP_TRY(hipMemcpyAsync(film->px + (size_t)y0 * row_bytes, a->staging.p, (size_t)(y1 - y0) * row_bytes, hipMemcpyDeviceToHost, a->stream));
            }
        }
        sync_checked(*a);
    });
    lg_accel_free(a);
    return rc;
}

int lg_capture(const lg_scene *s, lg_film *film) { // lib.rs:55-104: the BVH is (re)built inside every capture
    CallScope call; // (one API call, whatever its bands, shares and host threads launch)
    // The reference splits the film over `scene.threads` CPU threads, 0 = all cores (lib.rs:58-62); here the film is
    // split over devices: the ones named with lg_set_devices, the one named with lg_set_device, or -- a process that
    // named none -- EVERY visible device, capped by `scene.threads` when that is non-zero.  Pixels are independent,
    // so the film is the same for any split.
    std::vector<int> devs = g_devices;
    if (devs.empty() && !g_device_chosen) {
        // the default: every visible device -- for films of 2^18 pixels and more.  A smaller film is a millisecond of work on one
        // GPU; building an accel per device and a communicator over them for it would cost seconds (and every rank of a
        // torchrun job that never named a device would do so on every GPU of the node): it stays on the HIP CURRENT device,
        // which is also what a caller that chose its device through the HIP runtime itself expects
        int n_vis = 0, cur = 0;
        if ((unsigned long long)film->w * film->h >= (1ull << 18)) {
            if (hipGetDeviceCount(&n_vis) == hipSuccess)
                for (int d = 0; d < n_vis; ++d) devs.push_back(d);
        } else if (hipGetDevice(&cur) == hipSuccess) devs.push_back(cur);
    }
    if (s->s.threads != 0 && devs.size() > s->s.threads) devs.resize(s->s.threads);
    if (devs.size() <= 1) {
        lg_accel *a = accel_from_on(s, devs.empty() ? g_device : devs[0]);
        if (!a) return 1;
        int rc = lg_capture_subset(0, 1, a, film);
        lg_accel_free(a);
        return rc;
    }
    const uint32_t n = (uint32_t)devs.size();
    const uint32_t block_rows = film->h % (64u * n) == 0 ? 64u : 0u;
    {   // more than one DISTINCT device: the shares are gathered on the first device over xGMI (one grouped RCCL exchange,
        // multi.cpp) and the film leaves the node with one D2H copy -- the north-star's gather, behind the reference's capture()
        bool distinct = false;
        for (uint32_t r = 1; r < n; ++r) distinct = distinct || devs[r] != devs[0];
        if (distinct && !std::getenv("LASGUN_CAPTURE_NO_RCCL")) {
            // RCCL missing or failing (no librccl, ncclCommInitAll refused, an exchange error) must not fail a capture
            // that the RCCL-free path below can serve: say why under LASGUN_DEBUG and go on
            lg_multi *m = lg_multi_create(s, devs.data(), (int)n, block_rows);
            int rc = m ? lg_multi_capture(m, film) : 1;
            if (m) { std::string e = rc ? tl_error : std::string(); lg_multi_free(m); if (rc) tl_error = e; }
            if (rc == 0) return 0;
            static bool told = false;
            if (!told && std::getenv("LASGUN_DEBUG")) {
                told = true;
                std::fprintf(stderr, "[lasgun] lg_capture: the RCCL gather is unavailable (%s); using one host thread and one D2H copy per device\n", tl_error.c_str());
            }
        }
    }
    std::vector<int> rcs(n, 0);
    std::vector<std::string> errs(n);
    std::vector<std::thread> workers;
    for (uint32_t r = 0; r < n; ++r)
        workers.emplace_back([&, r] { // one host thread per device, like the reference's one thread per core
            rcs[r] = capture_share(s, film, devs[r], n, r, block_rows);
            if (rcs[r]) errs[r] = tl_error;
        });
    for (auto &t : workers) t.join();
    for (uint32_t r = 0; r < n; ++r)
        if (rcs[r]) return fail("device " + std::to_string(devs[r]) + ": " + errs[r]);
    return 0;
}
lg_film *lg_render(const lg_scene *s, uint32_t w, uint32_t h) { // lib.rs:46-50
    lg_film *f = lg_film_new(w, h);
    if (lg_capture(s, f)) { lg_film_free(f); return nullptr; }
    return f;
}

int lg_capture_radiance(size_t k, size_t n, const lg_accel *a, uint32_t w, uint32_t h, double *rgb) {
    return guarded([&] {
        if (n == 0) throw Error("n must be > 0");
        std::lock_guard<std::mutex> g(a->mtx);
        use_device(a->device);
        size_t count = (size_t)w * h * 3;
        if (a->staging_rad.n < count) { sync_checked(*a); a->staging_rad.alloc(count); }
        HIP_TRY(hipMemcpyAsync(a->staging_rad.p, rgb, count * 8, hipMemcpyHostToDevice, a->stream));
        DParams P = base_params(*a, w, h);
        if (n == 1 && k == 0) set_rect(P, 0, 0, w, h);
        else set_subset(*a, a->stream, P, k, n, w, h);
        P.out_row0 = 0;
        P.out_radiance = a->staging_rad.p;
        enqueue(*a, P, false, a->stream);
        HIP_TRY(hipMemcpyAsync(rgb, a->staging_rad.p, count * 8, hipMemcpyDeviceToHost, a->stream));
        sync_checked(*a);
    });
}
static int capture_stats_impl(const lg_accel *a, uint32_t w, uint32_t h, uint32_t y0, uint32_t y1, uint32_t filter, lg_stats *out);
int lg_capture_stats(const lg_accel *a, uint32_t w, uint32_t h, uint32_t y0, uint32_t y1, lg_stats *out) {
    return capture_stats_impl(a, w, h, y0, y1, 0u, out);
}
int lg_capture_stats_kind(const lg_accel *a, uint32_t w, uint32_t h, uint32_t y0, uint32_t y1, int kind, lg_stats *out) {
    if (kind < 0 || kind > 2) return fail("kind: 0 all, 1 closest-hit traversals, 2 shadow traversals");
    return capture_stats_impl(a, w, h, y0, y1, (uint32_t)kind, out);
}
static int capture_stats_impl(const lg_accel *a, uint32_t w, uint32_t h, uint32_t y0, uint32_t y1, uint32_t filter, lg_stats *out) {
    return guarded([&] {
        if (y1 > h || y0 > y1) throw Error("bad row range");
        std::lock_guard<std::mutex> g(a->mtx);
        use_device(a->device);
        DParams P = base_params(*a, w, h);
        set_rect(P, 0, y0, w, y1);
        P.out_row0 = y0;
        P.stats_filter = filter;
        enqueue(*a, P, true, a->stream);
        DStats s;
        HIP_TRY(hipMemcpyAsync(&s, a->stats.p, sizeof s, hipMemcpyDeviceToHost, a->stream));
        sync_checked(*a);
        *out = lg_stats{s.primary_rays, s.shadow_rays, s.secondary_rays, s.nodes_tested, s.spheres_tested, s.cuboids_tested,
                        s.triangles_tested, s.accel_entries, s.hits};
    });
}

int lg_audit_prune(const lg_accel *a, uint32_t w, uint32_t h, uint32_t y0, uint32_t y1, lg_prune_audit *out) {
    return guarded([&] {
        if (y1 > h || y0 > y1) throw Error("bad row range");
        if (!out) throw Error("out is NULL");
        std::lock_guard<std::mutex> g(a->mtx);
        if (a->fast) throw Error("the audit is of the pruned REFERENCE walk: not available in fast mode");
        use_device(a->device);
        if (!a->flat.has_records) { // the audit is of the pruned walk as it runs when it is on: with the mesh leaves' records
            const int before = a->prune;
            a->prune = 1;
            try { rebuild_tables(a, false); } catch (...) { a->prune = before; throw; }
            a->prune = before;
        }
        DParams P = base_params(*a, w, h);
        set_rect(P, 0, y0, w, y1);
        P.out_row0 = y0;
        P.prune = 1u; P.audit = std::getenv("LASGUN_AUDIT_SABOTAGE") ? 2u : 1u; // (2: the walk skips by an unsound rule on purpose -- the audit's self-test)
        enqueue(*a, P, true, a->stream);
        DStats s;
        HIP_TRY(hipMemcpyAsync(&s, a->stats.p, sizeof s, hipMemcpyDeviceToHost, a->stream));
        sync_checked(*a);
        auto slack = [](unsigned long long stored) { // (complemented bits, 0 = no sample: k_mega.hip)
            if (stored == 0ull) return (double)INFINITY;
            const unsigned long long bits = ~stored;
            double v; std::memcpy(&v, &bits, sizeof v);
            return v;
        };
        double used = 0.0;
        std::memcpy(&used, &s.audit_used_nodes, sizeof used);
        *out = lg_prune_audit{s.audit_nodes, s.audit_runs, s.audit_prims, s.audit_violations, slack(s.audit_slack_nodes), slack(s.audit_slack_runs), used};
    });
}
int lg_accel_set_lds_scene(const lg_accel *a, int enabled) {
    std::lock_guard<std::mutex> lk(a->mtx);
    a->lds_scene = enabled != 0;
    return a->ldss_blocks ? 1 : 0; // 1: the scene's tables fit in LDS (the variant exists for this accel)
}
int lg_accel_set_wf_split(const lg_accel *a, int bands) {
    if (bands < 0 || bands > 8) return fail("bands must be 0 (default) .. 8 (more than 4 are rendered as 4)");
    std::lock_guard<std::mutex> g(a->mtx);
    a->wf_split = (unsigned)bands;
    return 0;
}
int lg_accel_set_streaming(const lg_accel *a, int enabled) {
    std::lock_guard<std::mutex> g(a->mtx);
    if (enabled < 0 || enabled > 3) return fail("streaming must be 0 (megakernel), 1 (default), 2 (wavefront pipeline) or 3 (queue organisation)");
    a->streaming = enabled != 0;
    a->streaming_forced = enabled == 2; // 2 = use it whatever the scene and the launch size (tests)
    a->queue = enabled == 3 ? 1 : enabled == 1 ? -1 : 0; // 3 = the queue organisation whatever the scene; 1 = the accel's defaults
    return 0;
}
int lg_accel_set_tile_order(const lg_accel *a, int order) { // the direction the megakernel and the queue organisation claim a launch's tiles in
    std::lock_guard<std::mutex> g(a->mtx);
    if (order < -1 || order > 2) return fail("tile order must be -1 (default: measured), 0 (top-down), 1 (bottom-up) or 2 (from the middle outwards)");
    a->tile_order = order;
    return 0;
}
int lg_accel_set_tile_parts(const lg_accel *a, int parts) { // the megakernel: a tile handed out whole or in parts of 64 / parts lanes
    std::lock_guard<std::mutex> g(a->mtx);
    if (!(parts == -1 || parts == 1 || parts == 2 || parts == 4 || parts == 8)) return fail("tile parts must be -1 (default), 1, 2, 4 or 8");
    a->tile_parts = parts;
    return 0;
}
int lg_accel_set_sample_order(const lg_accel *a, int order) { // a supersampled pixel's samples: side by side in one launch chain, or one after the other
    std::lock_guard<std::mutex> g(a->mtx);
    if (order < -1 || order > 1) return fail("sample order must be -1 (default), 0 (side by side) or 1 (one after the other)");
    a->sample_order = order;
    return 0;
}
int lg_accel_last_organisation(const lg_accel *a) { // what the accel's last launch ran as: 0 megakernel, 1 level by level, 2 queue, + 16 when its tiles were claimed bottom-up; -1 before the first
    std::lock_guard<std::mutex> g(a->mtx);
    return a->last_org;
}
int lg_accel_get_prune(const lg_accel *a) { // the EFFECTIVE setting of the pruned walk: what a render of this accel uses right now
    std::lock_guard<std::mutex> g(a->mtx);
    return (int)base_params(*a, 8, 8).prune;
}
int lg_accel_set_prune(const lg_accel *a, int enabled) {
    if (enabled < -1 || enabled > 1) return fail("prune must be -1 (default), 0 or 1");
    std::lock_guard<std::mutex> g(a->mtx);
    const int before = a->prune;
    a->prune = enabled;
    if (enabled == 1 && !a->flat.has_records) { // the mesh leaves' culling records were left out of this accel's tables (a small mesh): build them now
        int rc = guarded([&] { rebuild_tables(a, false); });
        if (rc) { a->prune = before; return rc; }
    }
    return 0;
}
int lg_accel_set_mode(const lg_accel *a, int mode) {
    if (mode != 0 && mode != 1) return fail("mode must be 0 (reference traversal) or 1 (fast)");
    std::lock_guard<std::mutex> g(a->mtx);
    if (mode == 1) {
        int rc = guarded([&] { rebuild_tables(a, true); });
        if (rc) return rc;
    }
    if (mode == 1 && !a->fast_available) return fail(a->fast_refusal);
    a->fast = mode == 1;
    return 0;
}
void lg_profile_enable(const lg_accel *a, int enabled) {
    std::lock_guard<std::mutex> g(a->mtx);
    a->profiling = enabled != 0;
}
int lg_profile_read(const lg_accel *a, double *total_ms, uint64_t *launches) {
    return guarded([&] {
        std::lock_guard<std::mutex> g(a->mtx);
        double total = 0.0;
        for (auto &e : a->events) {
            HIP_TRY(hipEventSynchronize(e.second));
            float ms = 0.f;
            HIP_TRY(hipEventElapsedTime(&ms, e.first, e.second));
            total += ms;
            (void)hipEventDestroy(e.first); (void)hipEventDestroy(e.second);
        }
        *total_ms = total;
        *launches = a->events.size();
        a->events.clear();
    });
}

int lg_profile_read_kinds(const lg_accel *a, double ms[5], uint64_t launches[5]) {
    return guarded([&] {
        std::lock_guard<std::mutex> g(a->mtx);
        for (int k = 0; k < 5; ++k) {
            double total = 0.0;
            for (auto &e : a->kind_events[k]) {
                HIP_TRY(hipEventSynchronize(e.second));
                float t = 0.f;
                HIP_TRY(hipEventElapsedTime(&t, e.first, e.second));
                if (std::getenv("LASGUN_DEBUG")) std::fprintf(stderr, "[lasgun] kind %d launch: %.3f ms\n", k, t);
                total += t;
                (void)hipEventDestroy(e.first); (void)hipEventDestroy(e.second);
            }
            ms[k] = total; launches[k] = a->kind_events[k].size();
            a->kind_events[k].clear();
        }
    });
}
int lg_accel_dump(const lg_accel *a, const double **f, size_t *nf, const int64_t **i, size_t *ni) {
    *f = a->flat.dump_f.data(); *nf = a->flat.dump_f.size();
    *i = a->flat.dump_i.data(); *ni = a->flat.dump_i.size();
    return 0;
}
// Host-only: run the HLBVH build + flattening of `scene` WITHOUT touching a device and hand back
// the structure dump (same format as lg_accel_dump).  Lets CPU-only tests check the host builder.
int lg_host_build_dump(const lg_scene *s, const double **f, size_t *nf, const int64_t **i, size_t *ni, uint64_t info[8]) {
    static thread_local FlatScene flat;
    return guarded([&] {
        flatten_scene(s->s, flat);
        *f = flat.dump_f.data(); *nf = flat.dump_f.size();
        *i = flat.dump_i.data(); *ni = flat.dump_i.size();
        info[0] = flat.nodes.size(); info[1] = flat.primref.size(); info[2] = flat.spheres.size(); info[3] = flat.cuboids.size();
        info[4] = flat.tri_v.size() / 3; info[5] = flat.accels.size(); info[6] = flat.max_stack; info[7] = flat.has_specular ? 1 : 0;
    });
}
// Host-only self-check of the fast mode's wide records (DNode4) against the binary fast trees they were collapsed from:
// out[0] records, out[1] children, out[2] leaves reached, out[3] the deepest stack a walk that pushes every child but one would need
// (frames of nested accels not counted), out[4] violations (a child box that does not contain its node's f64 box, a leaf reached
// twice or never, a link that is not a node of the tree), out[5] = FlatScene::max_stack_fast1.
// Host-only self-check of the triangle strips (DStrip) against the leaves they are made from: out = { mesh leaves with records, runs,
// triangles, strip entries, violations, hash of the records and the leaves' pad words, hash of the strips, 0 }.  Every triangle slot of a mesh leaf with records must come up in exactly one
// run, exactly once, as a STRIP_TRI entry whose three vertices (the two entries before it and its own) are the slot's three
// vertices in some order; the entry counts must match the run records.
int lg_host_check_strips(const lg_scene *s, uint64_t out[8]) {
    return guarded([&] {
        FlatScene flat;
        flatten_scene(s->s, flat, false);
        for (int k = 0; k < 8; ++k) out[k] = 0;
        std::vector<char> seen_node(flat.nodes.size(), 0);
        for (const DAccel &A : flat.accels) {
            if (!(A.flags & AF_MESH)) continue;
            std::vector<uint32_t> todo{0};
            while (!todo.empty()) {
                const uint32_t nidx = todo.back(); todo.pop_back();
                if (seen_node[A.node_base + nidx]) continue;
                seen_node[A.node_base + nidx] = 1;
                const DNode &nd = flat.nodes[A.node_base + nidx];
                if (!(nd.meta & NODE_LEAF)) { todo.push_back(nidx + 1); todo.push_back(nd.link); continue; }
                const uint32_t nrec = nd.pad >> 24, rec0 = nd.pad & 0x00FFFFFFu;
                if (nrec == 0) continue; // (a leaf without records: walked in the reference's order)
                out[0]++;
                const size_t first = (size_t)A.prim_base + nd.link, count = nd.meta & 0xFFFFu;
                std::vector<int> hits(count, 0);
                for (uint32_t r = rec0; r < rec0 + nrec; ++r) {
                    const DChunk &k = flat.chunks[r];
                    if (k.start == CHUNK_IS_GROUP) continue;
                    out[1]++;
                    const uint32_t ntri = k.count & 0xFFu, nent = k.count >> 8;
                    uint32_t tris = 0;
                    for (uint32_t e = k.pad; e < k.pad + nent; ++e) {
                        out[3]++;
                        const DStrip &E = flat.strips[e];
                        if (!(E.code & STRIP_TRI)) continue;
                        ++tris; out[2]++;
                        const uint32_t slot = E.code & STRIP_SLOT_MASK;
                        if (slot < first || slot >= first + count || e < k.pad + 2) { out[4]++; continue; }
                        hits[slot - first]++;
                        uint32_t want[3][3], got[3][3];
                        std::memcpy(want, flat.leaf_soup[slot].w, 36);
                        std::memcpy(got[0], &flat.strips[e - 2].x, 12); std::memcpy(got[1], &flat.strips[e - 1].x, 12); std::memcpy(got[2], &E.x, 12);
                        bool used[3] = {false, false, false};
                        int matched = 0;
                        for (int i = 0; i < 3; ++i)
                            for (int j = 0; j < 3; ++j)
                                if (!used[j] && std::memcmp(got[i], want[j], 12) == 0) { used[j] = true; ++matched; break; }
                        if (matched != 3) out[4]++;
                    }
                    if (tris != ntri) out[4]++;
                }
                for (size_t i = 0; i < count; ++i) if (hits[i] != 1) out[4]++;
            }
        }
        // FNV-1a over the tables as they would be uploaded (the threaded build must give what one thread gives) and over the leaves' pad words
        auto fnv = [](uint64_t h, const void *p, size_t n) { const unsigned char *b = static_cast<const unsigned char *>(p); for (size_t i = 0; i < n; ++i) { h ^= b[i]; h *= 1099511628211ull; } return h; };
        uint64_t h = 14695981039346656037ull;
        h = fnv(h, flat.chunks.data(), flat.chunks.size() * sizeof(DChunk));
        for (const DNode &nd : flat.nodes) h = fnv(h, &nd.pad, sizeof nd.pad);
        out[5] = h;
        out[6] = fnv(14695981039346656037ull, flat.strips.data(), flat.strips.size() * sizeof(DStrip));
    });
}
int lg_host_check_wide_records(const lg_scene *s, uint64_t out[8]) {
    return guarded([&] {
        FlatScene flat;
        flatten_scene(s->s, flat, true);
        for (int k = 0; k < 8; ++k) out[k] = 0;
        out[5] = flat.max_stack_fast1;
        std::vector<uint32_t> seen_tree;
        for (const DAccel &A : flat.accels) {
            if (std::find(seen_tree.begin(), seen_tree.end(), A.fnode_base) != seen_tree.end()) continue; // (mesh instances share a tree)
            seen_tree.push_back(A.fnode_base);
            // the binary tree: its nodes (children follow their parent; DNode::link = second child) and its leaves by first slot
            std::map<uint32_t, uint32_t> leaf_by_start; // first slot -> node
            std::set<uint32_t> interior;
            std::vector<uint32_t> bin{0u};
            while (!bin.empty()) {
                const uint32_t i = bin.back(); bin.pop_back();
                const DNode &d = flat.nodes[A.fnode_base + i];
                if (d.meta & NODE_LEAF) leaf_by_start[d.link] = i;
                else { interior.insert(i); bin.push_back(i + 1u); bin.push_back(d.link); }
            }
            if (flat.nodes[A.fnode_base].meta & NODE_LEAF) continue; // a single leaf: no record
            std::map<uint32_t, int> reached;
            struct It { uint32_t node, depth; };
            std::vector<It> st{{0u, 0u}};
            auto contains = [](const float box[6], const DNode &d) {
                for (int a = 0; a < 3; ++a)
                    if (!((double)box[a] <= d.bmin[a]) || !((double)box[3 + a] >= d.bmax[a])) return false;
                return true;
            };
            while (!st.empty()) {
                const It it = st.back(); st.pop_back();
                const DNode4 &w = flat.nodes4[A.fnode_base + it.node];
                out[0]++;
                uint32_t k = 0;
                for (int c = 0; c < WIDE; ++c) k += w.link[c] != NO_HIT ? 1u : 0u;
                if (k < 2u) out[4]++;
                out[3] = std::max<uint64_t>(out[3], it.depth + k - 1u);
                for (int c = 0; c < WIDE; ++c) {
                    if (w.link[c] == NO_HIT) continue;
                    out[1]++;
                    uint32_t node;
                    if (w.link[c] & WIDE_LEAF) {
                        const uint32_t start = w.link[c] & WIDE_START_MASK, cnt = (w.link[c] >> WIDE_COUNT_SHIFT) & 7u;
                        auto f = leaf_by_start.find(start);
                        if (f == leaf_by_start.end() || (flat.nodes[A.fnode_base + f->second].meta & 0xFFFFu) != cnt) { out[4]++; continue; }
                        node = f->second;
                        if (reached[start]++) out[4]++;
                        out[2]++;
                    } else {
                        node = w.link[c];
                        if (!interior.count(node)) { out[4]++; continue; }
                        st.push_back({node, it.depth + k - 1u});
                    }
                    if (!contains(w.box[c], flat.nodes[A.fnode_base + node])) out[4]++;
                }
            }
            if (reached.size() != leaf_by_start.size()) out[4]++;
        }
    });
}
int lg_accel_info(const lg_accel *a, uint64_t out[8]) {
    const FlatScene &f = a->flat;
    out[0] = f.nodes.size(); out[1] = f.primref.size(); out[2] = f.spheres.size(); out[3] = f.cuboids.size();
    out[4] = f.tri_v.size() / 3; out[5] = f.accels.size(); out[6] = a->fast ? a->stack_depth_fast1 : a->stack_depth; out[7] = a->device_bytes;
    return 0;
}

int lg_kat_intersect(int kind, const double *params, const char *obj_text, size_t obj_len, const double o[3], const double d[3], double out[8]) {
    return guarded([&] {
        use_device();
        DevBuf<double> dparams, dout;
        DevBuf<float> dpos;
        DevBuf<uint32_t> dtri;
        std::vector<double> pv(params, params + 8);
        dparams.upload(pv);
        dout.alloc(8);
        uint32_t ntri = 0;
        if (kind == 2) {
            Obj obj;
            parse_obj_text(obj_text, obj_len, obj);
            std::vector<uint32_t> tv;
            for (auto &t : obj.tri) tv.push_back(t.v);
            ntri = (uint32_t)(tv.size() / 3);
            dpos.upload(obj.position);
            dtri.upload(tv);
        } else if (kind != 0 && kind != 1) throw Error("bad kind");
        HIP_TRY(launch_kat(kind, dparams.p, dpos.p, dtri.p, ntri, V3{o[0], o[1], o[2]}, V3{d[0], d[1], d[2]}, dout.p, nullptr));
        HIP_TRY(hipMemcpy(out, dout.p, 8 * sizeof(double), hipMemcpyDeviceToHost));
    });
}
int lg_kat_surface_interaction(const double o[3], const double d[3], double t, const double dpdu[3], const double dpdv[3], double out_ng[3]) {
    return guarded([&] {
        use_device();
        DevBuf<double> dout;
        dout.alloc(3);
        HIP_TRY(launch_kat_si(V3{o[0], o[1], o[2]}, V3{d[0], d[1], d[2]}, t, V3{dpdu[0], dpdu[1], dpdu[2]}, V3{dpdv[0], dpdv[1], dpdv[2]}, dout.p, nullptr));
        HIP_TRY(hipMemcpy(out_ng, dout.p, 3 * sizeof(double), hipMemcpyDeviceToHost));
    });
}
int lg_trace_pixel(const lg_accel *a, uint32_t w, uint32_t h, uint32_t x, uint32_t y, int fast, double *out, size_t out_len) {
    return guarded([&] {
        if (x >= w || y >= h) throw Error("pixel outside the film");
        std::lock_guard<std::mutex> g(a->mtx);
        use_device(a->device);
        const size_t need = 7 + 2 * a->flat.lights.size();
        if (out_len < need) throw Error("output too small: 7 + 2 * lights doubles");
        if (fast) rebuild_tables(a, true);
        if (fast && !a->fast_available) throw Error(a->fast_refusal);
        DParams P = base_params(*a, w, h);
        DevBuf<double> dout, dlog;
        dout.alloc(need);
        HIP_TRY(hipMemset(dout.p, 0, need * sizeof(double)));
        const size_t log_n = 1 + 4 * 4000;
        if (out_len >= need + log_n) { dlog.alloc(log_n); HIP_TRY(hipMemset(dlog.p, 0, log_n * sizeof(double))); P.dbg_log = dlog.p; }
        HIP_TRY(launch_trace_pixel(P, fast != 0, fast ? a->stack_depth_fast1 : a->stack_depth, x, y, dout.p, a->stream));
        HIP_TRY(hipMemcpyAsync(out, dout.p, need * sizeof(double), hipMemcpyDeviceToHost, a->stream));
        if (P.dbg_log) HIP_TRY(hipMemcpyAsync(out + need, dlog.p, log_n * sizeof(double), hipMemcpyDeviceToHost, a->stream));
        sync_checked(*a);
    });
}
// Measured rates of the current device, GB/s: what 0 = HBM copy (16 B per lane, 1 GiB each way, read + written bytes),
// 1 = aggregate LDS read rate (ds_read_b128, every CU streaming).  The roofline's measured denominators (bench.py).
int lg_probe_rate(int what, double *gbps) {
    return guarded([&] {
        use_device();
        if (what != 0 && what != 1) throw Error("what: 0 = HBM copy, 1 = LDS read");
        hipEvent_t e0, e1;
        HIP_TRY(hipEventCreate(&e0)); HIP_TRY(hipEventCreate(&e1));
        double best = 0.0;
        if (what == 0) {
            const size_t bytes = 1ull << 30;
            DevBuf<uint8_t> a, b;
            a.alloc(bytes); b.alloc(bytes);
            HIP_TRY(hipMemset(a.p, 1, bytes)); HIP_TRY(hipMemset(b.p, 2, bytes));
            for (int rep = 0; rep < 5; ++rep) {
                HIP_TRY(hipEventRecord(e0, nullptr));
                HIP_TRY(launch_probe_copy(a.p, b.p, bytes, nullptr));
                HIP_TRY(hipEventRecord(e1, nullptr));
                HIP_TRY(hipEventSynchronize(e1));
                float ms = 0.f;
                HIP_TRY(hipEventElapsedTime(&ms, e0, e1));
                if (rep > 0 && ms > 0.f) best = std::max(best, 2.0 * (double)bytes / (ms * 1e-3) / 1e9);
            }
        } else {
            int cus = 0, dev = 0;
            HIP_TRY(hipGetDevice(&dev));
            HIP_TRY(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
            DevBuf<uint32_t> sink;
            sink.alloc((size_t)cus);
            const uint32_t iters = 4096;
            for (int rep = 0; rep < 4; ++rep) {
                HIP_TRY(hipEventRecord(e0, nullptr));
                HIP_TRY(launch_probe_lds((uint32_t)cus, iters, sink.p, nullptr));
                HIP_TRY(hipEventRecord(e1, nullptr));
                HIP_TRY(hipEventSynchronize(e1));
                float ms = 0.f;
                HIP_TRY(hipEventElapsedTime(&ms, e0, e1));
                const double bytes = (double)cus * 1024.0 * 16.0 * 16.0 * iters;
                if (rep > 0 && ms > 0.f) best = std::max(best, bytes / (ms * 1e-3) / 1e9);
            }
        }
        (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
        *gbps = best;
    });
}
int lg_math_eval(int op, size_t n, const double *a, const double *b, double *out) {
    return guarded([&] {
        use_device();
        if (op < 0 || op > 8) throw Error("bad op");
        DevBuf<double> da, db, dout;
        std::vector<double> va(a, a + n), vb(b, b + n);
        da.upload(va); db.upload(vb); dout.alloc(n);
        HIP_TRY(launch_math(op, n, da.p, db.p, dout.p, nullptr));
        HIP_TRY(hipMemcpy(out, dout.p, n * sizeof(double), hipMemcpyDeviceToHost));
    });
}

} // extern "C"

// Diagnostic (tools/queue_levels.py): the packets each recursion level of the queue organisation's LAST launch on `hip_stream` held
// (QC_COUNT of its control block) -- with the rays per level this says how full the 64-ray packets of the deeper levels are.
extern "C" int lg_debug_queue_packets(const lg_accel *a, void *hip_stream, unsigned long long out[8]) {
    return guarded([&] {
        std::lock_guard<std::mutex> g(a->mtx);
        use_device(a->device);
        HIP_TRY(hipDeviceSynchronize());
        for (int d = 0; d < 8; ++d) out[d] = 0;
        for (auto &c : a->ctxs) {
            if (c->key != (hipStream_t)hip_stream || c->wf_counters.n < QC_WORDS) continue;
            std::vector<uint32_t> w(QC_WORDS);
            HIP_TRY(hipMemcpy(w.data(), c->wf_counters.p, QC_WORDS * sizeof(uint32_t), hipMemcpyDeviceToHost));
            for (uint32_t d = 0; d < QC_MAX_LEVELS; ++d) out[d] = w[QC_LEVEL0 + QC_LEVEL_WORDS * d + QC_COUNT];
#ifdef LG_QIDLE // diagnostic build: [7] = the waves' idle time (100 MHz ticks, summed), k_queue.hip
            out[7] = (unsigned long long)w[QC_ERROR] | ((unsigned long long)w[QC_ERROR + 1] << 32);
#endif
        }
    });
}

#if defined(LG_PKT_STATS) || defined(LG_STAMPS) || defined(LG_QIDLE)
extern "C" int lg_debug_stats(const lg_accel *a, int clear, unsigned long long *out9) { // analysis builds only
    return guarded([&] {
        use_device(a->device);
        HIP_TRY(hipDeviceSynchronize());
        if (clear) HIP_TRY(hipMemset(a->stats.p, 0, 2 * sizeof(DStats)));
        else HIP_TRY(hipMemcpy(out9, a->stats.p, 2 * sizeof(DStats), hipMemcpyDeviceToHost));
    });
}
#endif
