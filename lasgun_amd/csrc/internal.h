// lasgun_amd/csrc/internal.h -- what the host-side units of liblasgun_hip share (round 6: capi.cpp was one 2,600-line file):
//   devmem.cpp  the process-wide pools (device blocks, streams, pinned staging, error words), the current device, the last error
//   accel.cpp   lg_accel_from: flatten, LDS images, upload; the tables rebuilt with what was left out of them
//   launch.cpp  one render enqueued: launch contexts, the three organisations, the rule and the measured choice (tune.cpp), addressing modes
//   capi.cpp    the C ABI of include/lasgun_hip.h over them
// No torch types, no C++ exceptions across the boundary, no CPU render path.
#pragma once
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

extern thread_local std::string tl_error; // devmem.cpp
extern int g_device;
extern bool g_device_chosen; // lg_set_device was called: single-device captures stay on that device
extern std::vector<int> g_devices; // lg_set_devices: the devices a host-film lg_capture is split over (empty = g_device)

inline int fail(const std::string &msg) {
    tl_error = msg;
    return 1;
}
#define HIP_TRY(expr)                                                                                                   \
    do {                                                                                                                \
        hipError_t _e = (expr);                                                                                         \
        if (_e != hipSuccess) throw Error(std::string(#expr) + ": " + hipGetErrorString(_e));                           \
    } while (0)

void use_device(int dev); // devmem.cpp
void use_device();

// Device allocations of 1 KiB and more are recycled through a small per-process pool (at most 4 GiB parked per
// device): capture() builds and drops an accel -- film staging, per-pixel state -- for every frame, like the
// reference, and hipMalloc / hipFree of those buffers would otherwise cost about a millisecond of each frame.
// Nothing relies on the contents of a fresh buffer: every buffer is written (kernel, memset or copy) before it is read.
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
extern DevPool &g_pool; // devmem.cpp

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
extern StreamPool &g_streams; // never destroyed (see g_pool)

// Pinned host staging for the small tables a *_device entry point uploads (a batch's k table, a lattice row table): the copy is
// enqueued on the caller's stream from memory that stays put until the copy is through, so the call only enqueues (round 5 used a blocking
// hipMemcpy, and a device-wide synchronise when a row table changed: ADVICE r5).  Blocks are powers of two, recycled per process.
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
extern PinnedPool &g_pinned; // never destroyed (see g_pool)
struct PinnedBuf {
    void *p = nullptr;
    size_t cap = 0;
    PinnedBuf() = default;
    PinnedBuf(const PinnedBuf &) = delete;
    PinnedBuf &operator=(const PinnedBuf &) = delete;
    void need(size_t bytes) { if (cap < bytes) { g_pinned.give(cap, p); p = nullptr; cap = 0; p = g_pinned.take(bytes, &cap); } }
    ~PinnedBuf() { g_pinned.give(cap, p); }
};


// Sticky error words of the queue organisation (k_queue.hip: a wave that gave up waiting for work).  They live in PINNED HOST memory
// that every device writes straight into (system-scope store), one word per accel, handed out from pages of 1024: the host reads
// a word without a HIP call -- after any synchronise, at the head of every enqueue, in lg_accel_synchronize -- and only the host
// ever clears it, after it has reported it.  (Round 4 kept the word among the per-launch control words: the memset before the
// next chunk or supersample erased it, and a launch on a caller's stream was looked at before it had finished -- ADVICE r4.)
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
extern ErrWords &g_err_words; // never destroyed (see g_pool)

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
constexpr size_t PRUNE_MIN_TRIS = 4096; // the pruned walk is the default from this many triangles in a mesh (accel.cpp; profiles/r05_prune_threshold.jsonl)

// ---- launch.cpp: one render enqueued (callers hold a.mtx and have made the accel's device current)
void check_queue_error(const lg_accel &a);
void sync_checked(const lg_accel &a);
DParams base_params(const lg_accel &a, uint32_t w, uint32_t h);
void ensure_aux_streams(const lg_accel &a, unsigned n);
void enqueue(const lg_accel &a, DParams &P, bool stats, hipStream_t stream);
void set_rect(DParams &P, uint32_t x0, uint32_t y0, uint32_t x1, uint32_t y1);
unsigned long long subset_count(unsigned long long area, unsigned long long k, unsigned long long n);
void set_subset(const lg_accel &a, hipStream_t stream, DParams &P, size_t k, size_t n, uint32_t w, uint32_t h);
// Several subsets {k_j + i*n} of one n as ONE render (lg_capture_subsets): the k values sorted and without repeats or empty subsets,
// the periods the longest subset has, and whether the batch is every pixel of the film (every k of 0 .. n-1: the frame itself).
struct SubsetBatch {
    std::vector<unsigned long long> ks;
    unsigned long long n = 1, periods = 0, items = 0;
    bool whole = false;
};
SubsetBatch make_batch(const size_t *ks, size_t count, size_t n, uint32_t w, uint32_t h);
void set_subsets(const lg_accel &a, DParams &P, const SubsetBatch &b, hipStream_t stream);
void subsets_enqueued(const lg_accel &a, hipStream_t stream);
void subsets_abandoned(const lg_accel &a, hipStream_t stream);
// Which API call is running (the measured choice counts calls, not launches: tune.h): a scope at every entry from outside
struct CallScope {
    CallScope();
    ~CallScope();
};
template <class F> int guarded(F f) {
    try {
        CallScope call;
        f();
        return 0;
    } catch (const std::exception &e) {
        return fail(e.what());
    }
}

// ---- accel.cpp
void rebuild_tables(const lg_accel *ca, bool fast);
lg_accel *accel_from_on(const lg_scene *s, int device);
