// lasgun_amd/csrc/multi.cpp -- one film on several GPUs of ONE process, gathered over xGMI with RCCL.
//
// The reference's capture() fans the film out over `scene.threads` CPU threads and joins them
// (/root/reference/src/lib.rs:55-104).  This is that fan-out on a node of MI355X: every device holds its own
// copy of the scene (KB .. tens of MB) and renders a share of the film -- 64-row blocks dealt round-robin,
// or contiguous row tiles -- into a tile in its own HBM; then ONE grouped RCCL exchange (ncclSend on every
// other device, ncclRecv on the root, all inside one ncclGroupStart / ncclGroupEnd) moves the tiles over
// xGMI straight into the film on the root device, each piece at its final offset.  The
// root's own share is rendered in place.  xGMI is point to point: the n-1 sends use n-1 different links
// into the root, 4*w*h/n bytes each (8 MiB per device at 4096^2 on 8 GPUs).
//
// Single process, single host thread: ncclCommInitAll builds one communicator per distinct device; every
// device works on its accel's own HIP stream, RCCL calls are enqueued on those streams, so a send starts
// when its tile is rendered and the root's receives overlap its own rendering.  RCCL is loaded with dlopen
// when the first multi-device capture is created: single-device users never load it, and a process that
// already carries a RCCL (PyTorch) shares that copy.
//
// Only the C ABI of include/lasgun_hip.h is used here (accels, device entry points, streams).
#include <dlfcn.h>
#include <hip/hip_runtime_api.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "../../include/lasgun_hip.h"

extern "C" void lg_set_last_error(const char *msg); // capi.cpp

namespace {

// ---- the RCCL entry points this file needs (rccl.h: ncclResult_t is an int enum, ncclUint8 = 1) ----
typedef void *comm_t;
struct Rccl {
    void *lib = nullptr;
    int (*CommInitAll)(comm_t *, int, const int *) = nullptr;
    int (*CommDestroy)(comm_t) = nullptr;
    int (*GroupStart)() = nullptr;
    int (*GroupEnd)() = nullptr;
    int (*Send)(const void *, size_t, int, int, comm_t, hipStream_t) = nullptr;
    int (*Recv)(void *, size_t, int, int, comm_t, hipStream_t) = nullptr;
    int (*AllGather)(const void *, void *, size_t, int, comm_t, hipStream_t) = nullptr;
    const char *(*GetErrorString)(int) = nullptr;
    std::string error;
    bool load() {
        if (lib) return true;
        for (const char *name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
            lib = dlopen(name, RTLD_NOW | RTLD_LOCAL);
            if (lib) break;
        }
        if (!lib) { error = std::string("cannot load RCCL (librccl.so.1): ") + dlerror(); return false; }
#define LG_SYM(field, sym)                                                                                             \
    field = reinterpret_cast<decltype(field)>(dlsym(lib, sym));                                                        \
    if (!field) { error = std::string("RCCL symbol missing: ") + sym; lib = nullptr; return false; }
        LG_SYM(CommInitAll, "ncclCommInitAll") LG_SYM(CommDestroy, "ncclCommDestroy") LG_SYM(GroupStart, "ncclGroupStart")
        LG_SYM(GroupEnd, "ncclGroupEnd") LG_SYM(Send, "ncclSend") LG_SYM(Recv, "ncclRecv") LG_SYM(GetErrorString, "ncclGetErrorString")
        LG_SYM(AllGather, "ncclAllGather")
#undef LG_SYM
        return true;
    }
};
Rccl g_rccl;
std::mutex g_rccl_mtx;
constexpr int NCCL_UINT8 = 1;
// Communicators are expensive to build (ncclCommInitAll takes a good fraction of a second) and capture() creates its
// multi-device state anew for every frame, like the reference rebuilds its BVH: one set per device list, kept for the
// life of the process.  Calls on one set are serialised (g_rccl_mtx).
std::map<std::vector<int>, std::vector<comm_t>> g_comms;

// What an exchange must undo on every exit path (lg_multi_capture_device*).
struct ExchangeGuard {
    bool group_open = false;
    std::vector<hipEvent_t> events;
    int caller_device = -1;
    ExchangeGuard() { if (hipGetDevice(&caller_device) != hipSuccess) caller_device = -1; }
    ~ExchangeGuard() {
        if (group_open) (void)g_rccl.GroupEnd();
        for (hipEvent_t ev : events) (void)hipEventDestroy(ev);
        if (caller_device >= 0) (void)hipSetDevice(caller_device);
    }
};

struct Share { // what one rank renders
    int device = 0;
    int comm_rank = -1;         // index of its device among the distinct devices (= rank in the communicator)
    lg_accel *accel = nullptr;
    void *tile = nullptr;       // device buffer of this rank's share (ranks > 0)
    size_t tile_bytes = 0;
};

} // namespace

struct lg_multi {
    const lg_scene *scene = nullptr;
    std::vector<Share> shares;
    std::vector<int> devices;   // distinct devices, in order of first appearance (devices[0] = root)
    std::vector<comm_t> comms;  // one communicator per distinct device (empty when there is only one)
    uint32_t block_rows = 64;
    bool force_rccl = false;    // LASGUN_MULTI_FORCE_RCCL=1: shares on the root's own device travel through RCCL too (self send / recv): 1-GPU rehearsal
    hipStream_t recv_stream = nullptr; // on the root: the receives (and local copies) run beside the root's own rendering
    void *own_film = nullptr;   // lg_multi_capture: device film on the root
    size_t own_film_bytes = 0;
    std::string error;
};

static int mfail(lg_multi *m, const std::string &msg) {
    if (m) m->error = msg;
    lg_set_last_error(msg.c_str());
    return 1;
}
#define HIP_OK(expr)                                                                                                   \
    do {                                                                                                               \
        hipError_t e_ = (expr);                                                                                        \
        if (e_ != hipSuccess) return mfail(m, std::string(#expr) + ": " + hipGetErrorString(e_));                      \
    } while (0)
#define NCCL_OK(expr)                                                                                                  \
    do {                                                                                                               \
        int r_ = (expr);                                                                                               \
        if (r_ != 0) return mfail(m, std::string(#expr) + ": " + g_rccl.GetErrorString(r_));                           \
    } while (0)

extern "C" {

lg_multi *lg_multi_create(const lg_scene *scene, const int *device_ids, int count, uint32_t block_rows) {
    int ndev = lg_device_count();
    if (count <= 0 || !device_ids) { lg_set_last_error("lg_multi_create: empty device list"); return nullptr; }
    for (int i = 0; i < count; ++i)
        if (device_ids[i] < 0 || device_ids[i] >= ndev) { lg_set_last_error("lg_multi_create: device index out of range"); return nullptr; }
    lg_multi *m = new lg_multi();
    m->scene = scene;
    m->block_rows = block_rows;
    const char *force = std::getenv("LASGUN_MULTI_FORCE_RCCL");
    m->force_rccl = force && force[0] == '1';
    for (int i = 0; i < count; ++i) {
        Share s;
        s.device = device_ids[i];
        int at = -1;
        for (size_t k = 0; k < m->devices.size(); ++k) if (m->devices[k] == s.device) at = (int)k;
        if (at < 0) { at = (int)m->devices.size(); m->devices.push_back(s.device); }
        s.comm_rank = at;
        m->shares.push_back(s);
    }
    { // Accel::from on every rank's device (host HLBVH build + flatten + upload), one host thread per rank
        std::vector<std::string> errs(m->shares.size());
        std::vector<std::thread> workers;
        for (size_t r = 0; r < m->shares.size(); ++r)
            workers.emplace_back([&, r] {
                m->shares[r].accel = lg_accel_from_on(scene, m->shares[r].device);
                if (!m->shares[r].accel) errs[r] = lg_last_error();
            });
        for (auto &t : workers) t.join();
        for (size_t r = 0; r < m->shares.size(); ++r)
            if (!m->shares[r].accel) { const std::string e = errs[r]; lg_multi_free(m); lg_set_last_error(e.c_str()); return nullptr; }
    }
    if (m->devices.size() > 1 || m->force_rccl) {
        std::lock_guard<std::mutex> g(g_rccl_mtx);
        if (!g_rccl.load()) { std::string e = g_rccl.error; lg_multi_free(m); lg_set_last_error(e.c_str()); return nullptr; }
        auto it = g_comms.find(m->devices);
        if (it == g_comms.end()) {
            std::vector<comm_t> comms(m->devices.size(), nullptr);
            int r = g_rccl.CommInitAll(comms.data(), (int)m->devices.size(), m->devices.data());
            if (r != 0) {
                std::string e = std::string("ncclCommInitAll: ") + g_rccl.GetErrorString(r);
                lg_multi_free(m);
                lg_set_last_error(e.c_str());
                return nullptr;
            }
            it = g_comms.emplace(m->devices, comms).first;
        }
        m->comms = it->second;
    }
    return m;
}

void lg_multi_free(lg_multi *m) {
    if (!m) return;
    for (Share &s : m->shares) {
        if (s.tile && hipSetDevice(s.device) == hipSuccess) { (void)hipDeviceSynchronize(); (void)hipFree(s.tile); }
        if (s.accel) lg_accel_free(s.accel);
    }
    if (!m->shares.empty() && hipSetDevice(m->shares[0].device) == hipSuccess) {
        if (m->own_film) (void)hipFree(m->own_film);
        if (m->recv_stream) (void)hipStreamDestroy(m->recv_stream);
    }
    delete m; // (the communicators stay in g_comms)
}

int lg_multi_rank_count(const lg_multi *m) { return (int)m->shares.size(); }
lg_accel *lg_multi_accel(const lg_multi *m, int rank) { return rank >= 0 && rank < (int)m->shares.size() ? m->shares[(size_t)rank].accel : nullptr; }
int lg_multi_uses_rccl(const lg_multi *m) { return m->comms.empty() ? 0 : 1; }

// The whole film into DEVICE memory of the root (the first device of the list).  Synchronous: on return the film is complete.
extern "C" void lg_internal_call_scope(int enter); // capi.cpp: which API call a launch belongs to (the measured choice counts calls)
namespace { struct CallScope { CallScope() { lg_internal_call_scope(1); } ~CallScope() { lg_internal_call_scope(0); } }; }

int lg_multi_capture_device(lg_multi *m, uint32_t w, uint32_t h, void *dev_rgba_on_root) {
    CallScope call; // ONE call, whatever its shares launch (two shares of one device are two launches of one kind)
    // declared in this order: on ANY exit path the guard first closes a still-open RCCL group (an error between GroupStart
    // and GroupEnd would otherwise leave every later RCCL call of this thread -- PyTorch's included -- queued and never
    // launched), destroys the events and puts the caller's device back; then the lock is released
    std::unique_lock<std::mutex> rccl_lock(g_rccl_mtx, std::defer_lock);
    ExchangeGuard guard;
    const uint32_t n = (uint32_t)m->shares.size();
    const size_t row_bytes = (size_t)w * 4;
    const bool interleaved = m->block_rows != 0 && h % (m->block_rows * n) == 0;
    const uint32_t b = m->block_rows;
    uint8_t *film = (uint8_t *)dev_rgba_on_root;
    // ---- every rank renders its share on its own device and stream
    struct Piece { uint32_t rank; size_t tile_off, film_off, bytes; };
    std::vector<Piece> pieces; // what has to travel to (or be copied on) the root
    for (uint32_t r = 0; r < n; ++r) {
        Share &s = m->shares[r];
        HIP_OK(hipSetDevice(s.device));
        void *stream = lg_accel_stream(s.accel);
        uint32_t y0 = 0, y1 = 0;
        size_t bytes;
        if (interleaved) bytes = (size_t)(h / n) * row_bytes;
        else {
            const uint32_t base = h / n, rem = h % n;
            y0 = r * base + (r < rem ? r : rem); y1 = y0 + base + (r < rem ? 1u : 0u);
            bytes = (size_t)(y1 - y0) * row_bytes;
        }
        if (r == 0 && !interleaved) { // the root's own tile is rendered in place
            if (y1 > y0 && lg_capture_rows_device(s.accel, w, h, y0, y1, 0, film, stream)) return mfail(m, lg_last_error());
            continue;
        }
        if (s.tile_bytes < bytes) {
            if (s.tile) { HIP_OK(hipStreamSynchronize((hipStream_t)stream)); HIP_OK(hipFree(s.tile)); s.tile = nullptr; }
            HIP_OK(hipMalloc(&s.tile, bytes ? bytes : 1));
            s.tile_bytes = bytes;
        }
        if (interleaved) {
            if (lg_capture_interleaved_device(s.accel, w, h, b, n, r, s.tile, stream)) return mfail(m, lg_last_error());
            for (uint32_t g = 0; g < h / (b * n); ++g) // block g of the compact tile is image block g*n + r
                pieces.push_back(Piece{r, (size_t)g * b * row_bytes, ((size_t)g * n + r) * b * row_bytes, (size_t)b * row_bytes});
        } else if (y1 > y0) {
            if (lg_capture_rows_device(s.accel, w, h, y0, y1, y0, s.tile, stream)) return mfail(m, lg_last_error());
            pieces.push_back(Piece{r, 0, (size_t)y0 * row_bytes, bytes});
        }
    }
    // ---- the gather: ONE grouped exchange over xGMI; pieces that are already on the root's device are copied there
    Share &root = m->shares[0];
    HIP_OK(hipSetDevice(root.device));
    if (!m->recv_stream) HIP_OK(hipStreamCreateWithFlags(&m->recv_stream, hipStreamNonBlocking));
    hipStream_t root_stream = (hipStream_t)lg_accel_stream(root.accel);
    bool any_rccl = false;
    for (const Piece &p : pieces) if (m->shares[p.rank].device != root.device || m->force_rccl) any_rccl = true;
    if (any_rccl && m->comms.empty()) return mfail(m, "internal: no communicator");
    // pieces rendered on the root's device by a stream other than the receiving one: order the transfer after their render
    for (uint32_t r = 0; r < n; ++r) {
        Share &s = m->shares[r];
        bool has = false;
        for (const Piece &p : pieces) has = has || p.rank == r;
        if (!has || s.device != root.device) continue;
        hipEvent_t ev;
        HIP_OK(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
        guard.events.push_back(ev);
        HIP_OK(hipEventRecord(ev, (hipStream_t)lg_accel_stream(s.accel)));
        HIP_OK(hipStreamWaitEvent(m->recv_stream, ev, 0));
    }
    if (any_rccl) { rccl_lock.lock(); NCCL_OK(g_rccl.GroupStart()); guard.group_open = true; }
    for (const Piece &p : pieces) {
        Share &s = m->shares[p.rank];
        if (s.device == root.device && !m->force_rccl) continue; // copied below
        // sender on its own device and stream (ordered after its render), receiver on the root's receiving stream
        const bool self = s.device == root.device;
        HIP_OK(hipSetDevice(s.device));
        NCCL_OK(g_rccl.Send((const uint8_t *)s.tile + p.tile_off, p.bytes, NCCL_UINT8, root.comm_rank, m->comms[(size_t)s.comm_rank],
                            self ? m->recv_stream : (hipStream_t)lg_accel_stream(s.accel)));
        HIP_OK(hipSetDevice(root.device));
        NCCL_OK(g_rccl.Recv(film + p.film_off, p.bytes, NCCL_UINT8, s.comm_rank, m->comms[(size_t)root.comm_rank], m->recv_stream));
    }
    if (any_rccl) { guard.group_open = false; NCCL_OK(g_rccl.GroupEnd()); }
    HIP_OK(hipSetDevice(root.device));
    if (!m->force_rccl)
        for (const Piece &p : pieces) {
            Share &s = m->shares[p.rank];
            if (s.device != root.device) continue;
            HIP_OK(hipMemcpyAsync(film + p.film_off, (const uint8_t *)s.tile + p.tile_off, p.bytes, hipMemcpyDeviceToDevice, m->recv_stream));
        }
    HIP_OK(hipStreamSynchronize(m->recv_stream));
    HIP_OK(hipStreamSynchronize(root_stream)); // the root's own share, rendered in place
    // the senders' streams have completed their sends once the root has received; make the tiles reusable explicitly
    for (uint32_t r = 1; r < n; ++r) {
        HIP_OK(hipSetDevice(m->shares[r].device));
        HIP_OK(hipStreamSynchronize((hipStream_t)lg_accel_stream(m->shares[r].accel)));
    }
    return 0; // (the guard restores the caller's device)
}

// Every rank's device receives the WHOLE film ("equivalently ncclAllGather if every rank wants the image" --
// a display per GPU, a later per-device pass over the full frame).  One device per rank (no repeats), equal shares: 64-row
// blocks dealt round-robin, or contiguous row tiles of a height that n divides.  ONE ncclAllGather per device inside one
// group, on that device's render stream; contiguous tiles land in row order as they arrive, interleaved blocks are put in
// row order by n strided device copies per device.  dev_rgba[r] = a w*h*4-byte buffer on rank r's device.
int lg_multi_capture_device_all(lg_multi *m, uint32_t w, uint32_t h, void *const *dev_rgba) {
    CallScope call;
    std::unique_lock<std::mutex> rccl_lock(g_rccl_mtx, std::defer_lock);
    ExchangeGuard guard;
    const uint32_t n = (uint32_t)m->shares.size();
    if (m->devices.size() != n) return mfail(m, "lg_multi_capture_device_all: one device per rank (a device may not repeat)");
    const size_t row_bytes = (size_t)w * 4;
    const uint32_t b = m->block_rows;
    const bool interleaved = b != 0 && h % (b * n) == 0;
    if (!interleaved && h % n != 0) return mfail(m, "lg_multi_capture_device_all: the film's height must be a multiple of the rank count");
    const uint32_t rows = h / n;
    const size_t tile_bytes = (size_t)rows * row_bytes;
    if (n > 1 && m->comms.empty()) return mfail(m, "internal: no communicator");
    std::vector<void *> gathered(n, nullptr); // interleaved: rank-major staging per device
    for (uint32_t r = 0; r < n; ++r) {
        Share &s = m->shares[r];
        HIP_OK(hipSetDevice(s.device));
        void *stream = lg_accel_stream(s.accel);
        if (interleaved) {
            const size_t need = tile_bytes * ((size_t)n + 1); // [own tile][n gathered tiles]
            if (s.tile_bytes < need) {
                if (s.tile) { HIP_OK(hipStreamSynchronize((hipStream_t)stream)); HIP_OK(hipFree(s.tile)); s.tile = nullptr; }
                HIP_OK(hipMalloc(&s.tile, need));
                s.tile_bytes = need;
            }
            gathered[r] = (uint8_t *)s.tile + tile_bytes;
            if (lg_capture_interleaved_device(s.accel, w, h, b, n, r, s.tile, stream)) return mfail(m, lg_last_error());
        } else { // rendered in place: rank r's tile is rows [r*rows, (r+1)*rows) of its own copy of the film
            if (lg_capture_rows_device(s.accel, w, h, r * rows, (r + 1) * rows, 0, dev_rgba[r], stream)) return mfail(m, lg_last_error());
        }
    }
    if (n > 1) {
        rccl_lock.lock();
        NCCL_OK(g_rccl.GroupStart());
        guard.group_open = true;
        for (uint32_t r = 0; r < n; ++r) {
            Share &s = m->shares[r];
            HIP_OK(hipSetDevice(s.device));
            const void *send = interleaved ? s.tile : (const void *)((const uint8_t *)dev_rgba[r] + (size_t)r * tile_bytes); // (in place: RCCL allows send == recv + rank * count)
            void *recv = interleaved ? gathered[r] : dev_rgba[r];
            NCCL_OK(g_rccl.AllGather(send, recv, tile_bytes, NCCL_UINT8, m->comms[(size_t)s.comm_rank], (hipStream_t)lg_accel_stream(s.accel)));
        }
        guard.group_open = false;
        NCCL_OK(g_rccl.GroupEnd());
    }
    for (uint32_t r = 0; r < n; ++r) {
        Share &s = m->shares[r];
        HIP_OK(hipSetDevice(s.device));
        hipStream_t stream = (hipStream_t)lg_accel_stream(s.accel);
        if (interleaved) { // rank q's compact tile holds image blocks q, q + n, ...: one strided copy per source rank
            const uint8_t *src0 = n > 1 ? (const uint8_t *)gathered[r] : (const uint8_t *)s.tile;
            for (uint32_t q = 0; q < n; ++q)
                HIP_OK(hipMemcpy2DAsync((uint8_t *)dev_rgba[r] + (size_t)q * b * row_bytes, (size_t)n * b * row_bytes, src0 + (size_t)q * tile_bytes,
                                        (size_t)b * row_bytes, (size_t)b * row_bytes, h / (b * n), hipMemcpyDeviceToDevice, stream));
        }
        HIP_OK(hipStreamSynchronize(stream));
    }
    return 0;
}

// The same into a HOST film: the gathered device film + one D2H copy from the root.
int lg_multi_capture(lg_multi *m, lg_film *film) {
    const uint32_t w = lg_film_width(film), h = lg_film_height(film);
    const size_t bytes = (size_t)w * h * 4;
    HIP_OK(hipSetDevice(m->shares[0].device));
    if (m->own_film_bytes < bytes) {
        if (m->own_film) HIP_OK(hipFree(m->own_film));
        m->own_film = nullptr;
        HIP_OK(hipMalloc(&m->own_film, bytes ? bytes : 1));
        m->own_film_bytes = bytes;
    }
    if (lg_multi_capture_device(m, w, h, m->own_film)) return 1;
    HIP_OK(hipMemcpy(lg_film_pixels(film), m->own_film, bytes, hipMemcpyDeviceToHost));
    return 0;
}

} // extern "C"
