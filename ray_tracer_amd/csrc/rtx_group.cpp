// One frame over several GPUs of a node from C/C++: the rtx_scene_group part of include/rtx.h.
//
// Takes the place of the reference's thread fan-out (src/main.cpp:34-52: eight std::threads, each
// owning a strip of columns of one shared canvas).  Here the scene is replicated in every selected
// GPU's HBM, the frame is cut into screen tiles dealt round-robin (tile k -> device k mod N), every
// device renders its tiles into a packed buffer, ONE RCCL exchange of those equally sized buffers
// moves them over xGMI (an all-gather, or sends to the root alone: rtx_gather_plan), and the root
// device writes the row-major frame.  One host thread drives all devices (every call below is
// asynchronous), with a single-process communicator per device from ncclCommInitAll.  A device has
// ONE exchange stream for its communicator -- RCCL wants a communicator's operations issued in one
// order on one stream; frames in flight render on their own streams and meet the exchange stream
// through events.  RCCL is bound at first use with dlopen, so the
// library itself loads on machines without it and shares the copy a host program (PyTorch) may
// already have loaded.
#include <dlfcn.h>
#include <hip/hip_runtime_api.h>
#include <rccl/rccl.h>

#include <cstdio>
#include <cstring>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "host_scene.hpp"
#include "rtx.h"

extern "C" void rtx_set_last_error_(const char *message);   // rtx_api.cpp

namespace {

int fail(int code, const std::string &msg) {
    rtx_set_last_error_(msg.c_str());
    return code;
}

#define GRP_HIP(expr)                                                                                  \
    do {                                                                                               \
        hipError_t e_ = (expr);                                                                        \
        if (e_ != hipSuccess) return fail(RTX_ERR_DEVICE, std::string(#expr) + ": " + hipGetErrorString(e_)); \
    } while (0)

// The RCCL entry points this file uses, resolved once.
struct Rccl {
    void *handle = nullptr;
    ncclResult_t (*CommInitAll)(ncclComm_t *, int, const int *) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*AllGather)(const void *, void *, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Send)(const void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Recv)(void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    const char *(*GetErrorString)(ncclResult_t) = nullptr;
    std::string error;
};

const Rccl &rccl() {
    static Rccl r;
    static std::once_flag once;
    std::call_once(once, [] {
        for (const char *name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
            r.handle = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
            if (r.handle) break;
        }
        if (!r.handle) { r.error = std::string("RCCL is not loadable: ") + dlerror(); return; }
        auto sym = [&](const char *n) { void *p = dlsym(r.handle, n); if (!p && r.error.empty()) r.error = std::string("RCCL lacks ") + n; return p; };
        r.CommInitAll = (decltype(r.CommInitAll))sym("ncclCommInitAll");
        r.CommDestroy = (decltype(r.CommDestroy))sym("ncclCommDestroy");
        r.AllGather = (decltype(r.AllGather))sym("ncclAllGather");
        r.Send = (decltype(r.Send))sym("ncclSend");
        r.Recv = (decltype(r.Recv))sym("ncclRecv");
        r.GroupStart = (decltype(r.GroupStart))sym("ncclGroupStart");
        r.GroupEnd = (decltype(r.GroupEnd))sym("ncclGroupEnd");
        r.GetErrorString = (decltype(r.GetErrorString))sym("ncclGetErrorString");
    });
    return r;
}

#define GRP_NCCL(expr)                                                                                                 \
    do {                                                                                                               \
        ncclResult_t e_ = (expr);                                                                                      \
        if (e_ != ncclSuccess) return fail(RTX_ERR_DEVICE, std::string(#expr) + ": " + rccl().GetErrorString(e_));     \
    } while (0)

struct DeviceGuard {
    int prev = -1;
    DeviceGuard() { (void)hipGetDevice(&prev); }
    ~DeviceGuard() { if (prev >= 0) (void)hipSetDevice(prev); }
};

}  // namespace

struct rtx_scene_group {
    static constexpr int kFrames = 4;               // frames that may be in flight (one render stream per device each)
    std::vector<int> devices;
    std::vector<rtx_scene *> scenes;
    std::vector<ncclComm_t> comms;
    struct PerDevice {
        hipStream_t stream[kFrames] = {};           // render streams, one per frame slot
        hipStream_t exchange = nullptr;             // the communicator's stream: every exchange (and the root's re-assembly)
        hipEvent_t rendered[kFrames] = {};          // slot's tiles are in d_local     (recorded on stream[f])
        hipEvent_t exchanged[kFrames] = {};         // slot's exchange no longer needs d_local / d_gather (recorded on exchange)
        void *d_local[kFrames] = {};                // this device's packed tiles
        void *d_gather[kFrames] = {};               // [rank][packed tiles] after the exchange (root only when gathering to the root)
        uint64_t local_bytes = 0, gather_bytes = 0;
    };
    std::vector<PerDevice> per;
    void *d_frame[kFrames] = {};                    // root device: the row-major frame
    void *h_frame = nullptr;                        // pinned, for rtx_render_sharded
    uint64_t frame_bytes = 0, h_frame_bytes = 0;
    int32_t gather = RTX_GATHER_ALL;
    std::mutex lock;
};

namespace {

int pick_devices(const int32_t *devices, int32_t n, std::vector<int> *out) {
    int ndev = 0;
    GRP_HIP(hipGetDeviceCount(&ndev));
    if (ndev < 1) return fail(RTX_ERR_DEVICE, "no HIP device");
    if (n <= 0 || !devices) {
        const int take = n > 0 ? n : ndev;
        if (take > ndev) return fail(RTX_ERR_INVALID, "more devices asked for than the node has");
        for (int i = 0; i < take; i++) out->push_back(i);
        return RTX_OK;
    }
    for (int32_t i = 0; i < n; i++) {
        if (devices[i] < 0 || devices[i] >= ndev) return fail(RTX_ERR_INVALID, "device ordinal out of range");
        for (int d : *out)
            if (d == devices[i]) return fail(RTX_ERR_INVALID, "a device is listed twice");
        out->push_back(devices[i]);
    }
    return RTX_OK;
}

// Joins whatever threads it holds when it goes out of scope, also when creating a later thread throws.
struct ThreadPool {
    std::vector<std::thread> threads;
    ~ThreadPool() { for (std::thread &t : threads) if (t.joinable()) t.join(); }
};

// Scene replicas (created by `make`, one host thread per device: each create call is synchronous),
// then the communicators, streams and events.  RCCL announces its version on standard output while it
// initialises (NCCL_DEBUG=VERSION and up); a driver that keeps its standard output for its own lines
// sets NCCL_DEBUG=WARN or redirects around this call itself -- the library does not touch the
// process's file descriptors.
template <typename Make>
int group_create(const int32_t *devices, int32_t n_devices, rtx_scene_group **out, Make make) {
    if (!out) return fail(RTX_ERR_INVALID, "NULL argument");
    *out = nullptr;
    const Rccl &nc = rccl();
    if (!nc.error.empty()) return fail(RTX_ERR_UNSUPPORTED, nc.error);
    std::unique_ptr<rtx_scene_group, void (*)(rtx_scene_group *)> g(new rtx_scene_group, rtx_scene_group_destroy);
    if (int rc = pick_devices(devices, n_devices, &g->devices)) return rc;
    const size_t n = g->devices.size();
    g->scenes.assign(n, nullptr);
    g->per.resize(n);
    if (const char *mode = getenv("RTX_GROUP_GATHER")) g->gather = (!strcmp(mode, "root") || !strcmp(mode, "1")) ? RTX_GATHER_ROOT : RTX_GATHER_ALL;
    std::vector<int> rcs(n, RTX_OK);
    std::vector<std::string> errors(n);
    {
        ThreadPool pool;
        for (size_t i = 0; i < n; i++)
            pool.threads.emplace_back([&, i] {
                rcs[i] = make(g->devices[i], &g->scenes[i]);
                if (rcs[i] != RTX_OK) errors[i] = rtx_last_error();      // the error text is per thread
            });
    }
    for (size_t i = 0; i < n; i++)
        if (rcs[i] != RTX_OK) return fail(rcs[i], "device " + std::to_string(g->devices[i]) + ": " + errors[i]);
    DeviceGuard restore;
    g->comms.assign(n, nullptr);
    {
        const ncclResult_t e = nc.CommInitAll(g->comms.data(), (int)n, g->devices.data());
        if (e != ncclSuccess) return fail(RTX_ERR_DEVICE, std::string("ncclCommInitAll: ") + nc.GetErrorString(e));
    }
    for (size_t i = 0; i < n; i++) {
        rtx_scene_group::PerDevice &d = g->per[i];
        GRP_HIP(hipSetDevice(g->devices[i]));
        int least = 0, greatest = 0;                 // numerically: greatest priority = the smaller number
        GRP_HIP(hipDeviceGetStreamPriorityRange(&least, &greatest));
        // Default priority.  A high-priority exchange stream would let the exchange's kernels pass the persistent render
        // launches that keep the chip full, but on this runtime it hangs as soon as the process has more live streams than
        // hardware queues (four frames in flight + this stream with the default four queues): RTX_GROUP_EXCHANGE_HIGH=1
        // asks for it all the same, for a process that sets GPU_MAX_HW_QUEUES above its stream count.
        const char *high = getenv("RTX_GROUP_EXCHANGE_HIGH");
        GRP_HIP(hipStreamCreateWithPriority(&d.exchange, hipStreamNonBlocking, high && *high == '1' ? greatest : least));
        for (int f = 0; f < rtx_scene_group::kFrames; f++) {
            GRP_HIP(hipStreamCreateWithFlags(&d.stream[f], hipStreamNonBlocking));
            GRP_HIP(hipEventCreateWithFlags(&d.rendered[f], hipEventDisableTiming));
            GRP_HIP(hipEventCreateWithFlags(&d.exchanged[f], hipEventDisableTiming));
        }
    }
    *out = g.release();
    return RTX_OK;
}

// Grows the per-slot buffers to what `p` needs.  Called with the group locked.  Before any buffer is
// released EVERY member is waited for: an exchange still running on a peer reads and writes this
// device's buffers over xGMI.
int reserve(rtx_scene_group *g, const rtx_render_params &shard, uint64_t frame_bytes) {
    const uint64_t local = rtx_output_pixels(&shard) * 3, n = g->devices.size();
    bool grow = frame_bytes > g->frame_bytes;
    for (size_t i = 0; i < n; i++) grow = grow || local > g->per[i].local_bytes;
    if (!grow) return RTX_OK;
    for (size_t i = 0; i < n; i++) {
        GRP_HIP(hipSetDevice(g->devices[i]));
        GRP_HIP(hipDeviceSynchronize());
    }
    for (size_t i = 0; i < n; i++) {
        rtx_scene_group::PerDevice &d = g->per[i];
        if (local <= d.local_bytes) continue;
        GRP_HIP(hipSetDevice(g->devices[i]));
        for (int f = 0; f < rtx_scene_group::kFrames; f++) {
            (void)hipFree(d.d_local[f]);
            (void)hipFree(d.d_gather[f]);
            d.d_local[f] = d.d_gather[f] = nullptr;
        }
        d.local_bytes = d.gather_bytes = 0;
        for (int f = 0; f < rtx_scene_group::kFrames; f++) {
            GRP_HIP(hipMalloc(&d.d_local[f], local));
            GRP_HIP(hipMalloc(&d.d_gather[f], local * n));
        }
        d.local_bytes = local;
        d.gather_bytes = local * n;
    }
    if (frame_bytes > g->frame_bytes) {
        GRP_HIP(hipSetDevice(g->devices[0]));
        for (int f = 0; f < rtx_scene_group::kFrames; f++) { (void)hipFree(g->d_frame[f]); g->d_frame[f] = nullptr; }
        g->frame_bytes = 0;
        for (int f = 0; f < rtx_scene_group::kFrames; f++) GRP_HIP(hipMalloc(&g->d_frame[f], frame_bytes));
        g->frame_bytes = frame_bytes;
    }
    return RTX_OK;
}

// Queues one frame on slot `f`: every device renders its tiles on the slot's stream, the exchange and the
// root's re-assembly follow on the exchange streams.
int submit(rtx_scene_group *g, const rtx_render_params *p, int f) {
    if (!g || !p) return fail(RTX_ERR_INVALID, "NULL argument");
    if (f < 0 || f >= rtx_scene_group::kFrames) return fail(RTX_ERR_INVALID, "frame slot must be 0..3");
    // the parameters are checked before any buffer is touched
    if (p->width < 0 || p->height < 0) return fail(RTX_ERR_INVALID, "negative canvas size");
    const int n = (int)g->devices.size();
    rtx_render_params shard = *p;
    shard.tile_count = n;
    shard.tile_rank = 0;
    shard.layout = RTX_LAYOUT_PACKED_TILES;
    if (rtx_output_pixels(&shard) == 0 && (p->width > 0 && p->height > 0)) return RTX_ERR_INVALID;   // the message is set
    const uint64_t frame_bytes = (uint64_t)p->width * (uint64_t)p->height * 3;
    if (frame_bytes == 0) return RTX_OK;
    DeviceGuard restore;
    if (int rc = reserve(g, shard, frame_bytes)) return rc;
    const Rccl &nc = rccl();
    const uint64_t local = rtx_output_pixels(&shard) * 3;
    for (int i = 0; i < n; i++) {
        rtx_scene_group::PerDevice &d = g->per[i];
        GRP_HIP(hipSetDevice(g->devices[i]));
        shard.tile_rank = i;
        // the slot's previous exchange has read d_local (an event never recorded yet waits for nothing)
        GRP_HIP(hipStreamWaitEvent(d.stream[f], d.exchanged[f], 0));
        // pixels outside the window and the padding of a rank with fewer tiles read 0
        GRP_HIP(hipMemsetAsync(d.d_local[f], 0, d.local_bytes, d.stream[f]));
        if (int rc = rtx_render_device(g->scenes[i], &shard, d.d_local[f], nullptr, nullptr, nullptr, nullptr, d.stream[f])) return rc;
        GRP_HIP(hipEventRecord(d.rendered[f], d.stream[f]));
        GRP_HIP(hipStreamWaitEvent(d.exchange, d.rendered[f], 0));
    }
    std::vector<rtx_gather_op> ops((size_t)2 * n + 2);
    const int32_t n_ops = rtx_gather_plan(n, local, g->gather, ops.data(), (int32_t)ops.size());
    if (n_ops < 0) return n_ops;
    GRP_NCCL(nc.GroupStart());
    for (int32_t k = 0; k < n_ops; k++) {
        const rtx_gather_op &op = ops[k];
        rtx_scene_group::PerDevice &d = g->per[op.member];
        ncclResult_t e = ncclSuccess;
        if (op.kind == RTX_GATHER_OP_ALLGATHER) e = nc.AllGather(d.d_local[f], d.d_gather[f], op.bytes, ncclUint8, g->comms[op.member], d.exchange);
        else if (op.kind == RTX_GATHER_OP_SEND) e = nc.Send((const char *)d.d_local[f] + op.src_offset, op.bytes, ncclUint8, op.peer, g->comms[op.member], d.exchange);
        else if (op.kind == RTX_GATHER_OP_RECV) e = nc.Recv((char *)d.d_gather[f] + op.dst_offset, op.bytes, ncclUint8, op.peer, g->comms[op.member], d.exchange);
        if (e != ncclSuccess) { (void)nc.GroupEnd(); return fail(RTX_ERR_DEVICE, std::string("RCCL exchange: ") + nc.GetErrorString(e)); }
    }
    GRP_NCCL(nc.GroupEnd());
    for (int32_t k = 0; k < n_ops; k++) {
        const rtx_gather_op &op = ops[k];
        if (op.kind != RTX_GATHER_OP_COPY) continue;     // a member's own tiles into its gathered buffer: no link involved
        rtx_scene_group::PerDevice &d = g->per[op.member];
        GRP_HIP(hipSetDevice(g->devices[op.member]));
        GRP_HIP(hipMemcpyAsync((char *)d.d_gather[f] + op.dst_offset, (const char *)d.d_local[f] + op.src_offset, op.bytes,
                               hipMemcpyDeviceToDevice, d.exchange));
    }
    GRP_HIP(hipSetDevice(g->devices[0]));
    shard.tile_rank = 0;
    if (int rc = rtx_untile(&shard, n, g->per[0].d_gather[f], g->d_frame[f], 3, g->per[0].exchange)) return rc;
    for (int i = 0; i < n; i++) {
        GRP_HIP(hipSetDevice(g->devices[i]));
        GRP_HIP(hipEventRecord(g->per[i].exchanged[f], g->per[i].exchange));
    }
    return RTX_OK;
}

}  // namespace

extern "C" {

// Who moves which bytes where when N members, each holding `local_bytes` of packed tiles, bring them together.
// RTX_GATHER_ALL: every member takes part in one all-gather and ends with all N buffers (rank-major).
// RTX_GATHER_ROOT: members 1..N-1 send to member 0, which receives each buffer at its rank's offset and copies its own:
// (N-1)/N of the frame crosses the links once, where the all-gather moves it to every member.
int32_t rtx_gather_plan(int32_t n_members, uint64_t local_bytes, int32_t mode, rtx_gather_op *ops, int32_t max_ops) {
    if (n_members < 1 || !ops) return fail(RTX_ERR_INVALID, "rtx_gather_plan: bad argument");
    if (mode != RTX_GATHER_ALL && mode != RTX_GATHER_ROOT) return fail(RTX_ERR_INVALID, "rtx_gather_plan: unknown mode");
    const int32_t need = mode == RTX_GATHER_ALL ? n_members : 2 * (n_members - 1) + 1;
    if (max_ops < need) return fail(RTX_ERR_INVALID, "rtx_gather_plan: room for too few operations");
    int32_t k = 0;
    if (mode == RTX_GATHER_ALL) {
        for (int32_t i = 0; i < n_members; i++) ops[k++] = rtx_gather_op{RTX_GATHER_OP_ALLGATHER, i, -1, 0, 0, 0, local_bytes};
        return k;
    }
    for (int32_t r = 1; r < n_members; r++) {
        ops[k++] = rtx_gather_op{RTX_GATHER_OP_RECV, 0, r, 0, 0, (uint64_t)r * local_bytes, local_bytes};
        ops[k++] = rtx_gather_op{RTX_GATHER_OP_SEND, r, 0, 0, 0, 0, local_bytes};
    }
    ops[k++] = rtx_gather_op{RTX_GATHER_OP_COPY, 0, 0, 0, 0, 0, local_bytes};
    return k;
}

int rtx_scene_group_set_gather(rtx_scene_group *g, int32_t mode) {
    if (!g) return fail(RTX_ERR_INVALID, "NULL argument");
    if (mode != RTX_GATHER_ALL && mode != RTX_GATHER_ROOT) return fail(RTX_ERR_INVALID, "unknown gather mode");
    std::lock_guard<std::mutex> hold(g->lock);
    g->gather = mode;
    return RTX_OK;
}

int rtx_scene_group_create(const rtx_scene_desc *desc, const int32_t *devices, int32_t n_devices, rtx_scene_group **out) {
    if (!desc) return fail(RTX_ERR_INVALID, "NULL argument");
    try {
        return group_create(devices, n_devices, out, [desc](int device, rtx_scene **scene) {
            rtx_scene_desc d = *desc;
            d.device = device;
            return rtx_scene_create(&d, scene);
        });
    } catch (const std::exception &e) {
        return fail(RTX_ERR_NOMEM, std::string("unexpected failure: ") + e.what());
    }
}

int rtx_scene_group_create_from_host(const rtx_host_scene *hs, const int32_t *devices, int32_t n_devices, rtx_scene_group **out) {
    if (!hs) return fail(RTX_ERR_INVALID, "NULL argument");
    try {
        return group_create(devices, n_devices, out, [hs](int device, rtx_scene **scene) { return rtx_scene_create_from_host(hs, device, scene); });
    } catch (const std::exception &e) {
        return fail(RTX_ERR_NOMEM, std::string("unexpected failure: ") + e.what());
    }
}

int32_t rtx_scene_group_size(const rtx_scene_group *g) { return g ? (int32_t)g->devices.size() : 0; }
int32_t rtx_scene_group_device(const rtx_scene_group *g, int32_t i) { return g && i >= 0 && i < (int32_t)g->devices.size() ? g->devices[i] : -1; }
rtx_scene *rtx_scene_group_scene(rtx_scene_group *g, int32_t i) { return g && i >= 0 && i < (int32_t)g->scenes.size() ? g->scenes[i] : nullptr; }

void rtx_scene_group_destroy(rtx_scene_group *g) {
    if (!g) return;
    DeviceGuard restore;
    for (size_t i = 0; i < g->devices.size(); i++) {
        if (hipSetDevice(g->devices[i]) != hipSuccess) continue;
        (void)hipDeviceSynchronize();
        if (i < g->per.size())
            for (int f = 0; f < rtx_scene_group::kFrames; f++) {
                (void)hipFree(g->per[i].d_local[f]);
                (void)hipFree(g->per[i].d_gather[f]);
                if (g->per[i].stream[f]) (void)hipStreamDestroy(g->per[i].stream[f]);
                if (g->per[i].rendered[f]) (void)hipEventDestroy(g->per[i].rendered[f]);
                if (g->per[i].exchanged[f]) (void)hipEventDestroy(g->per[i].exchanged[f]);
            }
        if (i < g->per.size() && g->per[i].exchange) (void)hipStreamDestroy(g->per[i].exchange);
        if (i == 0) {
            for (int f = 0; f < rtx_scene_group::kFrames; f++) (void)hipFree(g->d_frame[f]);
            if (g->h_frame) (void)hipHostFree(g->h_frame);
        }
    }
    for (ncclComm_t c : g->comms)
        if (c) (void)rccl().CommDestroy(c);
    for (rtx_scene *s : g->scenes) rtx_scene_destroy(s);
    delete g;
}

int rtx_render_sharded_device(rtx_scene_group *g, const rtx_render_params *p, int32_t slot, void **d_rgb_u8) {
    if (!g) return fail(RTX_ERR_INVALID, "NULL argument");
    try {
        std::lock_guard<std::mutex> hold(g->lock);
        if (int rc = submit(g, p, slot)) return rc;
        if (d_rgb_u8) *d_rgb_u8 = g->d_frame[slot];
        return RTX_OK;
    } catch (const std::exception &e) {
        return fail(RTX_ERR_NOMEM, std::string("unexpected failure: ") + e.what());
    }
}

int rtx_scene_group_sync(rtx_scene_group *g, int32_t slot) {
    if (!g) return fail(RTX_ERR_INVALID, "NULL argument");
    if (slot >= rtx_scene_group::kFrames) return fail(RTX_ERR_INVALID, "frame slot must be 0..3, or negative for all");
    std::lock_guard<std::mutex> hold(g->lock);
    DeviceGuard restore;
    for (size_t i = 0; i < g->devices.size(); i++) {
        GRP_HIP(hipSetDevice(g->devices[i]));
        for (int f = 0; f < rtx_scene_group::kFrames; f++)
            if (slot < 0 || slot == f) {
                GRP_HIP(hipStreamSynchronize(g->per[i].stream[f]));
                GRP_HIP(hipEventSynchronize(g->per[i].exchanged[f]));      // recorded behind the exchange and the root's re-assembly
            }
    }
    return RTX_OK;
}

int rtx_render_sharded(rtx_scene_group *g, const rtx_render_params *p, uint8_t *rgb_out) {
    if (!g || !rgb_out) return fail(RTX_ERR_INVALID, "NULL argument");
    try {
        std::lock_guard<std::mutex> hold(g->lock);
        if (int rc = submit(g, p, 0)) return rc;
        const uint64_t bytes = (uint64_t)p->width * (uint64_t)p->height * 3;
        if (bytes == 0) return RTX_OK;
        DeviceGuard restore;
        GRP_HIP(hipSetDevice(g->devices[0]));
        if (bytes > g->h_frame_bytes) {
            if (g->h_frame) (void)hipHostFree(g->h_frame);
            g->h_frame = nullptr;
            g->h_frame_bytes = 0;
            GRP_HIP(hipHostMalloc(&g->h_frame, bytes, hipHostMallocDefault));
            g->h_frame_bytes = bytes;
        }
        hipStream_t st = g->per[0].exchange;          // behind the re-assembly
        GRP_HIP(hipMemcpyAsync(g->h_frame, g->d_frame[0], bytes, hipMemcpyDeviceToHost, st));
        // the other devices are done once the root has everybody's tiles, but their streams are waited for as
        // well so that the next call may reuse their buffers
        for (size_t i = 0; i < g->devices.size(); i++) {
            GRP_HIP(hipSetDevice(g->devices[i]));
            GRP_HIP(hipStreamSynchronize(g->per[i].stream[0]));
            GRP_HIP(hipStreamSynchronize(g->per[i].exchange));
        }
        rtx::parallel_copy(rgb_out, g->h_frame, bytes);
        return RTX_OK;
    } catch (const std::exception &e) {
        return fail(RTX_ERR_NOMEM, std::string("unexpected failure: ") + e.what());
    }
}

}  // extern "C"
