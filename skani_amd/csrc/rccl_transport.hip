// rccl_transport.hip -- the collectives of the distributed triangle (dist.hip) on RCCL: device buffers, xGMI between the GPUs of a node.
//
// librccl.so.1 is loaded on first use (dlopen): single-GPU users of libskani_hip.so do not need it, and a process that already carries an RCCL
// (PyTorch ships one under the same soname) keeps using that copy.  The communicator runs on the context's stream.  xGMI is point-to-point
// (7 links per GPU), so the sketch exchange is one grouped send/recv per peer -- every link carries exactly the sketches its peer needs -- and
// the all-gathers (marker sets, small tables) are RCCL's ring/tree collectives.  Host-memory buffers of the small table exchanges are staged
// through device scratch: RCCL moves device memory only.
#include <dlfcn.h>

#include "internal.h"

// The handful of RCCL declarations this file uses, restated from RCCL's public C API (rccl.h; the NCCL 2 ABI): the library is reached through
// dlopen only, so neither its headers nor its import library are needed to build libskani_hip.so.
extern "C" {
typedef struct ncclComm* ncclComm_t;
typedef struct { char internal[128]; } ncclUniqueId;
typedef enum { ncclSuccess = 0 } ncclResult_t;                           // every other value is an error; its text comes from ncclGetErrorString
typedef enum { ncclInt8 = 0, ncclUint8 = 1 } ncclDataType_t;             // (bytes are all that travels here)
}

namespace skh {

namespace {

struct RcclApi {
    void* lib = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*AllGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Send)(const void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Recv)(void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
};

static void load_api(RcclApi& a) {
    void* h = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
    if (!h) h = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL);
    if (!h) h = dlopen("/opt/rocm/lib/librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
    if (!h) throw Error(std::string("cannot load librccl.so.1: ") + dlerror());
    auto sym = [&](const char* n) { void* p = dlsym(h, n); if (!p) throw Error(std::string("librccl lacks ") + n); return p; };
    a.GetUniqueId = (decltype(a.GetUniqueId))sym("ncclGetUniqueId");
    a.CommInitRank = (decltype(a.CommInitRank))sym("ncclCommInitRank");
    a.CommDestroy = (decltype(a.CommDestroy))sym("ncclCommDestroy");
    a.AllGather = (decltype(a.AllGather))sym("ncclAllGather");
    a.Send = (decltype(a.Send))sym("ncclSend");
    a.Recv = (decltype(a.Recv))sym("ncclRecv");
    a.GroupStart = (decltype(a.GroupStart))sym("ncclGroupStart");
    a.GroupEnd = (decltype(a.GroupEnd))sym("ncclGroupEnd");
    a.GetErrorString = (decltype(a.GetErrorString))sym("ncclGetErrorString");
    a.lib = h;
}
// (contexts of different host threads may create their communicators at the same moment: the function table is filled once, under the static's own guard; a load that
// throws leaves the guard open and the next caller tries again)
RcclApi& api() {
    static RcclApi a = [] { RcclApi x; load_api(x); return x; }();
    return a;
}

void rccl_check(ncclResult_t r, const char* what) {
    if (r != ncclSuccess) throw Error(std::string(what) + ": " + api().GetErrorString(r));
}

struct RcclTransport : Transport {
    ncclComm_t comm = nullptr;
    ~RcclTransport() override { if (comm) (void)api().CommDestroy(comm); }
    void all_gather(skh_ctx* ctx, const void* send, void* recv, size_t bytes, bool device) override {
        if (!bytes) return;
        const void* s = send; void* r = recv; DBuf<char> ds, dr;
        if (!device) { ds.alloc(bytes); dr.alloc(bytes * world); h2d(ds.p, send, bytes, ctx->stream); s = ds.p; r = dr.p; }
        rccl_check(api().AllGather(s, r, bytes, ncclUint8, comm, ctx->stream), "ncclAllGather");
        if (!device) d2h(recv, dr.p, bytes * world, ctx->stream);                    // synchronises
        else dsync(ctx->stream);
    }
    void all_to_all_v(skh_ctx* ctx, const void* send, const uint64_t* send_cnt, const uint64_t* send_off, void* recv, const uint64_t* recv_cnt,
                      const uint64_t* recv_off, bool device) override {
        uint64_t sb = 0, rb = 0;
        for (int r = 0; r < world; r++) { sb = std::max(sb, send_off[r] + send_cnt[r]); rb = std::max(rb, recv_off[r] + recv_cnt[r]); }
        if (send_cnt[rank] != recv_cnt[rank]) throw Error("all_to_all_v: own send and receive sizes differ");
        if (!device) {
            // host buffers: the rank's own share never visits the device (the result rows of a world of one -- 6.8 MB for config 4 -- made a PCIe round trip), and
            // only what peers send or get is staged
            if (send_cnt[rank]) memcpy((char*)recv + recv_off[rank], (const char*)send + send_off[rank], send_cnt[rank]);
            bool peers = false;
            for (int r = 0; r < world; r++) if (r != rank && (send_cnt[r] || recv_cnt[r])) peers = true;
            if (!peers) return;
            DBuf<char> ds(sb + 1), dr(rb + 1);
            for (int r = 0; r < world; r++) if (r != rank && send_cnt[r]) h2d(ds.p + send_off[r], (const char*)send + send_off[r], send_cnt[r], ctx->stream);   // (pageable source: through the pinned ring, dev.h)
            rccl_check(api().GroupStart(), "ncclGroupStart");
            for (int r = 0; r < world; r++) {
                if (r == rank) continue;
                if (send_cnt[r]) rccl_check(api().Send(ds.p + send_off[r], send_cnt[r], ncclUint8, r, comm, ctx->stream), "ncclSend");
                if (recv_cnt[r]) rccl_check(api().Recv(dr.p + recv_off[r], recv_cnt[r], ncclUint8, r, comm, ctx->stream), "ncclRecv");
            }
            rccl_check(api().GroupEnd(), "ncclGroupEnd");
            dsync(ctx->stream);
            for (int r = 0; r < world; r++) if (r != rank && recv_cnt[r]) d2h((char*)recv + recv_off[r], dr.p + recv_off[r], recv_cnt[r], ctx->stream);   // (pageable destination: through the pinned ring)                                                       // (the staging buffers go with this scope: nothing of theirs may still be queued)
            return;
        }
        const char* s = (const char*)send; char* rv = (char*)recv;
        if (send_cnt[rank]) d2d(rv + recv_off[rank], s + send_off[rank], send_cnt[rank], ctx->stream);   // own share: a device copy, not a message
        rccl_check(api().GroupStart(), "ncclGroupStart");
        for (int r = 0; r < world; r++) {
            if (r == rank) continue;
            if (send_cnt[r]) rccl_check(api().Send(s + send_off[r], send_cnt[r], ncclUint8, r, comm, ctx->stream), "ncclSend");
            if (recv_cnt[r]) rccl_check(api().Recv(rv + recv_off[r], recv_cnt[r], ncclUint8, r, comm, ctx->stream), "ncclRecv");
        }
        rccl_check(api().GroupEnd(), "ncclGroupEnd");
        dsync(ctx->stream);
    }
    // asynchronous form: the grouped send/recv runs on the context's SECOND stream behind an event of the first (the send buffer is complete there); _end
    // makes the first stream wait for it on the device -- the host never blocks, and what is queued on the first stream in between (the home pairs'
    // table build and chaining, dist.hip) runs beside the transfers
    std::unique_ptr<DevEvent> ev_ready, ev_t0, ev_t1, ev_w0, ev_w1; bool open = false, timed = false;
    void exchange_begin(skh_ctx* ctx, const void* send, const uint64_t* send_cnt, const uint64_t* send_off, void* recv, const uint64_t* recv_cnt,
                        const uint64_t* recv_off) override {
        if (open) throw Error("exchange_begin: an exchange is already open");
        if (!ev_ready) { ev_ready.reset(new DevEvent()); ev_t0.reset(new DevEvent()); ev_t1.reset(new DevEvent()); ev_w0.reset(new DevEvent()); ev_w1.reset(new DevEvent()); }
        const char* s = (const char*)send; char* rv = (char*)recv;
        ev_ready->record(ctx->stream); ev_ready->make_wait(ctx->stream2);
        ev_t0->record(ctx->stream2);
        if (send_cnt[rank]) {                                                       // own share: a device copy, not a message
            if (send_cnt[rank] != recv_cnt[rank]) throw Error("exchange: own send and receive sizes differ");
            d2d(rv + recv_off[rank], s + send_off[rank], send_cnt[rank], ctx->stream2);
        }
        rccl_check(api().GroupStart(), "ncclGroupStart");
        for (int r = 0; r < world; r++) {
            if (r == rank) continue;
            if (send_cnt[r]) rccl_check(api().Send(s + send_off[r], send_cnt[r], ncclUint8, r, comm, ctx->stream2), "ncclSend");
            if (recv_cnt[r]) rccl_check(api().Recv(rv + recv_off[r], recv_cnt[r], ncclUint8, r, comm, ctx->stream2), "ncclRecv");
        }
        rccl_check(api().GroupEnd(), "ncclGroupEnd");
        ev_t1->record(ctx->stream2);
        open = true; timed = false;
    }
    void exchange_end(skh_ctx* ctx) override {
        if (!open) return;
        open = false;
        ev_w0->record(ctx->stream); ev_t1->make_wait(ctx->stream); ev_w1->record(ctx->stream);
        timed = true;
    }
    void exchange_times(uint64_t* total_us, uint64_t* wait_us) override {
        if (total_us) *total_us = 0;
        if (wait_us) *wait_us = 0;
        if (!timed) return;
        ev_w1->wait();
        if (total_us) *total_us = (uint64_t)(DevEvent::ms(*ev_t0, *ev_t1) * 1000.f);
        if (wait_us) *wait_us = (uint64_t)(DevEvent::ms(*ev_w0, *ev_w1) * 1000.f);
    }
};

}  // namespace

}  // namespace skh

using namespace skh;

extern "C" {

int skh_comm_unique_id(uint8_t id[SKH_COMM_ID_BYTES]) {
    static_assert(sizeof(ncclUniqueId) == SKH_COMM_ID_BYTES, "RCCL unique id size");
    if (!id) return SKH_ERR_INVALID;
    try {
        ncclUniqueId u;
        rccl_check(api().GetUniqueId(&u), "ncclGetUniqueId");
        memcpy(id, &u, sizeof(u));
        return SKH_OK;
    } catch (...) { return SKH_ERR_DEVICE; }
}

int skh_comm_create_rccl(skh_ctx* ctx, const uint8_t id[SKH_COMM_ID_BYTES], int rank, int world, skh_comm** out) {
    if (!ctx || !id || !out) return SKH_ERR_INVALID;
    *out = nullptr;
    try {
        if (world < 1 || rank < 0 || rank >= world) { ctx->err = "bad rank / world size"; return SKH_ERR_INVALID; }
        PinScope scope(ctx->device, &ctx->ring);                                    // binds the thread to the context's device
        std::unique_ptr<RcclTransport> t(new RcclTransport());
        t->rank = rank; t->world = world;
        ncclUniqueId u; memcpy(&u, id, sizeof(u));
        rccl_check(api().CommInitRank(&t->comm, world, u, rank), "ncclCommInitRank");
        skh_comm* c = new skh_comm(); c->t = t.release();
        *out = c;
        return SKH_OK;
    } catch (const std::exception& e) { ctx->err = e.what(); return SKH_ERR_DEVICE; }
    catch (...) { ctx->err = "unknown error"; return SKH_ERR_INTERNAL; }
}

}  // extern "C"
