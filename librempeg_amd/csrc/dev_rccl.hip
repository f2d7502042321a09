// HIP side of libswscale_hip -- the RCCL step of sws_scale_frames() (BASELINE north_star / SURVEY 8e: "RCCL broadcast of filter tables over xGMI").
//
// The path has ONE exchange: the tables a context's kernels read (filter banks, plan rows, LUTs, geometry) are a pure function of the options, built on the
// host, and every GPU that converts frames of the call needs a copy.  Without this file each GPU gets them by a host -> device copy of its own (PCIe, n times
// the bytes through the host link).  With the option "rccl_tables" (or SWS_HIP_RCCL=1) the HOME GPU's blocks are uploaded once and ncclBroadcast() -- one
// grouped call per table block over a communicator of the GPUs in use, each rank on its context's stream -- carries them to the peers over xGMI; the peers'
// own planner runs have already laid out identical blocks (same sizes, same hashes: checked) and only their upload is replaced.  Frames never cross GPUs.
//
// librccl.so is dlopen()ed on first use: the library has no link-time dependency on it, and anything that goes wrong -- no librccl, a communicator that cannot
// be made, a block the home GPU does not hold, a failing call -- falls back to the host -> device copies, which are kept ready (TableDeferred::data).
// Opt-in and UNMEASURED ON HARDWARE: the build and test boxes of rounds 1 - 6 have one GPU (a communicator of one rank never reaches this code).
#include <dlfcn.h>

#include <cstdlib>
#include <map>
#include <mutex>
#include <string>
#include <vector>

#include <rccl/rccl.h>          // types and prototypes only; every call goes through the dlsym table below

#include "dev_internal.hpp"

namespace swship {

namespace {
struct RcclApi {
    void *lib = nullptr;
    decltype(&ncclCommInitAll) CommInitAll = nullptr;
    decltype(&ncclCommDestroy) CommDestroy = nullptr;
    decltype(&ncclGroupStart) GroupStart = nullptr;
    decltype(&ncclGroupEnd) GroupEnd = nullptr;
    decltype(&ncclBroadcast) Broadcast = nullptr;
    decltype(&ncclGetErrorString) GetErrorString = nullptr;
    bool ok = false;
};
struct RcclWorld {                 // (leaked on purpose, like the guard registry: contexts may outlive static destructors)
    std::mutex mu;
    RcclApi api;
    bool tried = false;
    std::map<std::vector<int>, std::vector<ncclComm_t>> comms;   // device list -> one communicator per rank
};
RcclWorld &world() { static RcclWorld *w = new RcclWorld(); return *w; }

bool load_api(RcclApi &a)
{
    for (const char *name : { "librccl.so.1", "librccl.so" }) {
        a.lib = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
        if (a.lib) break;
    }
    if (!a.lib) return false;
    a.CommInitAll = (decltype(a.CommInitAll))dlsym(a.lib, "ncclCommInitAll");
    a.CommDestroy = (decltype(a.CommDestroy))dlsym(a.lib, "ncclCommDestroy");
    a.GroupStart = (decltype(a.GroupStart))dlsym(a.lib, "ncclGroupStart");
    a.GroupEnd = (decltype(a.GroupEnd))dlsym(a.lib, "ncclGroupEnd");
    a.Broadcast = (decltype(a.Broadcast))dlsym(a.lib, "ncclBroadcast");
    a.GetErrorString = (decltype(a.GetErrorString))dlsym(a.lib, "ncclGetErrorString");
    a.ok = a.CommInitAll && a.CommDestroy && a.GroupStart && a.GroupEnd && a.Broadcast && a.GetErrorString;
    return a.ok;
}
} // namespace

bool rccl_tables_wanted(const SwsInternal *c)
{
    if (c->tune.rccl_tables) return true;
    static const bool env = std::getenv("SWS_HIP_RCCL") && std::atoi(std::getenv("SWS_HIP_RCCL")) > 0;
    return env;
}

// The peers' deferred table uploads (dev_state.hip table_put with DeviceState::defer_uploads) are delivered: by broadcast from the home state's blocks where
// that is possible, by the host -> device copy otherwise.  `states`: home first, then the peers that planned in this call.  Returns 0, or a HIP error of the
// fallback copies (an RCCL failure is not an error of the call).
int rccl_deliver_tables(SwsInternal *c, DeviceState *home, const std::vector<DeviceState *> &peers)
{
    bool any = false;
    for (DeviceState *d : peers) any = any || !d->deferred.empty();
    if (!any) return 0;
    // every deferred block needs a twin on the home GPU: same size, same contents
    bool twins = true;
    std::vector<std::vector<const TableRecord *>> src(peers.size());
    for (size_t r = 0; r < peers.size() && twins; r++)
        for (const TableDeferred &t : peers[r]->deferred) {
            const TableRecord *hit = nullptr;
            for (const TableRecord &h : home->tab_recs) if (h.bytes == t.bytes && h.hash == t.hash) { hit = &h; break; }
            if (!hit) { twins = false; break; }
            src[r].push_back(hit);
        }
    // ... and the peers the same list of blocks as each other (one broadcast per block, every rank in it)
    for (size_t r = 1; r < peers.size() && twins; r++) {
        twins = peers[r]->deferred.size() == peers[0]->deferred.size();
        for (size_t k = 0; twins && k < peers[r]->deferred.size(); k++) twins = src[r][k] == src[0][k];
    }
    bool done = false;
    if (twins) {
        RcclWorld &W = world();
        std::lock_guard<std::mutex> lk(W.mu);
        if (!W.tried) { W.tried = true; if (!load_api(W.api)) log_msg(c, 1, "rccl_tables: librccl.so not available (%s); tables go by host -> device copies\n", dlerror()); }
        if (W.api.ok) {
            std::vector<int> devs{ home->device };
            for (DeviceState *d : peers) devs.push_back(d->device);
            auto it = W.comms.find(devs);
            if (it == W.comms.end()) {
                std::vector<ncclComm_t> cm(devs.size(), nullptr);
                const ncclResult_t rc = W.api.CommInitAll(cm.data(), (int)devs.size(), devs.data());
                if (rc != ncclSuccess) { log_msg(c, 0, "rccl_tables: ncclCommInitAll over %zu GPUs failed (%s); tables go by host -> device copies\n", devs.size(), W.api.GetErrorString(rc)); cm.clear(); }
                it = W.comms.emplace(devs, std::move(cm)).first;
            }
            const std::vector<ncclComm_t> &cm = it->second;
            if (!cm.empty()) {
                done = true;
                for (size_t k = 0; k < peers[0]->deferred.size() && done; k++) {
                    const TableRecord *h = src[0][k];
                    ncclResult_t rc = W.api.GroupStart();
                    // rank 0 = the home GPU (root, in place), rank r + 1 = peer r; each on its own context's stream, behind what that stream already holds
                    if (rc == ncclSuccess) { (void)hipSetDevice(home->device); rc = W.api.Broadcast(h->dst, const_cast<void *>(h->dst), h->bytes, ncclUint8, 0, cm[0], home->stream); }
                    for (size_t r = 0; r < peers.size() && rc == ncclSuccess; r++) {
                        (void)hipSetDevice(peers[r]->device);
                        rc = W.api.Broadcast(peers[r]->deferred[k].dst, peers[r]->deferred[k].dst, h->bytes, ncclUint8, 0, cm[r + 1], peers[r]->stream);   // (the send side is read on the root only)
                    }
                    const ncclResult_t rc2 = W.api.GroupEnd();
                    if (rc != ncclSuccess || rc2 != ncclSuccess) {
                        log_msg(c, 0, "rccl_tables: broadcast failed (%s); tables go by host -> device copies\n", W.api.GetErrorString(rc != ncclSuccess ? rc : rc2));
                        done = false;
                    }
                }
                if (done) log_msg(c, 2, "rccl_tables: %zu table block(s) broadcast from GPU %d to %zu peer GPU(s)\n", peers[0]->deferred.size(), home->device, peers.size());
            }
        }
    }
    // the uploads are part of planning: the peers wait for theirs (like table_put); the fallback is the copy table_put would have made
    for (DeviceState *d : peers) {
        HIPCHK(hipSetDevice(d->device));
        if (!done)
            for (const TableDeferred &t : d->deferred) HIPCHK(hipMemcpyAsync(t.dst, t.data.data(), t.bytes, hipMemcpyHostToDevice, d->stream));
        HIPCHK(hipStreamSynchronize(d->stream));
        d->deferred.clear();
        d->defer_uploads = false;
    }
    if (done) { HIPCHK(hipSetDevice(home->device)); HIPCHK(hipStreamSynchronize(home->stream)); }
    return 0;
}

} // namespace swship
