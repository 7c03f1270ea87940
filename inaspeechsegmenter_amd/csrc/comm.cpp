// The single exchange step of the file-parallel archive path (include/iss.h, "multi-GPU"): one ncclAllGather of
// fixed-capacity int32 segment tables over RCCL/xGMI.  librccl is dlopen'ed on first use (single-GPU deployments do
// not need it, and inside a PyTorch process the already-loaded librccl.so.1 is the one that gets bound).
//
// Reference analogue: none on the device side -- the reference collects results through the file system of its Pyro
// workers (scripts/ina_speech_segmenter_pyro_client.py:64-74); SURVEY.md section 8(e) defines this exchange.
#include "iss_internal.h"
#include <dlfcn.h>
#include <chrono>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <thread>

namespace {

// the subset of rccl.h this file needs (types are ABI-stable: opaque comm pointer, 128-byte id, int enums)
typedef struct { char internal[128]; } NcclUniqueId;
typedef void* NcclComm;
enum { NCCL_SUCCESS = 0 };
enum { NCCL_INT32 = 2, NCCL_FLOAT64 = 8 };       // ncclDataType_t
enum { NCCL_MAX = 2 };                            // ncclRedOp_t

struct Rccl {
    void* h = nullptr;
    int (*GetUniqueId)(NcclUniqueId*) = nullptr;
    int (*CommInitRank)(NcclComm*, int, NcclUniqueId, int) = nullptr;
    int (*CommDestroy)(NcclComm) = nullptr;
    int (*CommAbort)(NcclComm) = nullptr;
    int (*CommCount)(const NcclComm, int*) = nullptr;
    int (*CommUserRank)(const NcclComm, int*) = nullptr;
    int (*CommGetAsyncError)(NcclComm, int*) = nullptr;
    int (*GetVersion)(int*) = nullptr;
    std::string path;
    int (*AllGather)(const void*, void*, size_t, int, NcclComm, hipStream_t) = nullptr;
    int (*AllReduce)(const void*, void*, size_t, int, int, NcclComm, hipStream_t) = nullptr;
    const char* (*GetErrorString)(int) = nullptr;
    std::string err;
};
Rccl g_rccl;
std::mutex g_mu;

bool load_rccl() {
    std::lock_guard<std::mutex> l(g_mu);
    if (g_rccl.h) return true;
    const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    for (const char* n : names) {
        g_rccl.h = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
        if (g_rccl.h) break;
        g_rccl.err = dlerror();
    }
    if (!g_rccl.h) return false;
#define ISS_SYM(field, name)                                                          \
    *(void**)(&g_rccl.field) = dlsym(g_rccl.h, name);                                 \
    if (!g_rccl.field) { g_rccl.err = "symbol " name " missing in librccl"; dlclose(g_rccl.h); g_rccl.h = nullptr; return false; }
    ISS_SYM(GetUniqueId, "ncclGetUniqueId")
    ISS_SYM(CommInitRank, "ncclCommInitRank")
    ISS_SYM(CommDestroy, "ncclCommDestroy")
    ISS_SYM(CommAbort, "ncclCommAbort")
    ISS_SYM(CommCount, "ncclCommCount")
    ISS_SYM(CommUserRank, "ncclCommUserRank")
    ISS_SYM(CommGetAsyncError, "ncclCommGetAsyncError")
    ISS_SYM(GetVersion, "ncclGetVersion")
    ISS_SYM(AllGather, "ncclAllGather")
    ISS_SYM(AllReduce, "ncclAllReduce")
    ISS_SYM(GetErrorString, "ncclGetErrorString")
#undef ISS_SYM
    Dl_info info;
    if (dladdr((void*)g_rccl.AllGather, &info) && info.dli_fname) g_rccl.path = info.dli_fname;
    return true;
}

// Wait for the collective(s) enqueued on the context's stream.  A peer that died before reaching the collective would
// leave hipStreamSynchronize blocked forever: poll instead, watch the communicator's asynchronous error state, and after
// ISS_COMM_TIMEOUT_S seconds (default 1800) abort the communicator (ncclCommAbort) so that THIS rank fails loudly too.
int wait_collective(iss_ctx* c, const char* what) {
    static const double limit = [] { const char* e = getenv("ISS_COMM_TIMEOUT_S"); const double v = e ? atof(e) : 0.0; return v > 0.0 ? v : 1800.0; }();
    const auto t0 = std::chrono::steady_clock::now();
    for (unsigned spin = 0;; ++spin) {
        const hipError_t q = hipStreamQuery(c->stream);
        if (q == hipSuccess) return ISS_OK;
        if (q != hipErrorNotReady) return iss_fail(c, ISS_EHIP, "%s: hipStreamQuery failed: %s", what, hipGetErrorString(q));
        int aerr = NCCL_SUCCESS;
        const bool bad = g_rccl.CommGetAsyncError((NcclComm)c->comm, &aerr) != NCCL_SUCCESS || aerr != NCCL_SUCCESS;
        const double waited = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        if (bad || waited > limit) {
            (void)g_rccl.CommAbort((NcclComm)c->comm);
            c->comm = nullptr; c->comm_rank = 0; c->comm_world = 1;
            if (bad) return iss_fail(c, ISS_EHIP, "%s: RCCL reported an asynchronous error (%s); communicator aborted", what, g_rccl.GetErrorString(aerr));
            return iss_fail(c, ISS_ETIMEOUT, "%s: no completion after %.0f s (ISS_COMM_TIMEOUT_S): a peer rank is gone or stuck; communicator aborted", what, waited);
        }
        if (spin < 2000) std::this_thread::yield(); else std::this_thread::sleep_for(std::chrono::microseconds(200));
    }
}

#define ISS_NCCL(c, call)                                                                        \
    do {                                                                                         \
        int r__ = (call);                                                                        \
        if (r__ != NCCL_SUCCESS)                                                                 \
            return iss_fail((c), ISS_EHIP, "%s failed: %s (%s:%d)", #call, g_rccl.GetErrorString(r__), __FILE__, __LINE__); \
    } while (0)

}  // namespace

extern "C" int iss_comm_unique_id(iss_ctx* c, uint8_t* id_out) {
    if (!c || !id_out) return iss_fail(c, ISS_EINVAL, "iss_comm_unique_id: NULL argument");
    if (!load_rccl()) return iss_fail(c, ISS_ENODEV, "cannot load librccl: %s", g_rccl.err.c_str());
    static_assert(sizeof(NcclUniqueId) == ISS_COMM_ID_BYTES, "");
    NcclUniqueId id;
    ISS_NCCL(c, g_rccl.GetUniqueId(&id));
    memcpy(id_out, &id, sizeof(id));
    return ISS_OK;
}

extern "C" int iss_comm_init(iss_ctx* c, const uint8_t* id_in, int32_t rank, int32_t world) {
    if (!c || !id_in || world < 1 || rank < 0 || rank >= world) return iss_fail(c, ISS_EINVAL, "iss_comm_init: bad argument");
    if (c->comm) return iss_fail(c, ISS_ESTATE, "iss_comm_init: communicator already initialised");
    if (!load_rccl()) return iss_fail(c, ISS_ENODEV, "cannot load librccl: %s", g_rccl.err.c_str());
    ISS_HIP(c, hipSetDevice(c->device));
    NcclUniqueId id;
    memcpy(&id, id_in, sizeof(id));
    NcclComm comm = nullptr;
    ISS_NCCL(c, g_rccl.CommInitRank(&comm, world, id, rank));
    c->comm = comm; c->comm_rank = rank; c->comm_world = world;
    return ISS_OK;
}

extern "C" int iss_comm_destroy(iss_ctx* c) {
    if (!c) return ISS_EINVAL;
    if (c->comm && g_rccl.CommDestroy) {
        (void)hipSetDevice(c->device);
        (void)hipStreamSynchronize(c->stream);
        (void)g_rccl.CommDestroy((NcclComm)c->comm);
    }
    c->comm = nullptr; c->comm_rank = 0; c->comm_world = 1;
    return ISS_OK;
}

extern "C" int iss_allgather_segments(iss_ctx* c, const int32_t* local_rows, int32_t n_local, int32_t capacity,
                                      int32_t* all_rows, int32_t* counts) {
    if (!c || n_local < -1 || capacity < 1 || (n_local > 0 && !local_rows) || !all_rows || !counts)
        return iss_fail(c, ISS_EINVAL, "iss_allgather_segments: bad argument");
    if (!c->comm) return iss_fail(c, ISS_ESTATE, "iss_allgather_segments: call iss_comm_init first");
    ISS_HIP(c, hipSetDevice(c->device));
    const int world = c->comm_world;
    const size_t slab = ((size_t)capacity + 1) * 4;               // int32 per rank: header row + capacity rows
    int rc;
    if ((rc = iss_reserve(c, c->comm_send, slab * 4))) return rc;
    if ((rc = iss_reserve(c, c->comm_recv, slab * 4 * world))) return rc;
    // header row (n_rows, capacity, rank, 0) + the first min(n_local, capacity) rows, through a pinned staging buffer
    std::vector<int32_t> buf(slab, 0);
    buf[0] = n_local; buf[1] = capacity; buf[2] = c->comm_rank;     // n_local == -1: "my local work failed" (see iss.h)
    const int k = n_local < capacity ? n_local : capacity;
    if (k > 0) memcpy(&buf[4], local_rows, (size_t)k * 16);
    void* pinned; int slot;
    if ((rc = iss_stage_host(c, buf.data(), slab * 4, &pinned, &slot))) return rc;
    ISS_HIP(c, hipMemcpyAsync(c->comm_send.p, pinned, slab * 4, hipMemcpyHostToDevice, c->stream));
    iss_stage_mark(c, slot);
    ISS_NCCL(c, g_rccl.AllGather(c->comm_send.p, c->comm_recv.p, slab, NCCL_INT32, (NcclComm)c->comm, c->stream));
    if ((rc = wait_collective(c, "iss_allgather_segments"))) return rc;      // (before the read-back: a copy into pageable memory blocks)
    std::vector<int32_t> got(slab * world);
    ISS_HIP(c, hipMemcpyAsync(got.data(), c->comm_recv.p, slab * 4 * world, hipMemcpyDeviceToHost, c->stream));
    ISS_HIP(c, hipStreamSynchronize(c->stream));
    for (int r = 0; r < world; ++r) {
        const int32_t* s = &got[(size_t)r * slab];
        if (s[1] != capacity || s[2] != r) return iss_fail(c, ISS_ESTATE, "iss_allgather_segments: rank %d sent capacity %d / rank %d (expected %d / %d)", r, s[1], s[2], capacity, r);
        counts[r] = s[0];                                         // -1: rank r reports that its local work failed
        const int kk = s[0] < capacity ? s[0] : capacity;
        if (kk > 0) memcpy(all_rows + (size_t)r * capacity * 4, s + 4, (size_t)kk * 16);
    }
    return ISS_OK;
}

extern "C" int iss_comm_allreduce_max(iss_ctx* c, double* value) {
    if (!c || !value) return iss_fail(c, ISS_EINVAL, "iss_comm_allreduce_max: NULL argument");
    if (!c->comm) return iss_fail(c, ISS_ESTATE, "iss_comm_allreduce_max: call iss_comm_init first");
    ISS_HIP(c, hipSetDevice(c->device));
    int rc;
    if ((rc = iss_reserve(c, c->comm_send, 64))) return rc;
    ISS_HIP(c, hipMemcpyAsync(c->comm_send.p, value, 8, hipMemcpyHostToDevice, c->stream));
    ISS_HIP(c, hipStreamSynchronize(c->stream));
    ISS_NCCL(c, g_rccl.AllReduce(c->comm_send.p, (char*)c->comm_send.p + 16, 1, NCCL_FLOAT64, NCCL_MAX, (NcclComm)c->comm, c->stream));
    if ((rc = wait_collective(c, "iss_comm_allreduce_max"))) return rc;
    ISS_HIP(c, hipMemcpyAsync(value, (char*)c->comm_send.p + 16, 8, hipMemcpyDeviceToHost, c->stream));
    ISS_HIP(c, hipStreamSynchronize(c->stream));
    return ISS_OK;
}

extern "C" int iss_comm_info(iss_ctx* c, int32_t* world, int32_t* rank, int32_t* version, char* lib_path, int32_t lib_path_len) {
    if (!c) return ISS_EINVAL;
    if (!c->comm) return iss_fail(c, ISS_ESTATE, "iss_comm_info: call iss_comm_init first");
    int n = 0, r = 0, v = 0;
    ISS_NCCL(c, g_rccl.CommCount((NcclComm)c->comm, &n));      // what RCCL itself says, not what iss_comm_init was told
    ISS_NCCL(c, g_rccl.CommUserRank((NcclComm)c->comm, &r));
    (void)g_rccl.GetVersion(&v);
    if (world) *world = n;
    if (rank) *rank = r;
    if (version) *version = v;
    if (lib_path && lib_path_len > 0) { strncpy(lib_path, g_rccl.path.c_str(), (size_t)lib_path_len - 1); lib_path[lib_path_len - 1] = 0; }
    return ISS_OK;
}
