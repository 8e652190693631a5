// hctr_comm / hctr_gather_labels: the one collective of the multi-GPU path for plain-C callers -
// one process per GPU, contiguous line shards, ONE all-gather of the packed decoded labels over RCCL (xGMI).
// The reference has no counterpart (its inference is single-device, test.py:143-148; NCCL appears only in training
// DDP, main.py:226-237). Python callers use torch.distributed instead (dist.py); both move the same packed layout.
//
// RCCL is bound at run time (dlopen): a process that already holds a copy (PyTorch ships its own librccl.so) keeps
// using that one, and single-GPU users never load it.
#include "../../include/hctr_hip.h"

#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <cstring>
#include <mutex>
#include <new>
#include <string>
#include <vector>

namespace {

struct Rccl {
    void* handle = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*AllGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
};

std::mutex g_mu;
Rccl g_rccl;
thread_local std::string g_comm_error;

int comm_fail(int code, const std::string& msg) noexcept {
    try { g_comm_error = msg; } catch (...) {}
    return code;
}

int load_rccl() {
    std::lock_guard<std::mutex> lk(g_mu);
    if (g_rccl.handle) return HCTR_OK;
    void* h = nullptr;
    for (const char* name : {"librccl.so.1", "librccl.so"}) {
        h = dlopen(name, RTLD_NOW | RTLD_NOLOAD);           // a copy the process already holds (PyTorch's)
        if (h) break;
    }
    if (!h)
        for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
            h = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
            if (h) break;
        }
    if (!h) {
        const char* de = dlerror();                             // (a second call would return NULL: it clears the error)
        return comm_fail(HCTR_ERR_STATE, std::string("librccl not found: ") + (de ? de : ""));
    }
    Rccl r;
    r.handle = h;
    r.GetUniqueId = (decltype(r.GetUniqueId))dlsym(h, "ncclGetUniqueId");
    r.CommInitRank = (decltype(r.CommInitRank))dlsym(h, "ncclCommInitRank");
    r.AllGather = (decltype(r.AllGather))dlsym(h, "ncclAllGather");
    r.CommDestroy = (decltype(r.CommDestroy))dlsym(h, "ncclCommDestroy");
    r.GetErrorString = (decltype(r.GetErrorString))dlsym(h, "ncclGetErrorString");
    if (!r.GetUniqueId || !r.CommInitRank || !r.AllGather || !r.CommDestroy || !r.GetErrorString)
        return comm_fail(HCTR_ERR_STATE, "librccl lacks an expected symbol");
    g_rccl = r;
    return HCTR_OK;
}

}  // namespace

struct hctr_comm {
    ncclComm_t comm = nullptr;
    int rank = 0, world = 1, device = 0;
    hipStream_t stream = nullptr;
    int32_t* d_send = nullptr;
    int32_t* d_recv = nullptr;
    size_t cap_send = 0, cap_recv = 0;          // elements
};

#define COMM_GUARD(...)                                                           \
    try { __VA_ARGS__ } catch (const std::bad_alloc&) { return comm_fail(HCTR_ERR_NOMEM, "out of host memory"); } \
    catch (...) { return comm_fail(HCTR_ERR_STATE, "unexpected C++ exception"); }

extern "C" {

const char* hctr_comm_last_error(void) { return g_comm_error.c_str(); }

int hctr_comm_unique_id(void* id128) {
    COMM_GUARD(
        if (!id128) return comm_fail(HCTR_ERR_ARG, "id128 is NULL");
        int rc = load_rccl();
        if (rc != HCTR_OK) return rc;
        ncclUniqueId id;
        ncclResult_t r = g_rccl.GetUniqueId(&id);
        if (r != ncclSuccess) return comm_fail(HCTR_ERR_HIP, std::string("ncclGetUniqueId: ") + g_rccl.GetErrorString(r));
        static_assert(sizeof(id) == HCTR_COMM_ID_BYTES, "unique id size");
        memcpy(id128, &id, sizeof(id));
        return HCTR_OK;
    )
}

int hctr_comm_create(hctr_comm** out, const void* id128, int rank, int world, int device) {
    COMM_GUARD(
        if (!out || !id128) return comm_fail(HCTR_ERR_ARG, "NULL argument");
        *out = nullptr;
        if (world < 1 || rank < 0 || rank >= world) return comm_fail(HCTR_ERR_ARG, "bad rank / world");
        int rc = load_rccl();
        if (rc != HCTR_OK) return rc;
        if (hipSetDevice(device) != hipSuccess) return comm_fail(HCTR_ERR_HIP, "hipSetDevice failed");
        hctr_comm* c = new hctr_comm();
        c->rank = rank; c->world = world; c->device = device;
        if (hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess) {
            delete c;
            return comm_fail(HCTR_ERR_HIP, "hipStreamCreate failed");
        }
        ncclUniqueId id;
        memcpy(&id, id128, sizeof(id));
        ncclResult_t r = g_rccl.CommInitRank(&c->comm, world, id, rank);
        if (r != ncclSuccess) {
            (void)hipStreamDestroy(c->stream);
            delete c;
            return comm_fail(HCTR_ERR_HIP, std::string("ncclCommInitRank: ") + g_rccl.GetErrorString(r));
        }
        *out = c;
        return HCTR_OK;
    )
}

void hctr_comm_destroy(hctr_comm* c) {
    if (!c) return;
    (void)hipSetDevice(c->device);
    if (c->comm) (void)g_rccl.CommDestroy(c->comm);
    if (c->d_send) (void)hipFree(c->d_send);
    if (c->d_recv) (void)hipFree(c->d_recv);
    if (c->stream) (void)hipStreamDestroy(c->stream);
    delete c;
}

int hctr_gather_labels(hctr_comm* c, const int32_t* labels, const int32_t* lengths, int n_local, int row_stride,
                       int lines_per_rank, int cap, int32_t* out_labels, int32_t* out_lengths) {
    COMM_GUARD(
        if (!c) return comm_fail(HCTR_ERR_ARG, "comm is NULL");
        // Arguments every rank passes identically (lines_per_rank, cap) decide the collective's size: if THEY are bad no
        // rank can enter it, so return at once. Anything wrong with this rank's own data (a NULL pointer, a sequence
        // longer than cap, a failed copy) must not leave the other ranks waiting in the all-gather: this rank still takes
        // part, with the sentinel -1 in the length slot of its first row, and EVERY rank returns HCTR_ERR_ARG.
        if (lines_per_rank < 1 || cap < 0) return comm_fail(HCTR_ERR_ARG, "bad shape (lines_per_rank >= 1, cap >= 0)");
        if (!out_labels || !out_lengths) return comm_fail(HCTR_ERR_ARG, "out_labels/out_lengths is NULL");
        if (hipSetDevice(c->device) != hipSuccess) return comm_fail(HCTR_ERR_HIP, "hipSetDevice failed");
        std::string local_err;
        if (n_local < 0 || n_local > lines_per_rank || row_stride < cap) local_err = "bad shape (0 <= n_local <= lines_per_rank, cap <= row_stride)";
        else if (n_local > 0 && (!labels || !lengths)) local_err = "labels/lengths is NULL";
        // packed like dist.pack_labels: [lines_per_rank][1 + cap] int32, column 0 = length, zero padded
        const size_t row = (size_t)1 + cap, per = (size_t)lines_per_rank * row;
        std::vector<int32_t> send(per, 0);
        for (int i = 0; local_err.empty() && i < n_local; ++i) {
            const int n = lengths[i];
            if (n < 0 || n > cap) { local_err = "a label sequence is longer than cap"; break; }
            send[i * row] = n;
            if (n) memcpy(&send[i * row + 1], labels + (size_t)i * row_stride, (size_t)n * 4);
        }
        if (!local_err.empty()) send[0] = -1;
        if (c->cap_send < per) {
            if (c->d_send) (void)hipFree(c->d_send);
            c->d_send = nullptr; c->cap_send = 0;
            if (hipMalloc((void**)&c->d_send, per * 4) != hipSuccess) return comm_fail(HCTR_ERR_NOMEM, "hipMalloc failed");
            c->cap_send = per;
        }
        if (c->cap_recv < per * c->world) {
            if (c->d_recv) (void)hipFree(c->d_recv);
            c->d_recv = nullptr; c->cap_recv = 0;
            if (hipMalloc((void**)&c->d_recv, per * c->world * 4) != hipSuccess) return comm_fail(HCTR_ERR_NOMEM, "hipMalloc failed");
            c->cap_recv = per * c->world;
        }
        std::vector<int32_t> recv(per * c->world);
        hipError_t e = hipMemcpyAsync(c->d_send, send.data(), per * 4, hipMemcpyHostToDevice, c->stream);
        if (e != hipSuccess) {                        // still join the collective (with whatever d_send holds) - see above
            local_err = std::string("H2D: ") + hipGetErrorString(e);
            const int32_t bad = -1;
            (void)hipMemcpy(c->d_send, &bad, 4, hipMemcpyHostToDevice);
        }
        ncclResult_t r = g_rccl.AllGather(c->d_send, c->d_recv, per, ncclInt32, c->comm, c->stream);
        if (r != ncclSuccess) {
            (void)hipStreamSynchronize(c->stream);
            return comm_fail(HCTR_ERR_HIP, std::string("ncclAllGather: ") + g_rccl.GetErrorString(r));
        }
        e = hipMemcpyAsync(recv.data(), c->d_recv, per * c->world * 4, hipMemcpyDeviceToHost, c->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
        if (e != hipSuccess) return comm_fail(HCTR_ERR_HIP, std::string("D2H: ") + hipGetErrorString(e));
        if (!local_err.empty()) return comm_fail(HCTR_ERR_ARG, local_err);
        for (int rk = 0; rk < c->world; ++rk)
            if (recv[(size_t)rk * per] < 0)
                return comm_fail(HCTR_ERR_ARG, "rank " + std::to_string(rk) + " reported bad arguments to hctr_gather_labels");
        const size_t total = (size_t)lines_per_rank * c->world;
        for (size_t i = 0; i < total; ++i) {
            const int n = recv[i * row];
            out_lengths[i] = n;
            memset(out_labels + i * cap, 0, (size_t)cap * 4);
            if (n > 0 && n <= cap) memcpy(out_labels + i * cap, &recv[i * row + 1], (size_t)n * 4);
        }
        return HCTR_OK;
    )
}

}  // extern "C"
