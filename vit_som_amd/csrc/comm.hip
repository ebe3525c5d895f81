// Data-parallel gradient exchange: one RCCL communicator per process (one process per GPU), sum all-reduce of a
// slice of the flat gradient arena on a caller-chosen HIP stream.  Replaces what the reference gets implicitly
// from Lightning's DDP wrapper (experiments/benchmarking/train_vit_som.py:44-45,86-91: `devices` > 1 -> DDPStrategy ->
// bucketed NCCL all-reduce of the gradients).
//
// RCCL is bound at run time (dlopen): libvitsom_hip.so has no link-time dependency on it, and a process that
// already carries an RCCL (torch.distributed's "nccl" backend) shares that copy instead of loading a second one.
#include "common.h"

#include <dlfcn.h>
#include <mutex>
#include <stdlib.h>

#include <rccl/rccl.h>

namespace vsom {
namespace {

struct Rccl {
    void* handle = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
};

std::mutex g_mu;
Rccl g_rccl;
ncclComm_t g_comm = nullptr;      // the per-process communicator (the only mutable global state of the library)
int g_world = 0, g_rank = -1;

int load_rccl() {
    if (g_rccl.handle) return VSOM_OK;
    const char* env = getenv("VSOM_RCCL_PATH");
    void* h = nullptr;
    if (env && *env) h = dlopen(env, RTLD_NOW | RTLD_GLOBAL);
    const char* names[] = {"librccl.so", "librccl.so.1"};
    for (int pass = 0; pass < 2 && !h; ++pass)            // pass 0: a copy the process already holds; pass 1: load one
        for (const char* n : names) {
            h = dlopen(n, RTLD_NOW | RTLD_GLOBAL | (pass == 0 ? RTLD_NOLOAD : 0));
            if (h) break;
        }
    if (!h) h = dlopen("/opt/rocm/lib/librccl.so", RTLD_NOW | RTLD_GLOBAL);
    VSOM_REQUIRE(h != nullptr, VSOM_EUNSUPPORTED, "vsom_comm: cannot load librccl.so (%s); set VSOM_RCCL_PATH", dlerror());
    Rccl r;
    r.handle = h;
    r.GetUniqueId = reinterpret_cast<decltype(r.GetUniqueId)>(dlsym(h, "ncclGetUniqueId"));
    r.CommInitRank = reinterpret_cast<decltype(r.CommInitRank)>(dlsym(h, "ncclCommInitRank"));
    r.AllReduce = reinterpret_cast<decltype(r.AllReduce)>(dlsym(h, "ncclAllReduce"));
    r.CommDestroy = reinterpret_cast<decltype(r.CommDestroy)>(dlsym(h, "ncclCommDestroy"));
    r.GetErrorString = reinterpret_cast<decltype(r.GetErrorString)>(dlsym(h, "ncclGetErrorString"));
    VSOM_REQUIRE(r.GetUniqueId && r.CommInitRank && r.AllReduce && r.CommDestroy && r.GetErrorString, VSOM_EUNSUPPORTED,
                 "vsom_comm: librccl.so lacks a required symbol");
    g_rccl = r;
    return VSOM_OK;
}

int rccl_status(ncclResult_t rc, const char* what) {
    if (rc == ncclSuccess) return VSOM_OK;
    set_error("%s: %s", what, g_rccl.GetErrorString ? g_rccl.GetErrorString(rc) : "RCCL error");
    return 1000 + (int)rc;            // positive like a hipError_t, outside its range
}

}  // namespace
}  // namespace vsom

using namespace vsom;

extern "C" {

int vsom_comm_unique_id(void* id_out) {
    VSOM_REQUIRE(id_out, VSOM_EINVAL, "comm_unique_id: null pointer");
    std::lock_guard<std::mutex> lk(g_mu);
    int rc = load_rccl();
    if (rc) return rc;
    ncclUniqueId id;
    rc = rccl_status(g_rccl.GetUniqueId(&id), "ncclGetUniqueId");
    if (rc) return rc;
    static_assert(sizeof(id) == VSOM_COMM_ID_BYTES, "unique id size");
    memcpy(id_out, &id, sizeof(id));
    return VSOM_OK;
}

int vsom_comm_init(const void* unique_id, int world_size, int rank) {
    VSOM_REQUIRE(unique_id && world_size > 0 && rank >= 0 && rank < world_size, VSOM_EINVAL,
                 "comm_init: bad arguments (world_size=%d rank=%d)", world_size, rank);
    std::lock_guard<std::mutex> lk(g_mu);
    VSOM_REQUIRE(g_comm == nullptr, VSOM_EINVAL, "comm_init: this process already holds a communicator (vsom_comm_destroy first)");
    int rc = load_rccl();
    if (rc) return rc;
    ncclUniqueId id;
    memcpy(&id, unique_id, sizeof(id));
    ncclComm_t c = nullptr;
    rc = rccl_status(g_rccl.CommInitRank(&c, world_size, id, rank), "ncclCommInitRank");
    if (rc) return rc;
    g_comm = c; g_world = world_size; g_rank = rank;
    return VSOM_OK;
}

int vsom_comm_allreduce_sum(float* buf, long n, vsom_stream_t stream) {
    VSOM_REQUIRE(buf && n >= 0, VSOM_EINVAL, "comm_allreduce_sum: bad arguments");
    VSOM_REQUIRE(g_comm != nullptr, VSOM_EINVAL, "comm_allreduce_sum: no communicator (vsom_comm_init first)");
    if (n == 0) return VSOM_OK;
    const int rc = rccl_status(g_rccl.AllReduce(buf, buf, (size_t)n, ncclFloat32, ncclSum, g_comm, stream), "ncclAllReduce");
    if (rc == VSOM_OK && g_tape_rec)                  // a recorded step re-issues its collectives too (same order on every rank)
        tape_push([=]() { (void)g_rccl.AllReduce(buf, buf, (size_t)n, ncclFloat32, ncclSum, g_comm, stream); });
    return rc;
}

int vsom_comm_info(int* world_size, int* rank) {
    if (world_size) *world_size = g_comm ? g_world : 0;
    if (rank) *rank = g_comm ? g_rank : -1;
    return VSOM_OK;
}

int vsom_comm_destroy(void) {
    std::lock_guard<std::mutex> lk(g_mu);
    if (!g_comm) return VSOM_OK;
    const int rc = rccl_status(g_rccl.CommDestroy(g_comm), "ncclCommDestroy");
    g_comm = nullptr; g_world = 0; g_rank = -1;
    return rc;
}

}  // extern "C"
