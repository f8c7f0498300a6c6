// comm.cpp -- HOST code: the data-parallel exchange step of the PPO update as C-ABI entry points (SURVEY.md section 8(b):
// rlppo_comm_init / rlppo_allreduce / rlppo_comm_destroy).  The reference has no counterpart (it is single-device); the
// exchange sums the flat [grad_policy | grad_value] arena over the ranks between the last backward of a batch and
// clip_grad_norm_ (ppo_learner.py:187-193 then runs replicated).  RCCL is resolved at run time with dlopen (the process
// normally already holds PyTorch's copy of librccl.so; nothing here links against it), the collective is enqueued in place on
// the CALLER's stream, and no host synchronisation happens: the call sits in the stream like any other launch of this library.
#include <dlfcn.h>
#include <hip/hip_runtime_api.h>
#include <rccl/rccl.h>
#include <stdint.h>
#include <string.h>

#include <string>

#include "../../include/rlppo.h"
#include "common.hpp"

namespace {
struct Rccl {
    void *handle = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId *) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*AllReduce)(const void *, void *, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    const char *(*GetErrorString)(ncclResult_t) = nullptr;
};
Rccl g_rccl;
ncclComm_t g_comm = nullptr;
int g_rank = -1, g_world = 0;
std::string g_path;

int load_rccl() {
    if (g_rccl.handle) return 0;
    const char *names[] = {g_path.empty() ? nullptr : g_path.c_str(), "librccl.so", "librccl.so.1"};
    void *h = nullptr;
    for (const char *nm : names)
        if (nm && (h = dlopen(nm, RTLD_NOW | RTLD_GLOBAL))) break;
    if (!h) {
        rlppo::set_error("comm: cannot load librccl.so (%s)", dlerror());
        return RLPPO_ERR_ARG;
    }
    Rccl r;
    r.handle = h;
    r.GetUniqueId = reinterpret_cast<decltype(r.GetUniqueId)>(dlsym(h, "ncclGetUniqueId"));
    r.CommInitRank = reinterpret_cast<decltype(r.CommInitRank)>(dlsym(h, "ncclCommInitRank"));
    r.AllReduce = reinterpret_cast<decltype(r.AllReduce)>(dlsym(h, "ncclAllReduce"));
    r.CommDestroy = reinterpret_cast<decltype(r.CommDestroy)>(dlsym(h, "ncclCommDestroy"));
    r.GetErrorString = reinterpret_cast<decltype(r.GetErrorString)>(dlsym(h, "ncclGetErrorString"));
    if (!r.GetUniqueId || !r.CommInitRank || !r.AllReduce || !r.CommDestroy || !r.GetErrorString) {
        rlppo::set_error("comm: librccl.so lacks an expected symbol");
        return RLPPO_ERR_ARG;
    }
    g_rccl = r;
    return 0;
}

int nccl_fail(const char *what, ncclResult_t rc) {
    rlppo::set_error("comm: %s failed: %s", what, g_rccl.GetErrorString ? g_rccl.GetErrorString(rc) : "?");
    return 2000 + (int)rc;
}
}  // namespace

extern "C" {

int rlppo_comm_set_library(const char *path) {
    g_path = path ? path : "";
    return 0;
}

int rlppo_comm_unique_id(void *id128) {
    if (!id128) return RLPPO_ERR_ARG;
    if (int rc = load_rccl()) return rc;
    ncclUniqueId id;
    const ncclResult_t r = g_rccl.GetUniqueId(&id);
    if (r != ncclSuccess) return nccl_fail("ncclGetUniqueId", r);
    static_assert(sizeof(id) == RLPPO_COMM_ID_BYTES, "unique id size");
    memcpy(id128, &id, sizeof(id));
    return 0;
}

int rlppo_comm_init(int32_t rank, int32_t world, const void *id128) {
    if (!id128 || world < 1 || rank < 0 || rank >= world) {
        rlppo::set_error("comm_init: rank=%d world=%d", rank, world);
        return RLPPO_ERR_ARG;
    }
    if (g_comm) {
        rlppo::set_error("comm_init: a communicator already exists (one per process)");
        return RLPPO_ERR_ARG;
    }
    if (int rc = load_rccl()) return rc;
    ncclUniqueId id;
    memcpy(&id, id128, sizeof(id));
    const ncclResult_t r = g_rccl.CommInitRank(&g_comm, world, id, rank);  // binds to the calling thread's current device
    if (r != ncclSuccess) {
        g_comm = nullptr;
        return nccl_fail("ncclCommInitRank", r);
    }
    g_rank = rank;
    g_world = world;
    return 0;
}

int rlppo_allreduce(void *stream, void *buf, int64_t n, int32_t is_f64) {
    if (!g_comm) {
        rlppo::set_error("allreduce: no communicator (rlppo_comm_init first)");
        return RLPPO_ERR_ARG;
    }
    if (n < 0 || (n > 0 && !buf)) return RLPPO_ERR_ARG;
    if (n == 0) return 0;
    const ncclResult_t r = g_rccl.AllReduce(buf, buf, (size_t)n, is_f64 ? ncclFloat64 : ncclFloat32, ncclSum, g_comm, (hipStream_t)stream);
    if (r != ncclSuccess) return nccl_fail("ncclAllReduce", r);
    return 0;
}

int rlppo_comm_destroy(void) {
    if (!g_comm) return 0;
    const ncclResult_t r = g_rccl.CommDestroy(g_comm);
    g_comm = nullptr;
    g_rank = -1;
    g_world = 0;
    if (r != ncclSuccess) return nccl_fail("ncclCommDestroy", r);
    return 0;
}
}
