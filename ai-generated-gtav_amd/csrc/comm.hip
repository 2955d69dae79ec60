// RCCL collectives behind the C-ABI (SURVEY.md 8(b) `gtav_comm_{init,allgather,allreduce,destroy}`, 8(e)): a host that is not
// Python runs the multi-GPU path — one all-gather of the final latents per clip, one all-reduce of the gradient arena per training
// step — without torch.distributed.  RCCL is opened at run time (dlopen), preferring a copy that is already loaded into the process
// (PyTorch ships its own librccl.so: two copies in one process must not both be bound), so libgtav_amd.so has no link-time
// dependency on it and single-GPU users never load it.  One communicator per process / device, like accelerate's one process per GPU.
#include "../../include/gtav_amd.h"
#include "common.h"

#include <dlfcn.h>
#include <cstring>
#include <rccl/rccl.h>

using namespace gtav;

namespace {
struct Rccl {
    void* lib = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*AllGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
};
Rccl g_rccl;

int rccl_load() {
    if (g_rccl.lib) return 0;
    const char* names[] = {"librccl.so", "librccl.so.1"};
    void* lib = nullptr;
    for (const char* n : names)
        if (!lib) lib = dlopen(n, RTLD_NOW | RTLD_NOLOAD);          // a copy the process already uses (e.g. torch's)
    for (const char* n : names)
        if (!lib) lib = dlopen(n, RTLD_NOW | RTLD_LOCAL);
    if (!lib) lib = dlopen("/opt/rocm/lib/librccl.so", RTLD_NOW | RTLD_LOCAL);
    GTAV_REQUIRE(lib, "gtav_comm: librccl.so not found (%s)", dlerror());
#define SYM(field, name)                                                     \
    g_rccl.field = (decltype(g_rccl.field))dlsym(lib, name);                 \
    GTAV_REQUIRE(g_rccl.field, "gtav_comm: librccl.so has no symbol %s", name)
    SYM(GetUniqueId, "ncclGetUniqueId");
    SYM(CommInitRank, "ncclCommInitRank");
    SYM(CommDestroy, "ncclCommDestroy");
    SYM(AllReduce, "ncclAllReduce");
    SYM(AllGather, "ncclAllGather");
    SYM(GetErrorString, "ncclGetErrorString");
#undef SYM
    g_rccl.lib = lib;
    return 0;
}
#define GTAV_CHECK_NCCL(expr)                                                                                        \
    do {                                                                                                             \
        ncclResult_t r_ = (expr);                                                                                    \
        if (r_ != ncclSuccess) {                                                                                     \
            set_error("%s failed: %s", #expr, g_rccl.GetErrorString ? g_rccl.GetErrorString(r_) : "?");              \
            return 1;                                                                                                \
        }                                                                                                            \
    } while (0)
}  // namespace

struct gtav_comm {
    ncclComm_t comm = nullptr;
    int nranks = 1, rank = 0;
};

extern "C" {

int gtav_comm_unique_id(void* id128) {
    GTAV_REQUIRE(id128, "comm_unique_id: null argument");
    static_assert(sizeof(ncclUniqueId) == 128, "ncclUniqueId is 128 bytes");
    if (int rc = rccl_load()) return rc;
    GTAV_CHECK_NCCL(g_rccl.GetUniqueId((ncclUniqueId*)id128));
    return 0;
}

int gtav_comm_init(gtav_comm** out, int32_t nranks, int32_t rank, const void* id128) {
    GTAV_REQUIRE(out && id128 && nranks >= 1 && rank >= 0 && rank < nranks, "comm_init: bad argument (nranks %d, rank %d)", nranks, rank);
    if (int rc = rccl_load()) return rc;
    ncclUniqueId id;
    memcpy(&id, id128, sizeof(id));
    gtav_comm* c = new gtav_comm();
    c->nranks = nranks;
    c->rank = rank;
    ncclResult_t r = g_rccl.CommInitRank(&c->comm, nranks, id, rank);   // uses the calling thread's current HIP device
    if (r != ncclSuccess) {
        set_error("ncclCommInitRank failed: %s", g_rccl.GetErrorString(r));
        delete c;
        return 1;
    }
    *out = c;
    return 0;
}

int gtav_comm_allreduce_f32(gtav_comm* c, float* buf_dev, int64_t count, int32_t average, void* stream) {
    GTAV_REQUIRE(c && buf_dev && count >= 0, "comm_allreduce: bad argument");
    GTAV_CHECK_NCCL(g_rccl.AllReduce(buf_dev, buf_dev, (size_t)count, ncclFloat32, average ? ncclAvg : ncclSum, c->comm, (hipStream_t)stream));
    return 0;
}

int gtav_comm_allgather(gtav_comm* c, const void* send_dev, void* recv_dev, int64_t bytes_per_rank, void* stream) {
    GTAV_REQUIRE(c && send_dev && recv_dev && bytes_per_rank >= 0, "comm_allgather: bad argument");
    GTAV_CHECK_NCCL(g_rccl.AllGather(send_dev, recv_dev, (size_t)bytes_per_rank, ncclInt8, c->comm, (hipStream_t)stream));
    return 0;
}

int gtav_comm_destroy(gtav_comm* c) {
    if (!c) return 0;
    if (c->comm && g_rccl.CommDestroy) (void)g_rccl.CommDestroy(c->comm);
    delete c;
    return 0;
}

}  // extern "C"
