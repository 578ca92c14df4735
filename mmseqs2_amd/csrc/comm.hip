// Communicator of a multi-GPU run, owned by the library: RCCL (xGMI inside a node) reached through dlopen, so that a
// single-GPU caller never loads it.  One communicator spans the contexts that hold the shards of ONE target database
// (the reference's analogue is the MPI world of Prefiltering::runMpiSplits, Prefiltering.cpp:605-689):
//   * one process per GPU (bench.py under torch.distributed.run, an MPI build of mmseqs): mmgpu_comm_unique_id on one rank,
//     the 128 bytes travel by the host's own means, mmgpu_comm_init_rank on every rank;
//   * one process, several devices (the patched `mmseqs` binary): mmgpu_init_multi (multi_api.hip) -> ncclCommInitAll.
// The collectives themselves are enqueued on the context's stream by the exchange functions (pf_api.hip, mmgpu_api.hip):
// no host synchronisation anywhere on the data path.
#include <dlfcn.h>
#include <cstring>

#include "mmgpu_internal.h"

namespace mmgpu {

namespace {

struct NcclId { char internal[MMGPU_COMM_ID_BYTES]; };
static_assert(sizeof(NcclId) == 128, "ncclUniqueId is 128 bytes (rccl.h: NCCL_UNIQUE_ID_BYTES)");

struct RcclApi {
    void *lib = nullptr;
    int (*GetUniqueId)(NcclId *) = nullptr;
    int (*CommInitRank)(void **, int, NcclId, int) = nullptr;
    int (*CommInitAll)(void **, int, const int *) = nullptr;
    int (*CommDestroy)(void *) = nullptr;
    int (*AllGather)(const void *, void *, size_t, int, void *, hipStream_t) = nullptr;
    int (*GroupStart)() = nullptr;
    int (*GroupEnd)() = nullptr;
    const char *(*GetErrorString)(int) = nullptr;
    std::string error;
};

RcclApi *rccl() {
    static RcclApi api;
    static bool tried = false;
    if (tried) return &api;
    tried = true;
    const char *names[] = {getenv("MMGPU_RCCL_LIB"), "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    for (const char *n : names) {
        if (!n || !*n) continue;
        api.lib = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
        if (api.lib) break;
        api.error = dlerror();
    }
    if (!api.lib) return &api;
#define MMGPU_SYM(field, name)                                                     \
    api.field = reinterpret_cast<decltype(api.field)>(dlsym(api.lib, name));       \
    if (!api.field) { api.error = std::string("librccl lacks ") + name; api.lib = nullptr; return &api; }
    MMGPU_SYM(GetUniqueId, "ncclGetUniqueId")
    MMGPU_SYM(CommInitRank, "ncclCommInitRank")
    MMGPU_SYM(CommInitAll, "ncclCommInitAll")
    MMGPU_SYM(CommDestroy, "ncclCommDestroy")
    MMGPU_SYM(AllGather, "ncclAllGather")
    MMGPU_SYM(GroupStart, "ncclGroupStart")
    MMGPU_SYM(GroupEnd, "ncclGroupEnd")
    MMGPU_SYM(GetErrorString, "ncclGetErrorString")
#undef MMGPU_SYM
    return &api;
}

int nccl_fail(const char *what, int rc) {
    RcclApi *r = rccl();
    return fail(MMGPU_ERR_HIP, std::string(what) + ": " + (r->GetErrorString ? r->GetErrorString(rc) : "RCCL error"));
}

}  // namespace

int comm_require_rccl() {
    RcclApi *r = rccl();
    if (!r->lib) return fail(MMGPU_ERR_STATE, "RCCL is not available (dlopen librccl.so.1: " + r->error + ")");
    return MMGPU_OK;
}

int comm_group_start() {
    if (int e = comm_require_rccl()) return e;
    const int rc = rccl()->GroupStart();
    return rc ? nccl_fail("ncclGroupStart", rc) : MMGPU_OK;
}

int comm_group_end() {
    const int rc = rccl()->GroupEnd();
    return rc ? nccl_fail("ncclGroupEnd", rc) : MMGPU_OK;
}

int comm_init_all(mmgpu_ctx **ctxs, int n) {
    if (int e = comm_require_rccl()) return e;
    std::vector<int> dev(n);
    std::vector<void *> comms(n, nullptr);
    for (int i = 0; i < n; i++) dev[i] = ctxs[i]->device;
    const int rc = rccl()->CommInitAll(comms.data(), n, dev.data());
    if (rc) return nccl_fail("ncclCommInitAll", rc);
    for (int i = 0; i < n; i++) {
        Comm *c = new Comm();
        c->nccl = comms[i];
        c->rank = i;
        c->n_ranks = n;
        c->transport = "rccl";
        ctxs[i]->comm = c;
    }
    return MMGPU_OK;
}

// `bytes` from every rank, rank r's block at recv + r * bytes; on the context's stream.  One rank: a device copy.
int comm_allgather(mmgpu_ctx *c, const void *send, void *recv, size_t bytes) {
    if (bytes == 0) return MMGPU_OK;
    Comm *m = c->comm;
    if (!m || m->n_ranks == 1) {
        if (m && m->nccl && getenv("MMGPU_COMM_SELF_RCCL")) {   // one-rank self-test of the RCCL transport
            const int rc = rccl()->AllGather(send, recv, bytes, /*ncclUint8*/ 1, m->nccl, c->stream);
            return rc ? nccl_fail("ncclAllGather", rc) : MMGPU_OK;
        }
        if (send != recv) HIP_TRY(hipMemcpyAsync(recv, send, bytes, hipMemcpyDeviceToDevice, c->stream));
        return MMGPU_OK;
    }
    if (!m->nccl) return fail(MMGPU_ERR_STATE, "comm_allgather: this communicator's transport is driven by mmgpu_multi_* calls only");
    const int rc = rccl()->AllGather(send, recv, bytes, /*ncclUint8*/ 1, m->nccl, c->stream);
    return rc ? nccl_fail("ncclAllGather", rc) : MMGPU_OK;
}

void comm_free(mmgpu_ctx *c) {
    if (!c || !c->comm) return;
    if (c->comm->nccl && rccl()->lib) (void)rccl()->CommDestroy(c->comm->nccl);
    delete c->comm;
    c->comm = nullptr;
}

}  // namespace mmgpu

using mmgpu::fail;

extern "C" int mmgpu_comm_unique_id(uint8_t *id) {
    if (!id) return fail(MMGPU_ERR_ARG, "mmgpu_comm_unique_id: NULL argument");
    if (int e = mmgpu::comm_require_rccl()) return e;
    mmgpu::NcclId nid;
    const int rc = mmgpu::rccl()->GetUniqueId(&nid);
    if (rc) return mmgpu::nccl_fail("ncclGetUniqueId", rc);
    memcpy(id, nid.internal, MMGPU_COMM_ID_BYTES);
    return MMGPU_OK;
}

extern "C" int mmgpu_comm_init_rank(mmgpu_ctx *c, const uint8_t *id, int rank, int n_ranks) {
    if (!c || !id) return fail(MMGPU_ERR_ARG, "mmgpu_comm_init_rank: NULL argument");
    if (n_ranks < 1 || n_ranks > 64 || rank < 0 || rank >= n_ranks) return fail(MMGPU_ERR_ARG, "mmgpu_comm_init_rank: rank / n_ranks out of range (1..64 ranks)");
    if (c->comm) return fail(MMGPU_ERR_STATE, "mmgpu_comm_init_rank: the context already has a communicator");
    if (int e = mmgpu::comm_require_rccl()) return e;
    HIP_TRY(hipSetDevice(c->device));
    mmgpu::NcclId nid;
    memcpy(nid.internal, id, MMGPU_COMM_ID_BYTES);
    void *comm = nullptr;
    const int rc = mmgpu::rccl()->CommInitRank(&comm, n_ranks, nid, rank);
    if (rc) return mmgpu::nccl_fail("ncclCommInitRank", rc);
    mmgpu::Comm *m = new mmgpu::Comm();
    m->nccl = comm;
    m->rank = rank;
    m->n_ranks = n_ranks;
    m->transport = "rccl";
    c->comm = m;
    return MMGPU_OK;
}

extern "C" int mmgpu_comm_info(mmgpu_ctx *c, int *rank, int *n_ranks, char *transport, int cap) {
    if (!c) return fail(MMGPU_ERR_ARG, "mmgpu_comm_info: NULL context");
    if (rank) *rank = c->comm ? c->comm->rank : 0;
    if (n_ranks) *n_ranks = c->comm ? c->comm->n_ranks : 1;
    if (transport && cap > 0) {
        strncpy(transport, c->comm ? c->comm->transport.c_str() : "none", (size_t)cap - 1);
        transport[cap - 1] = 0;
    }
    return MMGPU_OK;
}

extern "C" void mmgpu_comm_destroy(mmgpu_ctx *c) {
    if (!c) return;
    (void)hipSetDevice(c->device);
    mmgpu::comm_free(c);
}
