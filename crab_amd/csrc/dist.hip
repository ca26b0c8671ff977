// crab_gather_results: the ONE collective of the path behind the C-ABI (SURVEY.md 8e, 8b export list) - per-clip sharding needs nothing but a
// gather of fixed-size result records {clip id, ids[n_new], optional first-step logits} to a root rank.  RCCL over xGMI: every peer has its own
// direct link to the root (~153 GB/s), the payload is KBs to tens of MB, so a plain ncclGather is the right shape (no ring, no all-reduce).
//
// RCCL is loaded LAZILY (dlopen at the first crab_dist_* call): libcrab_hip.so keeps no link-time dependency on it, a single-GPU caller never
// touches it, and a process that already holds a librccl (PyTorch ships one) shares that copy through the loader.  The reference has no
// collective on its inference path (scripts/finetune/inference_hyper_lora.py:1466-1479 loops on one device); crab_amd/parallel.py is the
// torch.distributed form of the same gather, this file is the one a C-only caller (examples/*.c) uses.
#include "crab_internal.h"
#include <dlfcn.h>
#include <stdlib.h>

namespace {

typedef int ncclResult_t;                        // ncclSuccess == 0 (rccl.h)
typedef struct ncclComm* ncclComm_t;
typedef struct { char internal[CRAB_DIST_ID_BYTES]; } ncclUniqueId;
enum { NCCL_UINT8 = 1 };                         // ncclDataType_t: ncclUint8 (rccl.h)

struct Rccl {
    void* h = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*Gather)(const void*, void*, size_t, int, int, ncclComm_t, hipStream_t) = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
    char why[256] = {0};
};

Rccl* rccl() {
    // function-local static with an initialiser: C++11 guarantees one thread runs it, the others wait (two contexts on two host threads may
    // both reach their first crab_dist_* call at once)
    static Rccl* inst = []() {
        static Rccl r;
        const char* names[] = {getenv("CRAB_RCCL_LIB"), "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
        for (const char* n : names) {
            if (!n || !*n) continue;
            r.h = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
            if (r.h) break;
        }
        if (!r.h) { snprintf(r.why, sizeof(r.why), "librccl.so not found (%s); set CRAB_RCCL_LIB", dlerror()); return &r; }
        r.GetUniqueId = (decltype(r.GetUniqueId))dlsym(r.h, "ncclGetUniqueId");
        r.CommInitRank = (decltype(r.CommInitRank))dlsym(r.h, "ncclCommInitRank");
        r.CommDestroy = (decltype(r.CommDestroy))dlsym(r.h, "ncclCommDestroy");
        r.Gather = (decltype(r.Gather))dlsym(r.h, "ncclGather");
        r.GetErrorString = (decltype(r.GetErrorString))dlsym(r.h, "ncclGetErrorString");
        if (!r.GetUniqueId || !r.CommInitRank || !r.CommDestroy || !r.Gather) {
            snprintf(r.why, sizeof(r.why), "librccl.so lacks ncclGetUniqueId / ncclCommInitRank / ncclCommDestroy / ncclGather");
            r.h = nullptr;
        }
        return &r;
    }();
    return inst;
}

int rccl_fail(crab_ctx* ctx, Rccl* r, const char* what, ncclResult_t rc) {
    char msg[384];
    snprintf(msg, sizeof(msg), "%s: RCCL error %d (%s)", what, rc, r->GetErrorString ? r->GetErrorString(rc) : "?");
    return crab_fail(ctx, CRAB_E_HIP, msg);
}

}  // namespace

struct crab_comm {
    ncclComm_t comm;
    int world, rank, device;
};

extern "C" {

int crab_dist_unique_id(crab_ctx* ctx, void* id_out) {
    if (!ctx) return CRAB_E_INVALID;
    if (!id_out) return crab_fail(ctx, CRAB_E_INVALID, "dist_unique_id: null output");
    Rccl* r = rccl();
    if (!r->h) return crab_fail(ctx, CRAB_E_UNSUPPORTED, r->why);
    ncclUniqueId id;
    ncclResult_t rc = r->GetUniqueId(&id);
    if (rc) return rccl_fail(ctx, r, "ncclGetUniqueId", rc);
    memcpy(id_out, &id, CRAB_DIST_ID_BYTES);
    return CRAB_OK;
}

int crab_dist_init(crab_ctx* ctx, const void* id, int world, int rank, crab_comm** out) {
    if (!ctx) return CRAB_E_INVALID;
    if (!id || !out || world < 1 || rank < 0 || rank >= world) return crab_fail(ctx, CRAB_E_INVALID, "dist_init: id, out, 0 <= rank < world");
    *out = nullptr;
    Rccl* r = rccl();
    if (!r->h) return crab_fail(ctx, CRAB_E_UNSUPPORTED, r->why);
    CRAB_HIP_TRY(ctx, hipSetDevice(ctx->device));                  // one process per GPU: the communicator lives on the context's device
    ncclUniqueId uid;
    memcpy(&uid, id, CRAB_DIST_ID_BYTES);
    ncclComm_t c = nullptr;
    ncclResult_t rc = r->CommInitRank(&c, world, uid, rank);
    if (rc) return rccl_fail(ctx, r, "ncclCommInitRank", rc);
    crab_comm* cc = (crab_comm*)calloc(1, sizeof(crab_comm));
    if (!cc) { r->CommDestroy(c); return crab_fail(ctx, CRAB_E_INVALID, "dist_init: out of host memory"); }
    cc->comm = c; cc->world = world; cc->rank = rank; cc->device = ctx->device;
    *out = cc;
    return CRAB_OK;
}

int crab_gather_results(crab_ctx* ctx, void* stream, crab_comm* comm, const void* send, int64_t bytes_per_rank, void* recv, int root) {
    if (!ctx) return CRAB_E_INVALID;
    if (!comm || !send || bytes_per_rank <= 0 || root < 0 || root >= comm->world || (comm->rank == root && !recv))
        return crab_fail(ctx, CRAB_E_INVALID, "gather_results: comm, send, positive bytes_per_rank, 0 <= root < world, recv on the root");
    Rccl* r = rccl();
    if (!r->h) return crab_fail(ctx, CRAB_E_UNSUPPORTED, r->why);
    ncclResult_t rc = r->Gather(send, recv, (size_t)bytes_per_rank, NCCL_UINT8, root, comm->comm, (hipStream_t)stream);
    if (rc) return rccl_fail(ctx, r, "ncclGather", rc);
    return CRAB_OK;
}

int crab_dist_world(const crab_comm* comm) { return comm ? comm->world : 0; }
int crab_dist_rank(const crab_comm* comm) { return comm ? comm->rank : -1; }

void crab_dist_destroy(crab_comm* comm) {
    if (!comm) return;
    Rccl* r = rccl();
    if (r->h && comm->comm) r->CommDestroy(comm->comm);
    free(comm);
}

}  // extern "C"
