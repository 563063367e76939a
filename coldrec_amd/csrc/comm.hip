// RCCL entry points of the C ABI: the two exchange steps of the multi-GPU path for a consumer of
// libcoldrec_hip.so that is not a PyTorch process -- all-gather of the packed per-shard top-k lists (eval) and all-reduce
// of fp32 buffers (the 4 batch sums and the dense gradient of the data-parallel step).  The reference is single-process
// (no call site to cite); coldrec_amd's own host layer uses torch.distributed (backend "nccl" = RCCL) for the same two
// collectives, these wrappers exist so that the .so alone is a complete multi-GPU boundary.
//
// librccl is NOT a link-time dependency: it is dlopen'ed on the first crh_comm_* call (an already loaded copy -- e.g.
// PyTorch's -- is reused), so single-GPU users never map it.
#include <dlfcn.h>
#include <rccl/rccl.h>
#include <string.h>

#include <mutex>
#include <string>
#include <new>

#include "crh_common.h"

namespace {

struct Rccl {
    void* handle = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*AllGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
    bool ok = false;
};

Rccl g_rccl;
std::once_flag g_once;
std::string g_rccl_err;      // dlerror() of the last failed dlopen, captured once (a second dlerror() call returns NULL)

void load_rccl() {
    static const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    void* h = nullptr;
    for (const char* n : names)                      // a copy that is already mapped (PyTorch's) wins
        if ((h = dlopen(n, RTLD_NOW | RTLD_NOLOAD))) break;
    if (!h)
        for (const char* n : names) {
            if ((h = dlopen(n, RTLD_NOW | RTLD_LOCAL))) break;
            const char* e = dlerror();
            if (e) g_rccl_err = e;
        }
    if (!h) return;
    g_rccl.handle = h;
#define CRH_SYM(field, sym) g_rccl.field = reinterpret_cast<decltype(g_rccl.field)>(dlsym(h, sym))
    CRH_SYM(GetUniqueId, "ncclGetUniqueId");
    CRH_SYM(CommInitRank, "ncclCommInitRank");
    CRH_SYM(CommDestroy, "ncclCommDestroy");
    CRH_SYM(AllGather, "ncclAllGather");
    CRH_SYM(AllReduce, "ncclAllReduce");
    CRH_SYM(GetErrorString, "ncclGetErrorString");
#undef CRH_SYM
    g_rccl.ok = g_rccl.GetUniqueId && g_rccl.CommInitRank && g_rccl.CommDestroy && g_rccl.AllGather && g_rccl.AllReduce;
}

bool need_rccl(const char* who) {
    std::call_once(g_once, load_rccl);
    if (!g_rccl.ok) crh_set_error("%s: librccl.so could not be loaded (%s)", who, g_rccl_err.empty() ? "symbols missing" : g_rccl_err.c_str());
    return g_rccl.ok;
}

#define CRH_NCCL(call, who)                                                                          \
    do {                                                                                             \
        ncclResult_t r_ = (call);                                                                    \
        if (r_ != ncclSuccess) {                                                                     \
            crh_set_error("%s: %s failed: %s", who, #call, g_rccl.GetErrorString ? g_rccl.GetErrorString(r_) : "?"); \
            return CRH_ERR_HIP;                                                                      \
        }                                                                                            \
    } while (0)

}  // namespace

struct crh_comm {
    ncclComm_t comm;
    int rank, world;
};

extern "C" int crh_comm_unique_id(void* id128_host) {
    CRH_CHECK_ARG(id128_host, "crh_comm_unique_id: NULL buffer");
    if (!need_rccl("crh_comm_unique_id")) return CRH_ERR_HIP;
    ncclUniqueId id;
    CRH_NCCL(g_rccl.GetUniqueId(&id), "crh_comm_unique_id");
    memcpy(id128_host, &id, sizeof(id));
    return CRH_OK;
}

extern "C" crh_comm* crh_comm_init(int rank, int world, const void* id128_host) {
    if (rank < 0 || world < 1 || rank >= world || !id128_host) {
        crh_set_error("crh_comm_init: bad arguments (rank %d of %d)", rank, world);
        return nullptr;
    }
    if (!need_rccl("crh_comm_init")) return nullptr;
    crh_comm* c = new (std::nothrow) crh_comm();
    if (!c) return nullptr;
    ncclUniqueId id;
    memcpy(&id, id128_host, sizeof(id));
    const ncclResult_t r = g_rccl.CommInitRank(&c->comm, world, id, rank);
    if (r != ncclSuccess) {
        crh_set_error("crh_comm_init: ncclCommInitRank failed: %s", g_rccl.GetErrorString ? g_rccl.GetErrorString(r) : "?");
        delete c;
        return nullptr;
    }
    c->rank = rank;
    c->world = world;
    return c;
}

extern "C" int crh_comm_destroy(crh_comm* c) {
    if (!c) return CRH_OK;
    const ncclResult_t r = g_rccl.CommDestroy(c->comm);
    delete c;
    if (r != ncclSuccess) {
        crh_set_error("crh_comm_destroy: ncclCommDestroy failed");
        return CRH_ERR_HIP;
    }
    return CRH_OK;
}

extern "C" int crh_comm_rank(const crh_comm* c) { return c ? c->rank : -1; }
extern "C" int crh_comm_world(const crh_comm* c) { return c ? c->world : -1; }

// In-place sum over the ranks of n fp32 values (the 4 batch sums; the (U+I) x d gradient table), on `stream`.
extern "C" int crh_comm_allreduce_f32(crh_comm* c, float* buf, int64_t n, void* stream) {
    CRH_CHECK_ARG(c && buf && n > 0, "crh_comm_allreduce_f32: bad arguments");
    CRH_NCCL(g_rccl.AllReduce(buf, buf, (size_t)n, ncclFloat32, ncclSum, c->comm, reinterpret_cast<hipStream_t>(stream)),
             "crh_comm_allreduce_f32");
    return CRH_OK;
}

// Exchange step of the item-row-sharded evaluation: every rank contributes its shard's (n_users, k) scores and global
// ids; gathered_* are laid out [rank][user][k] -- exactly crh_merge_topk's input with n_lists = world.
// ONE collective: the rank's lists are packed into (n_users, 2k) 32-bit words (k score bit patterns, then k ids, per
// user -- the layout coldrec_amd/eval.py sends through torch.distributed), one ncclAllGather of int32 moves them, and the
// gathered words are split back into the two arrays.  xGMI is point-to-point: a second collective would pay the
// launch + ring latency twice for a payload (8 k bytes per user) that is latency-bound to begin with.
namespace {

__global__ void pack_topk_kernel(const float* __restrict__ score, const int32_t* __restrict__ idx, int64_t n, int k,
                                 int32_t* __restrict__ packed) {
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;          // one (user, slot) word pair
    if (t >= n * k) return;
    const int64_t u = t / k;
    const int j = (int)(t - u * k);
    packed[u * 2 * k + j] = __float_as_int(score[t]);
    packed[u * 2 * k + k + j] = idx[t];
}

__global__ void unpack_topk_kernel(const int32_t* __restrict__ gathered, int64_t rows, int k, float* __restrict__ score,
                                   int32_t* __restrict__ idx) {
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;          // rows = world * n_users
    if (t >= rows * k) return;
    const int64_t r = t / k;
    const int j = (int)(t - r * k);
    score[t] = __int_as_float(gathered[r * 2 * k + j]);
    idx[t] = gathered[r * 2 * k + k + j];
}

}  // namespace

// Exchange step of the data-parallel touched-rows training step (crh_bpr_bwd_owned_f32 -> crh_rows_pack_f32 -> HERE ->
// crh_rows_unpack_f32): every rank contributes its `cap` slots of (row id, d floats); gathered_* are laid out [rank][slot],
// exactly what crh_rows_unpack_f32 takes with n = world * cap.  Two collectives on the caller's stream (ids: 4 bytes per
// slot, rows: 4 d bytes per slot; no packing pass over the 13 MB of rows).
extern "C" int crh_comm_allgather_rows(crh_comm* c, const int32_t* ids, const float* rows, int64_t cap, int d,
                                       int32_t* gathered_ids, float* gathered_rows, void* stream) {
    CRH_CHECK_ARG(c && ids && rows && gathered_ids && gathered_rows && cap > 0 && d > 0, "crh_comm_allgather_rows: bad arguments");
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    CRH_NCCL(g_rccl.AllGather(ids, gathered_ids, (size_t)cap, ncclInt32, c->comm, st), "crh_comm_allgather_rows (ids)");
    CRH_NCCL(g_rccl.AllGather(rows, gathered_rows, (size_t)cap * (size_t)d, ncclFloat32, c->comm, st), "crh_comm_allgather_rows (rows)");
    return CRH_OK;
}

extern "C" size_t crh_comm_allgather_topk_workspace_bytes(int world, int64_t n_users, int k) {
    if (world < 1 || n_users < 0 || k < 1) return 0;
    return (size_t)(world + 1) * (size_t)n_users * 2 * (size_t)k * sizeof(int32_t);
}

extern "C" int crh_comm_allgather_topk(crh_comm* c, const float* score, const int32_t* idx, int64_t n_users, int k,
                                       float* gathered_score, int32_t* gathered_idx, void* workspace,
                                       size_t workspace_bytes, void* stream) {
    CRH_CHECK_ARG(c && score && idx && gathered_score && gathered_idx && n_users > 0 && k >= 1 && k <= CRH_MAX_K,
                  "crh_comm_allgather_topk: bad arguments");
    CRH_CHECK_ARG(workspace && workspace_bytes >= crh_comm_allgather_topk_workspace_bytes(c->world, n_users, k),
                  "crh_comm_allgather_topk: workspace smaller than crh_comm_allgather_topk_workspace_bytes()");
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    const int64_t n = n_users * k;
    int32_t* mine = static_cast<int32_t*>(workspace);                       // (n_users, 2k)
    int32_t* all = mine + 2 * n;                                            // (world, n_users, 2k)
    hipLaunchKernelGGL(pack_topk_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, score, idx, n_users, k, mine);
    CRH_HIP(hipGetLastError());
    CRH_NCCL(g_rccl.AllGather(mine, all, (size_t)(2 * n), ncclInt32, c->comm, st), "crh_comm_allgather_topk");
    const int64_t rows = n_users * c->world;
    hipLaunchKernelGGL(unpack_topk_kernel, dim3((unsigned)((rows * k + 255) / 256)), dim3(256), 0, st, all, rows, k,
                       gathered_score, gathered_idx);
    CRH_HIP(hipGetLastError());
    return CRH_OK;
}
