// Native transport for the multi-GPU path: fluid_comm_t over RCCL (xGMI inside a node).
//
// RCCL is reached through dlopen of the librccl.so the caller names — in bench.py the one inside
// the running PyTorch, so the process holds exactly one RCCL — and every call is enqueued on the
// solver's own HIP stream: no host synchronisation and no Python inside the PCG loop.
// Message sizes here: block faces of a few cells' width per neighbour (256^3 in 2 x 2 x 2: 128^2 x 4 cells x 8 B =
// 0.5 MB, edges and corners a few KB), the gathered coarse level of the V-cycle (<= 0.6 MB) and 1-2 double
// all-reduces per PCG iteration — latency-bound, not bandwidth-bound.
#include <dlfcn.h>
#include <cstdio>
#include <cstring>
#include <string>

#include "../../include/fluid_hip.h"

namespace {

typedef void* ncclComm_t;
struct ncclUniqueId { char internal[128]; };
typedef int ncclResult_t;
enum { ncclUint8 = 1, ncclInt32 = 2, ncclInt64 = 4, ncclFloat32 = 7, ncclFloat64 = 8 };
enum { ncclSum = 0, ncclMax = 2, ncclMin = 3 };

struct Api {
    void* h = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*AllReduce)(const void*, void*, size_t, int, int, ncclComm_t, void*) = nullptr;
    ncclResult_t (*Send)(const void*, size_t, int, int, ncclComm_t, void*) = nullptr;
    ncclResult_t (*Recv)(void*, size_t, int, int, ncclComm_t, void*) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
};

thread_local std::string g_rccl_err;

bool load_api(const char* path, Api& a)
{
    a.h = dlopen(path && *path ? path : "librccl.so", RTLD_NOW | RTLD_GLOBAL);
    if (!a.h) { g_rccl_err = std::string("dlopen librccl: ") + dlerror(); return false; }
#define SYM(field, name)                                                             \
    a.field = (decltype(a.field))dlsym(a.h, name);                                   \
    if (!a.field) { g_rccl_err = std::string("dlsym ") + name + " failed"; return false; }
    SYM(GetUniqueId, "ncclGetUniqueId")
    SYM(CommInitRank, "ncclCommInitRank")
    SYM(CommDestroy, "ncclCommDestroy")
    SYM(AllReduce, "ncclAllReduce")
    SYM(Send, "ncclSend")
    SYM(Recv, "ncclRecv")
    SYM(GroupStart, "ncclGroupStart")
    SYM(GroupEnd, "ncclGroupEnd")
    SYM(GetErrorString, "ncclGetErrorString")
#undef SYM
    return true;
}

struct Ctx {
    Api api;
    ncclComm_t comm = nullptr;
    int rank = 0, size = 1;
};

// One grouped launch for all the neighbours of a halo exchange (<= 26 peers): receives posted first, then sends.
int cb_exchange(void* vctx, int32_t n, const int32_t* peer, const void* const* sbuf, const size_t* sbytes, void* const* rbuf,
                const size_t* rbytes, void* stream)
{
    Ctx* c = (Ctx*)vctx;
    bool any = false;
    for (int i = 0; i < n; ++i) any = any || sbytes[i] || rbytes[i];
    if (!any) return 0;
    ncclResult_t r = c->api.GroupStart();
    for (int i = 0; i < n && !r; ++i)
        if (rbytes[i]) r = c->api.Recv(rbuf[i], rbytes[i], ncclUint8, peer[i], c->comm, stream);
    for (int i = 0; i < n && !r; ++i)
        if (sbytes[i]) r = c->api.Send(sbuf[i], sbytes[i], ncclUint8, peer[i], c->comm, stream);
    ncclResult_t e = c->api.GroupEnd();
    if (!r) r = e;
    if (r) { g_rccl_err = std::string("rccl exchange: ") + c->api.GetErrorString(r); fprintf(stderr, "%s\n", g_rccl_err.c_str()); }
    return r;
}

int cb_allreduce(void* vctx, void* buf, int64_t count, int32_t dtype, int32_t op, void* stream)
{
    Ctx* c = (Ctx*)vctx;
    const int dt = dtype == FLUID_DT_F64 ? ncclFloat64 : (dtype == FLUID_DT_I32 ? ncclInt32 : (dtype == FLUID_DT_F32 ? ncclFloat32 : (dtype == FLUID_DT_U8 ? ncclUint8 : ncclInt64)));
    const int ro = op == FLUID_OP_SUM ? ncclSum : (op == FLUID_OP_MAX ? ncclMax : ncclMin);
    ncclResult_t r = c->api.AllReduce(buf, buf, (size_t)count, dt, ro, c->comm, stream);
    if (r) { g_rccl_err = std::string("rccl allreduce: ") + c->api.GetErrorString(r); fprintf(stderr, "%s\n", g_rccl_err.c_str()); }
    return r;
}

}  // namespace

extern "C" {

const char* fluid_rccl_last_error(void) { return g_rccl_err.c_str(); }

int fluid_rccl_unique_id(const char* librccl_path, void* id128)
{
    Api a;
    if (!id128 || !load_api(librccl_path, a)) return FLUID_ERR_HIP;
    ncclUniqueId id;
    ncclResult_t r = a.GetUniqueId(&id);
    if (r) { g_rccl_err = std::string("ncclGetUniqueId: ") + a.GetErrorString(r); return FLUID_ERR_HIP; }
    memcpy(id128, &id, sizeof(id));
    return FLUID_OK;
}

int fluid_rccl_comm_create(const char* librccl_path, const void* id128, int32_t rank, int32_t size, fluid_comm_t* out)
{
    if (!id128 || !out || size < 1 || rank < 0 || rank >= size) { g_rccl_err = "bad argument"; return FLUID_ERR_ARG; }
    Ctx* c = new Ctx();
    if (!load_api(librccl_path, c->api)) { delete c; return FLUID_ERR_HIP; }
    ncclUniqueId id;
    memcpy(&id, id128, sizeof(id));
    ncclResult_t r = c->api.CommInitRank(&c->comm, size, id, rank);  // binds to the CURRENT HIP device
    if (r) { g_rccl_err = std::string("ncclCommInitRank: ") + c->api.GetErrorString(r); delete c; return FLUID_ERR_HIP; }
    c->rank = rank;
    c->size = size;
    out->rank = rank;
    out->size = size;
    out->ctx = c;
    out->exchange = cb_exchange;
    out->allreduce = cb_allreduce;
    return FLUID_OK;
}

int fluid_rccl_comm_destroy(fluid_comm_t* cm)
{
    if (!cm || !cm->ctx) return FLUID_OK;
    Ctx* c = (Ctx*)cm->ctx;
    if (c->comm) c->api.CommDestroy(c->comm);
    delete c;
    cm->ctx = nullptr;
    return FLUID_OK;
}

}  // extern "C"
