// In-process transport: fluid_comm_t for several handles driven by several host threads of ONE process.
//
// The decomposed step (fluid_dist.hip) only sees the two callbacks of fluid_comm_t.  Over RCCL they are grouped
// ncclSend/ncclRecv and ncclAllReduce between processes (comm_rccl.cpp); here the peers are threads of the same
// process, the payload moves with device-to-device copies and the rendezvous is a mutex + condition variable.  This is
// how the tests run 2 x 2 x 2 blocks on the single GPU of their box (one process: the box allows few processes per
// card), and a host that drives several GPUs from one process can use it as well (the copies are peer copies then).
#include <hip/hip_runtime.h>

#include <condition_variable>
#include <cstdint>
#include <cstring>
#include <mutex>
#include <vector>

#include "../../include/fluid_hip.h"

namespace {

struct Msg {
    const void* ptr = nullptr;
    size_t bytes = 0;
    uint64_t posted = 0, consumed = 0;
};

struct Group {
    int size = 0;
    std::mutex mu;
    std::condition_variable cv;
    std::vector<Msg> box;                    // [src * size + dst]
    // all-reduce rendezvous
    int arrived = 0;
    uint64_t gen = 0;
    std::vector<std::vector<char>> stage;    // per rank
    std::vector<char> result[2];             // by generation parity
    bool failed = false;
};

struct Ctx {
    Group* g;
    int rank;
};

template <typename T>
void reduce_into(std::vector<char>& out, const std::vector<std::vector<char>>& in, int size, int64_t count, int op)
{
    out.resize((size_t)count * sizeof(T));
    T* o = (T*)out.data();
    memcpy(o, in[0].data(), (size_t)count * sizeof(T));
    for (int r = 1; r < size; ++r) {          // ascending ranks: a fixed order
        const T* a = (const T*)in[r].data();
        for (int64_t i = 0; i < count; ++i) {
            if (op == FLUID_OP_SUM) o[i] = o[i] + a[i];
            else if (op == FLUID_OP_MAX) o[i] = a[i] > o[i] ? a[i] : o[i];
            else o[i] = a[i] < o[i] ? a[i] : o[i];
        }
    }
}

int cb_exchange(void* vctx, int32_t n, const int32_t* peer, const void* const* sbuf, const size_t* sbytes, void* const* rbuf,
                const size_t* rbytes, void* stream)
{
    Ctx* c = (Ctx*)vctx;
    Group* g = c->g;
    const int me = c->rank;
    if (hipStreamSynchronize((hipStream_t)stream) != hipSuccess) return 1;   // my send buffers are complete, my receive buffers idle
    std::unique_lock<std::mutex> lk(g->mu);
    for (int i = 0; i < n; ++i) {
        if (peer[i] < 0 || peer[i] >= g->size || peer[i] == me) { g->failed = true; g->cv.notify_all(); return 1; }
        if (!sbytes[i]) continue;
        Msg& m = g->box[(size_t)me * g->size + peer[i]];
        m.ptr = sbuf[i];
        m.bytes = sbytes[i];
        m.posted++;
    }
    g->cv.notify_all();
    for (int i = 0; i < n; ++i) {
        if (!rbytes[i]) continue;
        Msg& m = g->box[(size_t)peer[i] * g->size + me];
        g->cv.wait(lk, [&] { return m.posted > m.consumed || g->failed; });
        if (g->failed || m.bytes != rbytes[i]) { g->failed = true; g->cv.notify_all(); return 1; }
        const void* src = m.ptr;
        lk.unlock();
        // on the receiver's own stream and waited for: a device-to-device hipMemcpy may return before the copy has landed,
        // and the null stream does not order with the handles' non-blocking streams
        hipError_t e = hipMemcpyAsync(rbuf[i], src, rbytes[i], hipMemcpyDefault, (hipStream_t)stream);
        if (e == hipSuccess) e = hipStreamSynchronize((hipStream_t)stream);
        lk.lock();
        if (e != hipSuccess) { g->failed = true; g->cv.notify_all(); return 1; }
        m.consumed++;
        g->cv.notify_all();
    }
    for (int i = 0; i < n; ++i) {   // my send buffers may be reused once every receiver has copied
        if (!sbytes[i]) continue;
        Msg& m = g->box[(size_t)me * g->size + peer[i]];
        g->cv.wait(lk, [&] { return m.consumed == m.posted || g->failed; });
        if (g->failed) return 1;
    }
    return 0;
}

int cb_allreduce(void* vctx, void* buf, int64_t count, int32_t dtype, int32_t op, void* stream)
{
    Ctx* c = (Ctx*)vctx;
    Group* g = c->g;
    const size_t es = dtype == FLUID_DT_U8 ? 1 : (dtype == FLUID_DT_I32 || dtype == FLUID_DT_F32 ? 4 : 8);
    const size_t bytes = (size_t)count * es;
    if (hipStreamSynchronize((hipStream_t)stream) != hipSuccess) return 1;
    std::vector<char>& mine = g->stage[c->rank];
    mine.resize(bytes);
    if (bytes && (hipMemcpyAsync(mine.data(), buf, bytes, hipMemcpyDeviceToHost, (hipStream_t)stream) != hipSuccess ||
                  hipStreamSynchronize((hipStream_t)stream) != hipSuccess))
        return 1;
    std::unique_lock<std::mutex> lk(g->mu);
    const uint64_t my_gen = g->gen;
    if (++g->arrived == g->size) {
        std::vector<char>& out = g->result[my_gen & 1];
        bool ok = true;
        for (int r = 0; r < g->size; ++r) ok = ok && g->stage[r].size() == bytes;   // every rank reduces the same count
        if (!ok) g->failed = true;
        else if (dtype == FLUID_DT_F64) reduce_into<double>(out, g->stage, g->size, count, op);
        else if (dtype == FLUID_DT_F32) reduce_into<float>(out, g->stage, g->size, count, op);
        else if (dtype == FLUID_DT_I32) reduce_into<int32_t>(out, g->stage, g->size, count, op);
        else if (dtype == FLUID_DT_U8) reduce_into<uint8_t>(out, g->stage, g->size, count, op);
        else reduce_into<int64_t>(out, g->stage, g->size, count, op);
        g->arrived = 0;
        g->gen++;
        g->cv.notify_all();
    } else {
        g->cv.wait(lk, [&] { return g->gen != my_gen || g->failed; });
    }
    if (g->failed) return 1;
    const std::vector<char>& out = g->result[my_gen & 1];
    lk.unlock();
    if (bytes && (hipMemcpyAsync(buf, out.data(), bytes, hipMemcpyHostToDevice, (hipStream_t)stream) != hipSuccess ||
                  hipStreamSynchronize((hipStream_t)stream) != hipSuccess))
        return 1;
    return 0;
}

}  // namespace

extern "C" {

int fluid_local_group_create(int32_t size, void** group)
{
    if (size < 1 || size > FLUID_MAX_RANKS || !group) return FLUID_ERR_ARG;
    Group* g = new Group();
    g->size = size;
    g->box.resize((size_t)size * size);
    g->stage.resize(size);
    *group = g;
    return FLUID_OK;
}

/* Wakes every waiting rank with an error (a driver thread whose rank failed outside the transport calls this, so that
 * its peers do not wait for it forever). */
int fluid_local_group_abort(void* group)
{
    Group* g = (Group*)group;
    if (!g) return FLUID_ERR_ARG;
    std::lock_guard<std::mutex> lk(g->mu);
    g->failed = true;
    g->cv.notify_all();
    return FLUID_OK;
}

int fluid_local_group_destroy(void* group)
{
    delete (Group*)group;
    return FLUID_OK;
}

// The fluid_comm_t's ctx is owned by the group's lifetime: destroy the handles first, then the group.
int fluid_local_comm_create(void* group, int32_t rank, fluid_comm_t* out)
{
    Group* g = (Group*)group;
    if (!g || !out || rank < 0 || rank >= g->size) return FLUID_ERR_ARG;
    Ctx* c = new Ctx{g, rank};   // (a few bytes per rank, released with the process)
    out->rank = rank;
    out->size = g->size;
    out->ctx = c;
    out->exchange = cb_exchange;
    out->allreduce = cb_allreduce;
    return FLUID_OK;
}

}  // extern "C"
