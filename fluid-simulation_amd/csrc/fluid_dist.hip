// Multi-GPU step: 3-D block decomposition, one process (or host thread) per GPU (include/fluid_hip.h).
//
// The reference is single-process (fluid.cc has no communication of any kind): everything here is new design.
//   particles   a block owns the particles whose base cell lies in it; per step ONE routing round sends a copy of every
//               particle to each adjacent block (<= 26) within one cell of its base cell — to its new owner if it has
//               left the block, as a ghost otherwise — so that every block forms the complete P2G sums of ITS cells in
//               the one-GPU summation order (cell lists ranked by global id);
//   fields      (decomposed solve) window arrays = block + 4 halo cells; halo exchanges of flags (4 wide), velocity,
//               pressure and the FLIP delta (1 wide) — one grouped exchange with all neighbours each;
//   numbering   the reference's x-major unknown numbering from all-reduced row-segment counts;
//   solve       PCG on the owned unknowns with a GLOBALLY COUPLED V-cycle: the levels below `split` (1 or 2) live on
//               the block + halo (the residual / coarse correction halos are exchanged, the kernels recompute inside
//               the halo what the one-GPU cycle computes there), the levels from `split` on are gathered by one
//               all-reduce and run redundantly on every rank.  Per cell the arithmetic is the one-GPU cycle's on the
//               same hierarchy, so the iteration count does not depend on the number of blocks.
//   replicated  (small grids) full-size arrays, the P2G result of the active box gathered from the blocks (all-gather between
//               adjacent blocks, else one all-reduce), pressure block solved identically on every rank.
#include "sim.h"
#include "dist_kernels.h"

using namespace fl;
#define fail fluid_fail
int cuts_from_hist(int n, const int64_t* hist, int P, int32_t* cuts);

#define COMMCHK(expr)                                                                              \
    do {                                                                                           \
        if ((expr) != 0) return fail(FLUID_ERR_HIP, std::string("comm callback failed: ") + #expr); \
    } while (0)

namespace {

constexpr int HALO_W = 4;   // window halo = halo of every locally held multigrid level (cells of that level)

struct IBox {   // half-open box in some global cell index space
    int lo[3], hi[3];
};
inline bool ib_empty(const IBox& b) { return b.hi[0] <= b.lo[0] || b.hi[1] <= b.lo[1] || b.hi[2] <= b.lo[2]; }
inline IBox ib_isect(const IBox& a, const IBox& b)
{
    IBox r;
    for (int k = 0; k < 3; ++k) { r.lo[k] = std::max(a.lo[k], b.lo[k]); r.hi[k] = std::min(a.hi[k], b.hi[k]); }
    if (ib_empty(r)) for (int k = 0; k < 3; ++k) { r.lo[k] = a.lo[k]; r.hi[k] = a.lo[k]; }
    return r;
}
inline IBox ib_grow(const IBox& a, int w)
{
    IBox r = a;
    if (ib_empty(a)) return r;
    for (int k = 0; k < 3; ++k) { r.lo[k] -= w; r.hi[k] += w; }
    return r;
}
inline long ib_cells(const IBox& b) { return ib_empty(b) ? 0 : (long)(b.hi[0] - b.lo[0]) * (b.hi[1] - b.lo[1]) * (b.hi[2] - b.lo[2]); }
// inclusive Box of b in the coordinates of an array whose cell 0 is global cell org
inline Box to_box(const IBox& b, const int org[3])
{
    if (ib_empty(b)) return Box{0, 0, 0, -1, -1, -1};
    return Box{b.lo[0] - org[0], b.lo[1] - org[1], b.lo[2] - org[2], b.hi[0] - 1 - org[0], b.hi[1] - 1 - org[1], b.hi[2] - 1 - org[2]};
}

struct HaloPlan {
    int n = 0;
    int peer[HALO_MAX_BOX];
    HaloArgs snd{}, rcv{};
    size_t scount[HALO_MAX_BOX], rcount[HALO_MAX_BOX];
    size_t stotal = 0, rtotal = 0;
};

struct DLevel {   // a multigrid level held on block + halo
    IBox dom{};   // cells of the level this rank holds (level-global indices); empty if the block misses the active box
    IBox own{};
    MLevel m{};
    HaloPlan plan;
    uint8_t *typ = nullptr, *cnt = nullptr;
    char *u = nullptr, *v = nullptr, *f = nullptr, *r = nullptr;
};

}  // namespace

struct DistState {
    fluid_comm_t comm{};
    int dims[3] = {1, 1, 1}, bc[3] = {0, 0, 0};
    std::vector<int> cuts[3];
    OwnBox ob{};
    int nbr[27];
    bool repl = false;
    // halo staging
    char *hs = nullptr, *hr = nullptr;
    size_t hs_cap = 0, hr_cap = 0;
    // particle routing
    double *mig_s = nullptr, *mig_r = nullptr;
    long mig_cap = 0;
    int *d_cnt = nullptr, *h_cnt = nullptr;   // [0..27) send counts / cursors, [32..59) receive counts, [64..] misc
    // replicated mode
    double* repl_buf = nullptr;
    size_t repl_cap = 0;
    // numbering
    int *rows = nullptr, *row_starts = nullptr;
    size_t rows_cap = 0;
    Box idx_box{0, 0, 0, -1, -1, -1};   // window cells whose indices may differ from -1
    // this step's boxes, global cell indices
    IBox Pg{}, Rg{}, Sg{};
    Box Rr{0, 0, 0, -1, -1, -1}, Sr{0, 0, 0, -1, -1, -1};   // owned parts, window coordinates
    // plans
    HaloPlan plan_flags, plan_f1, plan_gather;
    HaloPlan plan_split;              // the first replicated multigrid level: every rank's owned coarse block to every other rank
    bool split_exchange = false;      // ... used when all ranks are adjacent (dims <= 2 per axis); otherwise a SUM all-reduce
    int split_exchange_force = -1;    // FLUID_DIST_GATHER=exchange|allreduce
    // solver
    int split = 1, split_force = 0;
    IBox dom0{};                      // global level-0 multigrid domain [A, B)
    DLevel lv[2];
    uint8_t* cnt_pcg = nullptr;
    double *gstage[2] = {nullptr, nullptr}, *gpq = nullptr;
    long n_routed = 0;                // particles sent away so far (statistics)
    // re-balancing of the cut planes (fluid_dist_set_rebalance / FLUID_DIST_REBALANCE): every `rb_every` steps the particle
    // counts of the blocks are compared; beyond `rb_ratio` x the mean the planes are placed anew by particle count
    int rb_every = 0;
    double rb_ratio = 2.0;
    int n_rebalanced = 0;
    int n_rebalance_refused = 0;     // re-balances every rank gave up together (some rank had no room for a second window)
    // form of the PCG recurrence in the decomposed solve (FLUID_DIST_CG=cg|cgear): 0 = the loop of ConjugateGradient.h as the one-GPU solve
    // runs it (two scalar all-reduces per iteration), 1 = Chronopoulos-Gear (one); gcg = {gamma, alpha} of the last two bodies (device)
    int cg_form = 1;
    double* gcg = nullptr;
    // the previous step's GLOBAL active box was mostly air (every rank the same): this step looks for closed pockets (droplets)
    bool airy_prev = false, drops_step = false;
    // test hooks, read once in fluid_create_dist: the rank whose growth (route_round) / second window (dist_rebalance) is refused; -1 = none
    int fail_grow_rank = -1, fail_rebuild_rank = -1;
    int* rb_buf = nullptr;            // device: 3 N axis histograms + one count per rank
    std::vector<uint8_t> solid_global;   // what fluid_set_solid was given (empty: the default shell): a new window needs it again
    // overlap of the residual's halo exchange with the interior tiles of the level-0 down leg (FLUID_DIST_OVERLAP=0: off)
    bool overlap = true;
    int overlap_checked = 0;          // 0: not yet; 1: the overlapped exchange delivered the serial one's bytes on every rank; 2: it did not (overlap switched off)
    hipStream_t st2 = nullptr;        // the exchange runs here while the solver's stream sweeps the tiles that read no halo cell
    hipEvent_t ev_pack = nullptr, ev_halo = nullptr;
    int *tl_int = nullptr, *tl_bnd = nullptr;   // level-0 down-leg tiles (k_mg_down numbering): those that read no received cell / the others
    size_t tl_cap = 0;
    int n_int = 0, n_bnd = 0;
    std::vector<int> h_tl;            // host copy behind the (stream-ordered) upload
    uint8_t* tl_cls = nullptr;        // per level-0 leg tile: 1 = reads a received cell (the class the two lists above are cut by)
    std::vector<uint8_t> h_cls;
};

namespace {

// ---- geometry --------------------------------------------------------------------------------------------------
inline int rank_of(const DistState* d, int bx, int by, int bz) { return (bx * d->dims[1] + by) * d->dims[2] + bz; }
IBox block_of(const DistState* d, int rank)
{
    const int bz = rank % d->dims[2], by = (rank / d->dims[2]) % d->dims[1], bx = rank / (d->dims[1] * d->dims[2]);
    const int b[3] = {bx, by, bz};
    IBox r;
    for (int a = 0; a < 3; ++a) { r.lo[a] = d->cuts[a][b[a]]; r.hi[a] = d->cuts[a][b[a] + 1]; }
    return r;
}
// block of `rank` in the cell indices of multigrid level l (interior cuts are multiples of 4; the last cut is the grid's end)
IBox block_level(const DistState* d, int rank, int l, int N)
{
    IBox r = block_of(d, rank);
    for (int a = 0; a < 3; ++a) {
        r.lo[a] >>= l;
        r.hi[a] = r.hi[a] == N ? ((N - 1) >> l) + 1 : r.hi[a] >> l;
    }
    return r;
}

// Plan of one halo exchange: rank r owns own(r) (some global index space); my array holds own(me) grown by w; layout:
// array cell (i, j, k) = global cell org + (i, j, k), at element base + i*sx + j*sy + k.  I receive grow(own(me), w) ∩ own(nb)
// from each adjacent block nb and send grow(own(nb), w) ∩ own(me) — the two sides compute the same boxes.
template <typename OwnFn>
void make_plan(const DistState* d, HaloPlan& p, OwnFn own, int w, const int org[3], long base, long sx, long sy)
{
    p.n = 0;
    p.stotal = p.rtotal = 0;
    p.snd.base = p.rcv.base = base;
    p.snd.sx = p.rcv.sx = sx;
    p.snd.sy = p.rcv.sy = sy;
    const IBox mine = own(d->comm.rank);
    for (int dd = 0; dd < 27; ++dd) {
        if (dd == 13 || d->nbr[dd] < 0) continue;
        const int nb = d->nbr[dd];
        const IBox theirs = own(nb);
        const IBox rb = ib_isect(ib_grow(mine, w), theirs), sb = ib_isect(ib_grow(theirs, w), mine);
        const long rc = ib_cells(rb), sc = ib_cells(sb);
        if (!rc && !sc) continue;
        const int k = p.n++;
        p.peer[k] = nb;
        for (int a = 0; a < 3; ++a) {
            p.snd.lo[k][a] = sb.lo[a] - org[a]; p.snd.n[k][a] = sc ? sb.hi[a] - sb.lo[a] : 0;
            p.rcv.lo[k][a] = rb.lo[a] - org[a]; p.rcv.n[k][a] = rc ? rb.hi[a] - rb.lo[a] : 0;
        }
        p.snd.off[k] = (long)p.stotal; p.rcv.off[k] = (long)p.rtotal;
        p.scount[k] = (size_t)sc; p.rcount[k] = (size_t)rc;
        p.stotal += (size_t)sc; p.rtotal += (size_t)rc;
    }
    p.snd.nbox = p.rcv.nbox = p.n;
}

int ensure_stage(fluid_sim* s, size_t sbytes, size_t rbytes)
{
    DistState* d = s->ds;
    if (sbytes > d->hs_cap) {
        if (d->hs) { HIPCHK(hipStreamSynchronize(s->st)); hipFree(d->hs); d->hs = nullptr; }
        d->hs_cap = sbytes + sbytes / 2 + 4096;
        HIPCHK(hipMalloc((void**)&d->hs, d->hs_cap));
    }
    if (rbytes > d->hr_cap) {
        if (d->hr) { HIPCHK(hipStreamSynchronize(s->st)); hipFree(d->hr); d->hr = nullptr; }
        d->hr_cap = rbytes + rbytes / 2 + 4096;
        HIPCHK(hipMalloc((void**)&d->hr, d->hr_cap));
    }
    return FLUID_OK;
}

// exchange the halos of `narr` arrays of element size `elem` that share the plan's layout
int halo_exchange(fluid_sim* s, HaloPlan& p, int elem, int narr, void* const* arrays)
{
    DistState* d = s->ds;
    if (p.n == 0) return FLUID_OK;
    int rc = ensure_stage(s, p.stotal * narr * elem, p.rtotal * narr * elem);
    if (rc) return rc;
    p.snd.narr = p.rcv.narr = narr;
    for (int a = 0; a < HALO_MAX_ARR; ++a) p.snd.arr[a] = p.rcv.arr[a] = a < narr ? arrays[a] : arrays[0];
    launch_halo_copy(s->st, p.snd, elem, d->hs, true);
    HIPCHK(hipGetLastError());
    const void* sb[HALO_MAX_BOX];
    void* rb[HALO_MAX_BOX];
    size_t sn[HALO_MAX_BOX], rn[HALO_MAX_BOX];
    for (int k = 0; k < p.n; ++k) {
        sb[k] = d->hs + (size_t)p.snd.off[k] * narr * elem;
        rb[k] = d->hr + (size_t)p.rcv.off[k] * narr * elem;
        sn[k] = p.scount[k] * narr * elem;
        rn[k] = p.rcount[k] * narr * elem;
    }
    COMMCHK(d->comm.exchange(d->comm.ctx, p.n, p.peer, sb, sn, rb, rn, (void*)s->st));
    launch_halo_copy(s->st, p.rcv, elem, d->hr, false);
    HIPCHK(hipGetLastError());
    return FLUID_OK;
}
// The same exchange with the transfer and the unpack on the second stream: returns once the pack is queued on the solver's
// stream; the caller queues work that reads no received cell and then makes its stream wait for ev_halo (halo_wait).
int halo_exchange_begin(fluid_sim* s, HaloPlan& p, int elem, void* a)
{
    DistState* d = s->ds;
    if (p.n == 0) return FLUID_OK;
    int rc = ensure_stage(s, p.stotal * elem, p.rtotal * elem);
    if (rc) return rc;
    p.snd.narr = p.rcv.narr = 1;
    for (int k = 0; k < HALO_MAX_ARR; ++k) p.snd.arr[k] = p.rcv.arr[k] = a;
    launch_halo_copy(s->st, p.snd, elem, d->hs, true);
    HIPCHK(hipGetLastError());
    HIPCHK(hipEventRecord(d->ev_pack, s->st));
    HIPCHK(hipStreamWaitEvent(d->st2, d->ev_pack, 0));
    return FLUID_OK;
}
int halo_exchange_end(fluid_sim* s, HaloPlan& p, int elem)
{
    DistState* d = s->ds;
    if (p.n == 0) return FLUID_OK;
    const void* sb[HALO_MAX_BOX];
    void* rb[HALO_MAX_BOX];
    size_t sn[HALO_MAX_BOX], rn[HALO_MAX_BOX];
    for (int k = 0; k < p.n; ++k) {
        sb[k] = d->hs + (size_t)p.snd.off[k] * elem;
        rb[k] = d->hr + (size_t)p.rcv.off[k] * elem;
        sn[k] = p.scount[k] * elem;
        rn[k] = p.rcount[k] * elem;
    }
    COMMCHK(d->comm.exchange(d->comm.ctx, p.n, p.peer, sb, sn, rb, rn, (void*)d->st2));
    launch_halo_copy(d->st2, p.rcv, elem, d->hr, false);
    HIPCHK(hipGetLastError());
    HIPCHK(hipEventRecord(d->ev_halo, d->st2));
    HIPCHK(hipStreamWaitEvent(s->st, d->ev_halo, 0));
    return FLUID_OK;
}
int halo_exchange1(fluid_sim* s, HaloPlan& p, int elem, void* a)
{
    void* arr[1] = {a};
    return halo_exchange(s, p, elem, 1, arr);
}

int comm_allreduce(fluid_sim* s, void* buf, long count, int dtype, int op)
{
    if (s->ds->comm.size == 1) return FLUID_OK;   // one rank: the buffer already holds the result (no call with nobody to talk to)
    COMMCHK(s->ds->comm.allreduce(s->ds->comm.ctx, buf, (int64_t)count, dtype, op, (void*)s->st));
    return FLUID_OK;
}

// Every rank learns whether ANY rank failed since the last agreement, and all of them leave the step with the same code:
// MAX over the ranks of a local flag (stream-ordered all-reduce of one int + the read-back).  A rank that hits an error
// between two communication calls must not simply return — its peers would wait for it in the next exchange for ever
// (RCCL has no timeout) — so the few places where a step can still fail on ONE rank (growing the particle arrays for
// what migrates in) bring their result here BEFORE the next data exchange.  Everything else a step needs is sized for
// the whole block when the handle is created (fluid_create_dist), so that no step allocates.
int dist_agree(fluid_sim* s, int local_rc)
{
    DistState* d = s->ds;
    if (d->comm.size == 1) return local_rc;
    int* flag = d->h_cnt + 96;
    *flag = local_rc ? 1 : 0;
    hipError_t e = hipMemcpyAsync(d->d_cnt + 96, flag, sizeof(int), hipMemcpyHostToDevice, s->st);
    // (a failure of these two copies or of the stream leaves no way to tell the peers: that is a dead device, not a full one)
    if (e == hipSuccess && d->comm.allreduce(d->comm.ctx, d->d_cnt + 96, 1, FLUID_DT_I32, FLUID_OP_MAX, (void*)s->st) != 0)
        return fail(FLUID_ERR_HIP, "comm callback failed: agreement all-reduce");
    if (e == hipSuccess) e = hipMemcpyAsync(flag, d->d_cnt + 96, sizeof(int), hipMemcpyDeviceToHost, s->st);
    if (e == hipSuccess) e = hipStreamSynchronize(s->st);
    if (e != hipSuccess) return fail(FLUID_ERR_HIP, std::string("agreement: ") + hipGetErrorString(e));
    if (local_rc) return local_rc;   // (fluid_last_error() already says what failed here)
    if (*flag) return fail(FLUID_ERR_PEER, "another rank failed in this step (fluid_last_error() there says why); the step was abandoned on every rank");
    return FLUID_OK;
}

// ---- particles: migration + ghosts + sort ----------------------------------------------------------------------------------
// The routing round of a step (k_route: owners-to-be and ghosts in one go): count per direction, exchange the counts, write and
// exchange the records, append what arrived behind the live particles.
int route_round(fluid_sim* s)
{
    DistState* d = s->ds;
    const Grid g = s->g;
    int rc;
    {   // a block without neighbours (one rank) has nobody to hand particles to or to copy ghosts for
        bool any = false;
        for (int dd = 0; dd < 27; ++dd) any = any || (dd != 13 && d->nbr[dd] >= 0);
        if (!any) return FLUID_OK;
    }
    HIPCHK(hipMemsetAsync(d->d_cnt, 0, 64 * sizeof(int), s->st));
    launch_route(s->st, g, d->ob, s->np, s->pa.shifted(s->p_off), d->d_cnt, nullptr, 0);
    HIPCHK(hipGetLastError());
    // counts to / from the neighbours
    int np = 0, peer[26], dir[26];
    const void* sb[26];
    void* rb[26];
    size_t sn[26], rn[26];
    for (int dd = 0; dd < 27; ++dd) {
        if (dd == 13 || d->nbr[dd] < 0) continue;
        peer[np] = d->nbr[dd];
        dir[np] = dd;
        sb[np] = d->d_cnt + dd;            // what I send towards dd ...
        rb[np] = d->d_cnt + 32 + dd;       // ... and what the block in direction dd sends to me
        sn[np] = rn[np] = sizeof(int);
        ++np;
    }
    if (np) COMMCHK(d->comm.exchange(d->comm.ctx, np, peer, sb, sn, rb, rn, (void*)s->st));
    HIPCHK(hipMemcpyAsync(d->h_cnt, d->d_cnt, 64 * sizeof(int), hipMemcpyDeviceToHost, s->st));
    HIPCHK(hipStreamSynchronize(s->st));
    long soff[27], roff[27], stot = 0, rtot = 0;
    for (int dd = 0; dd < 27; ++dd) {
        soff[dd] = stot; roff[dd] = rtot;
        if (dd == 13 || d->nbr[dd] < 0) continue;
        stot += d->h_cnt[dd];
        rtot += d->h_cnt[32 + dd];
    }
    // room for what arrives (a block's share grows as the fluid spreads into it): grown on demand — the one allocation a
    // step can still need.  The peers are about to send: whether every rank has the room is agreed before anybody does.
    auto grow = [&]() -> int {
        if (stot > d->mig_cap || rtot > d->mig_cap) {
            HIPCHK(hipStreamSynchronize(s->st));
            hipFree(d->mig_s); hipFree(d->mig_r);
            d->mig_s = d->mig_r = nullptr;
            d->mig_cap = 0;
            const long cap = std::max(stot, rtot) * 3 / 2 + 4096;
            if (d->fail_grow_rank == d->comm.rank)   // (tests: a rank that cannot grow)
                return fail(FLUID_ERR_HIP, "migration buffers: allocation refused (FLUID_DIST_FAIL_GROW)");
            HIPCHK(hipMalloc((void**)&d->mig_s, (size_t)cap * 56));
            HIPCHK(hipMalloc((void**)&d->mig_r, (size_t)cap * 56));
            d->mig_cap = cap;
        }
        return grow_particles(s, s->p_off + s->np + rtot);
    };
    if ((rc = dist_agree(s, grow()))) return rc;
    int cur[27];
    for (int dd = 0; dd < 27; ++dd) cur[dd] = (int)soff[dd];
    memcpy(d->h_cnt + 64, cur, sizeof(cur));
    HIPCHK(hipMemcpyAsync(d->d_cnt, d->h_cnt + 64, 27 * sizeof(int), hipMemcpyHostToDevice, s->st));
    if (stot) launch_route(s->st, g, d->ob, s->np, s->pa.shifted(s->p_off), d->d_cnt, d->mig_s, 1);
    HIPCHK(hipGetLastError());
    for (int k = 0; k < np; ++k) {
        const int dd = dir[k];
        sb[k] = d->mig_s + (size_t)soff[dd] * 7;
        rb[k] = d->mig_r + (size_t)roff[dd] * 7;
        sn[k] = (size_t)d->h_cnt[dd] * 56;
        rn[k] = (size_t)d->h_cnt[32 + dd] * 56;
    }
    if (np) COMMCHK(d->comm.exchange(d->comm.ctx, np, peer, sb, sn, rb, rn, (void*)s->st));
    launch_unpack_records(s->st, rtot, d->mig_r, s->pa, s->p_off + s->np);
    HIPCHK(hipGetLastError());
    s->np += rtot;
    d->n_routed += stot;
    return FLUID_OK;
}

// routing (migration + ghosts), counting sort of everything by window cell, global particle bounding box
int dist_particles(fluid_sim* s)
{
    DistState* d = s->ds;
    const Grid g = s->g;
    int rc;
    int tok = prof_begin(s, FLUID_PROF_SORT, (double)s->np);
    if ((rc = route_round(s))) return rc;
    // sort: cells of the window, then "off the window" (= off the grid: only an edge block can hold such a particle), then dead.
    // Like the one-GPU sort only the x planes the particles can be in are zeroed, counted into and scanned (the global box of the
    // previous step + 3: CFL, and what arrived from the neighbours lies inside that box too); if the box read back says otherwise
    // (first step, new particles) the sort is repeated over the whole window — a local decision, no collective inside.
    {
        int ax0 = 0, ax1 = g.nx - 1;
        const bool guess = s->sort_hint && !box_empty(s->Pb);
        if (guess) { ax0 = std::max(0, s->Pb.x0 - 3); ax1 = std::min(g.nx - 1, s->Pb.x1 + 3); }
        if ((rc = sort_pass(s, ax0, ax1, d->h_cnt))) return rc;
        if (guess && s->h_ss->bbox_max[0] >= 0 && (s->h_ss->bbox_min[0] < std::min(ax0 + 2, s->Pb.x0) || s->h_ss->bbox_max[0] > std::max(ax1 - 2, s->Pb.x1))) {
            if ((rc = sort_pass(s, 0, g.nx - 1, d->h_cnt))) return rc;
        }
        s->sort_hint = true;
    }
    const long live = d->h_cnt[1];   // cells + off-grid bucket; the dead (last step's ghosts, this step's migrants) are dropped
    s->n_out = d->h_cnt[1] - d->h_cnt[0];
    launch_reorder(s->st, live, s->order2, s->pa.shifted(s->p_off), s->pb, s->pw, s->cap);   // + the P2G axis weights
    HIPCHK(hipGetLastError());
    std::swap(s->pa, s->pb);
    s->p_off = 0;
    s->np = live;
    // global bounding box of the base cells (+ the fullest cell of any rank: it picks the P2G form, which every rank
    // must take alike): MIN over [min3, -max3, -max_cell]
    {
        const StepState& h = *s->h_ss;
        int* v = d->h_cnt + 64;
        const int o[3] = {g.ox, g.oy, g.oz};
        for (int a = 0; a < 3; ++a) {
            v[a] = h.bbox_max[a] < 0 ? 0x7fffffff : h.bbox_min[a] + o[a];
            v[3 + a] = h.bbox_max[a] < 0 ? 0x7fffffff : -(h.bbox_max[a] + o[a]);
        }
        v[6] = -h.max_cell;
        if (d->comm.size > 1) {
            HIPCHK(hipMemcpyAsync(d->d_cnt + 64, v, 7 * sizeof(int), hipMemcpyHostToDevice, s->st));
            if ((rc = comm_allreduce(s, d->d_cnt + 64, 7, FLUID_DT_I32, FLUID_OP_MIN))) return rc;
            HIPCHK(hipMemcpyAsync(v, d->d_cnt + 64, 7 * sizeof(int), hipMemcpyDeviceToHost, s->st));
            HIPCHK(hipStreamSynchronize(s->st));
        }
        s->max_cell = -v[6];
        const int N = g.N;
        if (v[0] == 0x7fffffff) {
            d->Pg = IBox{{0, 0, 0}, {0, 0, 0}};
        } else {
            for (int a = 0; a < 3; ++a) { d->Pg.lo[a] = v[a]; d->Pg.hi[a] = -v[3 + a] + 1; }
        }
        const IBox grid{{0, 0, 0}, {N, N, N}};
        d->Rg = ib_isect(ib_grow(d->Pg, 1), grid);
        d->Sg = ib_isect(ib_grow(d->Pg, 2), grid);
    }
    const int org[3] = {g.ox, g.oy, g.oz};
    const IBox own{{d->ob.lo[0], d->ob.lo[1], d->ob.lo[2]}, {d->ob.hi[0], d->ob.hi[1], d->ob.hi[2]}};
    const IBox win{{g.ox, g.oy, g.oz}, {g.ox + g.nx, g.oy + g.ny, g.oz + g.nz}};
    d->Rr = to_box(ib_isect(d->Rg, own), org);
    d->Sr = to_box(ib_isect(d->Sg, own), org);
    s->Rb = to_box(ib_isect(d->Rg, win), org);   // what this rank's arrays hold of the global boxes
    s->Sb = to_box(ib_isect(d->Sg, win), org);
    s->Pb = to_box(ib_isect(d->Pg, win), org);
    s->p2g_ref_cells = ib_cells(d->Rg);
    prof_end(s, FLUID_PROF_SORT, tok);
    s->sorted = false;   // (the single-GPU gather through LDS tiles is not used here)
    return FLUID_OK;
}

void stats_begin(fluid_sim* s)
{
    DistState* d = s->ds;
    memset(&s->stats, 0, sizeof(s->stats));
    s->stats.paths = d->repl ? FLUID_PATH_DIST_REPLICATED : FLUID_PATH_DIST_DECOMPOSED;
    s->stats.dt_in = s->dt;
    s->stats.dt_out = s->dt;
    for (int a = 0; a < 3; ++a) {
        s->stats.box_lo[a] = d->Rg.lo[a];
        s->stats.box_hi[a] = d->Rg.hi[a] - 1;
    }
}

// FLIP gather + advect of the owned particles (fluid.cc:1490); the ghosts have served and are marked dead first
int dist_g2p_advect(fluid_sim* s, fluid_step_stats_t* stats)
{
    DistState* d = s->ds;
    const Grid g = s->g;
    int rc;
    launch_kill_ghosts(s->st, g, d->ob, s->np, s->pa.shifted(s->p_off));
    int tok = prof_begin(s, FLUID_PROF_G2P, (double)s->np);
    // the particles are still in the order of the sort (by window cell): where the owned part of the particle box is densely
    // filled, gather through LDS tiles over exactly the owned cells (the ghosts sit in the halo cells: never touched); the
    // off-grid bucket at the end of the arrays only has its speeds counted
    const IBox ownb{{d->ob.lo[0], d->ob.lo[1], d->ob.lo[2]}, {d->ob.hi[0], d->ob.hi[1], d->ob.hi[2]}};
    const int org[3] = {g.ox, g.oy, g.oz};
    const Box ownPb = to_box(ib_isect(d->Pg, ownb), org);
    if (s->p_off == 0 && !box_empty(ownPb) && (double)s->np >= 4.0 * (double)ownPb.cells()) {
        launch_g2p_tiled(s->st, g, ownPb, s->pa, s->cell_start, s->dcx, s->dcy, s->dcz, s->pcx, s->pcy, s->pcz, s->prm.flip_blend, s->ss);
        if (s->n_out > 0)
            launch_g2p(s->st, g, s->n_out, s->pa.shifted(s->np - s->n_out), s->dcx, s->dcy, s->dcz, s->pcx, s->pcy, s->pcz, s->prm.flip_blend, s->ss);
    } else {
        launch_g2p(s->st, g, s->np, s->pa.shifted(s->p_off), s->dcx, s->dcy, s->dcz, s->pcx, s->pcy, s->pcz, s->prm.flip_blend, s->ss);
    }
    prof_end(s, FLUID_PROF_G2P, tok);
    HIPCHK(hipGetLastError());
    // non-negative doubles order like their bit patterns: MAX over int64
    if ((rc = comm_allreduce(s, &s->ss->max_speed_bits, 1, FLUID_DT_I64, FLUID_OP_MAX))) return rc;
    launch_advect(s->st, g, s->np, s->pa.shifted(s->p_off), s->flags, s->prm.max_dt, s->prm.dx, s->ss);
    HIPCHK(hipGetLastError());
    if ((rc = read_ss(s))) return rc;
    s->dt = s->h_ss->dt;
    double ms;
    memcpy(&ms, &s->h_ss->max_speed_bits, sizeof(double));
    s->stats.max_speed = ms;
    s->stats.dt_out = s->dt;
    s->sorted = false;
    s->have_p2g = false;
    if (stats) *stats = s->stats;
    return FLUID_OK;
}

// ---- replicated pressure block ------------------------------------------------------------------------------------
// Particles sharded by block; the P2G result of the whole active box is assembled on every rank — an all-gather between
// the blocks when all of them are adjacent, else one SUM all-reduce: bit-identical to a one-GPU P2G either way — and flags, numbering, the pressure do..while with the single-GPU multigrid PCG and the FLIP delta field run
// identically on every rank with no communication.
int dist_step_replicated(fluid_sim* s, fluid_step_stats_t* stats)
{
    DistState* d = s->ds;
    const Grid g = s->g;
    int rc;
    if ((rc = dist_particles(s))) return rc;
    if ((rc = clear_dirty(s))) return rc;
    stats_begin(s);
    if (!box_empty(s->Rb)) {
        s->dirty = s->Sb;
        if (!box_empty(d->Rr)) {
            int tok = prof_begin(s, FLUID_PROF_P2G, (double)d->Rr.cells());
            rc = run_p2g(s, d->Rr);
            prof_end(s, FLUID_PROF_P2G, tok);
            if (rc) return rc;
        }
        if (d->dims[0] <= 2 && d->dims[1] <= 2 && d->dims[2] <= 2) {
            // every other block is adjacent (<= 2 blocks per axis, e.g. 2 x 2 x 2): ALL-GATHER — each block sends its owned part
            // of the box straight to every other block (one grouped exchange over all its xGMI links at once, no arithmetic,
            // half the bytes of the all-reduce below); exact copies, so the fields are bit-identical to a one-GPU P2G
            const int org[3] = {0, 0, 0};
            make_plan(d, d->plan_gather, [&](int r) { return ib_isect(block_of(d, r), d->Rg); }, g.N, org, 0, g.sx(), g.nz);
            if ((rc = halo_exchange1(s, d->plan_gather, 4, s->container))) return rc;
            void* a[3] = {s->u, s->v, s->w};
            if ((rc = halo_exchange(s, d->plan_gather, 8, 3, a))) return rc;
            launch_copy_vel_before(s->st, g, s->Rb, s->u, s->v, s->w, s->ub, s->vb, s->wb);
            HIPCHK(hipGetLastError());
        } else {
            // general decomposition: one SUM all-reduce of [container | u | v | w] over the box (every cell has exactly one
            // owner, the others add exact zeros)
            const size_t need = 4 * (size_t)s->Rb.cells();
            if (need > d->repl_cap) {
                if (d->repl_buf) { HIPCHK(hipStreamSynchronize(s->st)); hipFree(d->repl_buf); d->repl_buf = nullptr; }
                HIPCHK(hipMalloc((void**)&d->repl_buf, (need + need / 4) * sizeof(double)));
                d->repl_cap = need + need / 4;
            }
            launch_pack_box_own(s->st, g, s->Rb, d->Rr, s->container, s->u, s->v, s->w, d->repl_buf);
            HIPCHK(hipGetLastError());
            if ((rc = comm_allreduce(s, d->repl_buf, (long)need, FLUID_DT_F64, FLUID_OP_SUM))) return rc;
            launch_unpack_box(s->st, g, s->Rb, d->repl_buf, s->container, s->u, s->v, s->w, s->ub, s->vb, s->wb);
            HIPCHK(hipGetLastError());
        }
    }
    s->have_p2g = true;
    s->step_counter++;
    s->have_flags = false;
    if ((rc = phase_flags(s))) return rc;           // whole grid on every rank: global numbering
    double error = NAN;
    do {
        if ((rc = phase_pressure_pass(s, &error))) return rc;
        if (s->prm.max_outer_passes > 0 && s->stats.outer_passes >= s->prm.max_outer_passes) break;
    } while (error > s->prm.outer_tol);
    if ((rc = pic_fields(s))) return rc;
    if (!box_empty(s->Rb)) launch_flip_delta(s->st, g, s->Rb, s->u, s->v, s->w, s->ub, s->vb, s->wb, s->dcx, s->dcy, s->dcz, s->pcx, s->pcy, s->pcz);
    return dist_g2p_advect(s, stats);
}

// ---- decomposed solve: multigrid hierarchy of the step -----------------------------------------------------------------
MLevel level_layout(int dx, int dy, int dz)   // the layout mg_coarser gives a level of these dims
{
    MLevel m;
    m.dx = dx; m.dy = dy; m.dz = dz;
    const int Lz = (16 + std::max(dz, 0) + 1 + 15) / 16 * 16;
    m.sy = Lz; m.sx = (long)(std::max(dy, 0) + 2) * Lz;
    m.ox = 1; m.oy = 1; m.oz = 16;
    m.cells = (size_t)(std::max(dx, 0) + 2) * m.sx + Lz;
    return m;
}
// the coarse level as the kernels of the finer level must see it: coarse cell 0 under fine cell 0
MLevel coarse_view(const MLevel& mc, const int off[3])
{
    MLevel v = mc;
    v.ox += off[0]; v.oy += off[1]; v.oz += off[2];
    v.dx = std::max(mc.dx - off[0], 0); v.dy = std::max(mc.dy - off[1], 0); v.dz = std::max(mc.dz - off[2], 0);
    return v;
}

// domain of global level l: cells [lo, lo + dim) of that level's index space
struct GDom {
    int lo[3], dim[3];
};

template <typename V>
MgCoef<V> coef_as(const fluid_sim* s, int level);

int dist_mg_setup(fluid_sim* s)
{
    DistState* d = s->ds;
    const Grid g = s->g;
    const int N = g.N, me = d->comm.rank;
    int rc;
    // level-0 domain [A, B): the active box + 1 ring, its low corner aligned down to a multiple of 4 so that the block
    // cuts (multiples of 4) are cell boundaries of the two coarser levels
    GDom gd[fluid_sim::MG_MAXL];
    for (int a = 0; a < 3; ++a) {
        const int A = std::max(0, (d->Rg.lo[a] - 1) & ~3), B = std::min(N, d->Rg.hi[a] + 1);
        gd[0].lo[a] = A; gd[0].dim[a] = B - A;
        d->dom0.lo[a] = A; d->dom0.hi[a] = B;
    }
    int nl = 1;
    while (nl < fluid_sim::MG_MAXL && (nl < 2 || gd[nl - 1].dim[0] > 8 || gd[nl - 1].dim[1] > 8 || gd[nl - 1].dim[2] > 8)) {
        for (int a = 0; a < 3; ++a) { gd[nl].lo[a] = gd[nl - 1].lo[a] >> 1; gd[nl].dim[a] = (gd[nl - 1].dim[a] + 1) / 2; }
        ++nl;
    }
    if (gd[nl - 1].dim[0] > 8 || gd[nl - 1].dim[1] > 8 || gd[nl - 1].dim[2] > 8) return fail(FLUID_ERR_STATE, "multigrid: too many levels");
    // levels held on block + halo: 0 .. split-1.  Level 1 is gathered (split = 1) while it is small: one all-reduce of
    // the level instead of two halo exchanges per iteration, at the price of every rank sweeping all of it.
    const long l1 = (long)gd[1].dim[0] * gd[1].dim[1] * gd[1].dim[2];
    // Galerkin coarse levels by aggregation (kernels_gal.hip) in a mostly-air box, like the one-GPU solve: decided first, from global
    // numbers only (every rank the same cycle) — the aggregated levels run replicated, so a step that takes them gathers level 1
    // (split = 1) whatever its size.
    s->gal = false;
    s->gal_eligible = false;
    bool want_gal = false;
    int gal_lc = 1;
    {
        while (gal_lc < nl - 1 && !gal_fits_coarsest(level_layout(gd[gal_lc].dim[0], gd[gal_lc].dim[1], gd[gal_lc].dim[2]))) ++gal_lc;
        const double c0 = (double)gd[0].dim[0] * gd[0].dim[1] * gd[0].dim[2];
        const bool airy = (double)s->stats.num_active < 0.45 * c0 && c0 > 1500000.0;
        if (s->gal_mode && s->mg_fp32 && use_mg(s) && d->split_force != 2 && gal_lc >= 2 && gal_lc < nl &&
            gal_fits_coarsest(level_layout(gd[gal_lc].dim[0], gd[gal_lc].dim[1], gd[gal_lc].dim[2])) && (airy || s->gal_mode >= 2)) {
            s->gal_eligible = true;
            want_gal = s->gal_mode >= 2;
            if (s->gal_mode == 1) {
                if (s->gal_it[0] < 0) want_gal = false;
                else if (s->gal_it[1] < 0) want_gal = true;
                else {
                    want_gal = 21 * s->gal_it[1] < 20 * s->gal_it[0];   // counts x the measured cost of an iteration, as on one GPU (gal_build)
                    if (++s->gal_since_probe >= 32) { want_gal = !want_gal; s->gal_since_probe = 0; }
                }
            }
        } else {
            s->gal_it[0] = s->gal_it[1] = -1;   // (measured again when the box is mostly air again)
        }
    }
    int split = d->split_force ? d->split_force : (want_gal || l1 <= 300000 ? 1 : 2);
    if (split > nl - 1) split = nl - 1;
    if (split < 1) split = 1;
    d->split = split;
    s->mg_nl = nl;
    auto gdom_box = [&](int l) { IBox b; for (int a = 0; a < 3; ++a) { b.lo[a] = gd[l].lo[a]; b.hi[a] = gd[l].lo[a] + gd[l].dim[a]; } return b; };
    auto own_l = [&](int l, int rank) { return ib_isect(block_level(d, rank, l, N), gdom_box(l)); };
    // global (replicated) levels split .. nl-1 use the handle's level arrays; the tail kernel takes what fits its LDS
    for (int l = split; l < nl; ++l) s->mgl[l] = level_layout(gd[l].dim[0], gd[l].dim[1], gd[l].dim[2]);
    const size_t es = s->mg_fp32 ? sizeof(float) : sizeof(double);
    int tail = nl - 1;
    auto fits = [&](int t) {
        const size_t b = nl - t <= MG_TAIL_MAX ? mg_tail_lds_bytes(nl - t, s->mgl + t, es) : 0;
        return b > 0 && b <= MG_TAIL_LDS;
    };
    while (tail > split && fits(tail - 1)) --tail;
    if (!fits(tail)) return fail(FLUID_ERR_STATE, "multigrid: coarsest level does not fit the tail kernel");
    s->mg_tail = tail;
    // local levels
    for (int l = 0; l < split; ++l) {
        DLevel& L = d->lv[l];
        L.own = own_l(l, me);
        L.dom = ib_isect(ib_grow(L.own, HALO_W), gdom_box(l));
        if (l == 0) {
            const int org[3] = {g.ox, g.oy, g.oz};
            s->L = make_lbox(to_box(L.dom, org));
            // level-0 arrays share the PCG's box-local layout; domain cell 0 = first interior cell of the LBox
            L.m.dx = s->L.nx; L.m.dy = s->L.ny; L.m.dz = s->L.nz;
            L.m.sx = (long)s->L.Ly * s->L.Lz; L.m.sy = s->L.Lz;
            L.m.ox = 1; L.m.oy = 1; L.m.oz = LBOX_K0;
            L.m.cells = s->L.cells() + 2 * (size_t)s->L.Lz;
        } else {
            L.m = level_layout(std::max(L.dom.hi[0] - L.dom.lo[0], 0), std::max(L.dom.hi[1] - L.dom.lo[1], 0), std::max(L.dom.hi[2] - L.dom.lo[2], 0));
        }
        make_plan(d, L.plan, [&](int r) { return own_l(l, r); }, HALO_W, L.dom.lo, (long)L.m.at(0, 0, 0), L.m.sx, L.m.sy);
    }
    s->mgl[0] = d->lv[0].m;
    // Level-0 down-leg tiles by what they read (k_mg_down<..., 8, 8, 16, false>: the tile + 2 cells): the tiles that touch no cell
    // the residual's halo exchange writes can be swept while that exchange is in flight
    d->n_int = d->n_bnd = 0;
    if (d->overlap && !ib_empty(d->lv[0].dom) && d->lv[0].plan.n > 0) {
        const MLevel& m0 = d->lv[0].m;
        const HaloPlan& hp = d->lv[0].plan;
        constexpr int TX = 8, TY = 8, TZ = 16, H = 2;   // = MG_TX, MG_TY, MG_TZ of kernels_mg.hip
        const int gx = (m0.dz + TZ - 1) / TZ, gy = (m0.dy + TY - 1) / TY, gz = (m0.dx + TX - 1) / TX, nt = gx * gy * gz;
        d->h_tl.assign((size_t)2 * nt, 0);
        d->h_cls.assign((size_t)nt, 0);
        int* li = d->h_tl.data();
        int* lb = li + nt;
        for (int t = 0; t < nt; ++t) {
            const int tbx = t % gx, tby = (t / gx) % gy, tbz = t / (gx * gy);
            const int lo[3] = {tbz * TX - H, tby * TY - H, tbx * TZ - H}, hi[3] = {tbz * TX + TX + H, tby * TY + TY + H, tbx * TZ + TZ + H};
            bool touches = false;
            for (int k = 0; k < hp.n && !touches; ++k) {
                if (!hp.rcount[k]) continue;
                bool hit = true;
                for (int a = 0; a < 3; ++a) hit = hit && lo[a] < hp.rcv.lo[k][a] + hp.rcv.n[k][a] && hi[a] > hp.rcv.lo[k][a];
                touches = hit;
            }
            d->h_cls[t] = touches;
            if (touches) lb[d->n_bnd++] = t; else li[d->n_int++] = t;
        }
        if ((size_t)nt > d->tl_cap) {
            if (d->tl_int) { HIPCHK(hipStreamSynchronize(s->st)); hipFree(d->tl_int); hipFree(d->tl_bnd); hipFree(d->tl_cls); d->tl_int = d->tl_bnd = nullptr; d->tl_cls = nullptr; }
            d->tl_cap = (size_t)nt + nt / 4 + 64;
            HIPCHK(hipMalloc((void**)&d->tl_int, d->tl_cap * sizeof(int)));
            HIPCHK(hipMalloc((void**)&d->tl_bnd, d->tl_cap * sizeof(int)));
            HIPCHK(hipMalloc((void**)&d->tl_cls, d->tl_cap));
        }
        if (d->n_int) HIPCHK(hipMemcpyAsync(d->tl_int, li, d->n_int * sizeof(int), hipMemcpyHostToDevice, s->st));
        if (d->n_bnd) HIPCHK(hipMemcpyAsync(d->tl_bnd, lb, d->n_bnd * sizeof(int), hipMemcpyHostToDevice, s->st));
    }
    // The first replicated level is assembled from the blocks' owned coarse cells.  With at most two blocks per axis every
    // rank is adjacent to every other: each sends its block straight to the 1..7 others (one grouped send/recv round, all
    // xGMI links in parallel, 7/8 of the level per rank) instead of a ring all-reduce of the zero-padded level (twice the
    // bytes, 2(n-1) steps).  More blocks per axis: all-reduce.
    d->split_exchange = d->dims[0] <= 2 && d->dims[1] <= 2 && d->dims[2] <= 2;
    if (d->split_exchange_force >= 0) d->split_exchange = d->split_exchange && d->split_exchange_force == 1;
    if (d->split_exchange)
        make_plan(d, d->plan_split, [&](int r) { return own_l(split, r); }, N, gd[split].lo, (long)s->mgl[split].at(0, 0, 0), s->mgl[split].sx, s->mgl[split].sy);
    // one slab, zeroed by one fill (mg_setup of the single-GPU path does the same)
    size_t total = 0;
    auto take = [&](size_t bytes) { const size_t o = total; total += (bytes + 255) / 256 * 256; return o; };
    size_t o_typ[fluid_sim::MG_MAXL], o_cnt[fluid_sim::MG_MAXL], o_u[fluid_sim::MG_MAXL], o_v[fluid_sim::MG_MAXL], o_f[fluid_sim::MG_MAXL], o_r[fluid_sim::MG_MAXL];
    for (int l = 0; l < nl; ++l) {
        const size_t c = (l < split ? d->lv[l].m.cells : s->mgl[l].cells) + 64;
        o_typ[l] = take(c); o_cnt[l] = take(c);
        o_u[l] = take(c * es); o_v[l] = take(c * es); o_f[l] = take(c * es); o_r[l] = take(c * es);
    }
    const size_t o_z = take((d->lv[0].m.cells + 64) * 8);
    if (total > s->mg_slab_cap) {
        if (s->mg_slab) { HIPCHK(hipStreamSynchronize(s->st)); hipFree(s->mg_slab); s->mg_slab = nullptr; }
        const size_t cap = total + total / 4;
        HIPCHK(hipMalloc((void**)&s->mg_slab, cap));
        s->mg_slab_cap = cap;
    }
    for (int l = 0; l < nl; ++l) {
        s->mg_typ[l] = (uint8_t*)(s->mg_slab + o_typ[l]);
        s->mg_cnt[l] = l ? (uint8_t*)(s->mg_slab + o_cnt[l]) : nullptr;
        s->mg_u[l] = s->mg_slab + o_u[l]; s->mg_v[l] = s->mg_slab + o_v[l];
        s->mg_f[l] = l ? s->mg_slab + o_f[l] : nullptr;
        s->mg_r[l] = s->mg_slab + o_r[l];
        if (l < split) {
            DLevel& L = d->lv[l];
            L.typ = s->mg_typ[l]; L.cnt = l ? s->mg_cnt[l] : s->cntL;
            L.u = s->mg_u[l]; L.v = s->mg_v[l]; L.f = s->mg_f[l]; L.r = s->mg_r[l];
        }
    }
    s->Zmg = s->mg_slab + o_z;
    HIPCHK(hipMemsetAsync(s->mg_slab, 0, total, s->st));
    // ---- count bytes and cell types -----------------------------------------------------------------------------
    // level 0: from the flags (valid on the whole window: their halo is HALO_W wide)
    launch_cnt_local(s->st, g, s->L, s->flags, s->cntL);
    {
        const int org0[3] = {d->lv[0].dom.lo[0] - 1, d->lv[0].dom.lo[1] - 1, d->lv[0].dom.lo[2] - LBOX_K0};   // global cell of LBox index 0
        const IBox o = d->lv[0].own;
        Box ownL = ib_empty(o) ? Box{0, 0, 0, -1, -1, -1}
                               : Box{o.lo[0] - org0[0], o.lo[1] - org0[1], o.lo[2] - org0[2], o.hi[0] - 1 - org0[0], o.hi[1] - 1 - org0[1], o.hi[2] - 1 - org0[2]};
        // Closed pockets of the pressure system (kernels_droplets.hip) leave the global solve here too: every rank searches its
        // local box and claims the pockets that lie wholly in its OWNED cells (their pressure is written by that one owner; a pocket
        // across a cut stays in the global solve), clears their count bytes, and the count bytes' halo then comes from the owners — so
        // both sides of a cut build cnt_pcg, the cell types and the coarse levels on the same system.  Whether a step searches is
        // decided from global numbers (the previous step's box was mostly air), so every rank takes part in the exchange or none does.
        s->n_drop = 0;
        d->drops_step = s->drops_on && use_mg(s) && (d->airy_prev || s->lists_force == 1);
        if (d->drops_step) {
            if (!s->drop_cells) {
                HIPCHK(hipMalloc((void**)&s->drop_ctr, (size_t)(64 * DROP_NCTR + DROP_NCTR + 1) * sizeof(int)));
                HIPCHK(hipMalloc((void**)&s->drop_n, (size_t)DROP_CAP * sizeof(int)));
                HIPCHK(hipMalloc((void**)&s->drop_cells, (size_t)DROP_CAP * 64 * sizeof(int)));
            }
            HIPCHK(hipMemsetAsync(&s->ss->n_drop, 0, sizeof(int), s->st));
            if (!ib_empty(o)) launch_drop_find(s->st, s->L, s->cntL, s->drop_ctr, s->drop_ctr + 64 * DROP_NCTR, &s->ss->n_drop, s->drop_n, s->drop_cells, &ownL);
            HIPCHK(hipGetLastError());
            if ((rc = halo_exchange1(s, d->lv[0].plan, 1, s->cntL))) return rc;
            if ((rc = read_ss(s))) return rc;
            s->n_drop = ib_empty(o) ? 0 : std::min(s->h_ss->n_drop, DROP_CAP);
            if (s->n_drop > 0) s->stats.paths |= FLUID_PATH_DROPLETS;
        }
        launch_cnt_pcg(s->st, s->L, ownL, s->cntL, d->cnt_pcg);
        // Mostly-air box (global numbers of the previous step; or pinned): the level-0 legs and the w = A z sweep of this rank run over the
        // lists of their tiles that hold an unknown, like the one-GPU solve's; the down leg's two lists (tiles that read no received cell /
        // tiles that do: the residual's halo exchange flies behind the first) are cut from the active tiles.  Rank-local: every launch
        // shape and partial count that follows from them stays inside the rank.
        s->lists_on = false;
        const bool want_lists = use_mg(s) && !ib_empty(o) && s->lists_force != 0 && (d->airy_prev || s->lists_force == 1);
        if (want_lists) {
            const MLevel m0 = d->lv[0].m;
            const int n_mg = mg_up_blocks(m0), n_sq = sq_tile_count(s->L);
            const size_t need = (size_t)3 * n_mg + n_sq;
            if (need > s->tl_cap) {
                HIPCHK(hipStreamSynchronize(s->st));
                hipFree(s->tl_flags); hipFree(s->tl_mg); hipFree(s->tl_sq);
                s->tl_flags = nullptr; s->tl_mg = s->tl_sq = nullptr;
                const size_t cap = need + need / 4;
                HIPCHK(hipMalloc((void**)&s->tl_flags, cap));
                HIPCHK(hipMalloc((void**)&s->tl_mg, cap * sizeof(int)));
                HIPCHK(hipMalloc((void**)&s->tl_sq, cap * sizeof(int)));
                s->tl_cap = cap;
            }
            const bool split = d->n_int + d->n_bnd == n_mg && d->tl_cls;   // the overlap's two lists exist for this step
            uint8_t *act = s->tl_flags, *fi = act + n_mg, *fb = fi + n_mg, *fsq = fb + n_mg;
            launch_mg_tile_flags(s->st, m0, s->cntL, act);
            launch_compact_flags(s->st, act, n_mg, s->tl_mg, &s->ss->n_tl_mg);
            if (split) {
                HIPCHK(hipMemcpyAsync(d->tl_cls, d->h_cls.data(), (size_t)n_mg, hipMemcpyHostToDevice, s->st));
                launch_split_flags(s->st, act, d->tl_cls, n_mg, fi, fb);
                launch_compact_flags(s->st, fi, n_mg, d->tl_int, &s->ss->n_tl_int);
                launch_compact_flags(s->st, fb, n_mg, d->tl_bnd, &s->ss->n_tl_bnd);
            }
            launch_sq_tile_flags(s->st, s->L, d->cnt_pcg, fsq);
            launch_compact_flags(s->st, fsq, n_sq, s->tl_sq, &s->ss->n_tl_sq);
            HIPCHK(hipGetLastError());
            if ((rc = read_ss(s))) return rc;
            s->n_tl_mg = s->h_ss->n_tl_mg; s->n_tl_sq = s->h_ss->n_tl_sq;
            if (split) { d->n_int = s->h_ss->n_tl_int; d->n_bnd = s->h_ss->n_tl_bnd; }
            s->lists_on = s->n_tl_mg > 0 && s->n_tl_sq > 0;
            if (s->lists_on) s->stats.paths |= FLUID_PATH_TILE_LISTS;
            else if (split) { d->n_int = d->n_bnd = 0; }   // (nothing to sweep on this rank)
        }
    }
    if (!ib_empty(d->lv[0].dom))
        launch_mg_type_local(s->st, g, d->lv[0].m, d->lv[0].dom.lo[0] - g.ox, d->lv[0].dom.lo[1] - g.oy, d->lv[0].dom.lo[2] - g.oz, s->flags, s->cntL,
                             d->lv[0].typ);
    HIPCHK(hipGetLastError());
    for (int l = 1; l < nl; ++l) {
        if (l < split) {
            // held on block + halo: types of the children's cells, halos from the neighbours, then the counts
            DLevel &F = d->lv[l - 1], &C = d->lv[l];
            if (!ib_empty(F.dom)) {
                const int off[3] = {(F.dom.lo[0] >> 1) - C.dom.lo[0], (F.dom.lo[1] >> 1) - C.dom.lo[1], (F.dom.lo[2] >> 1) - C.dom.lo[2]};
                launch_mg_coarsen_types(s->st, F.m, F.typ, coarse_view(C.m, off), C.typ);
            }
            if ((rc = halo_exchange1(s, C.plan, 1, C.typ))) return rc;
            launch_mg_counts(s->st, C.m, C.typ, C.cnt);
            if ((rc = halo_exchange1(s, C.plan, 1, C.cnt))) return rc;
        } else if (l == split) {
            // gathered: every rank types the coarse cells under its own fine cells, one all-reduce assembles the level
            DLevel& F = d->lv[l - 1];
            const GDom& G = gd[l];
            if (!ib_empty(F.dom)) {
                const int off[3] = {(F.dom.lo[0] >> 1) - G.lo[0], (F.dom.lo[1] >> 1) - G.lo[1], (F.dom.lo[2] >> 1) - G.lo[2]};
                launch_mg_coarsen_types(s->st, F.m, F.typ, coarse_view(s->mgl[l], off), s->mg_typ[l]);
            }
            HIPCHK(hipGetLastError());
            if (d->split_exchange) {
                if ((rc = halo_exchange1(s, d->plan_split, 1, s->mg_typ[l]))) return rc;
            } else {
                launch_mask_outside<uint8_t>(s->st, s->mgl[l], to_box(own_l(l, me), G.lo), s->mg_typ[l]);
                HIPCHK(hipGetLastError());
                if ((rc = comm_allreduce(s, s->mg_typ[l], (long)s->mgl[l].cells, FLUID_DT_U8, FLUID_OP_SUM))) return rc;
            }
            launch_mg_counts(s->st, s->mgl[l], s->mg_typ[l], s->mg_cnt[l]);
        } else {
            launch_mg_coarsen(s->st, s->mgl[l - 1], s->mg_typ[l - 1], s->mgl[l], s->mg_typ[l], s->mg_cnt[l]);
        }
        HIPCHK(hipGetLastError());
    }
    // ---- the Galerkin levels' coefficients ------------------------------------------------------------------------------------
    // With split = 1 every level below level 0 is replicated, so only level 1's coefficients need the blocks: each rank forms
    // them for the coarse cells under its owned fine cells (the children and their face neighbours lie inside its halo: cuts are
    // multiples of 4) and the owners' values are gathered like the level's right-hand side, once per step; the levels below are
    // coarsened identically everywhere.
    if (want_gal && split == 1) {
        const int lc = gal_lc;
        size_t total_g = 0, off[fluid_sim::MG_MAXL][6];
        for (int l = 1; l <= lc; ++l)
            for (int q = 0; q < 6; ++q) {
                off[l][q] = total_g;
                total_g += (q < 5 ? ((q < 4 ? sizeof(float) : 1) * (s->mgl[l].cells + 64) + 255) / 256 * 256 : ((size_t)gal_tile_count(s->mgl[l]) + 255) / 256 * 256);
            }
        if (total_g > s->gal_slab_cap) {
            if (s->gal_slab) { HIPCHK(hipStreamSynchronize(s->st)); hipFree(s->gal_slab); s->gal_slab = nullptr; }
            HIPCHK(hipMalloc((void**)&s->gal_slab, total_g + total_g / 4));
            s->gal_slab_cap = total_g + total_g / 4;
        }
        HIPCHK(hipMemsetAsync(s->gal_slab, 0, total_g, s->st));
        for (int l = 1; l <= lc; ++l) {
            for (int q = 0; q < 4; ++q) s->gal_c[l][q] = (float*)(s->gal_slab + off[l][q]);
            s->gal_cnt[l] = (uint8_t*)(s->gal_slab + off[l][4]);
            s->gal_tfl[l] = (uint8_t*)(s->gal_slab + off[l][5]);
        }
        DLevel& F0 = d->lv[0];
        if (!ib_empty(F0.dom)) {
            const int off1[3] = {(F0.dom.lo[0] >> 1) - gd[1].lo[0], (F0.dom.lo[1] >> 1) - gd[1].lo[1], (F0.dom.lo[2] >> 1) - gd[1].lo[2]};
            launch_gal_level1(s->st, F0.m, s->cntL, coef_as<float>(s, 0), coarse_view(s->mgl[1], off1), s->gal_c[1][0], s->gal_c[1][1], s->gal_c[1][2], s->gal_c[1][3],
                              s->gal_cnt[1]);
            HIPCHK(hipGetLastError());
        }
        for (int q = 0; q < 5; ++q) {   // the owners' coarse cells to everyone
            void* arr = q < 4 ? (void*)s->gal_c[1][q] : (void*)s->gal_cnt[1];
            if (d->split_exchange) {
                if ((rc = halo_exchange1(s, d->plan_split, q < 4 ? 4 : 1, arr))) return rc;
            } else {
                if (q < 4) launch_mask_outside<float>(s->st, s->mgl[1], to_box(own_l(1, me), gd[1].lo), (float*)arr);
                else launch_mask_outside<uint8_t>(s->st, s->mgl[1], to_box(own_l(1, me), gd[1].lo), (uint8_t*)arr);
                HIPCHK(hipGetLastError());
                if ((rc = comm_allreduce(s, arr, (long)s->mgl[1].cells, q < 4 ? FLUID_DT_F32 : FLUID_DT_U8, FLUID_OP_SUM))) return rc;
            }
        }
        for (int l = 2; l <= lc; ++l)
            launch_gal_coarsen(s->st, s->mgl[l - 1], s->gal_c[l - 1][0], s->gal_c[l - 1][1], s->gal_c[l - 1][2], s->gal_c[l - 1][3], s->gal_cnt[l - 1], s->mgl[l],
                               s->gal_c[l][0], s->gal_c[l][1], s->gal_c[l][2], s->gal_c[l][3], s->gal_cnt[l]);
        for (int l = 1; l < lc; ++l) launch_gal_tile_flags(s->st, s->mgl[l], s->gal_cnt[l], s->gal_tfl[l]);
        HIPCHK(hipGetLastError());
        s->gal = true;
        s->gal_lc = lc;
        s->gal_dt = s->dt;
    }
    return FLUID_OK;
}

template <typename V>
MgCoef<V> coef_as(const fluid_sim* s, int level)
{
    const MgCoef<double> c = mg_coef(s, level);
    MgCoef<V> o;
    for (int k = 0; k < 7; ++k) { o.diag[k] = (V)c.diag[k]; o.inv[k] = (V)c.inv[k]; }
    o.off = (V)c.off;
    return o;
}

// z0 = M^-1 rhs0: the V(2,2) cycle of kernels_mg.hip over the decomposed hierarchy.  rhs0 (the PCG residual) must be
// valid on the whole local level-0 domain (its halo exchanged by the caller); z0 is valid on the owned cells + 1 ring.
// zf: z0 is a float array (the float cycle's result as it is: the Chronopoulos-Gear loop's kernels convert)
template <typename V>
int dist_vcycle_t(fluid_sim* s, const double* rhs0, void* z0v, double* part_rz, bool halo_pending, bool zf = false)
{
    double* const z0 = (double*)z0v;
    DistState* d = s->ds;
    const int nl = s->mg_nl, tail = s->mg_tail, split = d->split;
    const PcgState* ps = s->ps;
    auto U = [&](int l) { return (V*)s->mg_u[l]; };
    auto W = [&](int l) { return (V*)s->mg_v[l]; };
    auto F = [&](int l) { return (V*)s->mg_f[l]; };
    auto R = [&](int l) { return (V*)s->mg_r[l]; };
    int rc;
    // the level under local level l as that level's kernels see it, and the global cell of its cell 0
    auto under = [&](int l) {
        const DLevel& Lf = d->lv[l];
        int off[3];
        if (l + 1 < split) for (int a = 0; a < 3; ++a) off[a] = (Lf.dom.lo[a] >> 1) - d->lv[l + 1].dom.lo[a];
        else for (int a = 0; a < 3; ++a) off[a] = (Lf.dom.lo[a] >> 1) - (d->dom0.lo[a] >> (l + 1));
        return coarse_view(l + 1 < split ? d->lv[l + 1].m : s->mgl[l + 1], off);
    };
    if (s->gal && sizeof(V) == 4) {
        // Galerkin coarse levels (split = 1): level 0 by the same legs with its own coefficients — the down leg also forms the coarse
        // right-hand side (the sum of its residual over each coarse cell's children: no halo), the up leg takes the parent's value —,
        // the gathered level 1 and everything below by the kernels of kernels_gal.hip, identically on every rank
        typedef float G;
        auto GU = [&](int l) { return (G*)s->mg_u[l]; };
        auto GW = [&](int l) { return (G*)s->mg_v[l]; };
        auto GF = [&](int l) { return (G*)s->mg_f[l]; };
        auto C = [&](int l, int q) { return (const float*)s->gal_c[l][q]; };
        const int lc = s->gal_lc;
        DLevel& L = d->lv[0];
        const MLevel mc = ib_empty(L.dom) ? s->mgl[1] : under(0);
        if (!ib_empty(L.dom)) {
            if (halo_pending) {
                if (d->n_int) launch_mg_down<G, double>(s->st, L.m, L.cnt, rhs0, GU(0), (G*)s->mg_r[0], mc, nullptr, GF(1), coef_as<G>(s, 0), ps, d->tl_int, d->n_int, true);
                if ((rc = halo_exchange_end(s, L.plan, sizeof(double)))) return rc;
                if (d->n_bnd) launch_mg_down<G, double>(s->st, L.m, L.cnt, rhs0, GU(0), (G*)s->mg_r[0], mc, nullptr, GF(1), coef_as<G>(s, 0), ps, d->tl_bnd, d->n_bnd, true);
            } else {
                launch_mg_down<G, double>(s->st, L.m, L.cnt, rhs0, GU(0), (G*)s->mg_r[0], mc, nullptr, GF(1), coef_as<G>(s, 0), ps, s->lists_on ? s->tl_mg : nullptr,
                                          s->n_tl_mg, true);
            }
            HIPCHK(hipGetLastError());
        } else if (halo_pending) {
            if ((rc = halo_exchange_end(s, L.plan, sizeof(double)))) return rc;
        }
        if (d->split_exchange) {
            if ((rc = halo_exchange1(s, d->plan_split, sizeof(G), GF(1)))) return rc;
        } else {
            int lo[3];
            for (int a = 0; a < 3; ++a) lo[a] = d->dom0.lo[a] >> 1;
            IBox gb;
            for (int a = 0; a < 3; ++a) { gb.lo[a] = lo[a]; gb.hi[a] = lo[a] + (a == 0 ? s->mgl[1].dx : (a == 1 ? s->mgl[1].dy : s->mgl[1].dz)); }
            const IBox o = ib_isect(block_level(d, d->comm.rank, 1, s->g.N), gb);
            launch_mask_outside<G>(s->st, s->mgl[1], to_box(o, lo), GF(1));
            HIPCHK(hipGetLastError());
            if ((rc = comm_allreduce(s, GF(1), (long)s->mgl[1].cells, FLUID_DT_F32, FLUID_OP_SUM))) return rc;
        }
        for (int l = 1; l < lc; ++l) launch_gal_down(s->st, s->mgl[l], s->gal_tfl[l], C(l, 0), C(l, 1), C(l, 2), C(l, 3), GF(l), GU(l), s->mgl[l + 1], GF(l + 1), ps);
        launch_gal_coarsest(s->st, s->mgl[lc], C(lc, 0), C(lc, 1), C(lc, 2), C(lc, 3), GF(lc), GU(lc), s->gal_sweeps, ps);
        for (int l = lc - 1; l >= 1; --l)
            launch_gal_up(s->st, s->mgl[l], s->gal_tfl[l], C(l, 0), C(l, 1), C(l, 2), C(l, 3), GF(l), GU(l), GW(l), s->mgl[l + 1], l + 1 == lc ? GU(l + 1) : GW(l + 1),
                          (float)s->gal_wc, ps);
        if (!ib_empty(L.dom)) {
            const int tok = prof_begin(s, FLUID_PROF_MG_UP0, (double)ib_cells(L.dom));
            if (zf)
                launch_mg_up<G, double, float>(s->st, L.m, L.cnt, rhs0, (const G*)GU(0), (float*)z0v, mc, GW(1), coef_as<G>(s, 0), part_rz, ps, s->gal_wc,
                                               s->lists_on ? s->tl_mg : nullptr, s->n_tl_mg, d->cnt_pcg, 1);
            else
                launch_mg_up<G, double, double>(s->st, L.m, L.cnt, rhs0, (const G*)GU(0), z0, mc, GW(1), coef_as<G>(s, 0), part_rz, ps, s->gal_wc,
                                                s->lists_on ? s->tl_mg : nullptr, s->n_tl_mg, d->cnt_pcg, 1);
            prof_end(s, FLUID_PROF_MG_UP0, tok);
        }
        HIPCHK(hipGetLastError());
        return FLUID_OK;
    }
    // ---- down: levels on block + halo ----
    for (int l = 0; l < split; ++l) {
        DLevel& L = d->lv[l];
        const MLevel mc = ib_empty(L.dom) ? s->mgl[l + 1] : under(l);
        const uint8_t* cc = l + 1 < split ? d->lv[l + 1].cnt : s->mg_cnt[l + 1];
        if (!ib_empty(L.dom)) {
            if (l == 0) {
                if (halo_pending) {
                    // interior tiles while the residual's halo is in flight on the second stream, the rest once it has landed
                    if (d->n_int) launch_mg_down<V, double>(s->st, L.m, L.cnt, rhs0, U(0), R(0), mc, nullptr, nullptr, coef_as<V>(s, 0), ps, d->tl_int, d->n_int);
                    if ((rc = halo_exchange_end(s, L.plan, sizeof(double)))) return rc;
                    if (d->n_bnd) launch_mg_down<V, double>(s->st, L.m, L.cnt, rhs0, U(0), R(0), mc, nullptr, nullptr, coef_as<V>(s, 0), ps, d->tl_bnd, d->n_bnd);
                } else {
                    launch_mg_down<V, double>(s->st, L.m, L.cnt, rhs0, U(0), R(0), mc, nullptr, nullptr, coef_as<V>(s, 0), ps, s->lists_on ? s->tl_mg : nullptr, s->n_tl_mg);
                }
                launch_mg_restrict<V>(s->st, L.m, (const V*)R(0), mc, cc, F(1), ps);
            } else {
                launch_mg_down<V, V>(s->st, L.m, L.cnt, (const V*)F(l), U(l), R(l), mc, cc, F(l + 1), coef_as<V>(s, l), ps);
            }
            HIPCHK(hipGetLastError());
        }
        if (l + 1 < split) {
            if ((rc = halo_exchange1(s, d->lv[l + 1].plan, sizeof(V), F(l + 1)))) return rc;
        } else {
            // gather of the first replicated level: the owners' coarse cells to everyone (block exchange), or my coarse
            // cells, zeros elsewhere, and one SUM all-reduce
            if (d->split_exchange) {
                if ((rc = halo_exchange1(s, d->plan_split, sizeof(V), F(l + 1)))) return rc;
            } else {
                int lo[3];
                for (int a = 0; a < 3; ++a) lo[a] = d->dom0.lo[a] >> (l + 1);
                IBox gb;
                for (int a = 0; a < 3; ++a) { gb.lo[a] = lo[a]; gb.hi[a] = lo[a] + (a == 0 ? s->mgl[l + 1].dx : (a == 1 ? s->mgl[l + 1].dy : s->mgl[l + 1].dz)); }
                const IBox o = ib_isect(block_level(d, d->comm.rank, l + 1, s->g.N), gb);
                launch_mask_outside<V>(s->st, s->mgl[l + 1], to_box(o, lo), F(l + 1));
                HIPCHK(hipGetLastError());
                if ((rc = comm_allreduce(s, F(l + 1), (long)s->mgl[l + 1].cells, sizeof(V) == 4 ? FLUID_DT_F32 : FLUID_DT_F64, FLUID_OP_SUM))) return rc;
            }
        }
    }
    // ---- replicated levels: down, tail, up (identical on every rank) ----
    for (int l = split; l < tail; ++l)
        launch_mg_down<V, V>(s->st, s->mgl[l], s->mg_cnt[l], (const V*)F(l), U(l), R(l), s->mgl[l + 1], s->mg_cnt[l + 1], F(l + 1), coef_as<V>(s, l), ps);
    {
        V off[fluid_sim::MG_MAXL];
        for (int l = tail; l < nl; ++l) off[l] = (V)mg_coef(s, l).off;
        launch_mg_tail<V>(s->st, nl - tail, (const V*)F(tail), s->mgl + tail, s->mg_cnt + tail, U(tail), off + tail, s->mg_csweeps, ps, s->mg_wc[3]);
    }
    for (int l = tail - 1; l >= split; --l) {
        const V* ec = l + 1 == tail ? U(l + 1) : W(l + 1);
        launch_mg_up<V, V, V>(s->st, s->mgl[l], s->mg_cnt[l], (const V*)F(l), (const V*)U(l), W(l), s->mgl[l + 1], ec, coef_as<V>(s, l), nullptr, ps,
                              s->mg_wc[l == 1 ? 1 : 2]);
    }
    HIPCHK(hipGetLastError());
    // ---- up: levels on block + halo ----
    for (int l = split - 1; l >= 0; --l) {
        DLevel& L = d->lv[l];
        const V* ec = l + 1 == tail ? U(l + 1) : W(l + 1);
        if (!ib_empty(L.dom)) {
            const MLevel mc = under(l);
            if (l == 0) {
                const int tok = prof_begin(s, FLUID_PROF_MG_UP0, (double)ib_cells(L.dom));
                if constexpr (std::is_same<V, float>::value) {
                    if (zf)
                        launch_mg_up<float, double, float>(s->st, L.m, L.cnt, rhs0, (const float*)U(0), (float*)z0v, mc, ec, coef_as<float>(s, 0), part_rz, ps, s->mg_wc[0],
                                                           s->lists_on ? s->tl_mg : nullptr, s->n_tl_mg, d->cnt_pcg);
                }
                if (!zf)
                    launch_mg_up<V, double, double>(s->st, L.m, L.cnt, rhs0, (const V*)U(0), z0, mc, ec, coef_as<V>(s, 0), part_rz, ps, s->mg_wc[0],
                                                    s->lists_on ? s->tl_mg : nullptr, s->n_tl_mg, d->cnt_pcg);
                prof_end(s, FLUID_PROF_MG_UP0, tok);
            } else {
                launch_mg_up<V, V, V>(s->st, L.m, L.cnt, (const V*)F(l), (const V*)U(l), W(l), mc, ec, coef_as<V>(s, l), nullptr, ps, s->mg_wc[l == 1 ? 1 : 2]);
            }
            HIPCHK(hipGetLastError());
        }
        if (l > 0 && (rc = halo_exchange1(s, L.plan, sizeof(V), W(l)))) return rc;   // the correction the level above interpolates from
    }
    return FLUID_OK;
}

// PCG of ConjugateGradient.h:28-90 on the owned unknowns; z = the decomposed V-cycle (or Eigen's diagonal preconditioner).
// Per iteration: one halo exchange of r, the V-cycle's exchanges / gather, and two small all-reduces ({|r|^2, r.z}, s.q).
int dist_solve(fluid_sim* s)
{
    typedef double T;
    DistState* d = s->ds;
    const Grid g = s->g;
    const LBox L = s->L;
    T *X = (T*)s->X, *R = (T*)s->R, *Q = (T*)s->Q, *Z = (T*)s->Zmg;
    T* Sx[2] = {(T*)s->S[0], (T*)s->S[1]};
    const uint8_t* cnt = d->cnt_pcg;
    const Coef<T> cf = make_coef<T>(s);
    const double tol = s->prm.cg_tol;
    const bool mg = use_mg(s);
    long max_it = s->prm.cg_max_iters > 0 ? s->prm.cg_max_iters : 2 * (long)s->stats.num_active;
    if (max_it < 1) max_it = 1;
    const bool lists = mg && s->lists_on;
    const int nxr = pcg_xr_blocks(L), nsq = lists && d->cg_form == 1 ? pcg_list_blocks(s->n_tl_sq) : pcg_sq_blocks(L);
    const int n_rz = ib_empty(d->lv[0].dom) ? 0 : (lists ? s->n_tl_mg : mg_up_blocks(d->lv[0].m));
    const double cells = (double)ib_cells(d->lv[0].own);
    int rc;
    int tsolve = prof_begin(s, FLUID_PROF_SOLVE, cells);
    // start: x0 = 0, or (multigrid path, like the one-GPU solve) the previous pressure — every rank's owned values, its ring from
    // the owners (the pressure halo exchange carries the guess too), so that r0 = b - A x0 is the same vector on both sides of a cut
    const bool guess = mg && s->warm && s->have_guess;
    const double *g1, *g2;
    double gca, gcb;
    s->start_guess(g1, g2, gca, gcb);
    if (guess) launch_pcg_init_guess<T>(s->st, g, L, cnt, s->diver, g1, g2, gca, gcb, X, R, cf, s->part_bb, s->part_rz[1], s->ps);
    else launch_pcg_init<T>(s->st, g, L, cnt, s->diver, X, R, cf, s->part_bb, s->part_rz[0], s->ps);
    // Once per handle, with real peers: the residual's halo exchanged serially on the solver's stream and then again the overlapped way
    // (pack on the solver's stream, transfer and unpack on the second) must deliver the same bytes on every rank — one communicator is
    // driven from two streams there.  If any rank sees a difference every rank switches the overlap off (MAX all-reduce of the verdict).
    if (mg && d->overlap && !d->overlap_checked && d->comm.size > 1) {
        HaloPlan& hp = d->lv[0].plan;
        const size_t nb = hp.n ? (size_t)hp.rtotal * sizeof(T) : 0;
        std::vector<unsigned char> a(nb), b(nb);
        if ((rc = halo_exchange1(s, hp, sizeof(T), R))) return rc;
        if (nb) HIPCHK(hipMemcpyAsync(a.data(), d->hr, nb, hipMemcpyDeviceToHost, s->st));
        if (nb) HIPCHK(hipMemsetAsync(d->hr, 0xFF, nb, s->st));
        if (hp.n && !ib_empty(d->lv[0].dom)) {
            if ((rc = halo_exchange_begin(s, hp, sizeof(T), R))) return rc;
            if ((rc = halo_exchange_end(s, hp, sizeof(T)))) return rc;
        } else if ((rc = halo_exchange1(s, hp, sizeof(T), R))) return rc;
        if (nb) HIPCHK(hipMemcpyAsync(b.data(), d->hr, nb, hipMemcpyDeviceToHost, s->st));
        HIPCHK(hipStreamSynchronize(s->st));
        int* flag = d->h_cnt + 97;
        *flag = nb && memcmp(a.data(), b.data(), nb) != 0 ? 1 : 0;
        HIPCHK(hipMemcpyAsync(d->d_cnt + 97, flag, sizeof(int), hipMemcpyHostToDevice, s->st));
        if ((rc = comm_allreduce(s, d->d_cnt + 97, 1, FLUID_DT_I32, FLUID_OP_MAX))) return rc;
        HIPCHK(hipMemcpyAsync(flag, d->d_cnt + 97, sizeof(int), hipMemcpyDeviceToHost, s->st));
        HIPCHK(hipStreamSynchronize(s->st));
        d->overlap_checked = *flag ? 2 : 1;
        if (*flag) d->overlap = false;
    }
    long it = 0;
    const int pclass = s->pass_class();
    // Bodies launched before the first look at the device: what the class's previous solve needed (identical on every rank).  Eigen's count i
    // means i + 1 bodies ran; the Chronopoulos-Gear loop polls with a test of its own (k_pcg_poll_stage) and needs no further body to
    // notice the end, the standard loop notices it at the head of the next body.
    const bool cgear_loop = mg && d->cg_form == 1;
    long batch = s->mg_last_iters_k[pclass] > 5 ? s->mg_last_iters_k[pclass] + (cgear_loop ? 1 : 0) : 8;
    int polls = 0;
    bool done = false;
    while (!done) {
        for (long k = 0; k < batch && it < max_it; ++k, ++it) {
            const int cur = (int)(it & 1), prv = cur ^ 1;
            // the residual's halo: exchanged on the second stream behind the interior tiles of the level-0 down leg (every rank takes
            // the same branch: the switch is a parameter of the run, and a rank without a level-0 domain has an empty plan)
            const bool ovl = mg && d->overlap && d->lv[0].plan.n > 0 && !ib_empty(d->lv[0].dom);
            const bool cgear = mg && d->cg_form == 1;
            if (ovl) rc = halo_exchange_begin(s, d->lv[0].plan, sizeof(T), R);
            else rc = halo_exchange1(s, d->lv[0].plan, sizeof(T), R);
            if (rc) return rc;
            if (cgear) {
                // Chronopoulos-Gear: z = M^-1 r, w = A z, then ONE all-reduce of {|r|^2 of the previous body (|b|^2 before the first),
                // gamma = r.z, delta = w.z (, |r0|^2 of a solve started from a guess)} and one fused update of s, q = A s, x, r
                const bool zf = s->mg_fp32 && sizeof(T) == 8;   // z stays float in memory
                if (s->mg_fp32) rc = dist_vcycle_t<float>(s, R, Z, s->mg_part, ovl, zf);
                else rc = dist_vcycle_t<double>(s, R, Z, s->mg_part, ovl);
                if (rc) return rc;
                T* Wv = Sx[1];
                int tok = prof_begin(s, FLUID_PROF_PCG_SQ, cells);
                if (zf) launch_pcg_az_dist_zf(s->st, L, cnt, (const float*)Z, (double*)Wv, make_coef<double>(s), s->part_pq, s->ps, lists ? s->tl_sq : nullptr, s->n_tl_sq);
                else launch_pcg_az_dist<T>(s->st, L, cnt, Z, Wv, cf, s->part_pq, s->ps, lists ? s->tl_sq : nullptr, s->n_tl_sq);
                prof_end(s, FLUID_PROF_PCG_SQ, tok);
                const bool g0 = it == 0 && guess;
                launch_sum4(s->st, it == 0 ? s->part_bb : s->part_rr, nxr, s->mg_part, n_rz, s->part_pq, nsq, s->part_rz[1], g0 ? nxr : 0, d->gstage[cur]);
                if ((rc = comm_allreduce(s, d->gstage[cur], g0 ? 4 : 3, FLUID_DT_F64, FLUID_OP_SUM))) return rc;
                tok = prof_begin(s, FLUID_PROF_PCG_XR, cells);
                if (zf)
                    launch_pcg_cgear_upd_zf(s->st, L, cnt, (double*)X, (double*)R, (double*)Sx[0], (double*)Q, (const float*)Z, (const double*)Wv, d->gstage[cur], d->gcg, cur,
                                            s->part_rr, s->ps, it == 0 ? (guess ? 2 : 1) : 0, tol);
                else
                    launch_pcg_cgear_upd<T>(s->st, L, cnt, X, R, Sx[0], Q, Z, Wv, d->gstage[cur], d->gcg, cur, s->part_rr, s->ps, it == 0 ? (guess ? 2 : 1) : 0, tol);
                prof_end(s, FLUID_PROF_PCG_XR, tok);
                continue;
            }
            // gstage[cur] = {|r|^2 of the previous body (|b|^2 before the first), r.z of this one}
            if (mg) {
                if (s->mg_fp32) rc = dist_vcycle_t<float>(s, R, Z, s->mg_part, ovl);
                else rc = dist_vcycle_t<double>(s, R, Z, s->mg_part, ovl);
                if (rc) return rc;
                launch_sum2(s->st, it == 0 ? s->part_bb : s->part_rr, nxr, s->mg_part, n_rz, d->gstage[cur], d->gstage[cur] + 1);
            } else {
                launch_sum2(s->st, it == 0 ? s->part_bb : s->part_rr, nxr, s->part_rz[0], nxr, d->gstage[cur], d->gstage[cur] + 1);
            }
            const bool g0 = it == 0 && guess;   // the first body of a warm-started solve also needs |r0|^2 (ConjugateGradient.h:51-56)
            if (g0) launch_sum2(s->st, s->part_rz[1], nxr, s->part_rz[1], 0, d->gstage[cur] + 2, nullptr);
            if ((rc = comm_allreduce(s, d->gstage[cur], g0 ? 3 : 2, FLUID_DT_F64, FLUID_OP_SUM))) return rc;
            int tok = prof_begin(s, FLUID_PROF_PCG_SQ, cells);
            launch_pcg_sq_dist<T>(s->st, L, cnt, mg ? Z : R, Sx[prv], Sx[cur], Q, cf, d->gstage[cur], d->gstage[cur] + 1, g0 ? d->gstage[cur] + 2 : d->gstage[prv] + 1,
                                  s->part_pq, s->ps, it == 0 ? (guess ? 2 : 1) : 0, tol, mg ? 1 : 0);
            prof_end(s, FLUID_PROF_PCG_SQ, tok);
            launch_sum2(s->st, s->part_pq, nsq, s->part_pq, 0, d->gpq, nullptr);
            if ((rc = comm_allreduce(s, d->gpq, 1, FLUID_DT_F64, FLUID_OP_SUM))) return rc;
            tok = prof_begin(s, FLUID_PROF_PCG_XR, cells);
            launch_pcg_xr_dist<T>(s->st, L, cnt, X, R, Sx[cur], Q, cf, d->gstage[cur] + 1, d->gpq, s->part_rr, s->part_rz[0], s->ps);
            prof_end(s, FLUID_PROF_PCG_XR, tok);
        }
        HIPCHK(hipGetLastError());
        if (cgear_loop) {
            launch_pcg_poll_stage(s->st, s->part_rr, nxr, s->ps, d->gpq);
            if ((rc = comm_allreduce(s, d->gpq, 2, FLUID_DT_F64, FLUID_OP_SUM))) return rc;
            launch_pcg_poll_test(s->st, d->gpq, s->ps);
        } else {
            // the break test of the last body sits at the head of the next SQ launch; every rank must leave at the same iteration
            if ((rc = comm_allreduce(s, &s->ps->done, 1, FLUID_DT_I32, FLUID_OP_MAX))) return rc;
        }
        HIPCHK(hipMemcpyAsync(s->h_ps, s->ps, sizeof(PcgState), hipMemcpyDeviceToHost, s->st));
        HIPCHK(hipStreamSynchronize(s->st));
        done = s->h_ps->done || it >= max_it;
        batch = cgear_loop && ++polls <= 2 ? 1 : 2;   // (a poll costs about what a body costs)
    }
    int iters = s->h_ps->iters;
    const double rr = s->h_ps->rr;
    if (!s->h_ps->done) iters = (int)max_it;
    launch_store_pressure<T>(s->st, g, L, cnt, X, s->pressure, mg && s->warm ? s->p_guess2 : nullptr, s->ps);
    launch_drop_solve(s->st, g, L, s->n_drop, s->drop_ctr + 64 * DROP_NCTR, s->drop_n, s->drop_cells, s->flags, s->diver, make_coef<double>(s), tol, s->pressure,
                      mg && s->warm ? s->p_guess2 : nullptr, &s->ss->n_drop_fail);
    if (mg && s->warm) s->rotate_guess();   // (the pressure halo exchange that follows carries the new p_guess)
    s->have_guess = mg && s->warm;
    HIPCHK(hipGetLastError());
    prof_end(s, FLUID_PROF_SOLVE, tsolve);
    s->stats.cg_iters_last = iters;
    s->stats.cg_iters += iters;
    s->mg_last_iters_k[pclass] = iters;
    if (s->gal) s->stats.paths |= FLUID_PATH_MG_GALERKIN;
    if (s->gal_eligible && s->stats.outer_passes == 0) s->gal_it[s->gal ? 1 : 0] = iters;   // the first pass of the step: what the two cycles are compared by
    s->stats.relres = s->h_ps->bb > 0 ? std::sqrt(rr / s->h_ps->bb) : 0.0;
    if (s->h_ps->breakdown) return fail(FLUID_ERR_SOLVER, "PCG breakdown: s.As <= 0 or NaN");
    return FLUID_OK;
}

// ---- decomposed step ----------------------------------------------------------------------------------------------------
int dist_step_decomposed(fluid_sim* s, fluid_step_stats_t* stats)
{
    DistState* d = s->ds;
    const Grid g = s->g;
    const int N = g.N, me = d->comm.rank;
    int rc;
    if ((rc = dist_particles(s))) return rc;           // fluid.cc:1378-1413, first half: who holds what
    if ((rc = clear_dirty(s))) return rc;
    stats_begin(s);
    s->step_counter++;
    const int org[3] = {g.ox, g.oy, g.oz};
    const bool any = !ib_empty(d->Rg);
    if (any) s->dirty = s->Sb;
    if (any && !box_empty(d->Rr)) {
        int tok = prof_begin(s, FLUID_PROF_P2G, (double)d->Rr.cells());
        rc = run_p2g(s, d->Rr);
        prof_end(s, FLUID_PROF_P2G, tok);
        if (rc) return rc;
        HIPCHK(hipGetLastError());
    }
    // 1-wide halos of this step's fields: owned cells of the box the fields live in
    make_plan(d, d->plan_f1, [&](int r) { return ib_isect(block_of(d, r), d->Sg); }, 1, org, 0, g.sx(), g.nz);
    {
        void* a[6] = {s->u, s->v, s->w, s->ub, s->vb, s->wb};   // velBeforeUpdate = the same values (fluid.cc:1455)
        if ((rc = halo_exchange(s, d->plan_f1, 8, 6, a))) return rc;
    }
    // ---- flags on the owned block, halo from the owners; the reference's unknown numbering (:1416-1433) ----
    const Box ownW = to_box(IBox{{d->ob.lo[0], d->ob.lo[1], d->ob.lo[2]}, {d->ob.hi[0], d->ob.hi[1], d->ob.hi[2]}}, org);
    launch_flags_box(s->st, g, ownW, s->solid, s->container, s->flags);
    HIPCHK(hipGetLastError());
    if ((rc = halo_exchange1(s, d->plan_flags, 1, s->flags))) return rc;
    {
        launch_fill_box_int(s->st, g, d->idx_box, s->indices, -1);
        d->idx_box = d->Rr;
        const int RX = any ? d->Rg.hi[0] - d->Rg.lo[0] : 0, RY = any ? d->Rg.hi[1] - d->Rg.lo[1] : 0, nseg = d->dims[2];
        const size_t nrow = (size_t)RX * RY * nseg;
        if (nrow + 4 > d->rows_cap) {
            HIPCHK(hipStreamSynchronize(s->st));
            hipFree(d->rows); hipFree(d->row_starts);
            d->rows = d->row_starts = nullptr;
            d->rows_cap = nrow + nrow / 4 + 1024;
            HIPCHK(hipMalloc((void**)&d->rows, d->rows_cap * sizeof(int)));
            HIPCHK(hipMalloc((void**)&d->row_starts, d->rows_cap * sizeof(int)));
        }
        if (nrow) {
            HIPCHK(hipMemsetAsync(d->rows, 0, nrow * sizeof(int), s->st));
            launch_row_counts(s->st, g, d->Rr, d->Rg.lo[0], d->Rg.lo[1], RY, nseg, d->bc[2], s->flags, d->rows);
            HIPCHK(hipGetLastError());
            if ((rc = comm_allreduce(s, d->rows, (long)nrow, FLUID_DT_I32, FLUID_OP_SUM))) return rc;
            launch_exclusive_scan(s->st, d->rows, d->row_starts, (long)nrow, s->scan_sums, &s->ss->num_active);
            launch_row_number(s->st, g, d->Rr, d->Rg.lo[0], d->Rg.lo[1], RY, nseg, d->bc[2], s->flags, d->row_starts, s->indices);
            HIPCHK(hipGetLastError());
        } else {
            HIPCHK(hipMemsetAsync(&s->ss->num_active, 0, sizeof(int), s->st));
        }
        if ((rc = read_ss(s))) return rc;
        s->stats.num_active = s->h_ss->num_active;
        s->last_num_active = s->stats.num_active;
    }
    // ---- local solver layout, multigrid hierarchy ----
    const bool airy_now = any && (double)s->stats.num_active < 0.45 * (double)ib_cells(d->Rg) && ib_cells(d->Rg) > 1500000;
    if (any) {
        if ((rc = dist_mg_setup(s))) return rc;
        const size_t lb = (s->L.cells() + 2 * (size_t)s->L.Lz) * solver_elem(s);
        HIPCHK(zero_search(s, lb));
    }
    d->airy_prev = airy_now;
    s->have_p2g = s->have_flags = true;
    // ---- pressure do..while (:1457-1484) ----
    double error = NAN;
    do {
        const double dt = s->dt;
        if (any) {
            if (!box_empty(d->Rr))
                launch_rhs_div(s->st, g, d->Rr, s->flags, s->u, s->v, s->w, s->rhs, s->diver, s->prm.dx, s->prm.gravity[0] * dt, s->prm.gravity[1] * dt,
                               s->prm.gravity[2] * dt);
            if ((rc = dist_solve(s))) return rc;
            {
                void* a[2] = {s->pressure, s->p_guess};   // (+ the next solve's starting guess: ring values from their owners)
                if ((rc = halo_exchange(s, d->plan_f1, 8, s->have_guess ? 2 : 1, a))) return rc;
            }
            if (s->make_q()) {   // (both guesses carry their owners' ring values: so does q, over the whole window)
                launch_axpby_box(s->st, g, Box{0, 0, 0, g.nx - 1, g.ny - 1, g.nz - 1}, 1.0, s->p_guess, -(1.0 - s->prm.update_frac), s->p_guess2, s->p_q);
                s->q_step = s->step_counter;
            }
            const double dtp = dt * s->prm.update_frac, k = dtp / (s->prm.rho * s->prm.dx);
            if (!box_empty(d->Sr))
                launch_vel_update(s->st, g, d->Sr, s->flags, s->pressure, s->u, s->v, s->w, k, s->prm.gravity[0] * dtp, s->prm.gravity[1] * dtp,
                                  s->prm.gravity[2] * dtp);
            {
                void* a[3] = {s->u, s->v, s->w};
                if ((rc = halo_exchange(s, d->plan_f1, 8, 3, a))) return rc;
            }
            if (!box_empty(d->Rr)) {
                launch_rhs_div(s->st, g, d->Rr, s->flags, s->u, s->v, s->w, s->rhs, s->diver2, s->prm.dx, s->prm.gravity[0] * dt, s->prm.gravity[1] * dt,
                               s->prm.gravity[2] * dt);
                launch_err_norm(s->st, g, d->Rr, s->flags, s->diver, s->diver2, s->part_err, s->ss);
            } else {
                HIPCHK(hipMemsetAsync(&s->ss->err_num, 0, 2 * sizeof(double), s->st));
            }
            HIPCHK(hipGetLastError());
            if ((rc = comm_allreduce(s, &s->ss->err_num, 2, FLUID_DT_F64, FLUID_OP_SUM))) return rc;
            if ((rc = read_ss(s))) return rc;
            if (s->h_ss->n_drop_fail > 0) s->stats.paths |= FLUID_PATH_DROPLETS_SHORT;
            error = std::sqrt(s->h_ss->err_num) / std::sqrt(s->h_ss->err_den);
        }
        s->stats.error = error;
        s->stats.outer_passes++;
        if (s->prm.max_outer_passes > 0 && s->stats.outer_passes >= s->prm.max_outer_passes) break;
    } while (error > s->prm.outer_tol);
    // ---- FLIP delta field on the owned cells, halo from the owners, gather + advect (:1490) ----
    if ((rc = pic_fields(s))) return rc;
    if (any && !box_empty(d->Rr))
        launch_flip_delta(s->st, g, d->Rr, s->u, s->v, s->w, s->ub, s->vb, s->wb, s->dcx, s->dcy, s->dcz, s->pcx, s->pcy, s->pcz);
    HIPCHK(hipGetLastError());
    {
        void* a[6] = {s->dcx, s->dcy, s->dcz, s->pcx, s->pcy, s->pcz};
        if ((rc = halo_exchange(s, d->plan_f1, 8, s->pcx ? 6 : 3, a))) return rc;
    }
    (void)N; (void)me;
    return dist_g2p_advect(s, stats);
}

}  // namespace

// Cut planes of one axis from the histogram of the particles' base cells: slabs of about equal count, interior cuts multiples
// of 4, every block >= 8 cells (fluid_decomp's rules).  The running count is taken from the histogram's prefix at the plane
// actually chosen (a plane moved by the rounding or the width rules used to leave the count where the search had stopped).
int cuts_from_hist(int n, const int64_t* hist, int P, int32_t* cuts)
{
    if (P < 1 || (P > 1 && n < 8 * P + 8)) return fail(FLUID_ERR_ARG, "too many blocks along an axis (each needs >= 8 cells)");
    std::vector<int64_t> prefix(n + 1, 0);
    for (int x = 0; x < n; ++x) prefix[x + 1] = prefix[x] + hist[x];
    const int64_t np = prefix[n];
    cuts[0] = 0;
    for (int r = 1; r < P; ++r) {
        const int64_t target = np * r / P;
        int x = cuts[r - 1];
        while (x < n && prefix[x + 1] <= target) ++x;
        int b = (x + 2) & ~3;                                          // nearest multiple of 4
        const int lo_b = cuts[r - 1] + 8, hi_b = ((n - 8 * (P - r)) & ~3);   // >= 8 cells for this block and for the ones above
        if (b < lo_b) b = (lo_b + 3) & ~3;
        if (b > hi_b) b = hi_b;
        cuts[r] = b;
    }
    cuts[P] = n;
    for (int r = 0; r < P; ++r)
        if (P > 1 && cuts[r + 1] - cuts[r] < 8) return fail(FLUID_ERR_ARG, "grid too small for this many blocks");
    return FLUID_OK;
}

namespace {

// Re-balancing ("load imbalance ... slab-by-fluid-count split").  The cut planes of a run are placed by the particle
// counts at upload; the water then falls and spreads, and in the settled pool of the drop scene the upper blocks hold nothing
// (profiles/r02: particles per block [148010, 157256, 0, 0, 159910, 170880, 0, 0]).  Between two steps nothing but the particles
// and dt is state (every field is rebuilt by the next step), so the planes can be moved by building the windows anew:
//   1. SUM all-reduce of the three axis histograms of the live particles' base cells + one count per rank;
//   2. if the fullest block holds more than `rb_ratio` x the mean and the histograms give other planes: every rank computes the
//      same new planes;
//   3. all-to-all of the particles by their new owner (one grouped exchange of counts, one of 56-byte records, staged through the
//      host — this runs once in many steps);
//   4. a handle for the new window is created beside the old one and receives the particles; whether EVERY rank got that far is
//      agreed (dist_agree) — if not, all of them drop the new handle and go on with the old planes — then the new handle's
//      contents take the place of the old one's.
// The next solve starts from x = 0 (the stored pressure belongs to the old windows); results do not depend on the planes
// beyond the tolerance of the solve (tests/test_gpu_dist.py).
int dist_rebalance(fluid_sim* s)
{
    DistState* d = s->ds;
    const int N = s->g.N, R = d->comm.size, me = d->comm.rank;
    int rc;
    // Everything that can fail on ONE rank before the first collective is brought to dist_agree first: a rank that returned
    // from here alone would leave its peers in the all-reduce below for ever.
    int64_t live = 0;
    std::vector<double> pos, vel;
    std::vector<uint32_t> ids;
    std::vector<int> h(3 * (size_t)N + R, 0);
    const int lo = s->g.lo;
    auto gather_local = [&]() -> int {
        live = fluid_download_particles_ids(s, nullptr, nullptr, nullptr);
        if (live < 0) return fail(FLUID_ERR_HIP, "rebalance: counting the live particles failed");
        pos.resize(3 * (size_t)live); vel.resize(3 * (size_t)live); ids.resize((size_t)live);
        if (live && fluid_download_particles_ids(s, pos.data(), vel.data(), ids.data()) != live) return fail(FLUID_ERR_HIP, "rebalance: download failed");
        // 1. histograms of the base cells (fluid.cc:267: round half away from zero) and the per-rank counts, summed over the ranks
        for (int64_t i = 0; i < live; ++i)
            for (int a = 0; a < 3; ++a) {
                long b = std::lround(pos[3 * i + a]) - lo;
                b = b < 0 ? 0 : (b > N - 1 ? N - 1 : b);
                h[(size_t)a * N + b]++;
            }
        h[3 * (size_t)N + me] = (int)live;
        HIPCHK(hipMemcpyAsync(d->rb_buf, h.data(), h.size() * sizeof(int), hipMemcpyHostToDevice, s->st));
        return FLUID_OK;
    };
    if ((rc = dist_agree(s, gather_local()))) return rc;
    if ((rc = comm_allreduce(s, d->rb_buf, (long)h.size(), FLUID_DT_I32, FLUID_OP_SUM))) return rc;
    HIPCHK(hipMemcpyAsync(h.data(), d->rb_buf, h.size() * sizeof(int), hipMemcpyDeviceToHost, s->st));
    HIPCHK(hipStreamSynchronize(s->st));
    // 2. the same decision on every rank
    int64_t total = 0, most = 0;
    for (int r = 0; r < R; ++r) { total += h[3 * (size_t)N + r]; most = std::max<int64_t>(most, h[3 * (size_t)N + r]); }
    if (total == 0 || (double)most * R <= d->rb_ratio * (double)total) return FLUID_OK;
    std::vector<int32_t> nc[3];
    bool same = true;
    for (int a = 0; a < 3; ++a) {
        std::vector<int64_t> hist(N);
        for (int x = 0; x < N; ++x) hist[x] = h[(size_t)a * N + x];
        nc[a].assign(d->dims[a] + 1, 0);
        if (cuts_from_hist(N, hist.data(), d->dims[a], nc[a].data())) return FLUID_OK;   // (cannot happen: the current planes satisfy the rules)
        for (int b = 0; b <= d->dims[a]; ++b) same = same && nc[a][b] == d->cuts[a][b];
    }
    if (same) return FLUID_OK;
    // 3. particles to their new owners
    auto owner = [&](const double* p) {
        int b[3];
        for (int a = 0; a < 3; ++a) {
            long c = std::lround(p[a]) - lo;
            c = c < 0 ? 0 : (c > N - 1 ? N - 1 : c);
            int k = 0;
            while (k + 1 < d->dims[a] && c >= nc[a][k + 1]) ++k;
            b[a] = k;
        }
        return rank_of(d, b[0], b[1], b[2]);
    };
    std::vector<int> dest((size_t)live), scount(R, 0), rcount(R, 0);
    for (int64_t i = 0; i < live; ++i) { dest[i] = owner(&pos[3 * i]); scount[dest[i]]++; }
    int np = 0, peer[FLUID_MAX_RANKS];
    const void* sb[FLUID_MAX_RANKS];
    void* rb[FLUID_MAX_RANKS];
    size_t sn[FLUID_MAX_RANKS], rn[FLUID_MAX_RANKS];
    for (int r = 0; r < R; ++r) if (r != me) peer[np++] = r;
    if (np) {
        HIPCHK(hipMemcpyAsync(d->rb_buf, scount.data(), R * sizeof(int), hipMemcpyHostToDevice, s->st));
        for (int k = 0; k < np; ++k) { sb[k] = d->rb_buf + peer[k]; rb[k] = d->rb_buf + FLUID_MAX_RANKS + peer[k]; sn[k] = rn[k] = sizeof(int); }
        COMMCHK(d->comm.exchange(d->comm.ctx, np, peer, sb, sn, rb, rn, (void*)s->st));
        HIPCHK(hipMemcpyAsync(rcount.data(), d->rb_buf + FLUID_MAX_RANKS, R * sizeof(int), hipMemcpyDeviceToHost, s->st));
        HIPCHK(hipStreamSynchronize(s->st));
    }
    rcount[me] = 0;
    std::vector<long> soff(R + 1, 0), roff(R + 1, 0);
    for (int r = 0; r < R; ++r) { soff[r + 1] = soff[r] + (r == me ? 0 : scount[r]); roff[r + 1] = roff[r] + rcount[r]; }
    const long stot = soff[R], rtot = roff[R];
    std::vector<double> srec(7 * (size_t)stot), rrec(7 * (size_t)rtot);
    {
        std::vector<long> cur(soff.begin(), soff.end() - 1);
        for (int64_t i = 0; i < live; ++i) {
            if (dest[i] == me) continue;
            double* q = &srec[7 * (size_t)cur[dest[i]]++];
            for (int a = 0; a < 3; ++a) { q[a] = pos[3 * i + a]; q[3 + a] = vel[3 * i + a]; }
            q[6] = (double)ids[i];
        }
    }
    auto room = [&]() -> int {
        if (stot > d->mig_cap || rtot > d->mig_cap) {
            HIPCHK(hipStreamSynchronize(s->st));
            hipFree(d->mig_s); hipFree(d->mig_r);
            d->mig_s = d->mig_r = nullptr;
            d->mig_cap = 0;
            const long cap = std::max(stot, rtot) * 5 / 4 + 4096;
            HIPCHK(hipMalloc((void**)&d->mig_s, (size_t)cap * 56));
            HIPCHK(hipMalloc((void**)&d->mig_r, (size_t)cap * 56));
            d->mig_cap = cap;
        }
        return FLUID_OK;
    };
    if ((rc = dist_agree(s, room()))) return rc;
    if (np) {
        if (stot) HIPCHK(hipMemcpyAsync(d->mig_s, srec.data(), srec.size() * sizeof(double), hipMemcpyHostToDevice, s->st));
        for (int k = 0; k < np; ++k) {
            const int r = peer[k];
            sb[k] = d->mig_s + 7 * (size_t)soff[r]; sn[k] = (size_t)scount[r] * 56;
            rb[k] = d->mig_r + 7 * (size_t)roff[r]; rn[k] = (size_t)rcount[r] * 56;
        }
        COMMCHK(d->comm.exchange(d->comm.ctx, np, peer, sb, sn, rb, rn, (void*)s->st));
        if (rtot) HIPCHK(hipMemcpyAsync(rrec.data(), d->mig_r, rrec.size() * sizeof(double), hipMemcpyDeviceToHost, s->st));
        HIPCHK(hipStreamSynchronize(s->st));
    }
    const long mine = scount[me] + rtot;
    std::vector<double> npos(3 * (size_t)mine), nvel(3 * (size_t)mine);
    std::vector<uint32_t> nids((size_t)mine);
    long k = 0;
    for (int64_t i = 0; i < live; ++i)
        if (dest[i] == me) {
            for (int a = 0; a < 3; ++a) { npos[3 * k + a] = pos[3 * i + a]; nvel[3 * k + a] = vel[3 * i + a]; }
            nids[k++] = ids[i];
        }
    for (long j = 0; j < rtot; ++j, ++k) {
        for (int a = 0; a < 3; ++a) { npos[3 * k + a] = rrec[7 * j + a]; nvel[3 * k + a] = rrec[7 * j + 3 + a]; }
        nids[k] = (uint32_t)rrec[7 * j + 6];
    }
    // 4. the new window beside the old one
    fluid_sim* t = nullptr;
    auto build = [&]() -> int {
        fluid_decomp_t dc;
        for (int a = 0; a < 3; ++a) { dc.dims[a] = d->dims[a]; dc.cuts[a] = nc[a].data(); }
        if (d->fail_rebuild_rank == me) return fail(FLUID_ERR_HIP, "rebalance: second window refused (FLUID_DIST_FAIL_REBUILD)");
        int r2 = fluid_create_dist(&s->prm, &d->comm, &dc, &t);
        if (r2) return r2;
        if (!d->solid_global.empty() && (r2 = fluid_set_solid(t, d->solid_global.data()))) return r2;
        return fluid_upload_particles_ids(t, mine, npos.data(), nvel.data(), nids.data());
    };
    if ((rc = dist_agree(s, build()))) {
        // Some rank could not build its second window (normally: no room for it).  The outcome is decided by the AGREED flag alone,
        // so every rank takes the same way out: drop the new handle, forget the failed allocation (a runtime that keeps the last
        // error set until it is read would hand it to the next step's first check on this rank only) and go on with the old planes.
        if (t) fluid_destroy(t);
        (void)hipGetLastError();
        d->n_rebalance_refused++;
        return FLUID_OK;
    }
    DistState* nd = t->ds;
    nd->rb_every = d->rb_every; nd->rb_ratio = d->rb_ratio; nd->n_rebalanced = d->n_rebalanced + 1; nd->n_routed = d->n_routed;
    nd->solid_global.swap(d->solid_global);
    t->dt = s->dt;
    t->step_counter = s->step_counter;
    t->stats = s->stats;
    for (int c = 0; c < 3; ++c) t->mg_last_iters_k[c] = s->mg_last_iters_k[c];
    t->prof_every = s->prof_every;
    for (int c = 0; c < FLUID_PROF_COUNT; ++c) std::swap(t->prof[c], s->prof[c]);
    std::swap(*s, *t);      // the caller's handle now holds the new window
    fluid_destroy(t);       // ... and this one the old
    return FLUID_OK;
}

}  // namespace

int fl::dist_step(fluid_sim* s, fluid_step_stats_t* stats)
{
    HIPCHK(hipSetDevice(s->prm.device));
    int rc = s->ds->repl ? dist_step_replicated(s, stats) : dist_step_decomposed(s, stats);
    if (rc) return rc;
    DistState* d = s->ds;
    if (d->rb_every > 0 && d->comm.size > 1 && s->step_counter % d->rb_every == 0) {
        const int before = d->n_rebalanced;
        if ((rc = dist_rebalance(s))) return rc;
        if (s->ds->n_rebalanced != before && stats) stats->paths |= FLUID_PATH_DIST_REBALANCED;
    }
    return FLUID_OK;
}

void fl::dist_keep_solid(fluid_sim* s, const uint8_t* solid_global)
{
    if (s->ds) s->ds->solid_global.assign(solid_global, solid_global + (size_t)s->g.N * s->g.N * s->g.N);
}

void fl::dist_destroy(fluid_sim* s)
{
    DistState* d = s->ds;
    if (!d) return;
    void* ptrs[] = {d->hs, d->hr, d->mig_s, d->mig_r, d->d_cnt, d->repl_buf, d->rows, d->row_starts, d->cnt_pcg, d->gstage[0], d->gstage[1], d->gpq, d->gcg, d->tl_int, d->tl_bnd, d->tl_cls, d->rb_buf};
    for (void* p : ptrs) if (p) hipFree(p);
    if (d->st2) { hipStreamSynchronize(d->st2); hipStreamDestroy(d->st2); }
    if (d->ev_pack) hipEventDestroy(d->ev_pack);
    if (d->ev_halo) hipEventDestroy(d->ev_halo);
    if (d->h_cnt) hipHostFree(d->h_cnt);
    delete d;
    s->ds = nullptr;
}

extern "C" {

int fluid_create_dist(const fluid_params_t* p, const fluid_comm_t* comm, const fluid_decomp_t* dc, fluid_sim_t** out)
{
    if (!p || !comm || !dc || !out) return fail(FLUID_ERR_ARG, "null argument");
    if (comm->size < 1 || comm->size > FLUID_MAX_RANKS || comm->rank < 0 || comm->rank >= comm->size) return fail(FLUID_ERR_ARG, "bad rank/size");
    if (!comm->exchange || !comm->allreduce) return fail(FLUID_ERR_ARG, "comm callbacks missing");
    if (dc->dims[0] < 1 || dc->dims[1] < 1 || dc->dims[2] < 1 || (long)dc->dims[0] * dc->dims[1] * dc->dims[2] != comm->size)
        return fail(FLUID_ERR_ARG, "dims[0]*dims[1]*dims[2] must equal the number of ranks");
    for (int a = 0; a < 3; ++a) {
        const int32_t* c = dc->cuts[a];
        if (!c || c[0] != 0 || c[dc->dims[a]] != p->n) return fail(FLUID_ERR_ARG, "cuts must start at 0 and end at n");
        for (int b = 0; b < dc->dims[a]; ++b) {
            if (dc->dims[a] > 1 && c[b + 1] - c[b] < 8) return fail(FLUID_ERR_ARG, "every block must be at least 8 cells wide");
            if (b > 0 && (c[b] & 3)) return fail(FLUID_ERR_ARG, "interior cuts must be multiples of 4");
        }
    }
    DistState* d = new DistState();
    d->comm = *comm;
    for (int a = 0; a < 3; ++a) {
        d->dims[a] = dc->dims[a];
        d->cuts[a].assign(dc->cuts[a], dc->cuts[a] + dc->dims[a] + 1);
    }
    d->bc[2] = comm->rank % d->dims[2];
    d->bc[1] = (comm->rank / d->dims[2]) % d->dims[1];
    d->bc[0] = comm->rank / (d->dims[1] * d->dims[2]);
    for (int a = 0; a < 3; ++a) {
        d->ob.lo[a] = d->cuts[a][d->bc[a]];
        d->ob.hi[a] = d->cuts[a][d->bc[a] + 1];
        d->ob.has_lo[a] = d->bc[a] > 0;
        d->ob.has_hi[a] = d->bc[a] < d->dims[a] - 1;
    }
    for (int dd = 0; dd < 27; ++dd) {
        const int e[3] = {dd / 9 - 1, (dd / 3) % 3 - 1, dd % 3 - 1};
        int b[3];
        bool ok = dd != 13;
        for (int a = 0; a < 3; ++a) { b[a] = d->bc[a] + e[a]; ok = ok && b[a] >= 0 && b[a] < d->dims[a]; }
        d->nbr[dd] = ok ? rank_of(d, b[0], b[1], b[2]) : -1;
    }
    // AUTO: the decomposed solve pays >= 4 exchanges per PCG iteration, each a few tens of microseconds over xGMI; below
    // ~2 M unknowns per solve a single GPU runs the whole iteration in less than that, so small grids replicate the block.
    d->repl = p->dist_solve == FLUID_DIST_REPLICATED || (p->dist_solve == FLUID_DIST_AUTO && p->n < 384);
    if (const char* e = getenv("FLUID_DIST_SOLVE")) d->repl = atoi(e) == 0;   // developer knob, overrides the param
    if (const char* e = getenv("FLUID_DIST_SPLIT")) d->split_force = atoi(e);
    if (d->split_force < 0 || d->split_force > 2) d->split_force = 0;
    if (const char* e = getenv("FLUID_DIST_OVERLAP")) d->overlap = atoi(e) != 0;
    if (const char* e = getenv("FLUID_DIST_REBALANCE")) d->rb_every = std::max(0, atoi(e));
    if (const char* e = getenv("FLUID_DIST_CG")) d->cg_form = !strcmp(e, "cg") ? 0 : 1;
    if (const char* e = getenv("FLUID_DIST_FAIL_GROW")) d->fail_grow_rank = atoi(e);         // test hooks (tests/test_gpu_dist.py)
    if (const char* e = getenv("FLUID_DIST_FAIL_REBUILD")) d->fail_rebuild_rank = atoi(e);
    if (const char* e = getenv("FLUID_DIST_GATHER")) d->split_exchange_force = !strcmp(e, "exchange") ? 1 : (!strcmp(e, "allreduce") ? 0 : -1);
    Grid g;
    g.N = p->n;
    g.lo = -(p->n / 2);
    g.hi = g.lo + p->n - 1;
    if (d->repl) {
        g.nx = g.ny = g.nz = p->n;
        g.ox = g.oy = g.oz = 0;
    } else {
        int o[3], n[3];
        for (int a = 0; a < 3; ++a) {
            const int lo = std::max(0, d->ob.lo[a] - HALO_W), hi = std::min(p->n, d->ob.hi[a] + HALO_W);
            o[a] = lo; n[a] = hi - lo;
        }
        g.ox = o[0]; g.oy = o[1]; g.oz = o[2];
        g.nx = n[0]; g.ny = n[1]; g.nz = n[2];
    }
    fluid_params_t q = *p;
    int rc = fluid_create_window(&q, g, out);
    if (rc) { delete d; return rc; }
    fluid_sim* s = *out;
    s->dist = true;
    s->ds = d;
    auto bail = [&](const std::string& m) { fluid_destroy(s); *out = nullptr; return fail(FLUID_ERR_HIP, m); };
    if (dalloc(&d->gstage[0], (size_t)4) != hipSuccess || dalloc(&d->gstage[1], (size_t)4) != hipSuccess || dalloc(&d->gpq, (size_t)2) != hipSuccess || dalloc(&d->gcg, (size_t)4) != hipSuccess ||
        dalloc(&d->d_cnt, (size_t)128) != hipSuccess || dalloc(&d->cnt_pcg, s->lmax + 64) != hipSuccess ||
        hipHostMalloc((void**)&d->h_cnt, 128 * sizeof(int)) != hipSuccess)
        return bail("alloc of the decomposition's scratch failed");
    if (dalloc(&d->rb_buf, (size_t)3 * p->n + 2 * FLUID_MAX_RANKS + 64) != hipSuccess) return bail("rebalance scratch");
    if (hipStreamCreateWithFlags(&d->st2, hipStreamNonBlocking) != hipSuccess || hipEventCreateWithFlags(&d->ev_pack, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&d->ev_halo, hipEventDisableTiming) != hipSuccess)
        return bail("second stream / events");
    if (!d->repl) {
        const int org[3] = {g.ox, g.oy, g.oz};
        if (hipMemset(s->indices, 0xFF, s->ncell * sizeof(int)) != hipSuccess) return bail("indices fill failed");   // -1: no unknown yet (fluid.cc:1388)
        make_plan(d, d->plan_flags, [&](int r) { return block_of(d, r); }, HALO_W, org, 0, g.sx(), g.nz);
    }
    // Every buffer whose size follows the active box is sized NOW for a box that fills the whole block: a step then never
    // allocates (except for particles that migrate in, dist_agree), so no rank can fail alone between two exchanges.
    {
        const long N = p->n;
        const long on[3] = {d->ob.hi[0] - d->ob.lo[0], d->ob.hi[1] - d->ob.lo[1], d->ob.hi[2] - d->ob.lo[2]};
        auto shell = [&](int w) { return (on[0] + 2 * w) * (on[1] + 2 * w) * (on[2] + 2 * w) - on[0] * on[1] * on[2]; };
        size_t stage = 0;
        if (d->repl) {
            stage = (size_t)24 * N * N * N;                                  // the all-gather of u, v, w over the whole box
            d->repl_cap = 4 * (size_t)N * N * N;
            if (!(d->dims[0] <= 2 && d->dims[1] <= 2 && d->dims[2] <= 2) && hipMalloc((void**)&d->repl_buf, d->repl_cap * sizeof(double)) != hipSuccess)
                return bail("replicated mode: gather buffer");
        } else {
            stage = std::max((size_t)48 * shell(1), (size_t)8 * shell(HALO_W));   // six doubles, 1 wide; one double, HALO_W wide
            // the gathered level: level 1 while small, else level 2 — but a step that takes the Galerkin coarse levels gathers level 1
            // whatever its size (dist_mg_setup), so with them enabled the whole of level 1 is provided for
            const bool gal_possible = s->gal_mode != 0 && s->mg_fp32 && use_mg(s);
            const long lsplit = gal_possible ? (N / 2 + 2) * (N / 2 + 2) * (N / 2 + 2) : std::max(300000L, (N / 4 + 2) * (N / 4 + 2) * (N / 4 + 2));
            stage = std::max(stage, (size_t)8 * lsplit);
            d->rows_cap = (size_t)N * N * d->dims[2] + 1024;
            if (hipMalloc((void**)&d->rows, d->rows_cap * sizeof(int)) != hipSuccess || hipMalloc((void**)&d->row_starts, d->rows_cap * sizeof(int)) != hipSuccess)
                return bail("row numbering tables");
            // multigrid slab: the level-0 arrays never exceed the solver vectors' capacity (lmax); the deeper local levels add 1/7
            // of that, the gathered ones the level counted above; types + counts + u, v, f, r (the cycle's own precision) + z
            const size_t es = s->mg_fp32 ? 4 : 8;
            const size_t cap = (size_t)((double)(s->lmax + 4096) * 1.2 * (2 + 4 * es)) + 8 * (s->lmax + 4096) + (size_t)lsplit * 2 * (2 + 4 * es) + (1 << 20);
            if (hipMalloc((void**)&s->mg_slab, cap) != hipSuccess) return bail("multigrid slab");
            s->mg_slab_cap = cap;
            // ... and what round 4 added to the decomposed solve, for the same reason (no allocation inside a step): the Galerkin levels'
            // coefficients (4 floats + a count byte per cell of the replicated levels + their tile flags), the droplet buffers, the
            // active-tile lists of the level-0 legs (8 x 8 x 16 cells a tile) and of the A z sweep (4 x 8 x 32)
            if (gal_possible) {
                s->gal_slab_cap = (size_t)lsplit * 20 + (1 << 20);
                if (hipMalloc((void**)&s->gal_slab, s->gal_slab_cap) != hipSuccess) return bail("Galerkin coefficient slab");
            }
            if (s->drops_on && use_mg(s)) {
                if (hipMalloc((void**)&s->drop_ctr, (size_t)(64 * DROP_NCTR + DROP_NCTR + 1) * sizeof(int)) != hipSuccess ||
                    hipMalloc((void**)&s->drop_n, (size_t)DROP_CAP * sizeof(int)) != hipSuccess ||
                    hipMalloc((void**)&s->drop_cells, (size_t)DROP_CAP * 64 * sizeof(int)) != hipSuccess)
                    return bail("droplet buffers");
            }
            {
                const size_t ntile = s->lmax / 256 + 8192;   // generous: partial tiles along every edge of the local box
                s->tl_cap = 4 * ntile;
                d->tl_cap = ntile;
                if (hipMalloc((void**)&s->tl_flags, s->tl_cap) != hipSuccess || hipMalloc((void**)&s->tl_mg, s->tl_cap * sizeof(int)) != hipSuccess ||
                    hipMalloc((void**)&s->tl_sq, s->tl_cap * sizeof(int)) != hipSuccess || hipMalloc((void**)&d->tl_int, d->tl_cap * sizeof(int)) != hipSuccess ||
                    hipMalloc((void**)&d->tl_bnd, d->tl_cap * sizeof(int)) != hipSuccess || hipMalloc((void**)&d->tl_cls, d->tl_cap) != hipSuccess)
                    return bail("tile lists");
            }
        }
        if (ensure_stage(s, stage + 4096, stage + 4096)) return bail("halo staging buffers");
        const Box ownW = to_box(IBox{{d->ob.lo[0], d->ob.lo[1], d->ob.lo[2]}, {d->ob.hi[0], d->ob.hi[1], d->ob.hi[2]}}, (const int[3]){g.ox, g.oy, g.oz});
        const size_t need = p2g_part_doubles(ownW), ni = 4 + 4 * (size_t)p2g_max_items(ownW);
        if (hipMalloc((void**)&s->p2g_part, need * sizeof(double)) != hipSuccess || hipMalloc((void**)&s->p2g_items, ni * sizeof(int)) != hipSuccess ||
            hipMemset(s->p2g_items, 0, 4 * sizeof(int)) != hipSuccess)
            return bail("P2G partials / work list");
        s->p2g_part_cap = need; s->p2g_items_cap = ni;
        s->stats.outer_passes = 0;
        if (pic_fields(s)) return bail("PIC fields");
    }
    if (hipDeviceSynchronize() != hipSuccess) return bail("device sync failed");
    return FLUID_OK;
}

int fluid_window(fluid_sim_t* s, int32_t origin[3], int32_t dims[3], int32_t own_lo[3], int32_t own_hi[3])
{
    if (!s) return fail(FLUID_ERR_ARG, "null handle");
    const Grid g = s->g;
    const int o[3] = {g.ox, g.oy, g.oz}, n[3] = {g.nx, g.ny, g.nz};
    for (int a = 0; a < 3; ++a) {
        if (origin) origin[a] = o[a];
        if (dims) dims[a] = n[a];
        if (own_lo) own_lo[a] = s->ds ? s->ds->ob.lo[a] : 0;
        if (own_hi) own_hi[a] = s->ds ? s->ds->ob.hi[a] : g.N;
    }
    return FLUID_OK;
}

int fluid_upload_particles_ids(fluid_sim_t* s, int64_t n, const double* pos, const double* vel, const uint32_t* ids)
{
    if (!s || n < 0 || (n > 0 && (!pos || !ids))) return fail(FLUID_ERR_ARG, "bad particle arguments");
    HIPCHK(hipSetDevice(s->prm.device));
    // some room for ghosts and for particles that migrate in; both the particle arrays and the routing buffers grow on demand
    long slack = 1L << 16;
    if (const char* e = getenv("FLUID_DIST_SLACK")) slack = std::max(16L, atol(e));   // developer knob: tests force the growth paths with it
    int rc = alloc_particles(s, 2 * (long)n + slack);
    if (rc) return rc;
    DistState* d = s->ds;
    if (d && !d->mig_s) {
        d->mig_cap = (long)n / 2 + slack;
        HIPCHK(hipMalloc((void**)&d->mig_s, (size_t)d->mig_cap * 56));
        HIPCHK(hipMalloc((void**)&d->mig_r, (size_t)d->mig_cap * 56));
    }
    s->np = (long)n;
    s->p_off = 0;
    if (n > 0) {
        uint32_t* dids = (uint32_t*)s->order;  // staging: order[] is free between steps
        HIPCHK(hipMemcpyAsync(s->stage_pos, pos, 3 * n * sizeof(double), hipMemcpyHostToDevice, s->st));
        if (vel) HIPCHK(hipMemcpyAsync(s->stage_vel, vel, 3 * n * sizeof(double), hipMemcpyHostToDevice, s->st));
        HIPCHK(hipMemcpyAsync(dids, ids, n * sizeof(uint32_t), hipMemcpyHostToDevice, s->st));
        launch_unpack_ids(s->st, s->np, s->stage_pos, vel ? s->stage_vel : nullptr, dids, s->pa);
        HIPCHK(hipGetLastError());
        HIPCHK(hipStreamSynchronize(s->st));
    }
    s->sorted = s->have_p2g = s->have_flags = false;
    s->have_guess = false;  // a new particle set: the first solve starts from 0
    s->sort_hint = false;
    return FLUID_OK;
}

int64_t fluid_download_particles_ids(fluid_sim_t* s, double* pos, double* vel, uint32_t* ids)
{
    if (!s) return -1;
    if (hipSetDevice(s->prm.device) != hipSuccess) return -1;
    // the live particles (after a step the served ghosts are still in the arrays, marked dead)
    int* cursor = s->d_small;
    if (hipMemsetAsync(cursor, 0, sizeof(int), s->st) != hipSuccess) return -1;
    const bool fetch = pos && vel && ids;
    uint32_t* dids = (uint32_t*)s->order;
    launch_pack_live(s->st, s->np, s->pa.shifted(s->p_off), fetch ? s->stage_pos : nullptr, s->stage_vel, dids, cursor);
    int live = 0;
    if (hipMemcpyAsync(s->h_small, cursor, sizeof(int), hipMemcpyDeviceToHost, s->st) != hipSuccess) return -1;
    if (hipStreamSynchronize(s->st) != hipSuccess) return -1;
    live = s->h_small[0];
    if (!fetch || live == 0) return live;
    if (hipMemcpyAsync(pos, s->stage_pos, 3 * (size_t)live * sizeof(double), hipMemcpyDeviceToHost, s->st) != hipSuccess) return -1;
    if (hipMemcpyAsync(vel, s->stage_vel, 3 * (size_t)live * sizeof(double), hipMemcpyDeviceToHost, s->st) != hipSuccess) return -1;
    if (hipMemcpyAsync(ids, dids, (size_t)live * sizeof(uint32_t), hipMemcpyDeviceToHost, s->st) != hipSuccess) return -1;
    if (hipStreamSynchronize(s->st) != hipSuccess) return -1;
    return live;
}

int fluid_partition_blocks(int32_t n, int64_t np, const double* pos, const int32_t dims[3], int32_t* cuts_x, int32_t* cuts_y, int32_t* cuts_z)
{
    if (n < 8 || !dims || !cuts_x || !cuts_y || !cuts_z || (np > 0 && !pos)) return fail(FLUID_ERR_ARG, "bad argument");
    int32_t* cuts[3] = {cuts_x, cuts_y, cuts_z};
    const int lo = -(n / 2);
    for (int a = 0; a < 3; ++a) {
        std::vector<int64_t> hist(n, 0);
        for (int64_t i = 0; i < np; ++i) {
            long b = std::lround(pos[3 * i + a]) - lo;  // C round(): half away from zero, like the base cell (fluid.cc:267)
            b = b < 0 ? 0 : (b > n - 1 ? n - 1 : b);
            hist[b]++;
        }
        int rc = cuts_from_hist(n, hist.data(), dims[a], cuts[a]);
        if (rc) return rc;
    }
    return FLUID_OK;
}

int fluid_dist_set_rebalance(fluid_sim_t* s, int32_t every, double ratio)
{
    if (!s || !s->ds || every < 0 || !(ratio >= 1.0)) return fail(FLUID_ERR_ARG, "not a decomposed handle, or bad every / ratio");
    s->ds->rb_every = every;
    s->ds->rb_ratio = ratio;
    return FLUID_OK;
}

int fluid_dist_get_info(fluid_sim_t* s, int32_t* overlap, int32_t* cg_form, int32_t* n_refused)
{
    if (!s || !s->ds) return fail(FLUID_ERR_ARG, "fluid_dist_get_info: not a decomposed handle");
    if (overlap) *overlap = s->ds->overlap_checked;
    if (cg_form) *cg_form = s->ds->cg_form;
    if (n_refused) *n_refused = s->ds->n_rebalance_refused;
    return FLUID_OK;
}

int fluid_dist_get_cuts(fluid_sim_t* s, int32_t* cuts_x, int32_t* cuts_y, int32_t* cuts_z, int32_t* n_rebalanced)
{
    if (!s || !s->ds) return fail(FLUID_ERR_ARG, "not a decomposed handle");
    int32_t* c[3] = {cuts_x, cuts_y, cuts_z};
    for (int a = 0; a < 3; ++a)
        if (c[a]) for (size_t i = 0; i < s->ds->cuts[a].size(); ++i) c[a][i] = s->ds->cuts[a][i];
    if (n_rebalanced) *n_rebalanced = s->ds->n_rebalanced;
    return FLUID_OK;
}

}  // extern "C"
