// libfluid_hip.so — C ABI (include/fluid_hip.h) and host orchestration of the step
// fluid.cc:1378-1490 on one MI355X.  One handle = one HIP stream; all fields live in HBM.
#include "sim.h"
#include <algorithm>

using namespace fl;

static thread_local std::string g_err;
int fluid_fail(int code, const std::string& msg)
{
    g_err = msg;
    return code;
}
#define fail fluid_fail

Box fl::clip_dilate(const Box& b, int d, const Grid& g)
{
    Box r;
    r.x0 = b.x0 - d < 0 ? 0 : b.x0 - d;
    r.y0 = b.y0 - d < 0 ? 0 : b.y0 - d;
    r.z0 = b.z0 - d < 0 ? 0 : b.z0 - d;
    r.x1 = b.x1 + d > g.nx - 1 ? g.nx - 1 : b.x1 + d;
    r.y1 = b.y1 + d > g.ny - 1 ? g.ny - 1 : b.y1 + d;
    r.z1 = b.z1 + d > g.nz - 1 ? g.nz - 1 : b.z1 + d;
    return r;
}

// ---- profiling helpers ----------------------------------------------------------------------
int fl::prof_begin(fluid_sim* s, int k, double cells)
{
    ProfClass& p = s->prof[k];
    p.launches++;
    const bool per_iter = k == FLUID_PROF_PCG_SQ || k == FLUID_PROF_PCG_XR || k == FLUID_PROF_MG_UP0;
    const int every = per_iter ? s->prof_every : (s->prof_every >= 8 ? s->prof_every / 8 : (s->prof_every > 0 ? 1 : 0));
    if (every <= 0 || (p.launches - 1) % every) return -1;
    hipEvent_t a, b;
    if (hipEventCreate(&a) != hipSuccess || hipEventCreate(&b) != hipSuccess) return -1;
    hipEventRecord(a, s->st);
    p.e0.push_back(a);
    p.e1.push_back(b);
    p.pc.push_back(cells);
    return (int)p.e1.size() - 1;
}
void fl::prof_end(fluid_sim* s, int k, int tok)
{
    if (tok >= 0) hipEventRecord(s->prof[k].e1[tok], s->st);
}
static void prof_resolve(fluid_sim* s)
{
    for (int k = 0; k < FLUID_PROF_COUNT; ++k) {
        ProfClass& p = s->prof[k];
        for (size_t i = 0; i < p.e0.size(); ++i) {
            float ms = 0;
            hipEventSynchronize(p.e1[i]);
            if (hipEventElapsedTime(&ms, p.e0[i], p.e1[i]) == hipSuccess) {
                p.sampled++;
                p.ms += ms;
                p.cells += p.pc[i];
            }
            hipEventDestroy(p.e0[i]);
            hipEventDestroy(p.e1[i]);
        }
        p.e0.clear();
        p.e1.clear();
        p.pc.clear();
    }
}

// ---- memory ------------------------------------------------------------------------------------

static void free_particles(fluid_sim* s)
{
    for (Particles* p : {&s->pa, &s->pb}) {
        hipFree(p->px); hipFree(p->py); hipFree(p->pz); hipFree(p->vx); hipFree(p->vy); hipFree(p->vz); hipFree(p->pid);
        *p = Particles{};
    }
    hipFree(s->key); hipFree(s->slot); hipFree(s->order); hipFree(s->order2); hipFree(s->spid); hipFree(s->stage_pos); hipFree(s->stage_vel);
    s->order2 = nullptr; s->spid = nullptr; hipFree(s->pw);
    s->pw = nullptr;
    s->key = s->slot = s->order = nullptr;
    s->stage_pos = s->stage_vel = nullptr;
    s->cap = 0;
}

int fl::alloc_particles(fluid_sim* s, long n)
{
    if (n <= s->cap) return FLUID_OK;
    n = (n + 1) & ~1L;  // even: k_p2g_rows reads the particle arrays 16 B at a time
    free_particles(s);
    for (Particles* p : {&s->pa, &s->pb}) {
        HIPCHK(dalloc(&p->px, n)); HIPCHK(dalloc(&p->py, n)); HIPCHK(dalloc(&p->pz, n));
        HIPCHK(dalloc(&p->vx, n)); HIPCHK(dalloc(&p->vy, n)); HIPCHK(dalloc(&p->vz, n));
        HIPCHK(dalloc(&p->pid, n));
    }
    HIPCHK(dalloc(&s->key, n)); HIPCHK(dalloc(&s->slot, n)); HIPCHK(dalloc(&s->order, n)); HIPCHK(dalloc(&s->order2, n)); HIPCHK(dalloc(&s->spid, n));
    HIPCHK(dalloc(&s->stage_pos, 3 * n)); HIPCHK(dalloc(&s->stage_vel, 3 * n));
    HIPCHK(dalloc(&s->pw, 9 * n));
    HIPCHK(hipDeviceSynchronize());
    s->cap = n;
    return FLUID_OK;
}

// More room, the live particles kept (a decomposed run: a block's share grows as the fluid spreads into it).
int fl::grow_particles(fluid_sim* s, long need)
{
    if (need <= s->cap) return FLUID_OK;
    HIPCHK(hipStreamSynchronize(s->st));
    const long keep = s->p_off + s->np;
    Particles old = s->pa;
    s->pa = Particles{};
    const long ncap = ((need + need / 2) + 1) & ~1L;
    Particles nw{};
    HIPCHK(dalloc(&nw.px, ncap)); HIPCHK(dalloc(&nw.py, ncap)); HIPCHK(dalloc(&nw.pz, ncap));
    HIPCHK(dalloc(&nw.vx, ncap)); HIPCHK(dalloc(&nw.vy, ncap)); HIPCHK(dalloc(&nw.vz, ncap));
    HIPCHK(dalloc(&nw.pid, ncap));
    if (keep > 0) {
        HIPCHK(hipMemcpy(nw.px, old.px, keep * 8, hipMemcpyDeviceToDevice)); HIPCHK(hipMemcpy(nw.py, old.py, keep * 8, hipMemcpyDeviceToDevice));
        HIPCHK(hipMemcpy(nw.pz, old.pz, keep * 8, hipMemcpyDeviceToDevice)); HIPCHK(hipMemcpy(nw.vx, old.vx, keep * 8, hipMemcpyDeviceToDevice));
        HIPCHK(hipMemcpy(nw.vy, old.vy, keep * 8, hipMemcpyDeviceToDevice)); HIPCHK(hipMemcpy(nw.vz, old.vz, keep * 8, hipMemcpyDeviceToDevice));
        HIPCHK(hipMemcpy(nw.pid, old.pid, keep * 4, hipMemcpyDeviceToDevice));
    }
    hipFree(old.px); hipFree(old.py); hipFree(old.pz); hipFree(old.vx); hipFree(old.vy); hipFree(old.vz); hipFree(old.pid);
    free_particles(s);   // the second buffer and the scratch arrays hold nothing between steps
    s->pa = nw;
    Particles* p = &s->pb;
    HIPCHK(dalloc(&p->px, ncap)); HIPCHK(dalloc(&p->py, ncap)); HIPCHK(dalloc(&p->pz, ncap));
    HIPCHK(dalloc(&p->vx, ncap)); HIPCHK(dalloc(&p->vy, ncap)); HIPCHK(dalloc(&p->vz, ncap));
    HIPCHK(dalloc(&p->pid, ncap));
    HIPCHK(dalloc(&s->key, ncap)); HIPCHK(dalloc(&s->slot, ncap)); HIPCHK(dalloc(&s->order, ncap)); HIPCHK(dalloc(&s->order2, ncap)); HIPCHK(dalloc(&s->spid, ncap));
    HIPCHK(dalloc(&s->stage_pos, 3 * ncap)); HIPCHK(dalloc(&s->stage_vel, 3 * ncap));
    HIPCHK(dalloc(&s->pw, 9 * ncap));
    HIPCHK(hipDeviceSynchronize());
    s->cap = ncap;
    return FLUID_OK;
}

extern "C" {

const char* fluid_last_error(void) { return g_err.c_str(); }
const char* fluid_version(void) { return "libfluid_hip 0.1 gfx950"; }

int fluid_default_params(fluid_params_t* p)
{
    if (!p) return fail(FLUID_ERR_ARG, "null params");
    memset(p, 0, sizeof(*p));
    p->n = 121;
    p->device = 0;
    p->dx = 1.0;
    p->rho = 1.0;
    p->gravity[0] = 0; p->gravity[1] = -10; p->gravity[2] = 0;
    p->max_dt = 0.1;
    p->outer_tol = 0.1;
    p->update_frac = 0.1;
    p->cg_tol = 2.220446049250313e-16;
    p->cg_max_iters = 0;
    p->max_outer_passes = 0;
    p->precision = FLUID_PRECISION_FP64;
    p->preconditioner = FLUID_PRECOND_MG;
    p->flip_blend = 1.0;
    p->solve_start = FLUID_START_WARM;
    p->mg_precision = FLUID_MG_FP32;
    p->dist_solve = FLUID_DIST_AUTO;
    return FLUID_OK;
}

int fluid_destroy(fluid_sim_t* s)
{
    if (!s) return FLUID_OK;
    if (s->st) hipStreamSynchronize(s->st);
    if (s->ds) dist_destroy(s);
    prof_resolve(s);
    free_particles(s);
    void* ptrs[] = {s->solid, s->flags, s->container, s->rhs, s->diver, s->diver2, s->u, s->v, s->w, s->ub, s->vb, s->wb, s->dcx, s->dcy,
                    s->dcz, s->pressure, s->p_guess, s->p_guess2, s->p_q, s->indices, s->scan_sums, s->ipart, s->R, s->S[0], s->Q, s->X, s->mg_slab, s->mg_part, s->cntL, s->part_bb, s->part_rr,
                    s->part_rz[0], s->part_rz[1], s->part_pq, s->part_err, s->ps, s->cell_count, s->cell_start, s->ss,
                    s->pcx, s->pcy, s->pcz, s->p2g_part, s->p2g_items, s->p2g_crowd, s->tl_flags, s->tl_mg, s->tl_sq, s->d_small, s->row_flags, s->row_pos, s->row_list, s->drop_n, s->drop_cells, s->drop_ctr, s->gal_slab};
    for (void* p : ptrs) if (p) hipFree(p);
    if (s->h_ps) hipHostFree(s->h_ps);
    for (int i = 0; i < 2; ++i) if (s->ev_poll[i]) hipEventDestroy(s->ev_poll[i]);
    if (s->h_ss) hipHostFree(s->h_ss);
    if (s->h_small) hipHostFree(s->h_small);
    if (s->st) hipStreamDestroy(s->st);
    delete s;
    return FLUID_OK;
}

int fluid_create(const fluid_params_t* p, fluid_sim_t** out)
{
    if (!p || !out) return fail(FLUID_ERR_ARG, "null argument");
    Grid g;
    g.N = p->n;
    g.lo = -(p->n / 2);
    g.hi = g.lo + p->n - 1;
    g.nx = g.ny = g.nz = p->n;
    g.ox = g.oy = g.oz = 0;
    return fluid_create_window(p, g, out);
}

}  // extern "C"

// The handle's field arrays cover the window g (all of the grid on one GPU; block + halo on a rank of a decomposed run).
int fl::fluid_create_window(const fluid_params_t* p, const Grid& g, fluid_sim_t** out)
{
    if (!p || !out) return fail(FLUID_ERR_ARG, "null argument");
    if (p->n < 8 || p->n > 1024) return fail(FLUID_ERR_ARG, "n must be in [8,1024]");
    if (!(p->dx > 0) || !(p->rho > 0) || !(p->max_dt > 0)) return fail(FLUID_ERR_ARG, "dx, rho, max_dt must be > 0");
    if (p->precision != FLUID_PRECISION_FP64 && p->precision != FLUID_PRECISION_FP32) return fail(FLUID_ERR_ARG, "bad precision");
    if (!(p->flip_blend >= 0.0 && p->flip_blend <= 1.0)) return fail(FLUID_ERR_ARG, "flip_blend must be in [0,1] (1 = the reference's pure FLIP)");
    if (p->preconditioner != FLUID_PRECOND_MG && p->preconditioner != FLUID_PRECOND_JACOBI) return fail(FLUID_ERR_ARG, "bad preconditioner");
    if (p->solve_start != FLUID_START_WARM && p->solve_start != FLUID_START_ZERO) return fail(FLUID_ERR_ARG, "bad solve_start");
    if (p->mg_precision != FLUID_MG_FP32 && p->mg_precision != FLUID_MG_FP64) return fail(FLUID_ERR_ARG, "bad mg_precision");
    if (p->dist_solve < FLUID_DIST_AUTO || p->dist_solve > FLUID_DIST_REPLICATED || p->pad_ != 0) return fail(FLUID_ERR_ARG, "bad dist_solve / pad_");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
        return fail(FLUID_ERR_HIP, "no HIP device visible: libfluid_hip has no CPU path");
    if (p->device < 0 || p->device >= ndev) return fail(FLUID_ERR_ARG, "device ordinal out of range");
    HIPCHK(hipSetDevice(p->device));
    fluid_sim* s = new fluid_sim();
    s->prm = *p;
    s->g = g;
    s->ncell = g.cells();
    s->dt = p->max_dt;
    s->mg_fp32 = p->mg_precision == FLUID_MG_FP32;
    s->warm = p->solve_start == FLUID_START_WARM;
    if (const char* e = getenv("FLUID_MG_CSWEEPS")) s->mg_csweeps = atoi(e);       // developer knobs (tools/, experiments): they override the params
    if (const char* e = getenv("FLUID_TILE_LISTS")) s->lists_force = atoi(e) != 0;
    if (const char* e = getenv("FLUID_P2G_FORM")) s->p2g_force = !strcmp(e, "rows") ? 1 : (!strcmp(e, "tiles") ? 2 : (!strcmp(e, "crowd") ? 3 : 0));
    if (const char* e = getenv("FLUID_MG_GALERKIN")) s->gal_mode = atoi(e);
    if (const char* e = getenv("FLUID_DROPLETS")) s->drops_on = atoi(e) != 0;
    if (const char* e = getenv("FLUID_ROW_SWEEPS")) s->row_sweeps = atoi(e) != 0;
    if (const char* e = getenv("FLUID_DROPLETS_MIN")) s->drop_min = atoi(e);
    *out = nullptr;
    auto bail = [&](int rc) { fluid_destroy(s); return rc; };
#define A(expr) do { hipError_t e_ = (expr); if (e_ != hipSuccess) return bail(fail(FLUID_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(e_))); } while (0)
    A(hipStreamCreateWithFlags(&s->st, hipStreamNonBlocking));
    const size_t n = s->ncell;
    A(dalloc(&s->solid, n)); A(dalloc(&s->flags, n));
    A(dalloc(&s->container, n)); A(dalloc(&s->rhs, n)); A(dalloc(&s->diver, n)); A(dalloc(&s->diver2, n));
    A(dalloc(&s->u, n)); A(dalloc(&s->v, n)); A(dalloc(&s->w, n));
    A(dalloc(&s->ub, n)); A(dalloc(&s->vb, n)); A(dalloc(&s->wb, n));
    A(dalloc(&s->dcx, n)); A(dalloc(&s->dcy, n)); A(dalloc(&s->dcz, n));
    A(dalloc(&s->pressure, n));
    if (s->warm) {
        A(dalloc(&s->p_guess, n)); A(dalloc(&s->p_guess2, n));
        if (hipMemset(s->p_guess, 0, n * sizeof(double)) != hipSuccess || hipMemset(s->p_guess2, 0, n * sizeof(double)) != hipSuccess) return bail(FLUID_ERR_HIP);
        if (const char* e = getenv("FLUID_EXTRAPOLATE")) s->extrapolate = atoi(e) != 0;
        if (s->extrapolate) { A(dalloc(&s->p_q, n)); if (hipMemset(s->p_q, 0, n * sizeof(double)) != hipSuccess) return bail(FLUID_ERR_HIP); }
    }
    A(dalloc(&s->indices, n));
    A(dalloc(&s->scan_sums, (n + 2) / 2048 + 16));
    A(dalloc(&s->ipart, (size_t)1024 * 8));
    const size_t se = solver_elem(s);
    s->lmax = lbox_max_cells(std::max(g.nx, std::max(g.ny, g.nz)));  // >= n: the arrays double as N^3 scratch for fluid_stencil_apply
    const size_t ln = s->lmax;
    A(dalloc((char**)&s->R, ln * se)); A(dalloc((char**)&s->S[0], 2 * ln * se + 256));  // both search vectors: one slab, one memset per step
    s->S[1] = (char*)s->S[0] + ln * se;
    A(dalloc((char**)&s->Q, ln * se)); A(dalloc((char**)&s->X, ln * se)); A(dalloc(&s->cntL, ln + 64));
    A(dalloc(&s->mg_part, ln / 256 + 64));
    A(dalloc(&s->part_bb, (size_t)MAX_PARTIALS)); A(dalloc(&s->part_rr, (size_t)MAX_PARTIALS));
    A(dalloc(&s->part_rz[0], (size_t)MAX_PARTIALS)); A(dalloc(&s->part_rz[1], (size_t)MAX_PARTIALS));
    A(dalloc(&s->part_pq, (size_t)MAX_PARTIALS)); A(dalloc(&s->part_err, (size_t)2 * MAX_PARTIALS));
    A(dalloc(&s->ps, (size_t)1)); A(dalloc(&s->ss, (size_t)1));
    A(dalloc(&s->cell_count, n + 4)); A(dalloc(&s->cell_start, n + 4));
    A(dalloc(&s->d_small, (size_t)32));
    A(hipHostMalloc((void**)&s->h_small, 32 * sizeof(int)));
    A(hipHostMalloc((void**)&s->h_ps, 2 * sizeof(PcgState)));
    A(hipEventCreateWithFlags(&s->ev_poll[0], hipEventDisableTiming));
    A(hipEventCreateWithFlags(&s->ev_poll[1], hipEventDisableTiming));
    A(hipHostMalloc((void**)&s->h_ss, sizeof(StepState)));
#undef A
    // default solid shell: solid outside W (fluid.cc:1256-1266)
    std::vector<uint8_t> sol(n, 0);
    const int N = p->n;
    for (int x = 0; x < g.nx; ++x)
        for (int y = 0; y < g.ny; ++y)
            for (int z = 0; z < g.nz; ++z) {
                const int X = x + g.ox, Y = y + g.oy, Z = z + g.oz;   // global index
                if (X < 2 || X > N - 3 || Y < 2 || Y > N - 3 || Z < 2 || Z > N - 3) sol[g.idx(x, y, z)] = 1;
            }
    hipError_t e = hipMemcpy(s->solid, sol.data(), n, hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(s->flags, sol.data(), n, hipMemcpyHostToDevice);  // F_SOLID == 1
    if (e == hipSuccess) e = hipDeviceSynchronize();  // hipMemset above ran on the null stream; s->st does not wait for it
    if (e != hipSuccess) return bail(fail(FLUID_ERR_HIP, std::string("solid upload: ") + hipGetErrorString(e)));
    *out = s;
    return FLUID_OK;
}

extern "C" {

/* solid: the GLOBAL N^3 array on every rank of a decomposed run; each keeps its window */
int fluid_set_solid(fluid_sim_t* s, const uint8_t* solid)
{
    if (!s || !solid) return fail(FLUID_ERR_ARG, "null argument");
    const int N = s->g.N;
    const Grid g = s->g;
    for (int x = 0; x < N; ++x)
        for (int y = 0; y < N; ++y)
            for (int z = 0; z < N; ++z) {
                size_t c = ((size_t)x * N + y) * N + z;
                bool outsideW = x < 2 || x > N - 3 || y < 2 || y > N - 3 || z < 2 || z > N - 3;
                if (outsideW && !solid[c]) return fail(FLUID_ERR_ARG, "cells outside W=[lo+2,hi-2]^3 must be solid");
            }
    std::vector<uint8_t> sol(s->ncell);
    for (int x = 0; x < g.nx; ++x)
        for (int y = 0; y < g.ny; ++y)
            for (int z = 0; z < g.nz; ++z)
                sol[g.idx(x, y, z)] = solid[((size_t)(x + g.ox) * N + (y + g.oy)) * N + (z + g.oz)] ? 1 : 0;
    dist_keep_solid(s, solid);   // (a decomposed handle keeps the global array: re-balancing builds new windows from it)
    HIPCHK(hipStreamSynchronize(s->st));
    HIPCHK(hipMemcpy(s->solid, sol.data(), s->ncell, hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(s->flags, sol.data(), s->ncell, hipMemcpyHostToDevice));
    s->have_p2g = s->have_flags = false;
    s->flags_valid = false;  // the next flags pass sweeps the whole grid
    s->have_guess = false;   // a new obstacle: the next solve starts from 0
    return FLUID_OK;
}

int fluid_upload_particles(fluid_sim_t* s, int64_t n, const double* pos, const double* vel)
{
    if (!s || n < 0 || (n > 0 && !pos)) return fail(FLUID_ERR_ARG, "bad particle arguments");
    if (s->dist) return fail(FLUID_ERR_STATE, "decomposed run: use fluid_upload_particles_ids (global ids must be unique across ranks)");
    if (n > 0x7fffffffL) return fail(FLUID_ERR_ARG, "too many particles");
    HIPCHK(hipSetDevice(s->prm.device));
    int rc = alloc_particles(s, (long)n);
    if (rc) return rc;
    s->np = (long)n;
    s->p_off = 0;
    if (n > 0) {
        HIPCHK(hipMemcpyAsync(s->stage_pos, pos, 3 * n * sizeof(double), hipMemcpyHostToDevice, s->st));
        if (vel) HIPCHK(hipMemcpyAsync(s->stage_vel, vel, 3 * n * sizeof(double), hipMemcpyHostToDevice, s->st));
        launch_unpack_particles(s->st, s->np, s->stage_pos, vel ? s->stage_vel : nullptr, s->pa);
        HIPCHK(hipGetLastError());
        HIPCHK(hipStreamSynchronize(s->st));
    }
    s->sorted = s->have_p2g = s->have_flags = false;
    s->have_guess = false;  // a new particle set: the first solve starts from 0
    s->sort_hint = false;
    return FLUID_OK;
}

int fluid_download_particles(fluid_sim_t* s, double* pos, double* vel)
{
    if (!s || !pos || !vel) return fail(FLUID_ERR_ARG, "null argument");
    if (s->dist) return fail(FLUID_ERR_STATE, "decomposed run: use fluid_download_particles_ids");
    if (s->np == 0) return FLUID_OK;
    HIPCHK(hipSetDevice(s->prm.device));
    launch_pack_particles(s->st, s->np, s->pa.shifted(s->p_off), s->stage_pos, s->stage_vel);
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpyAsync(pos, s->stage_pos, 3 * s->np * sizeof(double), hipMemcpyDeviceToHost, s->st));
    HIPCHK(hipMemcpyAsync(vel, s->stage_vel, 3 * s->np * sizeof(double), hipMemcpyDeviceToHost, s->st));
    HIPCHK(hipStreamSynchronize(s->st));
    return FLUID_OK;
}

int64_t fluid_num_particles(fluid_sim_t* s) { return s ? s->np : -1; }
int fluid_set_dt(fluid_sim_t* s, double dt)
{
    if (!s || !(dt > 0)) return fail(FLUID_ERR_ARG, "dt must be > 0");
    s->dt = dt;
    return FLUID_OK;
}
int fluid_get_dt(fluid_sim_t* s, double* dt)
{
    if (!s || !dt) return fail(FLUID_ERR_ARG, "null argument");
    *dt = s->dt;
    return FLUID_OK;
}

// counter-based RNG (splitmix64 finaliser) -> U[0,1)
static inline double u01(uint64_t seed, uint64_t ctr)
{
    uint64_t z = seed * 0x9E3779B97F4A7C15ull + ctr * 0xBF58476D1CE4E5B9ull + 0x94D049BB133111EBull;
    z ^= z >> 30; z *= 0xBF58476D1CE4E5B9ull;
    z ^= z >> 27; z *= 0x94D049BB133111EBull;
    z ^= z >> 31;
    return (double)(z >> 11) * (1.0 / 9007199254740992.0);
}

int64_t fluid_scene_water_cube_drop(int32_t n, int32_t ppc, uint64_t seed, double* pos)
{
    if (n < 8 || ppc < 1) return -1;
    const int m = (int)std::lround((double)n * 41.0 / 121.0);  // 41 of 121 in the reference (fluid.cc:1176)
    const int c0 = -(m / 2), c1 = c0 + m - 1;
    const int lo = -(n / 2), hi = lo + n - 1;
    int64_t cnt = 0;
    uint64_t ctr = 0;
    for (int x = c0; x <= c1; ++x)
        for (int y = c0; y <= c1; ++y)
            for (int z = c0; z <= c1; ++z)
                for (int k = 0; k < ppc; ++k) {
                    // PointScatter.h:421-429: jitter around coord - 0.5
                    const double px = x - 0.5 + u01(seed, ctr), py = y - 0.5 + u01(seed, ctr + 1), pz = z - 0.5 + u01(seed, ctr + 2);
                    ctr += 3;
                    // PointList::add, fluid.cc:841 (|p| < boundary-2, generalised to lo+2 < p < hi-2)
                    if (px > lo + 2 && px < hi - 2 && py > lo + 2 && py < hi - 2 && pz > lo + 2 && pz < hi - 2) {
                        if (pos) { pos[3 * cnt] = px; pos[3 * cnt + 1] = py; pos[3 * cnt + 2] = pz; }
                        cnt++;
                    }
                }
    return cnt;
}

}  // extern "C"

// ---- phases ---------------------------------------------------------------------------------------
int fl::read_ss(fluid_sim* s)
{
    HIPCHK(hipMemcpyAsync(s->h_ss, s->ss, sizeof(StepState), hipMemcpyDeviceToHost, s->st));
    HIPCHK(hipStreamSynchronize(s->st));
    return FLUID_OK;
}

int fl::clear_dirty(fluid_sim* s)
{
    if (box_empty(s->dirty)) return FLUID_OK;
    const ZeroList z = {{s->container, s->rhs, s->diver, s->diver2}, {s->u, s->v, s->w, s->ub, s->vb, s->wb, s->pressure}};
    launch_zero_fields(s->st, z, s->g, s->dirty, !s->dist && s->row_sweeps);
    HIPCHK(hipGetLastError());
    s->dirty = Box{0, 0, 0, -1, -1, -1};
    return FLUID_OK;
}

// Buckets of the counting sort: the N^3 cells, then "off the grid", then "dead" (multi-GPU migrants).  Only the x planes
// [ax0, ax1] are zeroed, counted into and scanned: a particle moves at most one cell per step (dt <= dx / maxSpeed,
// fluid.cc:992-999), so the planes of the previous bounding box +- 3 hold every particle and every row a P2G window of
// this step can touch.  The bounding box read back below tells whether that held; if not (first step, new particles)
// the sort is simply repeated over the whole grid.
int fl::sort_pass(fluid_sim* s, int ax0, int ax1, int* h_tail)
{
    const Grid g = s->g;
    const long ncell = (long)s->ncell, n2 = g.sx();
    const long c0 = (long)ax0 * n2, c1 = (long)(ax1 + 1) * n2;
    launch_zero_step_state(s->st, s->ss, g.N);
    HIPCHK(hipMemsetAsync(s->cell_count + c0, 0, (c1 - c0) * sizeof(int), s->st));
    HIPCHK(hipMemsetAsync(s->cell_count + ncell, 0, 4 * sizeof(int), s->st));
    launch_bin_count(s->st, g, s->np, s->pa.shifted(s->p_off), s->key, s->slot, s->cell_count, s->ipart, s->ss);
    launch_exclusive_scan(s->st, s->cell_count + c0, s->cell_start + c0, c1 - c0, s->scan_sums, s->cell_start + c1);
    launch_sort_tail(s->st, s->cell_count, s->cell_start, c1, ncell);
    launch_bin_scatter(s->st, s->np, s->key, s->slot, s->cell_start, s->pa.shifted(s->p_off).pid, s->order, s->spid);
    launch_bin_rank(s->st, s->np, 0, s->key, s->cell_start, s->order, s->spid, s->order2);  // every position 0..np (cells, off-grid bucket)
    HIPCHK(hipGetLastError());
    if (h_tail) HIPCHK(hipMemcpyAsync(h_tail, s->cell_start + ncell, 3 * sizeof(int), hipMemcpyDeviceToHost, s->st));   // bucket starts (decomposed run)
    return read_ss(s);
}

static int phase_sort(fluid_sim* s)
{
    const Grid g = s->g;
    int tok = prof_begin(s, FLUID_PROF_SORT, (double)s->np);
    int ax0 = 0, ax1 = g.nx - 1;
    const bool guess = s->sort_hint && !box_empty(s->Pb);
    if (guess) {
        ax0 = std::max(0, s->Pb.x0 - 3);
        ax1 = std::min(g.nx - 1, s->Pb.x1 + 3);
    }
    int rc = sort_pass(s, ax0, ax1);
    if (rc) return rc;
    if (guess && s->h_ss->bbox_max[0] >= 0 && (s->h_ss->bbox_min[0] < std::min(ax0 + 2, s->Pb.x0) || s->h_ss->bbox_max[0] > std::max(ax1 - 2, s->Pb.x1))) {
        if ((rc = sort_pass(s, 0, g.nx - 1))) return rc;   // the guess did not hold
    }
    s->sort_hint = true;
    s->n_out = s->h_ss->n_out;
    s->max_cell = s->h_ss->max_cell;
    const StepState& h = *s->h_ss;
    if (h.bbox_max[0] < 0) {
        s->Pb = Box{0, 0, 0, -1, -1, -1};
    } else {
        s->Pb = Box{h.bbox_min[0], h.bbox_min[1], h.bbox_min[2], h.bbox_max[0], h.bbox_max[1], h.bbox_max[2]};
    }
    if (!box_empty(s->Pb)) {
        s->Rb = clip_dilate(s->Pb, 1, g);
        s->Sb = clip_dilate(s->Pb, 2, g);
    } else {
        s->Rb = s->Sb = s->Pb;
    }
    launch_reorder(s->st, s->np, s->order2, s->pa.shifted(s->p_off), s->pb, s->pw, s->cap);  // + the P2G axis weights
    HIPCHK(hipGetLastError());
    std::swap(s->pa, s->pb);
    s->p_off = 0;
    prof_end(s, FLUID_PROF_SORT, tok);
    s->sorted = true;
    return FLUID_OK;
}

// particle -> grid over box (k_p2g_rows + k_p2g_combine)
int fl::run_p2g(fluid_sim* s, const Box& box)
{
    // The tile form for piled particles and for a mostly empty box (the splash: under 30 % of the box were unknowns last step —
    // the row form pays its per-row latency for thousands of nearly empty rows).  Both inputs are the same on every rank.
    // (a decomposed run takes the GLOBAL active box here, so that every rank picks the form the one-GPU step would)
    const long ref_cells = s->p2g_ref_cells > 0 ? s->p2g_ref_cells : s->Rb.cells();
    const bool airy = s->last_num_active > 0 && (double)s->last_num_active < 0.3 * (double)ref_cells;
    const bool huge = (size_t)12 * sizeof(double) * (size_t)box.cells() > ((size_t)16 << 30);  // the row form's partials: 96 B per box cell
    const bool piled = s->max_cell > P2G_PILED || airy;
    if (s->p2g_force ? s->p2g_force == 2 : huge) {
        s->stats.paths |= FLUID_PATH_P2G_TILES;
        launch_p2g_tiles(s->st, s->g, box, s->pa, s->pw, s->cap, s->cell_start, s->flags, s->container, s->u, s->v, s->w, s->ub, s->vb, s->wb);
        return FLUID_OK;
    }
    // piled particles, mostly-air box: the crowded cells on the matrix cores first (k_p2g_crowd_sum), the rows walk the rest
    const bool crowd = s->p2g_force ? s->p2g_force == 3 : piled;
    if (crowd) {
        s->stats.paths |= FLUID_PATH_P2G_CROWD;
        const size_t nl = (size_t)s->cap / 16 + 64;   // pieces <= particles / 18
        if (nl > s->p2g_crowd_cap) {
            if (s->p2g_crowd) { HIPCHK(hipStreamSynchronize(s->st)); hipFree(s->p2g_crowd); s->p2g_crowd = nullptr; }
            HIPCHK(hipMalloc((void**)&s->p2g_crowd, nl * sizeof(int2)));
            s->p2g_crowd_cap = nl;
        }
    }
    const size_t need = p2g_part_doubles(box);
    if (need > s->p2g_part_cap) {
        if (s->p2g_part) { HIPCHK(hipStreamSynchronize(s->st)); hipFree(s->p2g_part); s->p2g_part = nullptr; }
        HIPCHK(hipMalloc((void**)&s->p2g_part, (need + need / 4) * sizeof(double)));
        s->p2g_part_cap = need + need / 4;
    }
    const size_t ni = 4 + 4 * (size_t)p2g_max_items(box);
    if (ni > s->p2g_items_cap) {
        if (s->p2g_items) { HIPCHK(hipStreamSynchronize(s->st)); hipFree(s->p2g_items); s->p2g_items = nullptr; }
        HIPCHK(hipMalloc((void**)&s->p2g_items, (ni + ni / 4) * sizeof(int)));
        HIPCHK(hipMemsetAsync(s->p2g_items, 0, 4 * sizeof(int), s->st));
        s->p2g_items_cap = ni + ni / 4;
    }
    launch_p2g(s->st, s->g, box, s->pa, s->pw, s->cap, s->cell_start, s->flags, s->p2g_part, s->p2g_items, s->container, s->u, s->v, s->w, s->ub,
               s->vb, s->wb, crowd ? s->p2g_crowd : nullptr, s->pb);
    return FLUID_OK;
}

static int phase_p2g(fluid_sim* s)
{
    HIPCHK(hipSetDevice(s->prm.device));
    int rc = phase_sort(s);
    if (rc) return rc;
    rc = clear_dirty(s);
    if (rc) return rc;
    memset(&s->stats, 0, sizeof(s->stats));
    s->stats.dt_in = s->dt;
    s->stats.dt_out = s->dt;
    for (int a = 0; a < 3; ++a) {
        s->stats.box_lo[a] = (&s->Rb.x0)[a];
        s->stats.box_hi[a] = (&s->Rb.x1)[a];
    }
    if (!box_empty(s->Rb)) {
        int tok = prof_begin(s, FLUID_PROF_P2G, (double)s->Rb.cells());   // k_p2g alone (the weights came with the sort's reorder pass)
        rc = run_p2g(s, s->Rb);
        prof_end(s, FLUID_PROF_P2G, tok);
        if (rc) return rc;
        HIPCHK(hipGetLastError());
        s->dirty = s->Sb;
    }
    s->have_p2g = true;
    s->step_counter++;
    s->have_flags = false;
    return FLUID_OK;
}

template <typename T>
Coef<T> fl::make_coef(const fluid_sim* s)
{
    // setA, fluid.cc:306: scale = dt/(rho dx dx); Adiag accumulates float(Adiag + scale); Aplus = float(-1*scale)
    const double scale = s->dt / (s->prm.rho * s->prm.dx * s->prm.dx);
    Coef<T> c;
    float acc = 0.0f;
    c.diag[0] = 0;
    c.inv[0] = 0;
    for (int k = 1; k <= 6; ++k) {
        acc = (float)((double)acc + scale);
        c.diag[k] = (T)acc;
        c.inv[k] = (T)1 / c.diag[k];
    }
    c.off = (T)(float)(-1 * scale);
    return c;
}
template Coef<double> fl::make_coef<double>(const fluid_sim*);   // fluid_dist.hip links against these
template Coef<float> fl::make_coef<float>(const fluid_sim*);

// Both search vectors of this step, zeroed by one fill (padding and non-unknowns must read 0).
hipError_t fl::zero_search(fluid_sim* s, size_t lb)
{
    const size_t step = (lb + 255) / 256 * 256;
    s->S[1] = (char*)s->S[0] + step;
    return hipMemsetAsync(s->S[0], 0, step + lb, s->st);
}

// ---- multigrid-preconditioned CG (single GPU, fp64) ----------------------------------------------
// (multi-GPU: the same V-cycle applied per slab, neighbour-slab unknowns treated as p = 0: block preconditioner, no halo traffic)

// Level hierarchy of this step: level 0 = the box-local solver layout, coarsened until <= 8^3.
static int gal_build(fluid_sim* s);
static int mg_setup(fluid_sim* s)
{
    s->mgl[0] = mg_level0(s->L);
    int nl = 1;
    while (nl < fluid_sim::MG_MAXL && (nl < 2 || s->mgl[nl - 1].dx > 8 || s->mgl[nl - 1].dy > 8 || s->mgl[nl - 1].dz > 8)) {
        s->mgl[nl] = mg_coarser(s->mgl[nl - 1]);
        ++nl;
    }
    if (s->mgl[nl - 1].dx > 8 || s->mgl[nl - 1].dy > 8 || s->mgl[nl - 1].dz > 8) return fail(FLUID_ERR_STATE, "multigrid: too many levels");
    s->mg_nl = nl;
    // levels small enough for one block (and everything coarser) run inside the tail kernel; level 0 never does
    // (the tail holds its levels in LDS: as many of the coarsest levels as fit)
    int tail = nl - 1;
    const size_t es = s->mg_fp32 ? sizeof(float) : sizeof(double);  // element size of the V-cycle's own arrays
    auto fits = [&](int t) {
        const size_t b = nl - t <= MG_TAIL_MAX ? mg_tail_lds_bytes(nl - t, s->mgl + t, es) : 0;
        return b > 0 && b <= MG_TAIL_LDS;
    };
    while (tail > 1 && fits(tail - 1)) --tail;
    if (!fits(tail)) return fail(FLUID_ERR_STATE, "multigrid: coarsest level does not fit the tail kernel");
    s->mg_tail = tail;
    // One slab for every per-step multigrid array (types, counts, u, v, f, r of each level and z of level 0), zeroed
    // by ONE fill: the layout changes with the box, everything outside the new domain must read as zero / solid, and
    // the V-cycle kernels write unknown cells only.  (30 separate fills cost ~5 us each.)
    size_t total = 0;
    auto take = [&](size_t bytes) { const size_t o = total; total += (bytes + 255) / 256 * 256; return o; };
    size_t o_typ[fluid_sim::MG_MAXL], o_cnt[fluid_sim::MG_MAXL], o_u[fluid_sim::MG_MAXL], o_v[fluid_sim::MG_MAXL], o_f[fluid_sim::MG_MAXL],
        o_r[fluid_sim::MG_MAXL];
    for (int l = 0; l < nl; ++l) {
        const size_t c = s->mgl[l].cells + 64;
        o_typ[l] = take(c); o_cnt[l] = take(c);
        o_u[l] = take(c * es); o_v[l] = take(c * es); o_f[l] = take(c * es); o_r[l] = take(c * es);
    }
    const size_t o_z = take((s->mgl[0].cells + 64) * 8);
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
    }
    s->Zmg = s->mg_slab + o_z;
    HIPCHK(hipMemsetAsync(s->mg_slab, 0, total, s->st));
    launch_mg_type0(s->st, s->g, s->L, s->mgl[0], s->flags, s->cntL, s->mg_typ[0]);
    for (int l = 1; l < nl; ++l) launch_mg_coarsen(s->st, s->mgl[l - 1], s->mg_typ[l - 1], s->mgl[l], s->mg_typ[l], s->mg_cnt[l]);
    { int rcg = gal_build(s); if (rcg) return rcg; }
    HIPCHK(hipGetLastError());
    return FLUID_OK;
}

MgCoef<double> fl::mg_coef(const fluid_sim* s, int level)
{
    MgCoef<double> c;
    if (level == 0) {
        const Coef<double> f = make_coef<double>(s);  // level 0 smooths with the reference's own coefficients
        for (int k = 0; k < 7; ++k) { c.diag[k] = f.diag[k]; c.inv[k] = f.inv[k]; }
        c.off = f.off;
        return c;
    }
    const double scale = s->dt / (s->prm.rho * s->prm.dx * s->prm.dx) / std::pow(4.0, level);  // h doubles per level
    c.off = -scale;
    c.diag[0] = 0; c.inv[0] = 0;
    for (int k = 1; k < 7; ++k) { c.diag[k] = scale * k; c.inv[k] = 1.0 / c.diag[k]; }
    return c;
}

// z = M^-1 r: V(2,2) cycle.  Level-0 rhs = `rhs0`; result in `z0`; part_rz gets the partials of rhs0.z0.
static int mg_rz_blocks(const fluid_sim* s) { return s->lists_on ? s->n_tl_mg : mg_up_blocks(s->mgl[0]); }

// One launch per leg and level (LDS-tiled kernels): down = both pre-sweeps + residual (+ the restriction for the
// levels in the middle), tail, up = prolongation + both post-sweeps.  V = the cycle's own arithmetic and storage type:
// float by default — M^-1 only has to be a fixed SPD operator close to A^-1, the PCG vectors, dot products and the
// residual recurrence around it stay double (same iteration counts as a double cycle, half the LDS and L2 traffic).
template <typename V>
static MgCoef<V> mg_coef_as(const fluid_sim* s, int level)
{
    const MgCoef<double> d = mg_coef(s, level);
    MgCoef<V> c;
    for (int k = 0; k < 7; ++k) { c.diag[k] = (V)d.diag[k]; c.inv[k] = (V)d.inv[k]; }
    c.off = (V)d.off;
    return c;
}
// (re)build the Galerkin coefficients of this step's levels (they carry dt: rebuilt by the solve if dt has changed since)
static int gal_build(fluid_sim* s)
{
    // Mostly-air box: the coarse levels of the cycle as Galerkin operators by aggregation (kernels_gal.hip) — the free surface stays where
    // it is on every level.  Level 0 with its own restriction launch only (big boxes), float cycle, one GPU.
    s->gal = false;
    s->gal_eligible = false;
    const int nl = s->mg_nl;
    int lc = 1;
    while (lc < nl - 1 && !gal_fits_coarsest(s->mgl[lc])) ++lc;
    if (!(s->gal_mode && s->lists_on && s->mg_fp32 && !s->dist && (long)s->mgl[0].dx * s->mgl[0].dy * s->mgl[0].dz > 200000 &&
          gal_fits_coarsest(s->mgl[lc]) && lc >= 2)) {
        s->gal_it[0] = s->gal_it[1] = -1;   // (measured again when the box is mostly air again)
        return FLUID_OK;
    }
    // Which cycle?  Aggregation wins where the re-discretised levels lose much of the water — rough, filmy pools and splashes: 22 iterations
    // against 31 at step 445 of the 256^3 drop, 61 against 72 (three passes) at step 210 — and loses on dense or flat water (30 against 21
    // in free fall, +8 % on a flat slab); the share of level-1 cells the re-discretised rule keeps does not separate the splash (0.80) from the
    // slab (0.8).  Both cycles give the same pressure, so the step simply measures: the first-pass iteration count of each is kept, the
    // cheaper one (count x cost per iteration) is used, and the other is looked at again every 32nd step.
    s->gal_eligible = true;
    bool use = s->gal_mode >= 2;
    if (s->gal_mode == 1) {
        if (s->gal_it[0] < 0) use = false;
        else if (s->gal_it[1] < 0) use = true;
        else {
            // counts weighted by the measured cost of an iteration (175 us against 170 at step 445, 512-thread k_gal_* legs + a one-block coarsest
            // level against the tail kernel: +3..5 %, profiles/r03/NOTES.md) — a fixed weight, not a timer: the choice stays a function of the state
            use = 21 * s->gal_it[1] < 20 * s->gal_it[0];
            if (++s->gal_since_probe >= 32) { use = !use; s->gal_since_probe = 0; }
        }
    }
    if (!use) return FLUID_OK;
    size_t total = 0, off[fluid_sim::MG_MAXL][6];
    for (int l = 1; l <= lc; ++l)
        for (int q = 0; q < 6; ++q) {
            off[l][q] = total;
            total += (q < 5 ? ((q < 4 ? sizeof(float) : 1) * (s->mgl[l].cells + 64) + 255) / 256 * 256 : ((size_t)gal_tile_count(s->mgl[l]) + 255) / 256 * 256);
        }
    if (total > s->gal_slab_cap) {
        if (s->gal_slab) { HIPCHK(hipStreamSynchronize(s->st)); hipFree(s->gal_slab); s->gal_slab = nullptr; }
        HIPCHK(hipMalloc((void**)&s->gal_slab, total + total / 4));
        s->gal_slab_cap = total + total / 4;
    }
    HIPCHK(hipMemsetAsync(s->gal_slab, 0, total, s->st));
    for (int l = 1; l <= lc; ++l) {
        for (int q = 0; q < 4; ++q) s->gal_c[l][q] = (float*)(s->gal_slab + off[l][q]);
        s->gal_cnt[l] = (uint8_t*)(s->gal_slab + off[l][4]);
        s->gal_tfl[l] = (uint8_t*)(s->gal_slab + off[l][5]);
    }
    launch_gal_level1(s->st, s->mgl[0], s->cntL, mg_coef_as<float>(s, 0), s->mgl[1], s->gal_c[1][0], s->gal_c[1][1], s->gal_c[1][2], s->gal_c[1][3], s->gal_cnt[1]);
    for (int l = 2; l <= lc; ++l)
        launch_gal_coarsen(s->st, s->mgl[l - 1], s->gal_c[l - 1][0], s->gal_c[l - 1][1], s->gal_c[l - 1][2], s->gal_c[l - 1][3], s->gal_cnt[l - 1], s->mgl[l],
                           s->gal_c[l][0], s->gal_c[l][1], s->gal_c[l][2], s->gal_c[l][3], s->gal_cnt[l]);
    for (int l = 1; l < lc; ++l) launch_gal_tile_flags(s->st, s->mgl[l], s->gal_cnt[l], s->gal_tfl[l]);
    s->gal = true;
    s->gal_lc = lc;
    s->gal_dt = s->dt;
    HIPCHK(hipGetLastError());
    return FLUID_OK;
}
// The cycle with Galerkin coarse levels (kernels_gal.hip): level 0 by the kernels of kernels_mg.hip (its own coefficients; the up leg
// takes the parent's value as the correction), levels 1 .. gal_lc - 1 by k_gal_down / k_gal_up, the coarsest by one block.
static int mg_vcycle_gal(fluid_sim* s, const double* rhs0, float* z0, double* part_rz)
{
    typedef float V;
    const PcgState* ps = s->ps;
    const int lc = s->gal_lc;
    auto U = [&](int l) { return (V*)s->mg_u[l]; };
    auto W = [&](int l) { return (V*)s->mg_v[l]; };
    auto F = [&](int l) { return (V*)s->mg_f[l]; };
    auto C = [&](int l, int q) { return (const float*)s->gal_c[l][q]; };
    const MLevel& m0 = s->mgl[0];
    // (the level-0 down leg forms the coarse right-hand side itself: the sum of the residual over each coarse cell's children, all inside its tile)
    launch_mg_down<V, double>(s->st, m0, s->cntL, rhs0, U(0), (V*)s->mg_r[0], s->mgl[1], nullptr, F(1), mg_coef_as<V>(s, 0), ps, s->tl_mg, s->n_tl_mg, true);
    for (int l = 1; l < lc; ++l) launch_gal_down(s->st, s->mgl[l], s->gal_tfl[l], C(l, 0), C(l, 1), C(l, 2), C(l, 3), F(l), U(l), s->mgl[l + 1], F(l + 1), ps);
    launch_gal_coarsest(s->st, s->mgl[lc], C(lc, 0), C(lc, 1), C(lc, 2), C(lc, 3), F(lc), U(lc), s->gal_sweeps, ps);
    for (int l = lc - 1; l >= 1; --l)
        launch_gal_up(s->st, s->mgl[l], s->gal_tfl[l], C(l, 0), C(l, 1), C(l, 2), C(l, 3), F(l), U(l), W(l), s->mgl[l + 1], l + 1 == lc ? U(l + 1) : W(l + 1),
                      (float)s->gal_wc, ps);
    const int tok = prof_begin(s, FLUID_PROF_MG_UP0, (double)s->Rb.cells());
    launch_mg_up<V, double, float>(s->st, m0, s->cntL, rhs0, (const V*)U(0), z0, s->mgl[1], W(1), mg_coef_as<V>(s, 0), part_rz, ps, s->gal_wc, s->tl_mg, s->n_tl_mg,
                                   nullptr, 1);
    prof_end(s, FLUID_PROF_MG_UP0, tok);
    HIPCHK(hipGetLastError());
    return FLUID_OK;
}

// z0: the result in the cycle's own type — a float cycle's z stays float in memory (k_pcg_sq_l converts, exactly)
template <typename V>
static int mg_vcycle_t(fluid_sim* s, const double* rhs0, V* z0, double* part_rz)
{
    const int nl = s->mg_nl, tail = s->mg_tail;
    const PcgState* ps = s->ps;
    auto U = [&](int l) { return (V*)s->mg_u[l]; };
    auto W = [&](int l) { return (V*)s->mg_v[l]; };
    auto F = [&](int l) { return (V*)s->mg_f[l]; };
    auto R = [&](int l) { return (V*)s->mg_r[l]; };
    for (int l = 0; l < tail; ++l) {
        const MLevel& m = s->mgl[l];
        // restriction inside the down kernel (halo 3): not at a big level 0, where 5x halo reads cost more than a launch
        const bool fold = l > 0 || (long)m.dx * m.dy * m.dz <= 200000;
        V* fc = fold ? F(l + 1) : nullptr;
        const uint8_t* cc = fold ? s->mg_cnt[l + 1] : nullptr;
        const bool lst = s->lists_on && !fold;
        if (l == 0) launch_mg_down<V, double>(s->st, m, s->cntL, rhs0, U(0), R(0), s->mgl[1], cc, fc, mg_coef_as<V>(s, 0), ps, lst ? s->tl_mg : nullptr, s->n_tl_mg);
        else launch_mg_down<V, V>(s->st, m, s->mg_cnt[l], (const V*)F(l), U(l), R(l), s->mgl[l + 1], cc, fc, mg_coef_as<V>(s, l), ps);
        if (!fold) launch_mg_restrict<V>(s->st, m, (const V*)R(l), s->mgl[l + 1], s->mg_cnt[l + 1], F(l + 1), ps);
    }
    {
        V off[fluid_sim::MG_MAXL];
        for (int l = tail; l < nl; ++l) off[l] = (V)mg_coef(s, l).off;
        launch_mg_tail<V>(s->st, nl - tail, (const V*)F(tail), s->mgl + tail, s->mg_cnt + tail, U(tail), off + tail, s->mg_csweeps, ps, s->mg_wc[3]);
    }
    for (int l = tail - 1; l >= 0; --l) {
        const MLevel& m = s->mgl[l];
        const V* ec = l + 1 == tail ? U(l + 1) : W(l + 1);  // out != u: neighbouring tiles still read u
        if (l == 0) {
            const int tok = prof_begin(s, FLUID_PROF_MG_UP0, (double)s->Rb.cells());
            launch_mg_up<V, double, V>(s->st, m, s->cntL, rhs0, (const V*)U(0), z0, s->mgl[1], ec, mg_coef_as<V>(s, 0), part_rz, ps, s->mg_wc[0],
                                       s->lists_on ? s->tl_mg : nullptr, s->n_tl_mg);
            prof_end(s, FLUID_PROF_MG_UP0, tok);
        } else {
            launch_mg_up<V, V, V>(s->st, m, s->mg_cnt[l], (const V*)F(l), (const V*)U(l), W(l), s->mgl[l + 1], ec, mg_coef_as<V>(s, l), nullptr, ps, s->mg_wc[l == 1 ? 1 : 2]);
        }
    }
    HIPCHK(hipGetLastError());
    return FLUID_OK;
}
static int mg_vcycle(fluid_sim* s, const double* rhs0, void* z0, double* part_rz)
{
    if (s->gal && s->lists_on) return mg_vcycle_gal(s, rhs0, (float*)z0, part_rz);
    return s->mg_fp32 ? mg_vcycle_t<float>(s, rhs0, (float*)z0, part_rz) : mg_vcycle_t<double>(s, rhs0, (double*)z0, part_rz);
}

// PCG loop of ConjugateGradient.h:28-90 with z = V-cycle(r); same start, stopping rule and cap as solve_impl.
static int solve_mg(fluid_sim* s)
{
    typedef double T;
    const Grid g = s->g;
    const LBox L = s->L;
    T* X = (T*)s->X;
    T* R = (T*)s->R;
    T* Q = (T*)s->Q;
    T* Sx[2] = {(T*)s->S[0], (T*)s->S[1]};
    T* Z = (T*)s->Zmg;  // z lives in its own level-0 array: the V-cycle uses mg_u[0]/mg_v[0]/mg_r[0] as scratch and writes z last
    const uint8_t* cnt = s->cntL;
    const Coef<T> cf = make_coef<T>(s);
    const double tol = s->prm.cg_tol;
    long max_it = s->prm.cg_max_iters > 0 ? s->prm.cg_max_iters : 2 * (long)s->stats.num_active;
    if (max_it < 1) max_it = 1;
    const double cells = (double)s->Rb.cells();
    const int sparse = (double)s->stats.num_active < 0.4 * (double)L.cells();  // mostly-air box: SQ / XR test the counts before loading
    const bool lists = s->lists_on;          // ... or, with the active-tile lists of this step, sweep those tiles only
    const bool rows = lists && s->rows_on && s->n_rows > 0;   // XR over the non-empty z rows instead of the SQ tiles
    // partials of the init kernel / of a list-mode XR launch (what the next SQ launch re-sums)
    const int n_init = pcg_xr_blocks(L), n_list = rows ? pcg_rows_blocks(s->n_rows) : pcg_list_blocks(s->n_tl_sq);
    // r.z partials come from the level-0 up leg, one per block; every block of the PCG kernels re-sums them from L2.
    // Above 1024 values one extra 1-block launch folds them into a single value first.
    const int n_rz_raw = mg_rz_blocks(s);
    const bool fold = n_rz_raw > 1024;
    const int n_rz = fold ? 1 : n_rz_raw;
    int rc;
    int tsolve = prof_begin(s, FLUID_PROF_SOLVE, cells);
    if (s->gal && s->gal_dt != s->dt) {   // dt changed since the coefficients were built (phase API): same decision, new coefficients
        const int keep = s->gal_mode; s->gal_mode = 2; rc = gal_build(s); s->gal_mode = keep;
        if (rc) return rc;
    }
    if (s->gal) s->stats.paths |= FLUID_PATH_MG_GALERKIN;
    // Start: x = 0 like the reference's cg.solve(b) — or, by default, the previous solve's pressure (Eigen's solveWithGuess
    // form of the same loop: r0 = b - A x0, same threshold tol^2 |b|^2).  The converged p does not depend on the start
    // beyond the tolerance; a settled pool needs far fewer iterations.  r0.r0 partials travel in part_rz[1] (unused by body 0).
    const bool guess = s->warm && s->have_guess;
    const double *g1, *g2;
    double gca, gcb;
    s->start_guess(g1, g2, gca, gcb);
    if (guess) launch_pcg_init_guess<T>(s->st, g, L, cnt, s->diver, g1, g2, gca, gcb, X, R, cf, s->part_bb, s->part_rz[1], s->ps);
    else launch_pcg_init<T>(s->st, g, L, cnt, s->diver, X, R, cf, s->part_bb, s->part_rr, s->ps);  // (its Jacobi r.z partials are unused)
    long it = 0;
    // Batching of the convergence poll.  Iteration counts barely change from one solve to the next (Eigen's count i means
    // i + 1 bodies ran), so the first batch runs exactly the bodies the previous solve needed without looking — one poll
    // (head-only launch + 40-byte copy + stream sync, ~50 us) per solve when the count repeats; after that every body is
    // polled.  Kernels of a finished solve exit at their first instruction, but each still costs a launch.
    const int pclass = s->pass_class();
    long batch = s->mg_last_iters_k[pclass] > 5 ? s->mg_last_iters_k[pclass] + 1 : 4;
    bool done = false;
    // SQ of one body: s' = z + beta s into Sx[cur], q = A s'.  z is in the cycle's own type (float by default)
    const bool zf = s->mg_fp32 || (s->gal && s->lists_on);
    auto sq = [&](int cur, const double* part_rr, int n_prev, const double* rz_new, const double* rz_old, int first, int sp) {
        const int prv = cur ^ 1;
        if (zf)
            launch_pcg_sq_zf(s->st, L, cnt, (const float*)s->Zmg, Sx[prv], Sx[cur], Q, cf, part_rr, n_prev, rz_new, rz_old, s->part_pq, s->ps, first, tol, n_rz, sp,
                             lists ? s->tl_sq : nullptr, s->n_tl_sq);
        else if (lists)
            launch_pcg_sq_list<T>(s->st, L, cnt, Z, Sx[prv], Sx[cur], Q, cf, part_rr, n_prev, rz_new, rz_old, s->part_pq, s->ps, first, tol, n_rz, 1, s->tl_sq, s->n_tl_sq);
        else
            launch_pcg_sq<T>(s->st, L, cnt, Z, Sx[prv], Sx[cur], Q, cf, part_rr, rz_new, rz_old, s->part_pq, s->ps, first, tol, n_rz, 1, sp, n_prev);
    };
    while (!done) {
        for (long k = 0; k < batch && it < max_it; ++k, ++it) {
            const int cur = (int)(it & 1), prv = cur ^ 1;
            if ((rc = mg_vcycle(s, R, s->Zmg, fold ? s->mg_part : s->part_rz[cur]))) return rc;
            if (fold) launch_sum2(s->st, s->mg_part, n_rz_raw, s->mg_part, 0, s->part_rz[cur], nullptr);
            int tok = prof_begin(s, FLUID_PROF_PCG_SQ, cells);
            sq(cur, it == 0 ? s->part_bb : s->part_rr, it == 0 ? n_init : (lists ? n_list : n_init), s->part_rz[cur], s->part_rz[prv], it == 0 ? (guess ? 2 : 1) : 0,
               sparse);
            prof_end(s, FLUID_PROF_PCG_SQ, tok);
            tok = prof_begin(s, FLUID_PROF_PCG_XR, cells);
            if (rows)
                launch_pcg_xr_rows<T>(s->st, L, cnt, X, R, Sx[cur], Q, cf, s->part_rz[cur], n_rz, s->part_pq, pcg_list_blocks(s->n_tl_sq), s->part_rr,
                                      s->part_err, s->ps, s->row_list, s->n_rows);
            else if (lists)
                launch_pcg_xr_list<T>(s->st, L, cnt, X, R, Sx[cur], Q, cf, s->part_rz[cur], n_rz, s->part_pq, s->part_rr, s->part_err, s->ps,
                                      s->tl_sq, s->n_tl_sq);
            else
                launch_pcg_xr<T>(s->st, L, cnt, X, R, Sx[cur], Q, cf, s->part_rz[cur], s->part_pq, s->part_rr, s->part_err, s->ps, n_rz, sparse);
            prof_end(s, FLUID_PROF_PCG_XR, tok);
        }
        HIPCHK(hipGetLastError());
        if (it < max_it) {
            // the break test of the last body sits at the head of the next SQ launch: a head-only launch (its s'/q are
            // overwritten by the real launch of that iteration if the solve goes on; the counter is put back below)
            const int cur = (int)(it & 1), prv = cur ^ 1;
            sq(cur, s->part_rr, lists ? n_list : n_init, s->part_rz[prv], s->part_rz[prv], 0, 0);
        }
        HIPCHK(hipMemcpyAsync(&s->h_ps[0], s->ps, sizeof(PcgState), hipMemcpyDeviceToHost, s->st));
        HIPCHK(hipStreamSynchronize(s->st));
        done = s->h_ps[0].done || it >= max_it;
        if (!done) {
            s->h_ps[1] = s->h_ps[0];   // pinned slot: stays valid while the copy is in flight
            s->h_ps[1].iters -= 1;     // the head-only launch counted a body that the next real launch counts again
            HIPCHK(hipMemcpyAsync(s->ps, &s->h_ps[1], sizeof(PcgState), hipMemcpyHostToDevice, s->st));
        }
        batch = 1;
    }
    int iters = s->h_ps->iters;
    const double rr = s->h_ps->rr;
    if (!s->h_ps->done) iters = (int)max_it;
    // the new solution goes into the buffer of the older guess, which then becomes the latest
    launch_store_pressure<T>(s->st, g, L, cnt, X, s->pressure, s->warm ? s->p_guess2 : nullptr, s->ps);
    launch_drop_solve(s->st, g, L, s->n_drop, s->drop_ctr + 64 * DROP_NCTR, s->drop_n, s->drop_cells, s->flags, s->diver, make_coef<double>(s), tol,
                      s->pressure, s->warm ? s->p_guess2 : nullptr, &s->ss->n_drop_fail);
    if (s->warm) s->rotate_guess();
    s->have_guess = s->warm;
    if (s->make_q()) {   // (one GPU and the replicated solve of a multi-GPU run; the decomposed step forms it after its halo exchange) q = p_1 - (1 - f) p_0 over the box, for the next steps' second passes (start_guess)
        launch_axpby_box(s->st, g, s->Rb, 1.0, s->p_guess, -(1.0 - s->prm.update_frac), s->p_guess2, s->p_q);
        s->q_step = s->step_counter;
    }
    HIPCHK(hipGetLastError());
    prof_end(s, FLUID_PROF_SOLVE, tsolve);
    s->stats.cg_iters_last = iters;
    s->stats.cg_iters += iters;
    s->mg_last_iters_k[pclass] = iters;
    if (s->gal_eligible && s->stats.outer_passes == 0) s->gal_it[s->gal ? 1 : 0] = iters;   // the first pass of the step: what the two cycles are compared by
    s->stats.relres = s->h_ps->bb > 0 ? std::sqrt(rr / s->h_ps->bb) : 0.0;
    if (s->h_ps->breakdown) return fail(FLUID_ERR_SOLVER, "PCG breakdown: s.As <= 0 or NaN");
    return FLUID_OK;
}

int fl::phase_flags(fluid_sim* s)
{
    HIPCHK(hipSetDevice(s->prm.device));
    // container > 0 only inside the active box, and outside [flag_x0, flag_x1] the flags / indices still hold what an
    // empty cell gets: sweep the x planes of this box and of the previous pass only (the unknown numbering is a prefix
    // count in linear order, so a contiguous plane range that holds every fluid cell numbers them like the full sweep)
    int x0 = 0, x1 = s->g.nx - 1;
    const bool none = box_empty(s->Sb);
    if (s->flags_valid) {
        x0 = none ? s->flag_x0 : (s->flag_x1 < s->flag_x0 ? s->Sb.x0 : std::min(s->Sb.x0, s->flag_x0));
        x1 = none ? s->flag_x1 : (s->flag_x1 < s->flag_x0 ? s->Sb.x1 : std::max(s->Sb.x1, s->flag_x1));
    }
    if (x1 >= x0) {
        launch_flags(s->st, s->g, s->solid, s->container, s->flags, x0, x1);
        launch_index_scan_range(s->st, s->g, s->flags, s->indices, s->scan_sums, &s->ss->num_active, x0, x1);
    } else {
        HIPCHK(hipMemsetAsync(&s->ss->num_active, 0, sizeof(int), s->st));
    }
    s->flag_x0 = none ? 0 : s->Sb.x0;
    s->flag_x1 = none ? -1 : s->Sb.x1;
    s->flags_valid = true;
    HIPCHK(hipGetLastError());
    bool built = false, drops = false;
    s->n_drop = 0;
    if (!box_empty(s->Rb)) {
        // box-local solver layout of this step: diag counts (and, in a mostly-air box, the lists of the tiles that hold an
        // unknown: their lengths come back with num_active in the one read below)
        s->L = make_lbox(s->Rb);
        launch_cnt_local(s->st, s->g, s->L, s->flags, s->cntL);
        if (use_mg(s) && (s->lists_force >= 0 ? s->lists_force == 1 : s->lists_hint)) {
            const MLevel m0 = mg_level0(s->L);
            const int n_mg = mg_up_blocks(m0), n_sq = sq_tile_count(s->L);
            const size_t need = (size_t)n_mg + n_sq;
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
            // the airborne droplets leave the system first: tiles, rows and coarse levels are built without them
            // (a few hundred droplets do not pay for the search: while the last search found fewer than drop_min, look every 8th
            // step only — the pressure is the same either way)
            if (s->drops_on && !s->dist && (s->drop_last < 0 || s->drop_last >= s->drop_min || ++s->drop_skipped >= 8)) {
                s->drop_skipped = 0;
                if (!s->drop_cells) {
                    HIPCHK(hipMalloc((void**)&s->drop_ctr, (size_t)(64 * DROP_NCTR + DROP_NCTR + 1) * sizeof(int)));
                    HIPCHK(hipMalloc((void**)&s->drop_n, (size_t)DROP_CAP * sizeof(int)));
                    HIPCHK(hipMalloc((void**)&s->drop_cells, (size_t)DROP_CAP * 64 * sizeof(int)));
                }
                launch_drop_find(s->st, s->L, s->cntL, s->drop_ctr, s->drop_ctr + 64 * DROP_NCTR, &s->ss->n_drop, s->drop_n, s->drop_cells);
                drops = true;
            }
            launch_mg_tile_flags(s->st, m0, s->cntL, s->tl_flags);
            launch_compact_flags(s->st, s->tl_flags, n_mg, s->tl_mg, &s->ss->n_tl_mg);
            launch_sq_tile_flags(s->st, s->L, s->cntL, s->tl_flags + n_mg);
            launch_compact_flags(s->st, s->tl_flags + n_mg, n_sq, s->tl_sq, &s->ss->n_tl_sq);
            if (s->rows_on) {
                const size_t nr = (size_t)pcg_row_count(s->L);
                if (nr > s->row_cap) {
                    HIPCHK(hipStreamSynchronize(s->st));
                    hipFree(s->row_flags); hipFree(s->row_pos); hipFree(s->row_list);
                    s->row_flags = s->row_pos = s->row_list = nullptr;
                    const size_t cap = nr + nr / 4 + 1024;
                    HIPCHK(hipMalloc((void**)&s->row_flags, cap * sizeof(int)));
                    HIPCHK(hipMalloc((void**)&s->row_pos, cap * sizeof(int)));
                    HIPCHK(hipMalloc((void**)&s->row_list, cap * sizeof(int)));
                    s->row_cap = cap;
                }
                launch_row_list(s->st, s->L, s->cntL, s->row_flags, s->row_pos, s->row_list, s->scan_sums, &s->ss->n_rows);
            }
            built = true;
        }
        HIPCHK(hipGetLastError());
    }
    int rc = read_ss(s);
    if (rc) return rc;
    s->stats.num_active = s->h_ss->num_active;
    s->last_num_active = s->stats.num_active;
    s->lists_on = false;
    s->lists_hint = false;
    if (!box_empty(s->Rb)) {
        // mostly air and big enough for the sweep over empty tiles to matter (a dense box keeps the XCD-ordered dense launches)
        const bool airy = (double)s->stats.num_active < 0.45 * (double)s->L.cells() && s->L.cells() > (size_t)1500000;
        s->lists_hint = airy;
        s->n_tl_mg = s->h_ss->n_tl_mg; s->n_tl_sq = s->h_ss->n_tl_sq; s->n_rows = s->h_ss->n_rows;
        s->n_drop = drops ? std::min(s->h_ss->n_drop, DROP_CAP) : 0;
        if (drops) s->drop_last = s->n_drop;
        if (s->n_drop > 0) s->stats.paths |= FLUID_PATH_DROPLETS;
        s->lists_on = built && s->n_tl_mg > 0 && s->n_tl_sq > 0 && (s->lists_force == 1 || airy);
        if (s->lists_on) s->stats.paths |= FLUID_PATH_TILE_LISTS;
        const size_t lb = (s->L.cells() + 2 * (size_t)s->L.Lz) * solver_elem(s);  // incl. the spare wrap rows
        HIPCHK(zero_search(s, lb));
        HIPCHK(hipGetLastError());
        if (use_mg(s)) {
            int rc2 = mg_setup(s);
            if (rc2) return rc2;
        }
    }
    s->have_flags = true;
    return FLUID_OK;
}

static int phase_rhs_div(fluid_sim* s, int which)
{
    if (!s->have_flags) return fail(FLUID_ERR_STATE, "rhs_div before flags_index");
    if (box_empty(s->Rb)) return FLUID_OK;
    const double dt = s->dt;
    launch_rhs_div(s->st, s->g, s->Rb, s->flags, s->u, s->v, s->w, s->rhs, which ? s->diver2 : s->diver, s->prm.dx,
                   s->prm.gravity[0] * dt, s->prm.gravity[1] * dt, s->prm.gravity[2] * dt, !s->dist && s->row_sweeps);  // gravity*dt, fluid.cc:420
    HIPCHK(hipGetLastError());
    return FLUID_OK;
}

template <typename T>
static int solve_impl(fluid_sim* s)
{
    const Grid g = s->g;
    const LBox L = s->L;
    T* X = (T*)s->X;
    T* R = (T*)s->R;
    T* Q = (T*)s->Q;
    T* Sx[2] = {(T*)s->S[0], (T*)s->S[1]};
    const uint8_t* cnt = s->cntL;
    const Coef<T> cf = make_coef<T>(s);
    const double tol = s->prm.cg_tol;
    long max_it = s->prm.cg_max_iters > 0 ? s->prm.cg_max_iters : 2 * (long)s->stats.num_active;  // IterativeSolverBase.h:362
    if (max_it < 1) max_it = 1;
    const double cells = (double)s->Rb.cells();  // units one launch processes = cells of the active box
    int tsolve = prof_begin(s, FLUID_PROF_SOLVE, cells);
    launch_pcg_init<T>(s->st, g, L, cnt, s->diver, X, R, cf, s->part_bb, s->part_rz[0], s->ps);
    // body 0
    int tok = prof_begin(s, FLUID_PROF_PCG_SQ, cells);
    launch_pcg_sq<T>(s->st, L, cnt, R, (const T*)nullptr, Sx[0], Q, cf, s->part_bb, nullptr, nullptr, s->part_pq, s->ps, 1, tol);
    prof_end(s, FLUID_PROF_PCG_SQ, tok);
    tok = prof_begin(s, FLUID_PROF_PCG_XR, cells);
    launch_pcg_xr<T>(s->st, L, cnt, X, R, Sx[0], Q, cf, s->part_rz[0], s->part_pq, s->part_rr, s->part_rz[1], s->ps);
    prof_end(s, FLUID_PROF_PCG_XR, tok);
    long it = 1;
    const int CHECK = 16;
    // The done flag is polled ONE BATCH BEHIND: batch b+1 is already queued when the host waits for the
    // state copy of batch b, so the stream never drains between batches (kernels of a finished solve
    // exit at their first instruction).
    bool done = false;
    int nb = 0;
    while (!done) {
        for (int k = 0; k < CHECK && it < max_it; ++k, ++it) {
            const int cur = (int)(it & 1), prv = cur ^ 1;
            tok = prof_begin(s, FLUID_PROF_PCG_SQ, cells);
            launch_pcg_sq<T>(s->st, L, cnt, R, Sx[prv], Sx[cur], Q, cf, s->part_rr, s->part_rz[cur], s->part_rz[prv], s->part_pq, s->ps, 0, tol);
            prof_end(s, FLUID_PROF_PCG_SQ, tok);
            tok = prof_begin(s, FLUID_PROF_PCG_XR, cells);
            launch_pcg_xr<T>(s->st, L, cnt, X, R, Sx[cur], Q, cf, s->part_rz[cur], s->part_pq, s->part_rr, s->part_rz[prv], s->ps);
            prof_end(s, FLUID_PROF_PCG_XR, tok);
        }
        HIPCHK(hipGetLastError());
        HIPCHK(hipMemcpyAsync(&s->h_ps[nb & 1], s->ps, sizeof(PcgState), hipMemcpyDeviceToHost, s->st));
        HIPCHK(hipEventRecord(s->ev_poll[nb & 1], s->st));
        if (nb >= 1) {
            HIPCHK(hipEventSynchronize(s->ev_poll[(nb - 1) & 1]));
            if (s->h_ps[(nb - 1) & 1].done) done = true;
        }
        if (it >= max_it) done = true;
        ++nb;
    }
    HIPCHK(hipMemcpyAsync(&s->h_ps[0], s->ps, sizeof(PcgState), hipMemcpyDeviceToHost, s->st));
    HIPCHK(hipStreamSynchronize(s->st));
    int iters = s->h_ps->iters;
    const double rr = s->h_ps->rr;
    if (!s->h_ps->done) {
        // cap reached without the break: Eigen leaves the loop with i == maxIters (ConjugateGradient.h:66)
        iters = (int)max_it;
    }
    launch_store_pressure<T>(s->st, g, L, cnt, X, s->pressure);
    launch_drop_solve(s->st, g, L, s->n_drop, s->drop_ctr + 64 * DROP_NCTR, s->drop_n, s->drop_cells, s->flags, s->diver, make_coef<double>(s), tol,
                      s->pressure, nullptr, &s->ss->n_drop_fail);
    HIPCHK(hipGetLastError());
    prof_end(s, FLUID_PROF_SOLVE, tsolve);
    s->stats.cg_iters_last = iters;
    s->stats.cg_iters += iters;
    s->stats.relres = s->h_ps->bb > 0 ? std::sqrt(rr / s->h_ps->bb) : 0.0;
    if (s->h_ps->breakdown) return fail(FLUID_ERR_SOLVER, "PCG breakdown: s.As <= 0 or NaN (matrix not SPD for these flags)");
    return FLUID_OK;
}

static int phase_solve(fluid_sim* s)
{
    if (!s->have_flags) return fail(FLUID_ERR_STATE, "solve before flags_index");
    if (box_empty(s->Rb)) return FLUID_OK;
    if (use_mg(s) && s->mg_nl >= 2) return solve_mg(s);  // a box that is already <= 8^3 has no coarser level: Jacobi
    return s->prm.precision == FLUID_PRECISION_FP32 ? solve_impl<float>(s) : solve_impl<double>(s);
}

static int phase_vel_update(fluid_sim* s)
{
    if (!s->have_flags) return fail(FLUID_ERR_STATE, "vel_update before flags_index");
    if (box_empty(s->Sb)) return FLUID_OK;
    const double dtp = s->dt * s->prm.update_frac;      // dt/10, fluid.cc:1475
    const double k = dtp / (s->prm.rho * s->prm.dx);    // :614
    launch_vel_update(s->st, s->g, s->Sb, s->flags, s->pressure, s->u, s->v, s->w, k, s->prm.gravity[0] * dtp, s->prm.gravity[1] * dtp,
                      s->prm.gravity[2] * dtp, !s->dist && s->row_sweeps);         // gravity*dt, :638
    HIPCHK(hipGetLastError());
    return FLUID_OK;
}

int fl::phase_pressure_pass(fluid_sim* s, double* error)
{
    int rc;
    if ((rc = phase_rhs_div(s, 0))) return rc;
    if ((rc = phase_solve(s))) return rc;
    if ((rc = phase_vel_update(s))) return rc;
    if ((rc = phase_rhs_div(s, 1))) return rc;
    double err = NAN;
    if (!box_empty(s->Rb)) {
        launch_err_norm(s->st, s->g, s->Rb, s->flags, s->diver, s->diver2, s->part_err, s->ss);
        HIPCHK(hipGetLastError());
        if ((rc = read_ss(s))) return rc;
        if (s->h_ss->n_drop_fail > 0) s->stats.paths |= FLUID_PATH_DROPLETS_SHORT;   // a pocket solved apart stopped short of the tolerance: reported, not hidden
        err = std::sqrt(s->h_ss->err_num) / std::sqrt(s->h_ss->err_den);  // fluid.cc:1483
    }
    s->stats.error = err;
    s->stats.outer_passes++;
    if (error) *error = err;
    return FLUID_OK;
}

// PIC blend (flip_blend < 1): three more cell fields, allocated when first needed
int fl::pic_fields(fluid_sim* s)
{
    if (s->prm.flip_blend >= 1.0 || s->pcx) return FLUID_OK;
    double** f[3] = {&s->pcx, &s->pcy, &s->pcz};
    for (double** q : f) {
        HIPCHK(hipMalloc((void**)q, s->ncell * sizeof(double)));
        HIPCHK(hipMemsetAsync(*q, 0, s->ncell * sizeof(double), s->st));
    }
    return FLUID_OK;
}

static int phase_flip_advect(fluid_sim* s)
{
    if (!s->have_p2g) return fail(FLUID_ERR_STATE, "flip_advect before p2g");
    HIPCHK(hipSetDevice(s->prm.device));
    int rcp = pic_fields(s);
    if (rcp) return rcp;
    if (!box_empty(s->Rb)) launch_flip_delta(s->st, s->g, s->Rb, s->u, s->v, s->w, s->ub, s->vb, s->wb, s->dcx, s->dcy, s->dcz, s->pcx, s->pcy, s->pcz);
    int tok = prof_begin(s, FLUID_PROF_G2P, (double)s->np);
    if (s->sorted && s->p_off == 0 && (double)s->np >= 4.0 * (double)s->Pb.cells()) {
        // sorted by base cell and not moved since, and the bounding box is densely filled (the falling cube: 8 per cell;
        // the splash: 0.7 per cell with up to 10^4 in one — there one block per tile is badly balanced and the
        // thread-per-particle kernel is faster; round 4 tried the dense tiles over a device-built work list with the spray
        // particle by particle: 357 / 407 us at steps 195 / 445 of the 256^3 drop against ~340 for this kernel alone, dropped,
        // profiles/r04/NOTES.md): gather through LDS tiles; the off-grid bucket (the array's tail) only has its speeds counted
        launch_g2p_tiled(s->st, s->g, s->Pb, s->pa, s->cell_start, s->dcx, s->dcy, s->dcz, s->pcx, s->pcy, s->pcz, s->prm.flip_blend, s->ss);
        if (s->n_out > 0)
            launch_g2p(s->st, s->g, s->n_out, s->pa.shifted(s->np - s->n_out), s->dcx, s->dcy, s->dcz, s->pcx, s->pcy, s->pcz, s->prm.flip_blend, s->ss);
    } else {
        launch_g2p(s->st, s->g, s->np, s->pa.shifted(s->p_off), s->dcx, s->dcy, s->dcz, s->pcx, s->pcy, s->pcz, s->prm.flip_blend, s->ss);
    }
    prof_end(s, FLUID_PROF_G2P, tok);
    launch_advect(s->st, s->g, s->np, s->pa.shifted(s->p_off), s->flags, s->prm.max_dt, s->prm.dx, s->ss);
    HIPCHK(hipGetLastError());
    int rc = read_ss(s);
    if (rc) return rc;
    s->dt = s->h_ss->dt;
    double ms;
    memcpy(&ms, &s->h_ss->max_speed_bits, sizeof(double));
    s->stats.max_speed = ms;
    s->stats.dt_out = s->dt;
    s->sorted = false;
    s->have_p2g = false;  // particles moved: fields are stale for a new gather
    return FLUID_OK;
}

extern "C" {

#define PHASE_GUARD(s)                                                                                        \
    if (!(s)) return fail(FLUID_ERR_ARG, "null handle");                                                      \
    if ((s)->dist) return fail(FLUID_ERR_STATE, "per-phase entry points are single-GPU only; a decomposed run steps with fluid_step")
int fluid_p2g(fluid_sim_t* s) { PHASE_GUARD(s); return phase_p2g(s); }
int fluid_flags_index(fluid_sim_t* s) { PHASE_GUARD(s); return phase_flags(s); }
int fluid_rhs_div(fluid_sim_t* s, int which) { PHASE_GUARD(s); return phase_rhs_div(s, which); }
int fluid_solve(fluid_sim_t* s) { PHASE_GUARD(s); return phase_solve(s); }
int fluid_vel_update(fluid_sim_t* s) { PHASE_GUARD(s); return phase_vel_update(s); }
int fluid_pressure_pass(fluid_sim_t* s, double* error) { PHASE_GUARD(s); return phase_pressure_pass(s, error); }
int fluid_flip_advect(fluid_sim_t* s) { PHASE_GUARD(s); return phase_flip_advect(s); }

/* fluid.cc:705-802 where the reference would call it (end of P2Gtransfer, fluid.cc:1147, commented out there): every cell inside W
 * that P2G left without a velocity gets the average of its defined neighbours, layer by layer, until the grid is full. */
int fluid_extrapolate(fluid_sim_t* s, int32_t* n_layers)
{
    PHASE_GUARD(s);
    if (!s->have_p2g) return fail(FLUID_ERR_STATE, "extrapolate before p2g");
    HIPCHK(hipSetDevice(s->prm.device));
    const Grid g = s->g;
    int* layer = s->indices;            // free between P2G and the flags pass, which rewrites it
    int* n_new = s->d_small;
    launch_extrap_init(s->st, g, s->solid, s->container, layer);
    int pass = 0;
    for (bool more = true; more;) {
        HIPCHK(hipMemsetAsync(n_new, 0, sizeof(int), s->st));
        for (int k = 0; k < 8; ++k) launch_extrap_layer(s->st, g, ++pass, layer, s->u, s->v, s->w, n_new);   // (a pass that finds nothing is a no-op)
        HIPCHK(hipGetLastError());
        HIPCHK(hipMemcpyAsync(s->h_small, n_new, sizeof(int), hipMemcpyDeviceToHost, s->st));
        HIPCHK(hipStreamSynchronize(s->st));
        more = s->h_small[0] != 0 && pass < 3 * g.N + 8;
    }
    // velBeforeUpdate is a copy of the grid taken after P2Gtransfer (fluid.cc:1455): it sees the extrapolated values
    const size_t bytes = s->ncell * sizeof(double);
    HIPCHK(hipMemcpyAsync(s->ub, s->u, bytes, hipMemcpyDeviceToDevice, s->st));
    HIPCHK(hipMemcpyAsync(s->vb, s->v, bytes, hipMemcpyDeviceToDevice, s->st));
    HIPCHK(hipMemcpyAsync(s->wb, s->w, bytes, hipMemcpyDeviceToDevice, s->st));
    // (s->indices held the layer numbers: the flags pass of this step must sweep the whole grid again)
    s->flags_valid = false;
    s->have_flags = false;
    s->dirty = Box{0, 0, 0, g.nx - 1, g.ny - 1, g.nz - 1};   // velocities everywhere now: the next step zeroes the whole grid
    HIPCHK(hipStreamSynchronize(s->st));
    if (n_layers) *n_layers = pass;
    return FLUID_OK;
}

/* fluid.cc:1053-1080: at most `per_cell` particles per base cell (index order); the others are parked outside the grid. */
int fluid_get_droplets(fluid_sim_t* s, int32_t* n_components, int64_t* cells, int32_t cap_components)
{
    if (!s) return fail(FLUID_ERR_ARG, "null handle");
    if (n_components) *n_components = s->n_drop;
    if (!cells || s->n_drop <= 0 || cap_components <= 0) return FLUID_OK;
    HIPCHK(hipSetDevice(s->prm.device));
    const int n = std::min(s->n_drop, (int)cap_components);
    std::vector<int> pre(DROP_NCTR + 1), an(DROP_CAP), ac((size_t)DROP_CAP * 64), hn(n), hc((size_t)n * 64);
    HIPCHK(hipStreamSynchronize(s->st));
    HIPCHK(hipMemcpy(pre.data(), s->drop_ctr + 64 * DROP_NCTR, pre.size() * sizeof(int), hipMemcpyDeviceToHost));
    HIPCHK(hipMemcpy(an.data(), s->drop_n, an.size() * sizeof(int), hipMemcpyDeviceToHost));
    HIPCHK(hipMemcpy(ac.data(), s->drop_cells, ac.size() * sizeof(int), hipMemcpyDeviceToHost));
    for (int d = 0, c = 0; d < n; ++d) {   // dense index -> slot, as k_drop_solve does
        while (c + 1 < DROP_NCTR && pre[c + 1] <= d) ++c;
        const size_t slot = (size_t)c * (DROP_CAP / DROP_NCTR) + (size_t)(d - pre[c]);
        hn[d] = an[slot];
        std::copy(ac.begin() + slot * 64, ac.begin() + slot * 64 + 64, hc.begin() + (size_t)d * 64);
    }
    const LBox L = s->L;
    for (int c = 0; c < n; ++c) {
        int64_t* o = cells + (size_t)c * 64;
        const int m = std::min(std::max(hn[c], 0), 64);
        for (int q = 0; q < m; ++q) {
            const long t = hc[(size_t)c * 64 + q];
            const int k = (int)(t % L.Lz), j = (int)((t / L.Lz) % L.Ly), i = (int)(t / ((long)L.Lz * L.Ly));
            o[q] = (int64_t)s->g.idx(L.x0 + i - 1, L.y0 + j - 1, L.z0 + k - LBOX_K0);
        }
        std::sort(o, o + m);
        for (int q = m; q < 64; ++q) o[q] = -1;
    }
    return FLUID_OK;
}

int fluid_resample(fluid_sim_t* s, int32_t per_cell, int64_t* n_parked)
{
    PHASE_GUARD(s);
    if (per_cell < 0) return fail(FLUID_ERR_ARG, "per_cell must be >= 0");
    if (s->np == 0) { if (n_parked) *n_parked = 0; return FLUID_OK; }
    HIPCHK(hipSetDevice(s->prm.device));
    int rc = phase_sort(s);             // cells contiguous, every cell's particles in ascending original index
    if (rc) return rc;
    HIPCHK(hipMemsetAsync(s->d_small, 0, sizeof(int), s->st));
    launch_resample(s->st, s->g, s->np - s->n_out, s->pa, s->cell_start, per_cell, s->g.hi - 10, (double)(s->g.hi + 40), s->d_small);
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpyAsync(s->h_small, s->d_small, sizeof(int), hipMemcpyDeviceToHost, s->st));
    HIPCHK(hipStreamSynchronize(s->st));
    if (n_parked) *n_parked = s->h_small[0];
    s->sorted = s->have_p2g = s->have_flags = false;   // positions changed
    s->sort_hint = false;
    return FLUID_OK;
}

int fluid_get_stats(fluid_sim_t* s, fluid_step_stats_t* st)
{
    if (!s || !st) return fail(FLUID_ERR_ARG, "null argument");
    *st = s->stats;
    return FLUID_OK;
}


int fluid_step(fluid_sim_t* s, fluid_step_stats_t* stats)
{
    if (!s) return fail(FLUID_ERR_ARG, "null handle");
    if (s->dist) return dist_step(s, stats);
    int rc;
    if ((rc = phase_p2g(s))) return rc;             // fluid.cc:1378-1413
    if ((rc = phase_flags(s))) return rc;           // :1416-1455
    double error = NAN;
    do {                                            // :1457
        if ((rc = phase_pressure_pass(s, &error))) return rc;
        if (s->prm.max_outer_passes > 0 && s->stats.outer_passes >= s->prm.max_outer_passes) break;
    } while (error > s->prm.outer_tol);             // :1484 (NaN ends the loop, inf continues)
    if ((rc = phase_flip_advect(s))) return rc;     // :1490
    if (stats) *stats = s->stats;
    return FLUID_OK;
}

static int field_info(fluid_sim* s, int field, void** ptr, size_t* bytes)
{
    const size_t n = s->ncell, se = solver_elem(s);
    switch (field) {
    case FLUID_FIELD_CONTAINER: case FLUID_FIELD_WEIGHTS: case FLUID_FIELD_OUTPUT: *ptr = s->container; *bytes = n * 4; return 0;
    case FLUID_FIELD_INDICES: *ptr = s->indices; *bytes = n * 4; return 0;
    case FLUID_FIELD_RHS: *ptr = s->rhs; *bytes = n * 4; return 0;
    case FLUID_FIELD_DIVER: *ptr = s->diver; *bytes = n * 4; return 0;
    case FLUID_FIELD_DIVER2: *ptr = s->diver2; *bytes = n * 4; return 0;
    case FLUID_FIELD_PRESSURE: *ptr = s->pressure; *bytes = n * 8; return 0;
    case FLUID_FIELD_SOLID: *ptr = s->solid; *bytes = n; return 0;
    case FLUID_FIELD_FLAGS: *ptr = s->flags; *bytes = n; return 0;
    case FLUID_FIELD_SEARCH: *ptr = s->S[0]; *bytes = n * se; return 0;
    case FLUID_FIELD_Q: *ptr = s->Q; *bytes = n * se; return 0;
    case FLUID_FIELD_VEL: case FLUID_FIELD_VEL_BEFORE: *ptr = nullptr; *bytes = 3 * n * 8; return 0;
    }
    return -1;
}

int fluid_download_field(fluid_sim_t* s, int field, void* dst, size_t bytes)
{
    if (!s || !dst) return fail(FLUID_ERR_ARG, "null argument");
    void* p;
    size_t nb;
    if (field_info(s, field, &p, &nb)) return fail(FLUID_ERR_ARG, "unknown field id");
    if (bytes != nb) return fail(FLUID_ERR_ARG, "byte count does not match the field size");
    HIPCHK(hipSetDevice(s->prm.device));
    HIPCHK(hipStreamSynchronize(s->st));
    if (field == FLUID_FIELD_VEL || field == FLUID_FIELD_VEL_BEFORE) {
        double* src[3] = {field == FLUID_FIELD_VEL ? s->u : s->ub, field == FLUID_FIELD_VEL ? s->v : s->vb, field == FLUID_FIELD_VEL ? s->w : s->wb};
        for (int a = 0; a < 3; ++a) HIPCHK(hipMemcpy((char*)dst + a * s->ncell * 8, src[a], s->ncell * 8, hipMemcpyDeviceToHost));
        return FLUID_OK;
    }
    HIPCHK(hipMemcpy(dst, p, nb, hipMemcpyDeviceToHost));
    return FLUID_OK;
}

int fluid_upload_field(fluid_sim_t* s, int field, const void* src, size_t bytes)
{
    if (!s || !src) return fail(FLUID_ERR_ARG, "null argument");
    void* p;
    size_t nb;
    if (field_info(s, field, &p, &nb)) return fail(FLUID_ERR_ARG, "unknown field id");
    if (bytes != nb) return fail(FLUID_ERR_ARG, "byte count does not match the field size");
    if (field == FLUID_FIELD_SOLID || field == FLUID_FIELD_FLAGS || field == FLUID_FIELD_INDICES || field == FLUID_FIELD_Q)
        return fail(FLUID_ERR_ARG, "field is not uploadable (use fluid_set_solid / fluid_flags_index)");
    HIPCHK(hipSetDevice(s->prm.device));
    HIPCHK(hipStreamSynchronize(s->st));
    if (field == FLUID_FIELD_VEL || field == FLUID_FIELD_VEL_BEFORE) {
        double* d[3] = {field == FLUID_FIELD_VEL ? s->u : s->ub, field == FLUID_FIELD_VEL ? s->v : s->vb, field == FLUID_FIELD_VEL ? s->w : s->wb};
        for (int a = 0; a < 3; ++a) HIPCHK(hipMemcpy(d[a], (const char*)src + a * s->ncell * 8, s->ncell * 8, hipMemcpyHostToDevice));
    } else {
        HIPCHK(hipMemcpy(p, src, nb, hipMemcpyHostToDevice));
    }
    // an uploaded field may be non-zero anywhere: widen every box to the whole grid
    s->Rb = s->Sb = Box{0, 0, 0, s->g.nx - 1, s->g.ny - 1, s->g.nz - 1};
    s->dirty = s->Rb;
    if (field == FLUID_FIELD_CONTAINER) { s->have_p2g = true; s->have_flags = false; }
    return FLUID_OK;
}

// reps launches of q = A s, timed with HIP events; with nsets > 1 the launches rotate over `nsets` separate copies of
// (s, q, flags) so that no launch finds its operands in the 256 MiB Infinity Cache (sets[i] = {s, q, flags} of copy i)
static int stencil_run(fluid_sim* s, int reps, int box_mode, int nsets, void* const (*sets)[3], float* avg_ms)
{
    const int N = s->g.N;
    const Box box = box_mode == 1 ? s->Rb : Box{0, 0, 0, N - 1, N - 1, N - 1};
    if (box_empty(box)) return fail(FLUID_ERR_STATE, "empty active box");
    // form of the dense sweep (developer use / tests): FLUID_MARCH_VARIANT, FLUID_MARCH_CX — launch_stencil_march (kernels_stencil.hip)
    const char* ev = getenv("FLUID_MARCH_VARIANT");
    const char* ec = getenv("FLUID_MARCH_CX");
    const int mv = ev ? atoi(ev) : 0, mc = ec ? atoi(ec) : 0;
    hipEvent_t e0, e1;
    HIPCHK(hipEventCreate(&e0));
    HIPCHK(hipEventCreate(&e1));
    HIPCHK(hipEventRecord(e0, s->st));
    for (int i = 0; i < reps; ++i) {
        const void* S = nsets > 1 ? sets[i % nsets][0] : s->S[0];
        void* Q = nsets > 1 ? sets[i % nsets][1] : s->Q;
        const uint8_t* fl = nsets > 1 ? (const uint8_t*)sets[i % nsets][2] : s->flags;
        // box_mode 0: dense sweep (LDS-DMA plane ring from 192^3 on, the lean march below; under 64^3 the tiled kernel);
        // 1: active box, tiled kernel; 2: dense sweep, tiled kernel
        if (s->prm.precision == FLUID_PRECISION_FP32) {
            if (!(box_mode == 0 && (N >= 64 || mv) && launch_stencil_march<float>(s->st, s->g, fl, (const float*)S, (float*)Q, make_coef<float>(s), mv, mc)))
                launch_stencil_apply<float>(s->st, s->g, box, fl, (const float*)S, (float*)Q, make_coef<float>(s));
        } else {
            if (!(box_mode == 0 && (N >= 64 || mv) && launch_stencil_march<double>(s->st, s->g, fl, (const double*)S, (double*)Q, make_coef<double>(s), mv, mc)))
                launch_stencil_apply<double>(s->st, s->g, box, fl, (const double*)S, (double*)Q, make_coef<double>(s));
        }
    }
    HIPCHK(hipEventRecord(e1, s->st));
    HIPCHK(hipGetLastError());
    HIPCHK(hipEventSynchronize(e1));
    float ms = 0;
    HIPCHK(hipEventElapsedTime(&ms, e0, e1));
    hipEventDestroy(e0);
    hipEventDestroy(e1);
    if (avg_ms) *avg_ms = ms / reps;
    return FLUID_OK;
}

int fluid_stencil_apply(fluid_sim_t* s, int reps, int box_mode, float* avg_ms)
{
    if (!s || reps < 1) return fail(FLUID_ERR_ARG, "bad argument");
    if (!s->have_flags) return fail(FLUID_ERR_STATE, "stencil_apply before flags_index");
    HIPCHK(hipSetDevice(s->prm.device));
    return stencil_run(s, reps, box_mode, 1, nullptr, avg_ms);
}

int fluid_stencil_apply_hbm(fluid_sim_t* s, int reps, int box_mode, int64_t footprint_bytes, int32_t* nsets_out, float* avg_ms)
{
    if (!s || reps < 1 || footprint_bytes < 0) return fail(FLUID_ERR_ARG, "bad argument");
    if (!s->have_flags) return fail(FLUID_ERR_STATE, "stencil_apply before flags_index");
    HIPCHK(hipSetDevice(s->prm.device));
    // one set = s + q + flags of the whole grid; as many sets as the requested footprint needs (at least 2: the handle's own arrays are not used)
    const size_t se = solver_elem(s), n = s->ncell;
    const size_t set_bytes = 2 * n * se + n;
    int nsets = (int)((footprint_bytes + (int64_t)set_bytes - 1) / (int64_t)set_bytes);
    nsets = std::max(2, std::min(nsets, 256));
    std::vector<void*> mem((size_t)3 * nsets, nullptr);
    std::vector<void*> flat(mem.size());
    auto release = [&]() { for (void* p : mem) if (p) hipFree(p); };
    for (int i = 0; i < nsets; ++i) {
        hipError_t e = hipMalloc(&mem[3 * i], n * se);
        if (e == hipSuccess) e = hipMalloc(&mem[3 * i + 1], n * se);
        if (e == hipSuccess) e = hipMalloc(&mem[3 * i + 2], n);
        if (e == hipSuccess) e = hipMemcpyAsync(mem[3 * i], s->S[0], n * se, hipMemcpyDeviceToDevice, s->st);
        if (e == hipSuccess) e = hipMemsetAsync(mem[3 * i + 1], 0, n * se, s->st);
        if (e == hipSuccess) e = hipMemcpyAsync(mem[3 * i + 2], s->flags, n, hipMemcpyDeviceToDevice, s->st);
        if (e != hipSuccess) { release(); return fail(FLUID_ERR_HIP, std::string("stencil_apply_hbm: ") + hipGetErrorString(e)); }
    }
    // untimed: three rotations over every set (first touch, address translations, clocks: the first timed launches of a fresh set of
    // allocations otherwise run 5-8 % under the steady state, tools/stencil_state.py)
    int rc = stencil_run(s, 3 * nsets, box_mode, nsets, (void* const (*)[3])mem.data(), nullptr);
    if (!rc) rc = stencil_run(s, reps, box_mode, nsets, (void* const (*)[3])mem.data(), avg_ms);
    // the result of the last launch, where fluid_download_field(FLUID_FIELD_Q) finds it
    if (!rc && hipMemcpyAsync(s->Q, mem[3 * ((reps - 1) % nsets) + 1], n * se, hipMemcpyDeviceToDevice, s->st) != hipSuccess) rc = fail(FLUID_ERR_HIP, "stencil_apply_hbm: copy back");
    hipStreamSynchronize(s->st);
    release();
    if (nsets_out) *nsets_out = nsets;
    return rc;
}

static int hook_device(int32_t device)
{
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return fail(FLUID_ERR_HIP, "no HIP device visible: libfluid_hip has no CPU path");
    if (device < 0 || device >= ndev) return fail(FLUID_ERR_ARG, "device ordinal out of range");
    HIPCHK(hipSetDevice(device));
    return FLUID_OK;
}

int fluid_spline_eval(int32_t device, int32_t which, int64_t n, const double* x, double* w)
{
    if (n < 0 || which < 0 || which > 3 || (n > 0 && (!x || !w))) return fail(FLUID_ERR_ARG, "bad argument");
    int rc = hook_device(device);
    if (rc || n == 0) return rc;
    double *dx = nullptr, *dw = nullptr;
    HIPCHK(hipMalloc((void**)&dx, n * sizeof(double)));
    HIPCHK(hipMalloc((void**)&dw, n * sizeof(double)));
    HIPCHK(hipMemcpy(dx, x, n * sizeof(double), hipMemcpyHostToDevice));
    launch_spline_eval(nullptr, which, (long)n, dx, dw);
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpy(w, dw, n * sizeof(double), hipMemcpyDeviceToHost));
    hipFree(dx); hipFree(dw);
    return FLUID_OK;
}

int fluid_dot_eval(int32_t device, int64_t n, const double* a, const double* b, double* out)
{
    if (n < 0 || !out || (n > 0 && (!a || !b))) return fail(FLUID_ERR_ARG, "bad argument");
    int rc = hook_device(device);
    if (rc) return rc;
    double *da = nullptr, *db = nullptr, *dp = nullptr;
    const int nb = 1024;   // partials of one PCG launch
    HIPCHK(hipMalloc((void**)&da, (n + 1) * sizeof(double)));
    HIPCHK(hipMalloc((void**)&db, (n + 1) * sizeof(double)));
    HIPCHK(hipMalloc((void**)&dp, (nb + 1) * sizeof(double)));
    if (n) {
        HIPCHK(hipMemcpy(da, a, n * sizeof(double), hipMemcpyHostToDevice));
        HIPCHK(hipMemcpy(db, b, n * sizeof(double), hipMemcpyHostToDevice));
    }
    launch_dot(nullptr, (long)n, da, db, dp, nb, dp + nb);
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpy(out, dp + nb, sizeof(double), hipMemcpyDeviceToHost));
    hipFree(da); hipFree(db); hipFree(dp);
    return FLUID_OK;
}

int fluid_profile_enable(fluid_sim_t* s, int sample_every)
{
    if (!s || sample_every < 0) return fail(FLUID_ERR_ARG, "bad argument");
    s->prof_every = sample_every;
    return FLUID_OK;
}
int fluid_profile_read(fluid_sim_t* s, int klass, int64_t* n_launches, int64_t* n_sampled, double* total_ms, double* cells)
{
    if (!s || klass < 0 || klass >= FLUID_PROF_COUNT) return fail(FLUID_ERR_ARG, "bad argument");
    prof_resolve(s);
    const ProfClass& p = s->prof[klass];
    if (n_launches) *n_launches = p.launches;
    if (n_sampled) *n_sampled = p.sampled;
    if (total_ms) *total_ms = p.ms;
    if (cells) *cells = p.cells;
    return FLUID_OK;
}
int fluid_profile_reset(fluid_sim_t* s)
{
    if (!s) return fail(FLUID_ERR_ARG, "null handle");
    prof_resolve(s);
    for (int k = 0; k < FLUID_PROF_COUNT; ++k) {
        s->prof[k].launches = s->prof[k].sampled = 0;
        s->prof[k].ms = s->prof[k].cells = 0;
    }
    return FLUID_OK;
}

}  // extern "C"
