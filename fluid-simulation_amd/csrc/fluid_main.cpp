// `fluid` — the program `./run.sh fluid` builds and runs (reference: run.sh:1-7, main() in
// fluid.cc:1151-1514).  Host C++ only: scene set-up, the 500-step loop, the same stdout lines
// (fluid.cc:1383-1386,1456,1486,1491,1499-1502) and one density grid per step; every step is
// one fluid_step() call into libfluid_hip.so (hand-written HIP, gfx950).
//
// Like the reference it takes no arguments.  Environment overrides (all optional):
//   FLUID_N (121)  FLUID_PPC (10)  FLUID_STEPS (500)  FLUID_SEED (0)  FLUID_DEVICE (0)  FLUID_FLIP_BLEND (1)
//   FLUID_OUT (simulation)  — directory for mygrids<i>.vdb (fluid.cc:1371,1503-1510); "" disables output; FLUID_RAW=1 adds .f32 dumps.
// Initial particles: with the defaults (N = 121, 10 per voxel) exactly the reference's — fill(CoordBBox(-20, 20)) scattered by
// UniformPointScatter with std::mt19937(FLUID_SEED) (fluid_scene_uniform_scatter: 689210 points); any other N / PPC takes the
// scaled synthetic cube (fluid_scene_water_cube_drop).
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <iostream>
#include <string>
#include <vector>
#include <sys/stat.h>

#include "fluid_hip.h"

static long env_long(const char* k, long d)
{
    const char* v = getenv(k);
    return v && *v ? atol(v) : d;
}

int main(int, char**)
{
    const auto t0 = std::chrono::steady_clock::now();
    fluid_params_t prm;
    fluid_default_params(&prm);
    prm.n = (int32_t)env_long("FLUID_N", 121);
    prm.device = (int32_t)env_long("FLUID_DEVICE", 0);
    const int ppc = (int)env_long("FLUID_PPC", 10);          // 10 points per voxel, fluid.cc:1349
    const int steps = (int)env_long("FLUID_STEPS", 500);     // fluid.cc:1368
    const uint64_t seed = (uint64_t)env_long("FLUID_SEED", 0);  // mt19937(0), fluid.cc:1348
    if (const char* b = getenv("FLUID_FLIP_BLEND")) prm.flip_blend = atof(b);  // default 1 = the reference's pure FLIP
    const char* outenv = getenv("FLUID_OUT");
    const std::string outdir = outenv ? outenv : "simulation";
    const bool raw_f32 = env_long("FLUID_RAW", 0) != 0;

    fluid_sim_t* sim = nullptr;
    if (fluid_create(&prm, &sim) != FLUID_OK) {
        std::cerr << "fluid_create: " << fluid_last_error() << std::endl;
        return 1;
    }
    const bool ref_scene = prm.n == 121 && ppc == 10;   // fluid.cc:1176,1347-1350
    const int32_t flo[3] = {-20, -20, -20}, fhi[3] = {20, 20, 20};
    const int64_t np = ref_scene ? fluid_scene_uniform_scatter(flo, fhi, 10.f, (uint32_t)seed, 60, nullptr)
                                 : fluid_scene_water_cube_drop(prm.n, ppc, seed, nullptr);
    std::vector<double> pos((size_t)3 * np);
    if (ref_scene) fluid_scene_uniform_scatter(flo, fhi, 10.f, (uint32_t)seed, 60, pos.data());
    else fluid_scene_water_cube_drop(prm.n, ppc, seed, pos.data());
    if (fluid_upload_particles(sim, np, pos.data(), nullptr) != FLUID_OK) {
        std::cerr << "fluid_upload_particles: " << fluid_last_error() << std::endl;
        return 1;
    }
    if (!outdir.empty()) mkdir(outdir.c_str(), 0755);  // the reference aborts when simulation/ is missing
    const size_t ncell = (size_t)prm.n * prm.n * prm.n;
    std::vector<float> out(outdir.empty() ? 0 : ncell);
    // file.write(grids) of fluid.cc:1508: `grids` is declared outside the loop (:1366) and receives every step's grid
    // (:1450), so the final mygrids.vdb holds all of them — streamed here, one grid appended per step
    fluid_vdb_writer_t* all = nullptr;
    std::string fin;
    if (!outdir.empty() && steps > 0) {
        const size_t slash = outdir.find_last_of('/');   // beside the output directory: ./mygrids.vdb for the default "simulation"
        fin = (slash == std::string::npos ? std::string() : outdir.substr(0, slash + 1)) + "mygrids.vdb";
        if (fluid_vdb_open(fin.c_str(), prm.n, steps, FLUID_VDB_ZIP_ACTIVE_MASK, &all) != FLUID_OK) { std::cerr << "cannot write " << fin << std::endl; return 1; }
    }

    double dt = prm.max_dt;  // fluid.cc:1367
    double simulationTime = 0;
    for (int i = 0; i < steps; ++i) {
        std::cout << "2" << std::endl;
        std::cout << "3" << std::endl;
        std::cout << "DT " << dt << std::endl;
        std::cout << "Before" << std::endl;
        fluid_step_stats_t st;
        if (fluid_step(sim, &st) != FLUID_OK) {
            std::cerr << "fluid_step: " << fluid_last_error() << std::endl;
            return 1;
        }
        std::cout << "After" << std::endl;
        dt = st.dt_out;
        std::cout << "DT " << dt << std::endl;
        std::cout << "Error:\t" << st.error << std::endl;
        std::cout << "Iteration:\t" << i + 1 << std::endl;
        simulationTime += dt;
        std::cout << "Time delta:\t" << simulationTime << std::endl;
        if (!outdir.empty()) {
            if (fluid_download_field(sim, FLUID_FIELD_OUTPUT, out.data(), ncell * sizeof(float)) != FLUID_OK) {
                std::cerr << "fluid_download_field: " << fluid_last_error() << std::endl;
                return 1;
            }
            // file2.write(grids2) of fluid.cc:1503-1504: simulation/mygrids<i>.vdb.  grids2 is declared inside the loop
            // (:1373), so each of these files holds exactly the grid of its step.
            const std::string fn = outdir + "/mygrids" + std::to_string(i) + ".vdb";
            const float* gp[1] = {out.data()};
            if (fluid_write_vdb(fn.c_str(), prm.n, 1, gp) != FLUID_OK) { std::cerr << "cannot write " << fn << std::endl; return 1; }
            if (fluid_vdb_append(all, out.data()) != FLUID_OK) { std::cerr << "cannot write " << fin << std::endl; return 1; }
            if (raw_f32) {  // FLUID_RAW=1: also the bare float32 dump (int32 n, then n^3 floats, z fastest)
                const std::string fr = outdir + "/mygrids" + std::to_string(i) + ".f32";
                FILE* f = fopen(fr.c_str(), "wb");
                if (!f) { std::cerr << "cannot write " << fr << std::endl; return 1; }
                int32_t n32 = prm.n;
                fwrite(&n32, sizeof(n32), 1, f);
                fwrite(out.data(), sizeof(float), ncell, f);
                fclose(f);
            }
        }
    }
    if (all && fluid_vdb_close(all) != FLUID_OK) { std::cerr << "cannot write " << fin << std::endl; return 1; }
    fluid_destroy(sim);
    const double sec = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    std::cout << "Time Taken " << sec / 60 << " minutes" << std::endl;  // fluid.cc:1513 (wall, not clock())
    return 0;
}
