// `mpm` — the program `./run.sh mpm` builds and runs (reference: run.sh:1-7, main() in mpm.cc:1020-1443).  Host C++ only:
// the scene (mpm.cc:1037-1052,1274-1278), the 500-step loop with the reference's stdout lines (mpm.cc:1313,1394-1396,
// 417,440-442,1405-1412,584,1418,1427) and one density grid per step; every step is one mpm_step() call into
// libfluid_hip.so (hand-written HIP, gfx950).
//
// Like the reference it takes no arguments.  Environment overrides (all optional):
//   MPM_STEPS (500)  MPM_PPV (400 points per voxel)  MPM_SEED (0)  MPM_DEVICE (0)  MPM_B (15)  MPM_LAYERS (4)
//   MPM_OUT (simulation) — directory for mygrids<i>.vdb (mpm.cc:1304-1306,1434-1435); "" disables output.
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <iostream>
#include <string>
#include <vector>
#include <sys/stat.h>

#include "mpm_hip.h"

static long env_long(const char* k, long d)
{
    const char* v = getenv(k);
    return v && *v ? atol(v) : d;
}

int main(int, char**)
{
    const auto t0 = std::chrono::steady_clock::now();
    mpm_params_t prm;
    mpm_default_params(&prm);
    prm.B = (int32_t)env_long("MPM_B", 15);
    prm.W = prm.B - 2;                                        // mpm.cc:1156: solid where |c| > 13
    prm.device = (int32_t)env_long("MPM_DEVICE", 0);
    const int steps = (int)env_long("MPM_STEPS", 500);        // mpm.cc:1301
    const float ppv = (float)env_long("MPM_PPV", 400);        // mpm.cc:1277
    const uint32_t seed = (uint32_t)env_long("MPM_SEED", 0);  // mpm.cc:1276
    const int layers = (int)env_long("MPM_LAYERS", 4);        // mpm.cc:1041: j = -13 .. -10
    const char* outenv = getenv("MPM_OUT");
    const std::string outdir = outenv ? outenv : "simulation";

    mpm_sim_t* sim = nullptr;
    if (mpm_create(&prm, &sim) != FLUID_OK) {
        std::cerr << "mpm_create: " << fluid_last_error() << std::endl;
        return 1;
    }
    const int64_t np = mpm_scene_cone(prm.B, prm.W, layers, ppv, seed, nullptr);
    if (np < 0) {
        std::cerr << "mpm_scene_cone: bad arguments" << std::endl;
        return 1;
    }
    std::vector<double> pos((size_t)3 * np);
    mpm_scene_cone(prm.B, prm.W, layers, ppv, seed, pos.data());
    if (mpm_upload_particles(sim, np, pos.data(), nullptr, nullptr) != FLUID_OK) {   // velocity (0, -50, 0), mpm.cc:484
        std::cerr << "mpm_upload_particles: " << fluid_last_error() << std::endl;
        return 1;
    }
    if (!outdir.empty()) mkdir(outdir.c_str(), 0755);
    const int n = 2 * prm.B + 1;
    const size_t ncell = (size_t)n * n * n;
    std::vector<float> out(outdir.empty() ? 0 : ncell);
    // file.write(grids), mpm.cc:1437: `grids` (mpm.cc:1294) receives every step's output grid (mpm.cc:1383)
    fluid_vdb_writer_t* all = nullptr;
    std::string fin;
    if (!outdir.empty() && steps > 0) {
        const size_t slash = outdir.find_last_of('/');
        fin = (slash == std::string::npos ? std::string() : outdir.substr(0, slash + 1)) + "mygrids.vdb";
        if (fluid_vdb_open(fin.c_str(), n, steps, FLUID_VDB_ZIP_ACTIVE_MASK, &all) != FLUID_OK) { std::cerr << "cannot write " << fin << std::endl; return 1; }
    }
    double dt = prm.dt0;   // mpm.cc:1295
    for (int i = 0; i < steps; ++i) {
        std::cout << "DT " << dt << std::endl;
        mpm_step_stats_t st;
        if (mpm_step(sim, &st) != FLUID_OK) {
            std::cerr << "mpm_step: " << fluid_last_error() << std::endl;
            return 1;
        }
        std::cout << "1" << std::endl << "GAH" << std::endl << "DAH" << std::endl << "2" << std::endl;
        // "Max Force " << Force << " " << maxMi << " " << maxForceCoeff2 << " " << yes   (Vec3d prints as [x, y, z])
        std::cout << "Max Force [" << st.max_force[0] << ", " << st.max_force[1] << ", " << st.max_force[2] << "] " << st.max_mi << " "
                  << st.max_force_coeff2 << " " << (st.any_active ? 1 : 0) << std::endl;
        std::cout << "GARR" << std::endl << "after" << std::endl;
        std::cout << "Error: " << st.cg_error << std::endl;
        std::cout << "5" << std::endl << "3" << std::endl;
        std::cout << "MAX " << st.max_grad << " " << st.max_fp << " " << st.max_fe << std::endl;
        std::cout << "4" << std::endl;
        dt = st.dt_out;
        std::cout << "DT " << dt << std::endl;
        std::cout << "Iteration:\t" << i + 1 << std::endl;
        if (!outdir.empty()) {
            if (mpm_download_field(sim, MPM_F_OUTPUT, out.data()) != FLUID_OK) {
                std::cerr << "mpm_download_field: " << fluid_last_error() << std::endl;
                return 1;
            }
            const std::string fn = outdir + "/mygrids" + std::to_string(i) + ".vdb";   // mpm.cc:1304,1434: one grid per file
            const float* gp[1] = {out.data()};
            if (fluid_write_vdb(fn.c_str(), n, 1, gp) != FLUID_OK) { std::cerr << "cannot write " << fn << std::endl; return 1; }
            if (fluid_vdb_append(all, out.data()) != FLUID_OK) { std::cerr << "cannot write " << fin << std::endl; return 1; }
        }
    }
    if (all && fluid_vdb_close(all) != FLUID_OK) { std::cerr << "cannot write " << fin << std::endl; return 1; }
    mpm_destroy(sim);
    const double sec = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    std::cout << "Time Taken " << sec / 60 << " minutes" << std::endl;   // mpm.cc:1442 (wall, not clock() / 15)
    return 0;
}
