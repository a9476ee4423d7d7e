// Host-only entry points under AddressSanitizer + UBSan (tests/test_host_sanitizers.py builds and runs this with g++;
// GPU sanitizers are not available on the pool).  Exercises the scatter (voxels, 8^3 tiles, a 128^3 tile, the filter,
// count-only calls, bad arguments) and the .vdb writer (one and two grids, odd size).
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "fluid_hip.h"
#include "mpm_hip.h"

static int fail(const char* what)
{
    std::fprintf(stderr, "FAILED: %s\n", what);
    return 1;
}

int main(int argc, char** argv)
{
    const char* dir = argc > 1 ? argv[1] : ".";
    {
        const int32_t lo[3] = {-20, -20, -20}, hi[3] = {20, 20, 20};
        const int64_t n = fluid_scene_uniform_scatter(lo, hi, 10.f, 0, 60, nullptr);
        if (n != 689210) return fail("reference scene count");
        std::vector<double> pos((size_t)3 * n);
        if (fluid_scene_uniform_scatter(lo, hi, 10.f, 0, 60, pos.data()) != n) return fail("reference scene fill");
    }
    {
        const int32_t lo[3] = {-130, -9, 3}, hi[3] = {5, 13, 3};   // mixed: a thin slab crossing the origin in x
        const int64_t n = fluid_scene_uniform_scatter(lo, hi, 2.f, 5, 0, nullptr);
        if (n != 2 * 136 * 23 * 1) return fail("slab count");
        std::vector<double> pos((size_t)3 * n);
        if (fluid_scene_uniform_scatter(lo, hi, 2.f, 5, 0, pos.data()) != n) return fail("slab fill");
    }
    {
        const int32_t lo[3] = {-130, -130, -130}, hi[3] = {5, 5, 5};  // holds the whole node [-128, -1]^3
        if (fluid_scene_uniform_scatter(lo, hi, 1.f, 3, 0, nullptr) != 136L * 136 * 136) return fail("128^3 tile count");
        const int32_t bad[3] = {6, 6, 6};
        if (fluid_scene_uniform_scatter(bad, hi, 1.f, 3, 0, nullptr) >= 0) return fail("lo > hi accepted");
        if (fluid_scene_uniform_scatter(lo, hi, 0.f, 3, 0, nullptr) >= 0) return fail("density 0 accepted");
        if (fluid_scene_uniform_scatter(nullptr, hi, 1.f, 3, 0, nullptr) >= 0) return fail("null box accepted");
    }
    {   // the snow cone of the MPM program: single voxels on both sides of the origin (four root nodes), count-only and filled
        const int64_t n = mpm_scene_cone(15, 13, 4, 400.f, 0, nullptr);
        if (n != 6205) return fail("cone count");
        std::vector<double> pos((size_t)3 * n);
        if (mpm_scene_cone(15, 13, 4, 400.f, 0, pos.data()) != n) return fail("cone fill");
        std::vector<double> big((size_t)3 * mpm_scene_cone(63, 61, 24, 3.f, 1, nullptr));
        if (mpm_scene_cone(63, 61, 24, 3.f, 1, big.data()) * 3 != (int64_t)big.size()) return fail("large cone fill");
        if (mpm_scene_cone(15, 13, 0, 400.f, 0, nullptr) >= 0 || mpm_scene_cone(15, 16, 4, 400.f, 0, nullptr) >= 0) return fail("bad cone accepted");
    }
    for (int n : {8, 21}) {
        std::vector<float> a((size_t)n * n * n), b(a.size());
        for (size_t i = 0; i < a.size(); ++i) { a[i] = (i % 7 == 0) ? 0.f : (float)(i % 13) * 0.25f; b[i] = (float)(i % 3); }
        const float* grids[2] = {a.data(), b.data()};
        char path[512];
        std::snprintf(path, sizeof path, "%s/san_%d.vdb", dir, n);
        if (fluid_write_vdb(path, n, n == 8 ? 1 : 2, grids) != FLUID_OK) return fail("fluid_write_vdb");
    }
    if (fluid_write_vdb("/nonexistent-dir/x.vdb", 8, 1, nullptr) == FLUID_OK) return fail("bad path accepted");
    {   // both compressions, and the streaming form the driver uses for the final mygrids.vdb (one grid appended per step)
        const int n = 13;
        std::vector<float> a((size_t)n * n * n);
        for (size_t i = 0; i < a.size(); ++i) a[i] = (i % 5 == 0) ? 0.f : (float)(i % 11);
        const float* g1[1] = {a.data()};
        char path[512];
        std::snprintf(path, sizeof path, "%s/san_mask.vdb", dir);
        if (fluid_write_vdb_ex(path, n, 1, g1, FLUID_VDB_ACTIVE_MASK) != FLUID_OK) return fail("fluid_write_vdb_ex");
        std::snprintf(path, sizeof path, "%s/san_stream.vdb", dir);
        fluid_vdb_writer_t* w = nullptr;
        if (fluid_vdb_open(path, n, 3, FLUID_VDB_ZIP_ACTIVE_MASK, &w) != FLUID_OK) return fail("fluid_vdb_open");
        for (int k = 0; k < 3; ++k)
            if (fluid_vdb_append(w, a.data()) != FLUID_OK) return fail("fluid_vdb_append");
        if (fluid_vdb_append(w, a.data()) == FLUID_OK) return fail("a fourth grid accepted");
        if (fluid_vdb_close(w) != FLUID_OK) return fail("fluid_vdb_close");
        if (fluid_vdb_open(path, n, 3, 7, &w) == FLUID_OK) return fail("bad compression accepted");
    }
    std::puts("host sanitizer run: ok");
    return 0;
}
