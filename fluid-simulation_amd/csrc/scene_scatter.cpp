// The reference's initial particles: what
//     fluidGrid->fill(CoordBBox(lo, hi), 0, true);                                   fluid.cc:1176
//     std::mt19937 mtRandi(seed);
//     UniformPointScatter<PointList, std::mt19937> scatteri(pos, pointsPerVolume, mtRandi);
//     scatteri(*fluidGrid);                                                          fluid.cc:1347-1350
// leaves in PointList::positions, restated from openvdb/tools/PointScatter.h:143-185,421-440, openvdb/math/Math.h:134-205
// and the tree's fill / ValueOn iteration order (tree/RootNode.h, InternalNode.h, LeafNode.h, TreeIterator.h), for a
// FloatTree (Tree4<float, 5, 4, 3>: nodes of 4096^3, 128^3 and 8^3 voxels).  Host code, no GPU.
//
// What the program does, in the order that fixes the random stream:
//  * sparse fill: a node the box covers completely becomes an active TILE (512 voxels for a leaf-sized one), a node it
//    cuts is descended into, down to single active voxels in the partly covered leaves;
//  * voxelCount = active voxels (tiles count all theirs); target = Index64(ppv * 1*1*1) * voxelCount points;
//  * target draws of std::uniform_int_distribution<uint64>(0, voxelCount-1) from a COPY of the engine (RandInt copies
//    Rand01's engine, which is itself a copy of the caller's: both streams start at the seed), sorted;
//  * the tree's active values are walked in ValueOn order (root table sorted by origin; inside a node strictly by
//    child index x-major/z-fastest, tiles and children interleaved; leaf voxels by offset), each with its voxel count;
//    the i-th sorted draw picks the value whose running count passes it: a voxel gives coord - 0.5 + getRand() per axis,
//    a tile bbox.min - 0.5 + extent * getRand(); getRand() = 0.5 + spread * (Rand01() - 0.5), spread = 1;
//  * the three getRand() calls are constructor arguments of Vec3R: g++ (run.sh:3) evaluates them right to left, so the
//    first draw lands in z (tests/test_scatter.py compiles the three-line check with the image's g++);
//  * PointList::add keeps a point iff |p| < boundary - 2 on every axis (fluid.cc:841).
// std::mt19937 and the two distributions are the C++ library's own: the numbers are those of the reference built with the
// same libstdc++ (the distributions' algorithms are implementation-defined).  The reference cannot be built in this image
// (OpenVDB needs TBB/Boost/Half), so this function is pinned by its invariants only: parity unpinned.
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <random>
#include <vector>

#include "fluid_hip.h"
#include "mpm_hip.h"

namespace {

struct Value {   // an active tile (dim > 1) or voxel (dim == 1) with its minimum corner
    int32_t x, y, z, dim;
};

struct BoxI {
    int32_t lo[3], hi[3];
};

// Active values of the filled box inside the node [o, o + dim)^3, in ValueOn order.  `dim` walks 4096 -> 128 -> 8 -> 1;
// children of a node are visited by index = (x * n + y) * n + z of their position in it.
void walk(const BoxI& b, int32_t ox, int32_t oy, int32_t oz, int32_t dim, std::vector<Value>& out)
{
    const int32_t o[3] = {ox, oy, oz};
    bool inside = true;
    for (int a = 0; a < 3; ++a) {
        if (o[a] + dim - 1 < b.lo[a] || o[a] > b.hi[a]) return;                         // untouched: inactive background
        if (o[a] < b.lo[a] || o[a] + dim - 1 > b.hi[a]) inside = false;
    }
    if (inside) {  // fill sets an active tile (a voxel at dim 1)
        out.push_back(Value{ox, oy, oz, dim});
        return;
    }
    const int32_t sub = dim == 4096 ? 128 : (dim == 128 ? 8 : 1), n = dim / sub;
    for (int32_t i = 0; i < n; ++i)
        for (int32_t j = 0; j < n; ++j)
            for (int32_t k = 0; k < n; ++k) walk(b, ox + i * sub, oy + j * sub, oz + k * sub, sub, out);
}

inline int32_t floor_to(int32_t v, int32_t dim) { return v & ~(dim - 1); }  // node origin (Coord & ~(DIM - 1))

}  // namespace

namespace {

// The draws of UniformPointScatter::operator() (PointScatter.h:143-185) over the active values `vals` in ValueOn order.
int64_t scatter_values(const std::vector<Value>& vals, float points_per_volume, uint32_t seed, int32_t boundary, double* pos)
{
    uint64_t voxels = 0;
    for (const Value& v : vals) voxels += (uint64_t)v.dim * v.dim * v.dim;
    if (voxels == 0) return 0;
    // PointScatter.h:151: Index64(mPointsPerVolume*dim[0]*dim[1]*dim[2])*mVoxelCount with voxelSize 1
    const uint64_t target = (uint64_t)((double)points_per_volume * 1.0 * 1.0 * 1.0) * voxels;

    const std::mt19937 caller(seed);
    std::mt19937 e01(caller);        // BasePointScatter::mRand01(randGen)
    std::mt19937 eint(e01);          // RandInt(mRand01.engine(), 0, voxelCount - 1)
    std::uniform_int_distribution<uint64_t> pick(0, voxels - 1);
    std::vector<uint64_t> list(target);
    for (uint64_t i = 0; i < target; ++i) list[i] = pick(eint);
    std::sort(list.begin(), list.end());

    std::uniform_real_distribution<double> r01;
    const double spread = 1.0;
    auto get_rand = [&]() { return 0.5 + spread * (r01(e01) - 0.5); };
    int64_t cnt = 0;
    size_t vi = 0;
    uint64_t n = (uint64_t)vals[0].dim * vals[0].dim * vals[0].dim;
    for (uint64_t i = 0; i < target; ++i) {
        while (n <= list[i]) {
            ++vi;
            n += (uint64_t)vals[vi].dim * vals[vi].dim * vals[vi].dim;
        }
        const Value& v = vals[vi];
        // dmin = coord (or bbox.min()) - 0.5; arguments right to left: z first
        const double rz = get_rand(), ry = get_rand(), rx = get_rand();
        double p[3];
        if (v.dim == 1) {
            p[0] = ((double)v.x - 0.5) + rx; p[1] = ((double)v.y - 0.5) + ry; p[2] = ((double)v.z - 0.5) + rz;
        } else {
            p[0] = ((double)v.x - 0.5) + v.dim * rx; p[1] = ((double)v.y - 0.5) + v.dim * ry; p[2] = ((double)v.z - 0.5) + v.dim * rz;
        }
        // indexToWorld of the unit linear transform is the identity; PointList::add (fluid.cc:841)
        if (boundary <= 0 ||
            (std::fabs(p[0]) < boundary - 2 && std::fabs(p[1]) < boundary - 2 && std::fabs(p[2]) < boundary - 2)) {
            if (pos) { pos[3 * cnt] = p[0]; pos[3 * cnt + 1] = p[1]; pos[3 * cnt + 2] = p[2]; }
            ++cnt;
        }
    }
    return cnt;
}

}  // namespace

extern "C" int64_t fluid_scene_uniform_scatter(const int32_t lo[3], const int32_t hi[3], float points_per_volume, uint32_t seed,
                                               int32_t boundary, double* pos)
{
    if (!lo || !hi || !(points_per_volume > 0)) return -1;
    BoxI b;
    for (int a = 0; a < 3; ++a) {
        if (lo[a] > hi[a] || lo[a] < -(1 << 20) || hi[a] > (1 << 20)) return -1;
        b.lo[a] = lo[a];
        b.hi[a] = hi[a];
    }
    // root table: std::map ordered by Coord::operator< (x, then y, then z) over the 4096^3 blocks the box touches
    std::vector<Value> vals;
    for (int32_t x = floor_to(b.lo[0], 4096); x <= b.hi[0]; x += 4096)
        for (int32_t y = floor_to(b.lo[1], 4096); y <= b.hi[1]; y += 4096)
            for (int32_t z = floor_to(b.lo[2], 4096); z <= b.hi[2]; z += 4096) walk(b, x, y, z, 4096, vals);
    return scatter_values(vals, points_per_volume, seed, boundary, pos);
}

// The snow cone of the MPM program (mpm.cc:1037-1052): single active voxels set with setValue, no tiles.  ValueOn order of
// voxels: root nodes by origin (x, y, z), then the child index inside the 4096^3 node, inside the 128^3 node, then the
// offset inside the 8^3 leaf — every index x-major, z fastest.
extern "C" int64_t mpm_scene_cone(int32_t B, int32_t W, int32_t layers, float points_per_voxel, uint32_t seed, double* pos)
{
    if (B < 3 || W < 1 || W > B || layers < 1 || layers > 2 * W + 1 || !(points_per_voxel > 0)) return -1;
    std::vector<Value> vals;
    for (int32_t i = -W; i <= W; ++i)
        for (int32_t j = -W; j <= -W + layers - 1; ++j)
            for (int32_t k = -W; k <= W; ++k) {
                const double r = (double)(j + W) / 2;
                if ((double)i * i + (double)k * k <= r * r) vals.push_back(Value{i, j, k, 1});
            }
    auto key = [](const Value& v, int64_t out[6]) {
        const int32_t c[3] = {v.x, v.y, v.z};
        int64_t root[3], i1 = 0, i2 = 0, i3 = 0;
        for (int a = 0; a < 3; ++a) {
            root[a] = floor_to(c[a], 4096);
            const int32_t in_root = c[a] - (int32_t)root[a];          // 0..4095
            i1 = i1 * 32 + (in_root >> 7);                            // 32^3 children of 128^3
            i2 = i2 * 16 + ((in_root & 127) >> 3);                    // 16^3 leaves
            i3 = i3 * 8 + (in_root & 7);                              // 8^3 voxels
        }
        out[0] = root[0], out[1] = root[1], out[2] = root[2], out[3] = i1, out[4] = i2, out[5] = i3;
    };
    std::sort(vals.begin(), vals.end(), [&](const Value& a, const Value& b) {
        int64_t ka[6], kb[6];
        key(a, ka), key(b, kb);
        for (int t = 0; t < 6; ++t)
            if (ka[t] != kb[t]) return ka[t] < kb[t];
        return false;
    });
    return scatter_values(vals, points_per_voxel, seed, B, pos);
}
