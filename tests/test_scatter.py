"""The reference's initial particles: fluid_scene_uniform_scatter against an independent restatement.

The reference cannot be built here (OpenVDB needs TBB/Boost/Half), so there is no golden output of its own: "parity
unpinned".  What can be checked: (1) the C++ (std::mt19937 + libstdc++'s distributions, recursive tree walk) against a
second restatement written differently (numpy's MT19937 raw stream = init_genrand(seed), the distributions' algorithms
spelled out from libstdc++ 11's bits/uniform_int_dist.h and bits/random.tcc, flat loops over the 8^3 blocks);
(2) the one compiler-dependent assumption — g++ evaluates the three getRand() constructor arguments right to left — with
the image's g++; (3) counts and bounds, including a box that holds a whole 128^3 tile.  No GPU.
"""
import os
import shutil
import subprocess

import numpy as np
import pytest

import __graft_entry__ as entry

fs = entry.load_package()


def _raw_stream(seed, n):
    """n raw 32-bit outputs of mt19937(seed) (init_genrand seeding, as std::mt19937(seed))."""
    rs = np.random.RandomState(seed)
    return rs.randint(0, 2 ** 32, size=n, dtype=np.uint64)


def _values_in_order(lo, hi, tiles=True):
    """Active values of fill(CoordBBox(lo, hi)) in ValueOn order, for boxes that stay inside the 128^3 blocks next to the
    origin: (x, y, z, dim) with dim 8 for a fully covered leaf (tile) and 1 for a voxel of a partly covered one."""
    assert -128 <= lo and hi <= 127
    out = []
    starts = [b for b in range(-128, 128, 8) if b + 7 >= lo and b <= hi]
    for rx in (-1, 0):          # root table: origin -4096 before 0, x then y then z
        for ry in (-1, 0):
            for rz in (-1, 0):
                def mine(b, r):
                    return (b < 0) == (r < 0)
                for bx in [b for b in starts if mine(b, rx)]:
                    for by in [b for b in starts if mine(b, ry)]:
                        for bz in [b for b in starts if mine(b, rz)]:
                            if tiles and lo <= bx and bx + 7 <= hi and lo <= by and by + 7 <= hi and lo <= bz and bz + 7 <= hi:
                                out.append((bx, by, bz, 8))
                                continue
                            for x in range(max(bx, lo), min(bx + 7, hi) + 1):
                                for y in range(max(by, lo), min(by + 7, hi) + 1):
                                    for z in range(max(bz, lo), min(bz + 7, hi) + 1):
                                        out.append((x, y, z, 1))
    return out


def _restated_scatter(lo, hi, ppv, seed, boundary, vals=None):
    vals = _values_in_order(lo, hi) if vals is None else vals
    counts = np.array([v[3] ** 3 for v in vals], dtype=np.int64)
    voxels = int(counts.sum())
    target = int(ppv) * voxels
    # std::uniform_int_distribution<uint64_t>(0, voxels - 1) on a 32-bit engine, libstdc++ 11: Lemire's nearly divisionless
    # method (_S_nd): product = u32 * range; reject while low32(product) < (2^32 - range) % range
    rng = voxels
    raw = _raw_stream(seed, target + 4096)
    thr = (2 ** 32 - rng) % rng
    ids = np.empty(target, dtype=np.int64)
    k = 0
    raw_l = raw.tolist()
    for i in range(target):
        while True:
            prod = raw_l[k] * rng
            k += 1
            if (prod & 0xFFFFFFFF) >= thr or (prod & 0xFFFFFFFF) >= rng:
                break
        ids[i] = prod >> 32
    ids.sort()
    # std::uniform_real_distribution<double> = generate_canonical<double, 53>: two draws, sum = a + b * 2^32 in double, / 2^64
    raw2 = _raw_stream(seed, 6 * target).astype(np.float64)
    r = (raw2[0::2] + raw2[1::2] * 4294967296.0) / 18446744073709551616.0
    r = np.where(r >= 1.0, np.nextafter(1.0, 0.0), r)
    g = 0.5 + 1.0 * (r - 0.5)                       # getRand(), spread 1
    gz, gy, gx = g[0::3], g[1::3], g[2::3]          # constructor arguments right to left
    ends = np.cumsum(counts)
    which = np.searchsorted(ends, ids, side="right")
    v = np.array(vals, dtype=np.float64)[which]
    dim = v[:, 3]
    pos = np.stack([(v[:, 0] - 0.5) + dim * gx, (v[:, 1] - 0.5) + dim * gy, (v[:, 2] - 0.5) + dim * gz], axis=1)
    keep = (np.abs(pos) < boundary - 2).all(axis=1) if boundary > 0 else np.ones(len(pos), bool)
    return pos[keep]


def test_lemire_rejection_rule_matches_libstdcxx():
    """The break condition above is libstdc++'s: `if (low < range) { threshold = -range % range; while (low < threshold) redo; }`."""
    rng = 68921
    thr = (2 ** 32 - rng) % rng
    for low in (0, thr - 1, thr, rng - 1, rng, 2 ** 32 - 1):
        redo = low < rng and low < thr
        assert redo == (not ((low >= thr) or (low >= rng)))


def test_reference_scene_against_the_restatement():
    pos = fs.reference_scatter()                     # lo -20, hi 20, 10.f, seed 0, boundary 60
    assert pos.shape == (689210, 3)                  # Index64(10.f) * 41^3
    ref = _restated_scatter(-20, 20, 10.0, 0, 60)
    assert ref.shape == pos.shape
    assert np.array_equal(pos, ref)                  # same doubles, point for point
    assert pos.min() >= -20.5 and pos.max() < 20.5
    cells = np.round(pos).astype(np.int64) + 20
    cnt = np.bincount((cells[:, 0] * 41 + cells[:, 1]) * 41 + cells[:, 2], minlength=41 ** 3)
    assert abs(cnt.mean() - 10.0) < 1e-9 and cnt.max() < 40


def test_other_seed_box_and_filter():
    a = fs.reference_scatter(lo=-9, hi=13, points_per_volume=3.0, seed=7, boundary=12)   # the filter cuts at |p| < 10
    b = _restated_scatter(-9, 13, 3.0, 7, 12)
    assert len(a) < 3 * 23 ** 3 and np.array_equal(a, b)
    assert np.abs(a).max() < 10.0


def test_random_boxes_against_the_restatement():
    """Seeded sweep over small boxes anywhere in the two 128^3 blocks around the origin (tiles, partial leaves, boxes that
    straddle the origin in some axes only), densities 1-3, with and without the boundary filter."""
    rng = np.random.default_rng(2024)
    for _ in range(25):
        side = int(rng.integers(1, 22))
        lo = int(rng.integers(-100, 100 - side))
        hi = lo + side - 1
        ppv = float(rng.integers(1, 4))
        seed = int(rng.integers(0, 2 ** 31))
        boundary = int(rng.choice([0, max(abs(lo), abs(hi)) + 1, 200]))
        a = fs.reference_scatter(lo=lo, hi=hi, points_per_volume=ppv, seed=seed, boundary=boundary)
        b = _restated_scatter(lo, hi, ppv, seed, boundary)
        assert a.shape == b.shape and np.array_equal(a, b), (lo, hi, ppv, seed, boundary)


def test_box_with_a_whole_128_tile():
    """[-130, 5]^3 covers the node [-128, -1]^3 completely: one active tile of 128^3 voxels at the second tree level."""
    n = 136
    pos = fs.reference_scatter(lo=-130, hi=5, points_per_volume=1.0, seed=3, boundary=0)
    assert pos.shape == (n ** 3, 3)
    assert pos.min() >= -130.5 and pos.max() < 5.5
    inside = ((pos >= -128.5) & (pos < -0.5)).all(axis=1).mean()
    assert abs(inside - (128 / n) ** 3) < 2e-3       # uniform over the box: the tile gets its share


def test_snow_cone_of_the_mpm_program():
    """mpm_scene_cone (mpm.cc:1037-1052,1274-1278): 16 single voxels set one by one (no tiles), walked in
    ValueOn order — here taken from the flat leaf-by-leaf enumeration of every voxel of [-16, 15]^3, filtered to the cone."""
    B, W = 15, 13
    cone = [(x, y, z, 1) for (x, y, z, d) in _values_in_order(-16, 15, tiles=False)
            if -W <= y <= -W + 3 and abs(x) <= W and abs(z) <= W and x * x + z * z <= ((y + W) / 2) ** 2]
    assert len(cone) == 16
    for ppv, seed in ((400.0, 0), (7.0, 5)):
        pos = fs.snow_cone(B=B, W=W, layers=4, points_per_voxel=ppv, seed=seed)
        ref = _restated_scatter(0, 0, ppv, seed, B, vals=cone)
        assert pos.shape == ref.shape and np.array_equal(pos, ref)
    assert len(fs.snow_cone()) == 6205
    assert fs.lib.mpm_scene_cone(15, 13, 0, 400.0, 0, None) < 0 and fs.lib.mpm_scene_cone(2, 1, 1, 400.0, 0, None) < 0


def test_bad_arguments():
    with pytest.raises(ValueError):
        fs.reference_scatter(lo=3, hi=2)
    with pytest.raises(ValueError):
        fs.reference_scatter(points_per_volume=0.0)


@pytest.mark.skipif(shutil.which("g++") is None, reason="needs g++")
def test_gxx_evaluates_constructor_arguments_right_to_left(tmp_path):
    """BasePointScatter::addPoint builds Vec3R(dmin[0] + getRand(), dmin[1] + getRand(), dmin[2] + getRand()): the order of
    the three calls is unspecified by the language; the reference is built by g++ (run.sh:3), which goes right to left."""
    src = tmp_path / "order.cpp"
    src.write_text("#include <cstdio>\nstruct V { double a, b, c; V(double x, double y, double z) : a(x), b(y), c(z) {} };\n"
                   "static int k = 0;\nstatic double f() { return (double)k++; }\n"
                   "int main() { const V v(1.0 + f(), 2.0 + f(), 3.0 + f()); std::printf(\"%g %g %g\\n\", v.a - 1, v.b - 2, v.c - 3); }\n")
    exe = tmp_path / "order"
    subprocess.run(["g++", "-std=c++11", "-O3", str(src), "-o", str(exe)], check=True)
    out = subprocess.run([str(exe)], check=True, capture_output=True, text=True).stdout.split()
    assert out == ["2", "1", "0"]
