"""`.vdb` writer: files re-read by tests/vdb_reader.py, an independent restatement of the library's
read side.  No GPU needed: the writer is host code in libfluid_hip.so.  (Parity against the real OpenVDB is unpinned:
there is no OpenVDB in the image and no sample .vdb in the reference tree.)"""
import numpy as np
import pytest

import vdb_reader


@pytest.fixture(scope="module")
def fs():
    import __graft_entry__ as entry
    return entry.load_package()


@pytest.mark.parametrize("n,compression", [(8, "zip"), (24, "zip"), (121, "zip"), (24, "active_mask"), (130, "active_mask")])
def test_round_trip(fs, tmp_path, n, compression):
    rng = np.random.default_rng(n)
    rho = rng.random((n, n, n), dtype=np.float32)
    rho[rng.random((n, n, n)) < 0.7] = 0.0                      # mostly empty, like a density grid
    path = tmp_path / f"g{n}.vdb"
    fs.write_vdb(path, rho, compression=compression)
    info, grids = vdb_reader.read(path)
    assert info["version"] == 224 and info["library"] == (4, 0) and info["has_offsets"] and info["metadata"] == {}
    assert len(grids) == 1
    g = grids[0]
    lo, hi = fs.grid_bounds(n)
    assert g.name == "" and g.unique_name == "\x1e0" and g.instance_parent == "" and g.background == 0.0
    assert g.map_type == "UniformScaleMap" and np.all(g.voxel_size == 1.0) and np.all(g.inv_twice_scale == 0.5)
    assert g.metadata["file_bbox_min"] == (lo, lo, lo) and g.metadata["file_bbox_max"] == (hi, hi, hi)
    assert g.metadata["file_voxel_count"] == n ** 3
    assert g.metadata["file_compression"] == ("zip + active values" if compression == "zip" else "active values")   # io/Compression.cc:48-58
    assert g.compression == (3 if compression == "zip" else 2)                                                        # io/Compression.h:78-81
    leaves_per_axis = len(range(lo & ~7, hi + 1, 8))
    assert g.metadata["file_mem_bytes"] > leaves_per_axis ** 3 * (2048 + 96)    # leaves + internal nodes + root (Tree::memUsage)
    assert g.tiles == []                                         # voxelizeActiveTiles(): leaves only
    vals, act = g.dense(lo - 3, hi + 3)                          # a margin: nothing active outside [lo,hi]^3
    inner = (slice(3, 3 + n),) * 3
    assert np.array_equal(vals[inner], rho)                      # bit-exact values
    assert act[inner].all() and act.sum() == n ** 3              # every cell of the box active (fluid.cc:1163), nothing else
    assert (vals[~act] == 0).all()
    # tree shape: 8^3 leaves aligned to multiples of 8 in index space, root children in ascending (x,y,z) order
    assert all(o[0] % 8 == 0 and o[1] % 8 == 0 and o[2] % 8 == 0 for o in g.leaves)
    per_axis = len(range(lo & ~7, hi + 1, 8))
    assert len(g.leaves) == per_axis ** 3
    assert g.root_children == sorted(g.root_children) and len(g.root_children) == (2 if lo < 0 <= hi else 1) ** 3


def test_several_grids_get_unique_names(fs, tmp_path):
    """The reference writes its growing grids2 vector (fluid.cc:1451,1503): unnamed grids are told apart by the
    "\\x1e<k>" suffix of io/Archive.cc:1196-1206."""
    n = 16
    gs = [np.full((n, n, n), float(k + 1), dtype=np.float32) for k in range(3)]
    path = tmp_path / "many.vdb"
    fs.write_vdb(path, gs)
    info, grids = vdb_reader.read(path)
    assert [g.unique_name for g in grids] == ["\x1e0", "\x1e1", "\x1e2"]
    lo, hi = fs.grid_bounds(n)
    for k, g in enumerate(grids):
        assert np.array_equal(g.dense(lo, hi)[0], gs[k])


def test_zip_shrinks_and_default_is_the_librarys(fs, tmp_path):
    n = 32
    rho = np.zeros((n, n, n), dtype=np.float32)
    rho[8:20, 8:20, 8:20] = 1.5                                  # a density blob: long runs of equal values
    a, b = tmp_path / "z.vdb", tmp_path / "m.vdb"
    fs.write_vdb(a, rho)                                         # default = ZIP | ACTIVE_MASK
    fs.write_vdb(b, rho, compression="active_mask")
    assert a.stat().st_size < b.stat().st_size / 2       # (the node masks stay raw)
    ga, gb = vdb_reader.read(a)[1][0], vdb_reader.read(b)[1][0]
    assert ga.compression == 3 and gb.compression == 2
    lo, hi = fs.grid_bounds(n)
    assert np.array_equal(ga.dense(lo, hi)[0], rho) and np.array_equal(gb.dense(lo, hi)[0], rho)


def test_stream_of_every_steps_grid(fs, tmp_path):
    """The reference's final mygrids.vdb holds EVERY step's grid (`grids` lives outside the loop, fluid.cc:1366,1450,1508):
    written as a stream, one grid appended per step; 500 grids re-read one by one."""
    n, steps = 8, 500
    path = tmp_path / "mygrids.vdb"
    w = fs.VdbStream(path, n, steps)
    rng = np.random.default_rng(3)
    want = []
    for k in range(steps):
        g = (rng.random((n, n, n), dtype=np.float32) * (rng.random((n, n, n)) < 0.3)).astype(np.float32)
        want.append(g)
        w.append(g)
    w.close()
    info, grids = vdb_reader.read(path)
    assert len(grids) == steps
    lo, hi = fs.grid_bounds(n)
    assert [g.unique_name for g in grids[:3]] == ["\x1e0", "\x1e1", "\x1e2"] and grids[-1].unique_name == "\x1e499"
    for k in (0, 1, 17, 250, 499):
        assert np.array_equal(grids[k].dense(lo, hi)[0], want[k])
    # closing before every announced grid was appended is an error (the header already holds the count)
    w2 = fs.VdbStream(tmp_path / "short.vdb", n, 3)
    w2.append(want[0])
    with pytest.raises(fs.FluidError):
        w2.close()


def test_bad_arguments(fs, tmp_path):
    with pytest.raises(fs.FluidError):
        fs.write_vdb(tmp_path / "no_such_dir" / "x.vdb", np.zeros((8, 8, 8), np.float32))
