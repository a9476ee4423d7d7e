"""Minimal reader of OpenVDB files of format version 224 holding FloatGrids (Tree_float_5_4_3) — test infrastructure.

An independent restatement of the READ side of the library the reference links (openvdb 4.0.2): io/Archive.cc
readHeader (:863-935) / readGridDescriptors, io/GridDescriptor.cc:75-104, MetaMap.cc readMeta, math/Maps.h
ScaleMap::read, tree/RootNode.h readTopology (:2291-2403) / readBuffers, tree/InternalNode.h readTopology
(:2198-2290), tree/LeafNode.h readTopology / readBuffers (:1331-1441), io/Compression.h readCompressedValues
(:323-456).  Supports compression flags NONE, ACTIVE_MASK and ZIP (no Blosc payloads), every mask-compression
metadata code, active tiles at the root and in internal nodes.  Returns per grid: name, metadata, transform and a
function that densifies a box."""
import struct
import numpy as np


class R:
    def __init__(self, b):
        self.b, self.p = b, 0

    def take(self, n):
        v = self.b[self.p:self.p + n]
        assert len(v) == n, "truncated file"
        self.p += n
        return v

    def u(self, fmt):
        v = struct.unpack_from("<" + fmt, self.b, self.p)
        self.p += struct.calcsize("<" + fmt)
        return v if len(v) > 1 else v[0]

    def string(self):
        return self.take(self.u("I")).decode("latin-1")


def _meta(r):
    out = {}
    for _ in range(r.u("I")):
        name, typ = r.string(), r.string()
        raw = r.take(r.u("I"))
        out[name] = {"string": lambda b: b.decode(), "vec3i": lambda b: struct.unpack("<3i", b), "int64": lambda b: struct.unpack("<q", b)[0],
                     "int32": lambda b: struct.unpack("<i", b)[0], "float": lambda b: struct.unpack("<f", b)[0], "bool": lambda b: b != b"\0",
                     "vec3d": lambda b: struct.unpack("<3d", b)}.get(typ, lambda b: b)(raw)
    return out


def _mask(r, log2dim):
    nbits = 1 << (3 * log2dim)
    words = np.frombuffer(r.take(nbits // 8), dtype="<u8")
    return np.unpackbits(words.view(np.uint8), bitorder="little").astype(bool)   # bit n of word n>>6


def _values(r, count, value_mask, child_mask, background, compression):
    """readCompressedValues for float, file version >= 222."""
    assert not compression & ~0x3, "Blosc payloads are not supported by this test reader"
    meta = r.u("b")
    inactive = [background, background]
    if meta == 1:
        inactive[0] = -background
    if meta in (2, 4, 5):
        inactive[0] = r.u("f")
        if meta == 5:
            inactive[1] = r.u("f")
    if meta == 3:
        inactive = [-background, background]  # mask selects between -background (off) and +background (on)
    sel = _mask(r, {512: 3, 4096: 4, 32768: 5}[count]) if meta in (3, 4, 5) else None
    mask_compressed = bool(compression & 0x2) and meta != 6
    nread = int(value_mask.sum()) if mask_compressed else count
    if compression & 0x1:
        # unzipFromStream (io/Compression.cc:103-150): int64 byte count; <= 0: that many raw bytes follow, > 0: zlib stream
        import zlib
        nz = r.u("q")
        raw = r.take(-nz) if nz <= 0 else zlib.decompress(r.take(nz))
        assert len(raw) == 4 * nread, (len(raw), nread)
        data = np.frombuffer(raw, dtype="<f4")
    else:
        data = np.frombuffer(r.take(4 * nread), dtype="<f4")
    if not mask_compressed:
        return data.copy()
    out = np.empty(count, dtype=np.float32)
    out[value_mask] = data
    off = ~value_mask
    if sel is None:
        out[off] = inactive[0]
    else:
        out[off] = np.where(sel[off], inactive[1], inactive[0])
    return out


class Grid:
    def __init__(self):
        self.leaves = {}      # origin -> (values(512,), mask(512,))
        self.tiles = []       # (origin, dim, value, active)

    def dense(self, lo, hi):
        """(values, active) over the index box [lo,hi]^3, z fastest."""
        n = hi - lo + 1
        v = np.full((n, n, n), self.background, dtype=np.float32)
        a = np.zeros((n, n, n), dtype=bool)
        def put(origin, dim, val, act):
            s = [slice(max(o, lo) - lo, min(o + dim - 1, hi) - lo + 1) for o in origin]
            src = [slice(max(o, lo) - o, min(o + dim - 1, hi) - o + 1) for o in origin]
            if any(x.start >= x.stop for x in s):
                return
            v[tuple(s)] = val[tuple(src)] if isinstance(val, np.ndarray) else val
            a[tuple(s)] = act[tuple(src)] if isinstance(act, np.ndarray) else act
        for origin, dim, val, act in self.tiles:
            put(origin, dim, val, act)
        for origin, (val, msk) in self.leaves.items():
            put(origin, 8, val.reshape(8, 8, 8), msk.reshape(8, 8, 8))
        return v, a


def _internal_topology(r, g, origin, log2dim, child_total, compression, order):
    """InternalNode::readTopology; child_total = log2 of the child's edge (7 for 16^3 of leaves... 3 for leaves)."""
    count = 1 << (3 * log2dim)
    child_mask, value_mask = _mask(r, log2dim), _mask(r, log2dim)
    vals = _values(r, count, value_mask, child_mask, g.background, compression)
    dim_child = 1 << child_total
    for n in range(count):
        x, y, z = n >> (2 * log2dim), (n >> log2dim) & ((1 << log2dim) - 1), n & ((1 << log2dim) - 1)
        org = (origin[0] + x * dim_child, origin[1] + y * dim_child, origin[2] + z * dim_child)
        if child_mask[n]:
            if child_total == 3:
                order.append(org)
                g.leaves[org] = [None, _mask(r, 3)]          # LeafNode::readTopology: the value mask
            else:
                _internal_topology(r, g, org, 4, 3, compression, order)
        elif value_mask[n] or vals[n] != g.background:
            g.tiles.append((org, dim_child, float(vals[n]), bool(value_mask[n])))


def read(path):
    r = R(open(path, "rb").read())
    magic, version, major, minor, has_offsets = r.u("q"), r.u("I"), r.u("I"), r.u("I"), r.u("b")
    assert magic == 0x56444220 and version >= 222, (hex(magic), version)
    uuid = r.take(36).decode()
    assert [len(x) for x in uuid.split("-")] == [8, 4, 4, 4, 12] and all(c in "0123456789abcdef-" for c in uuid), uuid
    info = {"version": version, "library": (major, minor), "has_offsets": bool(has_offsets), "uuid": uuid, "metadata": _meta(r)}
    grids = []
    for _ in range(r.u("i")):
        g = Grid()
        g.unique_name, g.type, g.instance_parent = r.string(), r.string(), r.string()
        g.name = g.unique_name.split("\x1e")[0]
        assert g.type == "Tree_float_5_4_3", g.type
        grid_pos, block_pos, end_pos = r.u("3q")
        if has_offsets:
            assert grid_pos == r.p, (grid_pos, r.p)
        compression = r.u("I")
        g.compression = compression
        g.metadata = _meta(r)
        g.map_type = r.string()
        assert g.map_type in ("UniformScaleMap", "ScaleMap"), g.map_type
        m = np.frombuffer(r.take(15 * 8), dtype="<f8").reshape(5, 3)
        g.scale, g.voxel_size, g.inv_scale, g.inv_scale_sqr, g.inv_twice_scale = m
        assert r.u("i") == 1                                   # buffer count
        g.background = r.u("f")
        ntiles, nchildren = r.u("2I")
        for _ in range(ntiles):
            org = r.u("3i"); val = r.u("f"); act = r.u("b")
            g.tiles.append((org, 4096, val, bool(act)))
        order = []
        g.root_children = []
        for _ in range(nchildren):
            org = r.u("3i")
            g.root_children.append(org)
            _internal_topology(r, g, org, 5, 7, compression, order)
        if has_offsets:
            assert block_pos == r.p, (block_pos, r.p)
        for org in order:                                      # LeafNode::readBuffers, same traversal order
            msk = _mask(r, 3)
            assert np.array_equal(msk, g.leaves[org][1])
            g.leaves[org][0] = _values(r, 512, msk, np.zeros(512, bool), g.background, compression)
        if has_offsets:
            assert end_pos == r.p, (end_pos, r.p)
        grids.append(g)
    assert r.p == len(r.b), "trailing bytes"
    return info, grids
