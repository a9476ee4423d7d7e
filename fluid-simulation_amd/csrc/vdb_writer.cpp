// .vdb writer for the step's output FloatGrid — host only, no GPU, no OpenVDB.
//
// Writes what `openvdb::io::File(name).write(grids)` of the reference's OpenVDB 4.0.2 writes for grids created as in
// fluid.cc:1161-1164 (FloatGrid::create(0), linear transform of voxel size 1, fill([lo,hi]^3, 0, active) +
// voxelizeActiveTiles, unnamed) — restated from the library's serialisation code, file format version 224:
//   header            io/Archive.cc:939-971   magic (int64), file version 224, library 4.0, has-offsets flag, 36-char uuid
//   file metadata     MetaMap.cc:117-136      (empty map: count 0)
//   grid count        io/Archive.cc:1167-1173
//   per grid          io/Archive.cc:1243-1328 descriptor (unique name, "Tree_float_5_4_3", instance parent; three stream
//                                             offsets, io/GridDescriptor.cc:53-73), compression flags (:704-723), metadata
//                                             with the file_* statistics (Grid.cc:446-457), transform (math/Transform.cc:
//                                             178-186, math/Maps.h ScaleMap::write), topology, buffers
//   tree              tree/Tree.h:1297-1301,1439-1443  buffer count 1, then the root
//   RootNode          tree/RootNode.h:2257-2288,2407-2412  background, #tiles, #children, children by ascending origin
//   InternalNode      tree/InternalNode.h:2175-2195,3032-3037  child mask, value mask, compressed tile values, children
//   LeafNode          tree/LeafNode.h:1321-1324,1444-1453  value mask (topology); value mask again + compressed values
//   node values       io/Compression.h:462-639  one metadata byte per node; with COMPRESS_ACTIVE_MASK and every inactive
//                                              value equal to the background only the active values follow
//   masks             util/NodeMasks.h:565-568  raw 64-bit words, bit n of word n>>6
//   offsets           tree/LeafNode.h:1049-1055, tree/InternalNode.h:3098-3103  x-major, z fastest
// Compression: COMPRESS_ZIP | COMPRESS_ACTIVE_MASK by default — the library's default without Blosc
// (io/Compression.h:78-81; io/Archive.cc DEFAULT_COMPRESSION_FLAGS) — every value buffer goes through zipToStream
// (io/Compression.cc:70-100: compress2 at Z_DEFAULT_COMPRESSION, an int64 byte count in front, negative and followed by the
// raw bytes when zipping did not shrink the buffer); FLUID_VDB_ACTIVE_MASK alone is kept for readers without zlib.
// Which grids go into which file (fluid.cc:1366-1373,1450-1451,1503-1508): `grids2` is declared INSIDE the step loop, so
// simulation/mygrids<i>.vdb holds exactly the grid of step i; `grids` is declared outside it and grows by one grid per
// step, so the final mygrids.vdb holds every step's grid (500 of them) — written here as a stream, one grid appended per
// step (fluid_vdb_open / _append / _close), names get the "\x1e<k>" suffixes of io/Archive.cc:1196-1206.
// Parity status: UNPINNED against the real library (no OpenVDB here to read the files back, no sample .vdb in the
// reference tree); tests/test_vdb.py re-reads the files with an independent restatement of the READ side.
#include <zlib.h>

#include <cstdint>
#include <cstdio>
#include <cstring>
#include <ctime>
#include <random>
#include <string>
#include <thread>
#include <vector>

#include "fluid_hip.h"

namespace {

// buffered stream to the file (a 512^3 grid is 0.6 GB: nothing is held in memory); offsets are patched by seeking back
struct Out {
    FILE* f = nullptr;
    size_t at = 0;
    bool ok = true;
    std::vector<char> buf;
    void flush()
    {
        if (!buf.empty() && fwrite(buf.data(), 1, buf.size(), f) != buf.size()) ok = false;
        buf.clear();
    }
    void raw(const void* p, size_t n)
    {
        const char* c = (const char*)p;
        buf.insert(buf.end(), c, c + n);
        at += n;
        if (buf.size() >= (size_t)8 << 20) flush();
    }
    template <typename T> void put(T v) { raw(&v, sizeof(T)); }
    void str(const std::string& s) { put<uint32_t>((uint32_t)s.size()); raw(s.data(), s.size()); }   // util/Name.h:57-63
    size_t pos() const { return at; }
    void patch64(size_t where, int64_t v)
    {
        flush();
        if (fseeko(f, (off_t)where, SEEK_SET) != 0 || fwrite(&v, 1, 8, f) != 8 || fseeko(f, 0, SEEK_END) != 0) ok = false;
    }
    // writeData (io/Compression.h:253-264): zipToStream with COMPRESS_ZIP, the raw bytes otherwise
    std::vector<unsigned char> zbuf;
    void data(const void* p, size_t n, uint32_t compression)
    {
        if (!(compression & 0x1)) { raw(p, n); return; }
        uLongf zn = compressBound((uLong)n);
        zbuf.resize(zn);
        const int st = compress2(zbuf.data(), &zn, (const Bytef*)p, (uLong)n, Z_DEFAULT_COMPRESSION);
        if (st == Z_OK && zn < n) {
            put<int64_t>((int64_t)zn);
            raw(zbuf.data(), zn);
        } else {
            put<int64_t>(-(int64_t)n);
            raw(p, n);
        }
    }
};

inline int floor_to(int v, int m) { return v & ~(m - 1); }  // origin of the node of size m (power of two) that holds v

constexpr int LEAF = 8, INT1 = 128, INT2 = 4096;             // Tree_float_5_4_3: 8^3 leaves, 16^3 of them, 32^3 of those
constexpr uint32_t COMPRESS_ZIP = 0x1, COMPRESS_ACTIVE_MASK = 0x2;   // io/Compression.h:79-80
constexpr int8_t NO_MASK_OR_INACTIVE_VALS = 0;               // io/Compression.h:94

struct Dense {
    int n, lo, hi;
    const float* v;
    bool inside(int x, int y, int z) const { return x >= lo && x <= hi && y >= lo && y <= hi && z >= lo && z <= hi; }
    float at(int x, int y, int z) const { return v[((size_t)(x - lo) * n + (y - lo)) * n + (z - lo)]; }
    bool overlaps(int ox, int oy, int oz, int dim) const
    {
        return ox + dim - 1 >= lo && ox <= hi && oy + dim - 1 >= lo && oy <= hi && oz + dim - 1 >= lo && oz <= hi;
    }
};

// value mask of the leaf at (ox,oy,oz): voxels inside the box are active (fill(..., active=true), fluid.cc:1163)
void leaf_mask(const Dense& g, int ox, int oy, int oz, uint64_t m[8])
{
    for (int w = 0; w < 8; ++w) m[w] = 0;
    for (int x = 0; x < 8; ++x)
        for (int y = 0; y < 8; ++y)
            for (int z = 0; z < 8; ++z)
                if (g.inside(ox + x, oy + y, oz + z)) {
                    const int n = (x << 6) + (y << 3) + z;
                    m[n >> 6] |= 1ull << (n & 63);
                }
}

// an internal node with no active tiles and background tile values: all-off value mask, metadata byte 0, no values
void internal_values(Out& o, size_t mask_bytes, uint32_t compression)
{
    std::vector<char> zero(mask_bytes, 0);
    o.raw(zero.data(), mask_bytes);          // mValueMask
    o.put<int8_t>(NO_MASK_OR_INACTIVE_VALS);   // writeCompressedValues: every inactive value == background, 0 active values
    o.data(nullptr, 0, compression);           // ... and the (empty) value array through writeData: an int64 0 with ZIP
}

// One leaf of the buffers pass: mask, active values, and (ZIP) the zipToStream framing of the values, prepared off the output
// stream so that the leaves of a 128^3 node can be compressed on several host threads (zlib at the library's default level costs
// ~20 us per leaf: 80 ms for a 121^3 grid on one thread, written twice per step by the driver); the bytes are those of the
// serial path, in the same order.
struct LeafJob {
    int lx, ly, lz;
    uint64_t vm[8];
    int na;
    float act[512];
    int64_t head;                       // zip: the int64 in front (compressed size, or -raw size)
    std::vector<unsigned char> z;       // zip: the compressed bytes (empty: the raw values follow)
};
void prepare_leaf(const Dense& g, LeafJob& j, uint32_t compression)
{
    leaf_mask(g, j.lx, j.ly, j.lz, j.vm);
    j.na = 0;
    for (int x = 0; x < 8; ++x)
        for (int y = 0; y < 8; ++y)
            for (int z = 0; z < 8; ++z)
                if (g.inside(j.lx + x, j.ly + y, j.lz + z)) j.act[j.na++] = g.at(j.lx + x, j.ly + y, j.lz + z);
    j.z.clear();
    j.head = 0;
    if (compression & COMPRESS_ZIP) {
        const size_t n = (size_t)j.na * sizeof(float);
        uLongf zn = compressBound((uLong)n);
        j.z.resize(zn);
        const int st = compress2(j.z.data(), &zn, (const Bytef*)j.act, (uLong)n, Z_DEFAULT_COMPRESSION);
        if (st == Z_OK && zn < n) { j.head = (int64_t)zn; j.z.resize(zn); }
        else { j.head = -(int64_t)n; j.z.clear(); }
    }
}
void flush_leaves(Out& o, const Dense& g, std::vector<LeafJob>& jobs, uint32_t compression)
{
    const size_t n = jobs.size();
    unsigned nt = std::thread::hardware_concurrency();
    nt = nt < 1 ? 1 : (nt > 16 ? 16 : nt);
    if (n < 64 || !(compression & COMPRESS_ZIP)) nt = 1;
    if (nt == 1) {
        for (auto& j : jobs) prepare_leaf(g, j, compression);
    } else {
        std::vector<std::thread> th;
        for (unsigned t = 0; t < nt; ++t)
            th.emplace_back([&, t] { for (size_t i = t; i < n; i += nt) prepare_leaf(g, jobs[i], compression); });
        for (auto& x : th) x.join();
    }
    for (auto& j : jobs) {
        o.raw(j.vm, sizeof(j.vm));
        o.put<int8_t>(NO_MASK_OR_INACTIVE_VALS);  // inactive voxels hold the background
        if (compression & COMPRESS_ZIP) {
            o.put<int64_t>(j.head);
            if (j.head > 0) o.raw(j.z.data(), j.z.size());
            else o.raw(j.act, (size_t)j.na * sizeof(float));
        } else {
            o.raw(j.act, (size_t)j.na * sizeof(float));
        }
    }
    jobs.clear();
}

void write_tree(Out& o, const Dense& g, bool buffers, uint32_t compression)
{
    std::vector<LeafJob> jobs;
    // root children in ascending (x,y,z) origin order (std::map<Coord>, math/Coord.h:180-185)
    for (int rx = floor_to(g.lo, INT2); rx <= g.hi; rx += INT2)
        for (int ry = floor_to(g.lo, INT2); ry <= g.hi; ry += INT2)
            for (int rz = floor_to(g.lo, INT2); rz <= g.hi; rz += INT2) {
                if (!buffers) {
                    const int32_t org[3] = {rx, ry, rz};
                    o.raw(org, sizeof(org));
                    std::vector<uint64_t> cm(32 * 32 * 32 / 64, 0);  // child mask of the 4096^3 node: 32^3 slots of 128^3
                    for (int a = 0; a < 32; ++a)
                        for (int b = 0; b < 32; ++b)
                            for (int c = 0; c < 32; ++c)
                                if (g.overlaps(rx + a * INT1, ry + b * INT1, rz + c * INT1, INT1)) {
                                    const int n = (a << 10) + (b << 5) + c;
                                    cm[n >> 6] |= 1ull << (n & 63);
                                }
                    o.raw(cm.data(), cm.size() * 8);
                    internal_values(o, cm.size() * 8, compression);
                }
                for (int a = 0; a < 32; ++a)
                    for (int b = 0; b < 32; ++b)
                        for (int c = 0; c < 32; ++c) {
                            const int ix = rx + a * INT1, iy = ry + b * INT1, iz = rz + c * INT1;
                            if (!g.overlaps(ix, iy, iz, INT1)) continue;
                            if (!buffers) {
                                uint64_t cm[64] = {};  // 16^3 slots of 8^3
                                for (int p = 0; p < 16; ++p)
                                    for (int q = 0; q < 16; ++q)
                                        for (int r = 0; r < 16; ++r)
                                            if (g.overlaps(ix + p * LEAF, iy + q * LEAF, iz + r * LEAF, LEAF)) {
                                                const int n = (p << 8) + (q << 4) + r;
                                                cm[n >> 6] |= 1ull << (n & 63);
                                            }
                                o.raw(cm, sizeof(cm));
                                internal_values(o, sizeof(cm), compression);
                            }
                            for (int p = 0; p < 16; ++p)
                                for (int q = 0; q < 16; ++q)
                                    for (int r = 0; r < 16; ++r) {
                                        const int lx = ix + p * LEAF, ly = iy + q * LEAF, lz = iz + r * LEAF;
                                        if (!g.overlaps(lx, ly, lz, LEAF)) continue;
                                        if (buffers) {   // (the leaves of this 128^3 node are written below, together)
                                            jobs.emplace_back();
                                            jobs.back().lx = lx; jobs.back().ly = ly; jobs.back().lz = lz;
                                            continue;
                                        }
                                        uint64_t vm[8];
                                        leaf_mask(g, lx, ly, lz, vm);
                                        o.raw(vm, sizeof(vm));
                                    }
                            if (buffers) flush_leaves(o, g, jobs, compression);
                        }
            }
}

template <typename T> void meta(Out& o, const char* name, const char* type, const T* v, uint32_t bytes)
{
    o.str(name); o.str(type); o.put<uint32_t>(bytes); o.raw(v, bytes);   // MetaMap.cc:126-135, Metadata.h:189-218
}

// Grid::memUsage() = Tree::memUsage() of the tree this file describes, as the library reports it in file_mem_bytes
// (Grid.h:778, tree/Tree.h:387, RootNode.h:1463-1472, InternalNode.h:1115-1123, LeafNode.h:1471-1476, LeafBuffer.h:377-387),
// with the x86-64 layouts of OpenVDB 4.0.2 (ABI 3+):
//   LeafNode<float,3>   sizeof 96 (16 B buffer object + 64 B value mask + 12 B origin, padded) + 2048 B of voxels
//   InternalNode<.,4>   4096 unions of 8 B + two 512 B masks + 12 B origin;  InternalNode<.,5>: 32768 x 8 + 2 x 4096 + 12
//   RootNode            sizeof 56 (std::map + background);  Tree: vptr + root + two accessor registries
//                       (tbb::concurrent_hash_map: its size depends on the TBB release — 568 B taken; informational metadata)
int64_t tree_mem_bytes(int lo, int hi)
{
    auto count = [&](int dim) { long c = 0; for (int x = floor_to(lo, dim); x <= hi; x += dim) ++c; return c * c * c; };
    const int64_t leaves = count(LEAF), n1 = count(INT1), n2 = count(INT2);
    const int64_t leaf = 96 + 2048, int1 = 4096 * 8 + 512 + 512 + 12, int2 = 32768 * 8 + 4096 + 4096 + 12;
    const int64_t root = 56, tree = 8 + 56 + 2 * 568;
    return tree + root + n2 * int2 + n1 * int1 + leaves * leaf;
}

struct Writer {
    Out o;
    int n = 0, n_grids = 0, written = 0;
    uint32_t compression = 0;
};

int writer_open(Writer& w, const char* path, int32_t n, int32_t n_grids, int32_t compression)
{
    w.n = n; w.n_grids = n_grids; w.written = 0;
    w.compression = (uint32_t)compression;
    Out& o = w.o;
    o.f = fopen(path, "wb");
    if (!o.f) return FLUID_ERR_ARG;
    // ---- header (io/Archive.cc:939-971) ----
    o.put<int64_t>(0x56444220);            // OPENVDB_MAGIC, version.h:83
    o.put<uint32_t>(224);                  // OPENVDB_FILE_VERSION, version.h:96
    o.put<uint32_t>(4); o.put<uint32_t>(0);  // library 4.0
    o.put<char>(1);                        // seekable: grid offsets follow each descriptor (io::File)
    {
        std::mt19937 ran((unsigned)(std::random_device()() + (unsigned)std::time(nullptr)));
        char u[37];
        const uint32_t a = ran(), b = ran(), c = ran(), d = ran();
        // random (version 4) uuid, textual form of boost::uuids::operator<<
        snprintf(u, sizeof(u), "%08x-%04x-4%03x-%04x-%04x%08x", a, b >> 16, b & 0xfff, 0x8000 | (c >> 18), c & 0xffff, d);
        o.raw(u, 36);
    }
    o.put<uint32_t>(0);                    // file-level metadata: empty map
    o.put<int32_t>(n_grids);
    return FLUID_OK;
}

int writer_append(Writer& w, const float* grid)
{
    if (w.written >= w.n_grids) return FLUID_ERR_STATE;
    Out& o = w.o;
    const int n = w.n, k = w.written;
    const int lo = -(n / 2), hi = lo + n - 1;
    const Dense g{n, lo, hi, grid};
    // ---- descriptor (io/GridDescriptor.cc:53-73); unnamed grids become "\x1e<k>" (io/Archive.cc:1196-1206) ----
    o.str(std::string("\x1e") + std::to_string(k));
    o.str("Tree_float_5_4_3");
    o.str("");                         // not an instance
    const size_t off = o.pos();
    o.put<int64_t>(0); o.put<int64_t>(0); o.put<int64_t>(0);
    o.patch64(off, (int64_t)o.pos());  // grid position
    o.put<uint32_t>(w.compression);
    // ---- grid metadata: the statistics Archive::writeGrid adds (std::map order = by name) ----
    const int32_t bmin[3] = {lo, lo, lo}, bmax[3] = {hi, hi, hi};
    const int64_t voxels = (int64_t)n * n * n, mem = tree_mem_bytes(lo, hi);
    const std::string comp = (w.compression & COMPRESS_ZIP) ? "zip + active values" : "active values";  // io/Compression.cc:48-58
    o.put<uint32_t>(5);
    meta(o, "file_bbox_max", "vec3i", bmax, 12);
    meta(o, "file_bbox_min", "vec3i", bmin, 12);
    meta(o, "file_compression", "string", comp.data(), (uint32_t)comp.size());
    meta(o, "file_mem_bytes", "int64", &mem, 8);
    meta(o, "file_voxel_count", "int64", &voxels, 8);
    // ---- transform: UniformScaleMap(1.0) = scale, voxel size, 1/scale, 1/scale^2, 1/(2 scale) ----
    o.str("UniformScaleMap");
    const double one[3] = {1, 1, 1}, half[3] = {0.5, 0.5, 0.5};
    o.raw(one, 24); o.raw(one, 24); o.raw(one, 24); o.raw(one, 24); o.raw(half, 24);
    // ---- topology ----
    o.put<int32_t>(1);                 // buffer count
    o.put<float>(0.0f);                // background
    o.put<uint32_t>(0);                // root tiles
    uint32_t nroot = 0;
    for (int x = floor_to(lo, INT2); x <= hi; x += INT2) ++nroot;
    o.put<uint32_t>(nroot * nroot * nroot);
    write_tree(o, g, false, w.compression);
    o.patch64(off + 8, (int64_t)o.pos());   // block position
    write_tree(o, g, true, w.compression);
    o.patch64(off + 16, (int64_t)o.pos());  // end position
    w.written++;
    return o.ok ? FLUID_OK : FLUID_ERR_ARG;
}

int writer_close(Writer& w)
{
    w.o.flush();
    const int rc = w.o.f ? fclose(w.o.f) : 0;
    w.o.f = nullptr;
    return (w.o.ok && rc == 0 && w.written == w.n_grids) ? FLUID_OK : FLUID_ERR_ARG;
}

}  // namespace

struct fluid_vdb_writer {
    Writer w;
};

extern "C" {

int fluid_vdb_open(const char* path, int32_t n, int32_t n_grids, int32_t compression, fluid_vdb_writer_t** out)
{
    if (!path || n < 1 || n > 4096 || n_grids < 1 || !out) return FLUID_ERR_ARG;
    if (compression != FLUID_VDB_ACTIVE_MASK && compression != FLUID_VDB_ZIP_ACTIVE_MASK) return FLUID_ERR_ARG;
    fluid_vdb_writer* h = new fluid_vdb_writer();
    const int rc = writer_open(h->w, path, n, n_grids, compression);
    if (rc) { delete h; return rc; }
    *out = h;
    return FLUID_OK;
}

int fluid_vdb_append(fluid_vdb_writer_t* h, const float* grid)
{
    if (!h || !grid) return FLUID_ERR_ARG;
    return writer_append(h->w, grid);
}

int fluid_vdb_close(fluid_vdb_writer_t* h)
{
    if (!h) return FLUID_OK;
    const int rc = writer_close(h->w);
    delete h;
    return rc;
}

int fluid_write_vdb_ex(const char* path, int32_t n, int32_t n_grids, const float* const* grids, int32_t compression)
{
    if (!grids) return FLUID_ERR_ARG;
    for (int k = 0; k < n_grids; ++k)
        if (!grids[k]) return FLUID_ERR_ARG;
    fluid_vdb_writer_t* h = nullptr;
    int rc = fluid_vdb_open(path, n, n_grids, compression, &h);
    if (rc) return rc;
    for (int k = 0; k < n_grids && !rc; ++k) rc = fluid_vdb_append(h, grids[k]);
    const int rc2 = fluid_vdb_close(h);
    return rc ? rc : rc2;
}

int fluid_write_vdb(const char* path, int32_t n, int32_t n_grids, const float* const* grids)
{
    return fluid_write_vdb_ex(path, n, n_grids, grids, FLUID_VDB_ZIP_ACTIVE_MASK);
}

}  // extern "C"
