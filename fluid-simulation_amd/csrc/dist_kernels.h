// Declarations of kernels_dist.hip (decomposed runs only).
#pragma once
#include <algorithm>

#include "common.h"

namespace fl {

constexpr int HALO_MAX_BOX = 26;   // the adjacent blocks of a 3-D block decomposition
constexpr int HALO_MAX_ARR = 6;    // arrays of one element size moved by one exchange (u, v, w, ub, vb, wb)

// One side (send or receive) of a halo exchange over a dense layout: element (i, j, k) of an array sits at
// base + i*sx + j*sy + k.  Box b covers lo[b] .. lo[b]+n[b]-1 and occupies, per array, n0*n1*n2 elements of the
// staging buffer from element off[b]*narr on (the arrays of a box follow one another: one message per peer).
struct HaloArgs {
    int nbox, narr;
    long base, sx, sy;
    int lo[HALO_MAX_BOX][3], n[HALO_MAX_BOX][3];
    long off[HALO_MAX_BOX];
    void* arr[HALO_MAX_ARR];
};
void launch_halo_copy(hipStream_t st, const HaloArgs& a, int elem, void* stage, bool pack);

// owned block in GLOBAL cell indices [lo, hi) and whether a neighbour block exists on each side
struct OwnBox {
    int lo[3], hi[3];
    int has_lo[3], has_hi[3];
};

void launch_flags_box(hipStream_t st, Grid g, Box box, const uint8_t* solid, const float* container, uint8_t* flags);
void launch_row_counts(hipStream_t st, Grid g, Box own, int rx0, int ry0, int RY, int nseg, int seg, const uint8_t* flags, int* rows);
void launch_row_number(hipStream_t st, Grid g, Box own, int rx0, int ry0, int RY, int nseg, int seg, const uint8_t* flags, const int* starts,
                       int* indices);
void launch_fill_box_int(hipStream_t st, Grid g, Box box, int* a, int v);
void launch_route(hipStream_t st, Grid g, OwnBox ob, long n, Particles p, int* cnt, double* rec, int pass);
void launch_kill_ghosts(hipStream_t st, Grid g, OwnBox ob, long n, Particles p);
void launch_pack_live(hipStream_t st, long n, Particles p, double* pos, double* vel, uint32_t* ids, int* cursor);
void launch_cnt_pcg(hipStream_t st, LBox L, Box own, const uint8_t* cnt, uint8_t* out);
template <typename T>
void launch_mask_outside(hipStream_t st, MLevel m, Box own, T* a);
void launch_mg_type_local(hipStream_t st, Grid g, MLevel m, int w0, int w1, int w2, const uint8_t* flags, const uint8_t* cnt, uint8_t* typ);
void launch_copy_vel_before(hipStream_t st, Grid g, Box box, const double* u, const double* v, const double* w, double* ub, double* vb, double* wb);
void launch_split_flags(hipStream_t st, const uint8_t* act, const uint8_t* cls, int n, uint8_t* fi, uint8_t* fb);
void launch_pack_box_own(hipStream_t st, Grid g, Box box, Box own, const float* container, const double* u, const double* v, const double* w, double* buf);

}  // namespace fl
