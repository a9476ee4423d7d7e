// Small device helpers shared by the dense stencil sweeps (kernels_pcg.hip, kernels_stencil.hip).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace fl {

// V elements of T = 16 bytes; the flag bytes of the same cells as one word
template <typename T, int V>
struct VecT {
    typedef T type __attribute__((ext_vector_type(V)));
};
template <int V>
struct FlagT;
template <>
struct FlagT<4> { typedef uint32_t type; };
template <>
struct FlagT<2> { typedef uint16_t type; };

// v where the mask word is all ones, +0 where it is zero
template <typename T>
__device__ __forceinline__ T and_mask(T v, int m)
{
    if constexpr (sizeof(T) == 4) return __builtin_bit_cast(T, __builtin_bit_cast(int, v) & m);
    else return __builtin_bit_cast(T, __builtin_bit_cast(long long, v) & (long long)m);   // m sign-extends
}

// static indices only: a runtime index into a by-value kernel argument would go through scratch
template <typename T, typename C>
__device__ __forceinline__ void load_coef(T* sdiag, T* sinv, const C& cf)
{
    if (threadIdx.x == 0) {
#pragma unroll
        for (int i = 0; i < 7; ++i) {
            sdiag[i] = cf.diag[i];
            sinv[i] = cf.inv[i];
        }
    }
}

}  // namespace fl
