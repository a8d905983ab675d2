// wg_splane.h -- the S-plane of a tensor (its bf16 hi and lo halves as two arrays of 8-channel units, [b][c / 8][p][8]) and the split
// that produces it: shared by the conv / weight-gradient kernels (wg_gemm16*.h) and the small kernels that write S-planes themselves
// (wg_small.h: the seam form of end_affine_kernel; wg_thin.h).
#pragma once
#include "wg_gemm.h"

typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
typedef float f32x2_t __attribute__((ext_vector_type(2)));
// two floats -> packed bf16 pair (round to nearest even): one v_cvt_pk_bf16_f32
__device__ __forceinline__ unsigned cvt_pk_bf16(float a, float b)
{
    f32x2_t v = {a, b};
    return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2_t));
}
// split two floats: packed hi pair and packed lo pair (element 0 in the low half)
__device__ __forceinline__ void split2(float a, float b, unsigned &hi, unsigned &lo)
{
    hi = cvt_pk_bf16(a, b);
    lo = cvt_pk_bf16(a - __uint_as_float(hi << 16), b - __uint_as_float(hi & 0xffff0000u));
}

struct SRef {
    unsigned short *hi;   // lo array at hi + lo_off
    size_t lo_off;        // = B * Cp * P elements
    int Cp, ch0;          // channel rows per item (multiple of 8), first channel (multiple of 8)
};
__device__ __forceinline__ size_t s_index(const SRef &r, const Geo &g, int b, int c, int t)
{
    const int cc = r.ch0 + c;
    return (((size_t)b * (r.Cp >> 3) + (cc >> 3)) * g.P + g.H + t) * 8 + (cc & 7);
}
