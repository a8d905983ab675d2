// wg_probe.h -- box calibration (wg_box_probe): a fixed matrix-pipe + LDS loop without global traffic.
//
// What a compute wave of the conv kernels does per 32-deep chunk, alone: a 64 x 64 output tile per wave, every operand fragment re-read
// from LDS (hi and lo images of A and B: 16 ds_read_b128), three v_mfma_f32_16x16x32_bf16 per fragment pair, eight waves per CU, random
// bf16 data (zeros would run at the full 2.4 GHz and say nothing: MI355X_MICROARCH.md, DVFS give-back items 1 and 7).  The in-kernel
// clock is shader cycles (s_memtime) over the 100 MHz wall clock (s_memrealtime) around the loop.  tools/experiments/shape_probe.hip is
// the stand-alone original; bench.py reports the numbers of this one as `box`.
#pragma once
#include "wg_gemm16q.h"

#define WG_BOX_CHUNKS 2048
#define WG_BOX_IMG (128 * 64)

// 65 536 units of random bf16 in (-2, 2): sign, exponent 125..127, random mantissa (a counter-based hash: no host copy)
__global__ void box_fill_kernel(u32x4 *rnd)
{
    const unsigned i = blockIdx.x * blockDim.x + threadIdx.x;
    u32x4 v;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        unsigned w = 0;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            unsigned x = (i * 8u + e * 2u + h) * 2654435761u + 0x9e3779b9u;
            x ^= x >> 16; x *= 0x85ebca6bu; x ^= x >> 13; x *= 0xc2b2ae35u; x ^= x >> 16;
            const unsigned m = x & 0x7f, ex = 125 + (x >> 7) % 3, sg = (x >> 11) & 1;
            w |= ((sg << 15) | (ex << 7) | m) << (16 * h);
        }
        v[e] = w;
    }
    rnd[i] = v;
}

__global__ __launch_bounds__(512) void box_probe_kernel(const u32x4 *__restrict__ rnd, float *out, unsigned long long *stamps)
{
    __shared__ __attribute__((aligned(16))) char smem[4 * WG_BOX_IMG];      // A hi, A lo, B hi, B lo: 128 rows of 64 bytes each
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int i = tid; i < 4 * WG_BOX_IMG / 16; i += 512) reinterpret_cast<u32x4 *>(smem)[i] = rnd[(blockIdx.x * 37 + i) & 65535];
    __syncthreads();
    const int wr = (wave >> 1) & 1, wc = wave & 1;
    auto rd = [&](int off) { return *reinterpret_cast<const bf16x8 *>(smem + off); };
    unsigned long long c0 = 0, w0 = 0;
    if (tid == 0) { c0 = clock64(); w0 = wall_clock64(); }
    f32x4 acc[4][4];
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 4; ++j)
            for (int q = 0; q < 4; ++q) acc[i][j][q] = 0.f;
    const int r16 = lane & 15, kg = lane >> 4;
    const int ao = wg16q_off(wr * 64 + r16, kg), bo = wg16q_off(wc * 64 + r16, kg);
    for (int c = 0; c < WG_BOX_CHUNKS; ++c) {
        bf16x8 ah[4], al[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) { ah[i] = rd(ao + i * 1024); al[i] = rd(WG_BOX_IMG + ao + i * 1024); }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const bf16x8 bh = rd(2 * WG_BOX_IMG + bo + j * 1024), bl = rd(3 * WG_BOX_IMG + bo + j * 1024);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al[i], bh, acc[i][j], 0, 0, 0);
                acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[i], bl, acc[i][j], 0, 0, 0);
                acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[i], bh, acc[i][j], 0, 0, 0);
            }
        }
    }
    float s = 0.f;
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 4; ++j)
            for (int q = 0; q < 4; ++q) s += acc[i][j][q];
    if (tid == 0) { stamps[2 * blockIdx.x] = clock64() - c0; stamps[2 * blockIdx.x + 1] = wall_clock64() - w0; }
    out[blockIdx.x * 512 + tid] = s;
}
