// WSRGlow conditioning front-end (model/wsrglow.py:8-18,27-50): HBM-bound gather/scatter kernels, no matrix work.
//
//   cond[b][ch][f], F = L/8 frames, 3659 channels:
//     ch in [0, 3200)     mu-law embedding:  table_mu[q(c[b][8f + j])][e],  ch = 400 j + e          (wsrglow.py:39)
//     ch in [3200, 3209)  |STFT16|[k]                                                                  (wsrglow.py:40-47)
//     ch in [3209, 3659)  phase embedding:   table_ang[a(angle(STFT16[k]))][e], ch = 3209 + 50 k + e   (wsrglow.py:48-49)
//   with c clipped to [-1, 1] first (wsrglow.py:38), reflect padding (4, 4), periodic Hann window of 16, hop 8, center=False.
//
// Forward: one workgroup = 64 frames x one slice of channels; the 17 quantiser decisions and 9 magnitudes of each frame are
// computed once per workgroup into LDS, then every store is coalesced along f (consecutive lanes = consecutive frames) and the
// table reads (410 KB + 24 KB) stay in L2.  Algorithmic bytes: 4 B per output element (+ the 4 L bytes of c).
// Backward: dtable[row][e] = sum over (b, f, group) with index == row of dcond[b][group*E + e][f]: each workgroup owns one
// (group, 16-column slice) pair, accumulates into an LDS copy of its table slice with LDS atomics while streaming dcond
// coalesced along f, and adds the slice to the output once.
#pragma once
#include <hip/hip_runtime.h>

#define WSR_MU 256
#define WSR_MU_DIM 400
#define WSR_BINS 9
#define WSR_ANG 120
#define WSR_ANG_DIM 50
#define WSR_CH_MAG (8 * WSR_MU_DIM)
#define WSR_CH_ANG (WSR_CH_MAG + WSR_BINS)
#define WSR_COND (WSR_CH_ANG + WSR_BINS * WSR_ANG_DIM)

__device__ __forceinline__ float wsr_clip(float x) { return fminf(fmaxf(x, -1.f), 1.f); }

// torchaudio.functional.mu_law_encoding with 256 channels, float32 step by step.  The value that is truncated is a function of its own
// (wsr_mu_pre / wsr_angle_pre): wg_wsr_cond_pre hands exactly these to a test, which can then tell a decision that fell on the other
// side of a bin edge within float noise from an indexing bug.
__device__ __forceinline__ float wsr_mu_pre(float x)
{
    const float mu = 255.f;
    const float sgn = (float)((x > 0.f) - (x < 0.f));
    const float x_mu = sgn * log1pf(mu * fabsf(x)) / log1pf(mu);
    return (x_mu + 1.f) / 2.f * mu + 0.5f;
}
__device__ __forceinline__ int wsr_mu_index(float x) { return (int)wsr_mu_pre(x); }
// AngleEmbedding.forward (wsrglow.py:16-17): ((angle / pi + 1) * 0.5 * (embed_num - 1)).long()
__device__ __forceinline__ float wsr_angle_pre(float ang) { return (ang / 3.14159274101257324f + 1.f) * 0.5f * (float)(WSR_ANG - 1); }
__device__ __forceinline__ int wsr_angle_index(float ang) { return (int)wsr_angle_pre(ang); }
__device__ __forceinline__ float wsr_padded(const float *c, int L, int i)
{
    int j = i - 4;
    j = j < 0 ? -j : j;
    j = j >= L ? 2 * (L - 1) - j : j;
    return wsr_clip(c[j]);
}
// cos / sin of 2 pi m / 16, m = 0..15 (exactly rounded constants)
__device__ __constant__ float WSR_COS16[16] = {1.f, 0.92387953251128674f, 0.70710678118654752f, 0.38268343236508977f, 0.f,
                                               -0.38268343236508977f, -0.70710678118654752f, -0.92387953251128674f, -1.f,
                                               -0.92387953251128674f, -0.70710678118654752f, -0.38268343236508977f, 0.f,
                                               0.38268343236508977f, 0.70710678118654752f, 0.92387953251128674f};
__device__ __constant__ float WSR_SIN16[16] = {0.f, 0.38268343236508977f, 0.70710678118654752f, 0.92387953251128674f, 1.f,
                                               0.92387953251128674f, 0.70710678118654752f, 0.38268343236508977f, 0.f,
                                               -0.38268343236508977f, -0.70710678118654752f, -0.92387953251128674f, -1.f,
                                               -0.92387953251128674f, -0.70710678118654752f, -0.38268343236508977f};

// windowed samples of frame f into x[16]
__device__ __forceinline__ void wsr_window(const float *cb, int L, int f, float (&x)[16])
{
#pragma unroll
    for (int n = 0; n < 16; ++n) x[n] = (0.5f - 0.5f * WSR_COS16[n]) * wsr_padded(cb, L, 8 * f + n);
}
// bin k of the 16-point DFT of x: (re, im); DC and Nyquist have an exact +0 imaginary part, as a real FFT returns them
__device__ __forceinline__ void wsr_bin(const float (&x)[16], int k, float &re, float &im)
{
    float r = 0.f, i = 0.f;
#pragma unroll
    for (int n = 0; n < 16; ++n) {
        const int m = (k * n) & 15;
        r = fmaf(x[n], WSR_COS16[m], r);
        i = fmaf(-x[n], WSR_SIN16[m], i);
    }
    re = r;
    im = (k == 0 || k == 8) ? 0.f : i;
}

#define WSR_FT 64          // frames per workgroup
__global__ __launch_bounds__(256) void wsr_cond_kernel(const float *c, int L, const float *mu_w, const float *ang_w, float *cond,
                                                       int ch_per_block)
{
    __shared__ short s_mu[8][WSR_FT];
    __shared__ short s_ang[WSR_BINS][WSR_FT];
    __shared__ float s_mag[WSR_BINS][WSR_FT];
    const int F = L >> 3, b = blockIdx.z, f0 = blockIdx.x * WSR_FT, tid = threadIdx.x;
    const float *cb = c + (size_t)b * L;
    if (tid < WSR_FT) {                                  // one frame per lane of wave 0
        const int f = f0 + tid;
        if (f < F) {
#pragma unroll
            for (int j = 0; j < 8; ++j) s_mu[j][tid] = (short)wsr_mu_index(wsr_clip(cb[8 * f + j]));
            float x[16];
            wsr_window(cb, L, f, x);
#pragma unroll
            for (int k = 0; k < WSR_BINS; ++k) {
                float re, im;
                wsr_bin(x, k, re, im);
                s_mag[k][tid] = hypotf(re, im);
                s_ang[k][tid] = (short)wsr_angle_index(atan2f(im, re));
            }
        }
    }
    __syncthreads();
    const int tx = tid & (WSR_FT - 1), ty = tid >> 6, f = f0 + tx;
    if (f >= F) return;
    const int ch0 = blockIdx.y * ch_per_block, ch1 = min(WSR_COND, ch0 + ch_per_block);
    float *ob = cond + (size_t)b * WSR_COND * F + f;
    for (int ch = ch0 + ty; ch < ch1; ch += 4) {
        float v;
        if (ch < WSR_CH_MAG) {
            const int j = ch / WSR_MU_DIM, e = ch - j * WSR_MU_DIM;
            v = mu_w[(int)s_mu[j][tx] * WSR_MU_DIM + e];
        } else if (ch < WSR_CH_ANG) {
            v = s_mag[ch - WSR_CH_MAG][tx];
        } else {
            const int k = (ch - WSR_CH_ANG) / WSR_ANG_DIM, e = (ch - WSR_CH_ANG) - k * WSR_ANG_DIM;
            v = ang_w[(int)s_ang[k][tx] * WSR_ANG_DIM + e];
        }
        ob[(size_t)ch * F] = v;
    }
}

// Diagnostics (wg_wsr_cond_pre): the two quantisers' values BEFORE truncation, computed by the very functions wsr_cond_kernel truncates:
// mu_pre[b][i] for every low-rate sample, ang_pre[b][k][f] for every STFT bin.  One thread per frame.
__global__ __launch_bounds__(256) void wsr_cond_pre_kernel(const float *c, int L, float *mu_pre, float *ang_pre)
{
    const int F = L >> 3, b = blockIdx.y, f = blockIdx.x * 256 + threadIdx.x;
    if (f >= F) return;
    const float *cb = c + (size_t)b * L;
#pragma unroll
    for (int j = 0; j < 8; ++j) mu_pre[(size_t)b * L + 8 * f + j] = wsr_mu_pre(wsr_clip(cb[8 * f + j]));
    float x[16];
    wsr_window(cb, L, f, x);
#pragma unroll
    for (int k = 0; k < WSR_BINS; ++k) {
        float re, im;
        wsr_bin(x, k, re, im);
        ang_pre[((size_t)b * WSR_BINS + k) * F + f] = wsr_angle_pre(atan2f(im, re));
    }
}

// Embedding-table gradients.  ANGLE = false: mu-law table (groups = 8 sample slots, E = 400); true: phase table (groups = 9 bins,
// E = 50).  grid = (ceil(E / 16), groups); block 256.  dtable must be zeroed by the caller (the host wrapper does).
template <bool ANGLE>
__global__ __launch_bounds__(256) void wsr_table_grad_kernel(const float *c, int B, int L, const float *dcond, float *dtable)
{
    constexpr int ROWS = ANGLE ? WSR_ANG : WSR_MU, E = ANGLE ? WSR_ANG_DIM : WSR_MU_DIM, ET = 16;
    constexpr int CH0 = ANGLE ? WSR_CH_ANG : 0;
    __shared__ float tab[ROWS][ET + 1];
    const int F = L >> 3, tid = threadIdx.x, e0 = blockIdx.x * ET, grp = blockIdx.y;
    const int ne = min(ET, E - e0);
    for (int i = tid; i < ROWS * (ET + 1); i += 256) (&tab[0][0])[i] = 0.f;
    __syncthreads();
    for (int b = 0; b < B; ++b) {
        const float *cb = c + (size_t)b * L;
        const float *gb = dcond + ((size_t)b * WSR_COND + CH0 + (size_t)grp * E + e0) * F;
        for (int f = tid; f < F; f += 256) {
            int row;
            if (ANGLE) {
                float x[16], re, im;
                wsr_window(cb, L, f, x);
                wsr_bin(x, grp, re, im);
                row = wsr_angle_index(atan2f(im, re));
            } else {
                row = wsr_mu_index(wsr_clip(cb[8 * f + grp]));
            }
            for (int e = 0; e < ne; ++e) atomicAdd(&tab[row][e], gb[(size_t)e * F + f]);
        }
    }
    __syncthreads();
    for (int i = tid; i < ROWS * ET; i += 256) {
        const int row = i / ET, e = i - row * ET;
        if (e < ne) {
            const float v = tab[row][e];
            if (v != 0.f) atomicAdd(&dtable[row * E + e0 + e], v);
        }
    }
}
