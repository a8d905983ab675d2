// wg_small.h -- the HBM-bound / tiny kernels around the two MFMA kernels: squeeze, 1x1 channel mixing,
// WN.end + affine coupling (+ its backward seed), upsampler, loss, weight packing, gradient finalisation.
#pragma once
#include "wg_gemm.h"
#include "wg_splane.h"       // SRef / s_index / split2: the seam form of end_affine_kernel writes the next WN's h_0 as an S-plane

#define WG_MAXC 32   // largest invertible-1x1 / end-conv row count handled by the small kernels

// ------------------------------------------------------------------------------------------------
// squeeze / unsqueeze   (waveglow.py:153,179): X[b][g][t] = audio[b][t*G + g]
// ------------------------------------------------------------------------------------------------
__global__ void squeeze_kernel(const float *__restrict__ audio, PRef X, Geo g, int G, int N)
{
    const int t = blockIdx.x * blockDim.x + threadIdx.x, b = blockIdx.y;
    if (t >= g.T) return;
    const float *src = audio + (size_t)b * N + (size_t)t * G;
    for (int c = 0; c < G; ++c) *paddr(X, g, b, c, t) = src[c];
}
__global__ void unsqueeze_kernel(PRef X, float *__restrict__ audio, Geo g, int G, int N)
{
    const int t = blockIdx.x * blockDim.x + threadIdx.x, b = blockIdx.y;
    if (t >= g.T) return;
    float *dst = audio + (size_t)b * N + (size_t)t * G;
    for (int c = 0; c < G; ++c) dst[c] = *paddr(X, g, b, c, t);
}

// plain [B][C][T] <-> plane (block-level API and mel planes)
__global__ void import_kernel(const float *__restrict__ src, PRef X, Geo g, int C)
{
    const int t = blockIdx.x * blockDim.x + threadIdx.x, c = blockIdx.y, b = blockIdx.z;
    if (t < g.T) *paddr(X, g, b, c, t) = src[((size_t)b * C + c) * g.T + t];
}
__global__ void export_kernel(PRef X, float *__restrict__ dst, Geo g, int C, float scale)
{
    const int t = blockIdx.x * blockDim.x + threadIdx.x, c = blockIdx.y, b = blockIdx.z;
    if (t < g.T) dst[((size_t)b * C + c) * g.T + t] = scale * *paddr(X, g, b, c, t);
}
__global__ void fill_rows_kernel(PRef X, Geo g, const float *__restrict__ perb, float cst)
{   // X[b][c][t] = (perb ? perb[b] : 0) + cst   for t < T
    const int t = blockIdx.x * blockDim.x + threadIdx.x, c = blockIdx.y, b = blockIdx.z;
    if (t < g.T) *paddr(X, g, b, c, t) = (perb ? perb[b] : 0.f) + cst;
}

// ------------------------------------------------------------------------------------------------
// invertible 1x1 conv as an in-place channel mix  (efficient_modules.py:40,53,237,239)
//   X[ch0+o][t] = sum_i Mx[o][i] * X[ch0+i][t]   (Mx = W, W^-1, W^T ...; `transpose` swaps indices)
// ------------------------------------------------------------------------------------------------
template <int C>
__global__ void mix_kernel(PRef X, const float *__restrict__ Mx, int transpose, Geo g)
{
    __shared__ float w[C * C];
    for (int e = threadIdx.x; e < C * C; e += blockDim.x) {
        const int o = e / C, i = e % C;
        w[e] = transpose ? Mx[i * C + o] : Mx[o * C + i];
    }
    __syncthreads();
    const int t = blockIdx.x * blockDim.x + threadIdx.x, b = blockIdx.y;
    if (t >= g.T) return;
    float x[C];
#pragma unroll
    for (int i = 0; i < C; ++i) x[i] = *paddr(X, g, b, i, t);
#pragma unroll
    for (int o = 0; o < C; ++o) {
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < C; ++i) s += w[o * C + i] * x[i];
        *paddr(X, g, b, o, t) = s;
    }
}

// Conv1x1Func.backward (efficient_modules.py:235-242) for c <= 8 in ONE pass over z and dz (a thread owns a time step):
//   x = W^-1 z (in place, :235-237), dx = W^T dz (in place, :239), and this block's share of dW = sum_t dz x^T (:240) into
//   part[block][c*c] -- five launches (two channel mixes, a 128x128-tile weight-gradient kernel for a c x c product, its slab
//   reduction) were 50 us per flow of dependent latency.  The split-K finalisation sums the blocks' shares in a fixed order.
#define WG_ICB_T 4
template <int C>
__global__ __launch_bounds__(256) void invconv_bwd_kernel(PRef X, PRef dX, const float *__restrict__ Wm, const float *__restrict__ Winv, Geo g,
                                                          float *__restrict__ part)
{
    __shared__ float w[C * C], wi[C * C];
    __shared__ float red[4][C * C];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int e = tid; e < C * C; e += 256) { w[e] = Wm[e]; wi[e] = Winv[e]; }
    __syncthreads();
    const int b = blockIdx.y;
    float acc[C * C];
#pragma unroll
    for (int e = 0; e < C * C; ++e) acc[e] = 0.f;
#pragma unroll
    for (int q = 0; q < WG_ICB_T; ++q) {                     // WG_ICB_T time steps per thread, 256 apart
        const int t = (blockIdx.x * WG_ICB_T + q) * 256 + tid;
        if (t >= g.T) continue;
        float z[C], dz[C], x[C];
#pragma unroll
        for (int i = 0; i < C; ++i) { z[i] = *paddr(X, g, b, i, t); dz[i] = *paddr(dX, g, b, i, t); }
#pragma unroll
        for (int o = 0; o < C; ++o) {
            float s = 0.f, d = 0.f;
#pragma unroll
            for (int i = 0; i < C; ++i) { s += wi[o * C + i] * z[i]; d += w[i * C + o] * dz[i]; }
            x[o] = s;
            *paddr(X, g, b, o, t) = s;
            *paddr(dX, g, b, o, t) = d;
        }
#pragma unroll
        for (int i = 0; i < C; ++i)
#pragma unroll
            for (int j = 0; j < C; ++j) acc[i * C + j] = fmaf(dz[i], x[j], acc[i * C + j]);
    }
#pragma unroll
    for (int e = 0; e < C * C; ++e) {
        float p = acc[e];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) p += __shfl_xor(p, o, 64);
        if (lane == 0) red[wave][e] = p;
    }
    __syncthreads();
    if (tid < C * C) part[((size_t)blockIdx.y * gridDim.x + blockIdx.x) * (C * C) + tid] = (red[0][tid] + red[1][tid]) + (red[2][tid] + red[3][tid]);
}

// LU (partial pivoting) of each flow's c x c weight: logdet (NaN if det<0, torch.logdet semantics) and inverse.
// One thread per matrix; c <= WG_MAXC.  out layout per matrix: [W (c*c) | Winv (c*c) | logdet | pad..] stride `ostride`.
struct LuJob {
    const float *W;
    int c;
};
#define WG_MAX_FLOWS 64
struct LuArgs {
    int n;
    LuJob job[WG_MAX_FLOWS];
    float *out;
    int ostride;
};
__global__ void lu_kernel(const LuArgs a)
{
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= a.n) return;
    const int c = a.job[k].c;
    const float *W = a.job[k].W;
    float *o = a.out + (size_t)k * a.ostride;
    float *A = o + 2 * WG_MAXC * WG_MAXC + 8;   // scratch copy (LU in place) inside this matrix's slot
    float *Wi = o + WG_MAXC * WG_MAXC;
    int perm[WG_MAXC];
    for (int i = 0; i < c * c; ++i) { o[i] = W[i]; A[i] = W[i]; }
    for (int i = 0; i < c; ++i) perm[i] = i;
    int sg = 1;
    float la = 0.f;
    for (int q = 0; q < c; ++q) {
        int p = q;
        float best = fabsf(A[q * c + q]);
        for (int r = q + 1; r < c; ++r) {
            const float v = fabsf(A[r * c + q]);
            if (v > best) { best = v; p = r; }
        }
        if (p != q) {
            for (int j = 0; j < c; ++j) { const float tmp = A[q * c + j]; A[q * c + j] = A[p * c + j]; A[p * c + j] = tmp; }
            const int ti = perm[q]; perm[q] = perm[p]; perm[p] = ti;
            sg = -sg;
        }
        const float piv = A[q * c + q];
        if (piv < 0.f) sg = -sg;
        la += logf(fabsf(piv));
        for (int r = q + 1; r < c; ++r) {
            const float f = A[r * c + q] / piv;
            A[r * c + q] = f;
            for (int j = q + 1; j < c; ++j) A[r * c + j] -= f * A[q * c + j];
        }
    }
    float yv[WG_MAXC];
    for (int col = 0; col < c; ++col) {
        for (int r = 0; r < c; ++r) {
            float s = (perm[r] == col) ? 1.f : 0.f;
            for (int j = 0; j < r; ++j) s -= A[r * c + j] * yv[j];
            yv[r] = s;
        }
        for (int r = c - 1; r >= 0; --r) {
            float s = yv[r];
            for (int j = r + 1; j < c; ++j) s -= A[r * c + j] * Wi[j * c + col];
            Wi[r * c + col] = s / A[r * c + r];
        }
    }
    o[2 * WG_MAXC * WG_MAXC] = sg > 0 ? la : __builtin_nanf("");
}

// ------------------------------------------------------------------------------------------------
// WN.end (1x1, skip_ch -> 2*in_ch, waveglow.py:105) fused with the affine coupling
// (efficient_modules.py:77-96) and, in backward mode, the seed of the WN backward (efficient_modules.py:132-148).
// One wave computes a [32 rows] x [32 time steps] tile of out = W_end . S straight from global memory
// (K = skip_ch, rows zero padded to 32); the tile goes through LDS so that log_s[j] and t[j] meet in one thread.
// ------------------------------------------------------------------------------------------------
enum { AFF_FWD = 0, AFF_REV = 1, AFF_BWD = 2, AFF_BWD_REV = 3, AFF_RAW = 4 };

struct AffineArgs {
    const float *endT;      // [Cs][32]  endT[k][m] = W_end[m][k] (m < 2*ic, else 0)
    const float *bias;      // nullable: end.bias [2*ic] (WN(bias=True))
    PRef S;                 // skip plane
    int Cs, ic;
    PRef X;                 // flow state, ch0 = first channel of this flow (xa rows [0,ic), xb rows [ic,2ic))
    PRef dX;                // gradient plane (BWD modes)
    PRef Gp;                // G plane (BWD modes): rows [0,ic) = d/dlog_s, [ic,2ic) = d/dt
    float *log_s_out;       // optional plain [B][ic][T] (block API): +log_s (FWD) or -log_s (REV)
    float *t_out;           // AFF_RAW only: plain [B][ic][T] t  (WN.forward output, waveglow.py:105)
    const float *dls_plain; // optional plain [B][ic][T] gradient of the returned log_s (block API, BWD modes)
    const float *dld;       // optional [B]: gradient of logdet[b] (model level, BWD)
    float *partial;         // [B][gridDim.x] per-block sums of +-log_s (FWD/REV), nullable
    Geo g;
    int mode;
    // SEAM instantiation only (synthesis: the inverse direction between two WNs, one launch instead of three): after the inverse affine the
    // inverse 1x1 conv on the flow's c = 2 ic channels (mixM: W^-1, c x c, row major), then WN.start of the NEXT flow visited on its
    // xa = plane channels [n_rel, n_rel + n_ic) relative to X's first channel (negative where an early output re-enters, waveglow.py:204-205)
    const float *mixM;
    const float *nW;        // next start conv, fp32 effective weights [n_C][n_ldw]
    int n_ldw, n_C, n_ic, n_rel;
    PRef nH;                // next WN's h_0, fp32 plane (p == nullptr: none)
    SRef nHS;               // ... and its S-plane
    // FROMG instantiation: out = sum_l Weff_l gate_l straight from the layers' gate S-planes -- Weff_l = W_end Wskip_l (2 ic x Cd per layer:
    // weff_kernel below), endT = Weff^T stacked along K, [nl * Cd][32] -- instead of W_end (sum_l Wskip_l gate_l): the 256-row skip sum
    // exists only to be contracted to these 2 ic rows again (model/waveglow.py:104-105), so it is neither computed nor read
    const unsigned short *gS[16];   // hi array of layer l's gate S-plane (Cd channels per item; lo at + g_lo_off)
    size_t g_lo_off;
    int Cd, nl;
    // SRC == 2: out = the sum of `nsrc` partial rows of 8 floats per column, [src][b][t][8] fp32 (src = layer x slot: what the gate convs'
    // epilogues left, wg_gemm16g.h wgg_gate_nb): 2 ic <= 8 only
    const float *part;
    int nsrc;
};

#define WG_AFF_T 64          // time steps per workgroup
#define WG_SEAM_MAXW 2048    // floats of the next start conv's weights a SEAM launch stages (n_C x n_ic)
#define WG_AFF_LD 64         // skip-channel loads a lane keeps in flight
// NR = accumulator rows kept per lane: 8 where 2 * ic <= 8 (WaveGlow: n_group 8), 32 otherwise.  (With 32 accumulators next to the 64
// loads in flight the kernel needed 250 VGPRs and 41 KB of LDS: two workgroups per CU, 38 us per launch at the training shape.)
template <int NR, bool SEAM = false, int SRC = 0>       // SRC: 0 = W_end . S, 1 = sum_l Weff_l gate_l from the gate planes, 2 = the gate convs' partial rows
__global__ __launch_bounds__(256) void end_affine_kernel(const AffineArgs a)
{
    // out[m][t] = sum_k W_end[m][k] S[k][t] for the 2*ic <= 32 rows of the end conv: far too few rows for a matrix tile to pay, and
    // the kernel is bound by reading S once.  A workgroup takes 64 time steps; its 4 waves split the skip channels, every lane
    // streams its quarter of S[:, t] (coalesced along t, 64 loads in flight) into 2*ic accumulators, LDS adds the four quarters.
    __shared__ float tile[NR][WG_AFF_T + 1];
    // phase 1: each wave's block of W_end^T rows ([WG_AFF_LD][NR] floats per wave); phase 2: the partial sums of waves 1-3
    constexpr int STAGE = (4 * WG_AFF_LD * NR > 3 * NR * (WG_AFF_T + 1)) ? 4 * WG_AFF_LD * NR : 3 * NR * (WG_AFF_T + 1);
    __shared__ __attribute__((aligned(16))) float stage[STAGE];
    float (*part)[NR][WG_AFF_T + 1] = reinterpret_cast<float (*)[NR][WG_AFF_T + 1]>(stage);
    __shared__ float red[4];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int b = blockIdx.y, t = blockIdx.x * WG_AFF_T + lane;
    const Geo g = a.g;
    const int rows = 2 * a.ic;
    float acc[NR];
#pragma unroll
    for (int m = 0; m < NR; ++m) acc[m] = 0.f;
    // SEAM: what the tail needs from memory is requested here, in front of the end conv's reduction, and used behind it -- the next
    // start conv's weights, W^-1, the flow's channels of this lane's time step (wave 0)
    float pre_sw[SEAM ? WG_SEAM_MAXW / 256 : 1], pre_mw = 0.f, pre_x[SEAM ? 8 : 1];
    if constexpr (SEAM) {
#pragma unroll
        for (int q = 0; q < WG_SEAM_MAXW / 256; ++q) {
            const int e = tid + 256 * q;
            pre_sw[q] = e < a.n_C * a.n_ic ? a.nW[(size_t)(e / a.n_ic) * a.n_ldw + (e % a.n_ic)] : 0.f;
        }
        if (tid < 4 * a.ic * a.ic) pre_mw = a.mixM[tid];
#pragma unroll
        for (int i = 0; i < 8; ++i) pre_x[i] = (wave == 0 && i < 2 * a.ic && t < g.T) ? *paddr(a.X, g, b, i, t) : 0.f;
    }
    if constexpr (SRC == 2) {
        // wave w adds its quarter of the sources in source order, sixteen 16-byte loads in flight; lanes = time steps, 32 bytes apart
        static_assert(SRC != 2 || NR == 8, "partial rows hold 8 floats");
        const int nq = (a.nsrc + 3) / 4, s0 = wave * nq, s1 = min(a.nsrc, s0 + nq);
        const size_t stride = (size_t)g.B * g.Tt * 8;
        const float *p0 = a.part + ((size_t)b * g.Tt + min(t, g.Tt - 1)) * 8;
        for (int sb = s0; sb < s1; sb += 8) {
            f32x4 v0[8], v1[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const float *q = p0 + (size_t)min(sb + u, s1 - 1) * stride;
                v0[u] = *reinterpret_cast<const f32x4 *>(q);
                v1[u] = *reinterpret_cast<const f32x4 *>(q + 4);
            }
#pragma unroll
            for (int u = 0; u < 8; ++u)
                if (sb + u < s1) {
                    acc[0] += v0[u][0]; acc[1] += v0[u][1]; acc[2] += v0[u][2]; acc[3] += v0[u][3];
                    acc[4] += v1[u][0]; acc[5] += v1[u][1]; acc[6] += v1[u][2]; acc[7] += v1[u][3];
                }
        }
    } else if constexpr (SRC == 1) {
        // K = nl * Cd gate channels, 8 per 16-byte unit of an S-plane ([c / 8][p][8] bf16, hi and lo arrays): a lane reads the units of
        // its time step -- consecutive lanes, consecutive units: 1 KB per wave instruction -- 8 units x (hi, lo) in flight, and rebuilds
        // gate = hi + lo (the operand the matrix kernels multiply: nothing is lost against the skip sum's own input)
        const int K = a.Cd * a.nl;
        const int kq = ((K / 8 + 3) / 4) * 8, k0 = wave * kq, k1 = min(K, k0 + kq);
        const size_t ti = (size_t)(g.H + min(t, g.Tt - 1)) * 8;
        const size_t gstride = (size_t)g.P * 8, item = (size_t)b * (a.Cd >> 3) * gstride;
        float *wl = stage + wave * (WG_AFF_LD * NR);
        for (int kk = 0; kk < kq; kk += WG_AFF_LD) {                     // (same trip count in every wave: barriers inside)
            const int k = k0 + kk;
            u32x4 vh[8], vl[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int ku = k + 8 * u;
                vh[u] = u32x4{0u, 0u, 0u, 0u}; vl[u] = vh[u];
                if (ku < k1) {
                    const int l = ku / a.Cd, ch = ku - l * a.Cd;         // (wave uniform: a unit never straddles two layers, Cd % 8 == 0)
                    const unsigned short *q = a.gS[l] + item + (size_t)(ch >> 3) * gstride + ti;
                    vh[u] = *reinterpret_cast<const u32x4 *>(q);
                    vl[u] = *reinterpret_cast<const u32x4 *>(q + a.g_lo_off);
                }
            }
            if (kk) __syncthreads();                                     // the previous block has been consumed
            for (int i = lane; i < WG_AFF_LD * (NR / 4); i += 64) {      // columns [0, NR) of rows k .. k+63 of endT ([K][32])
                const int u = i / (NR / 4), q = i - u * (NR / 4);
                float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
                if (k + u < k1) v = reinterpret_cast<const float4 *>(a.endT + (size_t)(k + u) * 32)[q];
                reinterpret_cast<float4 *>(wl)[i] = v;
            }
            __syncthreads();
#pragma unroll
            for (int u = 0; u < 8; ++u)
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const unsigned wh = vh[u][e >> 1], wlo = vl[u][e >> 1];
                    const float sv = (e & 1) ? __uint_as_float(wh & 0xffff0000u) + __uint_as_float(wlo & 0xffff0000u)
                                             : __uint_as_float(wh << 16) + __uint_as_float(wlo << 16);
                    const float4 *w = reinterpret_cast<const float4 *>(wl + (8 * u + e) * NR);      // same address in every lane: LDS broadcast
#pragma unroll
                    for (int q = 0; q < NR / 4; ++q) {
                        const float4 wq = w[q];
                        acc[4 * q] = fmaf(wq.x, sv, acc[4 * q]);         acc[4 * q + 1] = fmaf(wq.y, sv, acc[4 * q + 1]);
                        acc[4 * q + 2] = fmaf(wq.z, sv, acc[4 * q + 2]); acc[4 * q + 3] = fmaf(wq.w, sv, acc[4 * q + 3]);
                    }
                }
        }
    } else
    {
        const int kq = (a.Cs + 3) / 4, k0 = wave * kq, k1 = min(a.Cs, k0 + kq);
        const float *sp = paddr(a.S, g, b, 0, min(t, g.Tt - 1));        // columns in [T, Tt) read the zero padding; beyond Tt is clamped
        float *wl = stage + wave * (WG_AFF_LD * NR);
        // 64 loads in flight per lane (a whole quarter of the usual 256 skip channels at once): with one utterance the grid is 32
        // workgroups and the kernel's time is its chain of round trips.  The weights of those 64 channels come through LDS (one
        // coalesced copy per wave, then broadcast reads): as wave-uniform global loads they were a chain of 64 scalar-cache misses,
        // 25 us per launch = 10 % of a synthesis call
        for (int kk = 0; kk < kq; kk += WG_AFF_LD) {                     // (same trip count in every wave: barriers inside)
            const int k = k0 + kk;
            float sv[WG_AFF_LD];
#pragma unroll
            for (int u = 0; u < WG_AFF_LD; ++u) sv[u] = (k + u < k1) ? sp[(size_t)(k + u) * g.P] : 0.f;
            if (kk) __syncthreads();                                     // the previous block has been consumed
            for (int i = lane; i < WG_AFF_LD * (NR / 4); i += 64) {      // columns [0, NR) of rows k .. k+63 of endT ([Cs][32])
                const int u = i / (NR / 4), q = i - u * (NR / 4);
                float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
                if (k + u < k1) v = reinterpret_cast<const float4 *>(a.endT + (size_t)(k + u) * 32)[q];
                reinterpret_cast<float4 *>(wl)[i] = v;
            }
            __syncthreads();
#pragma unroll
            for (int u = 0; u < WG_AFF_LD; ++u) {
                const float4 *w = reinterpret_cast<const float4 *>(wl + u * NR);         // same address in every lane: LDS broadcast
#pragma unroll
                for (int q = 0; q < NR / 4; ++q) {
                    const float4 wq = w[q];
                    acc[4 * q] = fmaf(wq.x, sv[u], acc[4 * q]);         acc[4 * q + 1] = fmaf(wq.y, sv[u], acc[4 * q + 1]);
                    acc[4 * q + 2] = fmaf(wq.z, sv[u], acc[4 * q + 2]); acc[4 * q + 3] = fmaf(wq.w, sv[u], acc[4 * q + 3]);
                }
            }
        }
    }
    __syncthreads();                                                     // stage: weights -> partial sums
    if (wave > 0) {
#pragma unroll
        for (int m = 0; m < NR; ++m)
            if (m < rows) part[wave - 1][m][lane] = acc[m];
    }
    __syncthreads();
    if (wave == 0) {
#pragma unroll
        for (int m = 0; m < NR; ++m)
            if (m < rows) tile[m][lane] = (acc[m] + part[0][m][lane]) + (part[1][m][lane] + part[2][m][lane]) + (a.bias ? a.bias[m] : 0.f);
    }
    __syncthreads();

    const int ic = a.ic;
    float lsum = 0.f;
    if constexpr (SEAM) {
        // AFF_REV for the flow, W^-1 on its channels, the next WN's start conv -- three dependent ~5 us launches per flow of a synthesis
        // call otherwise (end_affine_kernel, mix_kernel, start_fwd_kernel: their arithmetic, expression for expression)
        __shared__ float xs[8][WG_AFF_T];                                // the flow's channels after the mix
        __shared__ float mw[64];
        __shared__ float sw[WG_SEAM_MAXW];
        const int c = 2 * ic;
        if (tid < c * c) mw[tid] = pre_mw;
#pragma unroll
        for (int q = 0; q < WG_SEAM_MAXW / 256; ++q) sw[tid + 256 * q] = pre_sw[q];
        __syncthreads();
        const bool live = t < g.T;
        if (wave == 0) {
            float x[8], y[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) x[i] = pre_x[i];
#pragma unroll
            for (int j = 0; j < 4; ++j)
                if (j < ic) {
                    const float ls = tile[j][lane], tt = tile[ic + j][lane];
                    // (x[ic + j] with a run-time ic: select, the array stays in registers)
                    float xb = 0.f;
#pragma unroll
                    for (int i = 0; i < 8; ++i) xb = (i == ic + j) ? x[i] : xb;
                    const float xr = (xb - tt) / expf(ls);                 // efficient_modules.py:94 / :167
#pragma unroll
                    for (int i = 0; i < 8; ++i) x[i] = (i == ic + j) ? xr : x[i];
                    if (live) lsum -= ls;
                }
#pragma unroll
            for (int o = 0; o < 8; ++o) {
                float sacc = 0.f;
#pragma unroll
                for (int i = 0; i < 8; ++i)
                    if (i < c && o < c) sacc += mw[o * c + i] * x[i];      // mix_kernel's sum
                y[o] = sacc;
            }
#pragma unroll
            for (int o = 0; o < 8; ++o)
                if (o < c) {
                    if (live) *paddr(a.X, g, b, o, t) = y[o];
                    xs[o][lane] = y[o];
                }
        }
        __syncthreads();
        if (live) {
            float xa[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int rel = a.n_rel + j;
                xa[j] = 0.f;
                if (j < a.n_ic) xa[j] = (rel >= 0 && rel < c) ? xs[rel][lane] : *paddr(a.X, g, b, rel, t);
            }
            for (int cg = wave; cg < a.n_C / 8; cg += 4) {                 // start_fwd_kernel's thread: one time step, 8 output channels
                float o[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int j = 0; j < 8; ++j)
                    if (j < a.n_ic) {
#pragma unroll
                        for (int e = 0; e < 8; ++e) o[e] = fmaf(sw[(cg * 8 + e) * a.n_ic + j], xa[j], o[e]);
                    }
                if (a.nH.p) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) *paddr(a.nH, g, b, cg * 8 + e, t) = o[e];
                }
                u32x4 h, l;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    unsigned hh, ll;
                    split2(o[2 * e], o[2 * e + 1], hh, ll);
                    h[e] = hh; l[e] = ll;
                }
                const size_t si = s_index(a.nHS, g, b, cg * 8, t);
                *reinterpret_cast<u32x4 *>(a.nHS.hi + si) = h;
                *reinterpret_cast<u32x4 *>(a.nHS.hi + a.nHS.lo_off + si) = l;
            }
        }
    } else
    for (int e = tid; e < ic * WG_AFF_T; e += 256) {
        const int j = e / WG_AFF_T, tl = e - j * WG_AFF_T;
        const int t = blockIdx.x * WG_AFF_T + tl;
        if (t >= g.T) continue;
        const float ls = tile[j][tl], tt = tile[ic + j][tl];
        if (a.mode == AFF_RAW) {
            const size_t q = ((size_t)b * ic + j) * g.T + t;
            a.log_s_out[q] = ls;
            a.t_out[q] = tt;
            continue;
        }
        float *xbp = paddr(a.X, g, b, ic + j, t);
        const float xb = *xbp;
        const size_t pidx = ((size_t)b * ic + j) * g.T + t;
        if (a.mode == AFF_FWD) {
            *xbp = xb * expf(ls) + tt;                                   // efficient_modules.py:81 / :110
            lsum += ls;
            if (a.log_s_out) a.log_s_out[pidx] = ls;
        } else if (a.mode == AFF_REV) {
            *xbp = (xb - tt) / expf(ls);                                 // :94 / :167
            lsum -= ls;
            if (a.log_s_out) a.log_s_out[pidx] = -ls;
        } else if (a.mode == AFF_BWD) {
            const float sc = expf(ls);
            const float xr = (xb - tt) / sc;                             // :133-134 rebuild xb from zb
            *xbp = xr;
            float *dp = paddr(a.dX, g, b, ic + j, t);
            const float dzb = *dp;
            float gl = dzb * xr * sc;                                    // :143-144 grad_outputs
            if (a.dld) gl += a.dld[b];
            if (a.dls_plain) gl += a.dls_plain[pidx];
            *paddr(a.Gp, g, b, j, t) = gl;
            *paddr(a.Gp, g, b, ic + j, t) = dzb;
            *dp = dzb * sc;                                              // :147
        } else {  // AFF_BWD_REV : InvAffineCouplingFunc.backward, X holds the block OUTPUT xo
            const float sc = expf(ls);
            const float zb = xb * sc + tt;                               // :194 rebuild zb from xb
            *xbp = zb;
            float *dp = paddr(a.dX, g, b, ic + j, t);
            const float dxb = *dp;
            float go = dxb * zb / sc;                                    // :202-203
            if (a.dls_plain) go += a.dls_plain[pidx];
            *paddr(a.Gp, g, b, j, t) = -go + dxb * tt / sc;
            *paddr(a.Gp, g, b, ic + j, t) = -dxb / sc;
            *dp = dxb / sc;                                              // :206
        }
    }
    if (a.partial) {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) lsum += __shfl_down(lsum, o, 64);
        if (lane == 0) red[wave] = lsum;
        __syncthreads();
        if (tid == 0) a.partial[(size_t)b * gridDim.x + blockIdx.x] = red[0] + red[1] + red[2] + red[3];
    }
}

// The affine coupling on plain arrays, for an AffineCouplingBlock whose transform is NOT this package's WN (efficient_modules.py:58-62
// takes any `transform_type`): the transform runs as the caller's torch module, this is the block's own arithmetic.
//   apply:     out = reverse ? (in - t) / exp(log_s) : in * exp(log_s) + t                                     (:81 / :94)
//   backward:  the block input rebuilt from its output, the seeds of the transform's backward (gradients w.r.t. ITS outputs log_s, t)
//              and the gradient of the passed-through half: AffineCouplingFunc.backward :132-148 / InvAffineCouplingFunc.backward :194-206
struct AffinePlainArgs {
    const float *in, *log_s, *t;       // [n]
    const float *dout, *dls;           // backward: gradient of the block output's second half; of the returned log_s (nullable)
    float *out;                        // apply: the output half; backward: the rebuilt input half
    float *g_ls, *g_t, *din;           // backward
    size_t n;
    int reverse, backward;
};
__global__ void affine_plain_kernel(const AffinePlainArgs a)
{
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= a.n) return;
    const float ls = a.log_s[i], tt = a.t[i], sc = expf(ls), v = a.in[i];
    if (!a.backward) {
        a.out[i] = a.reverse ? (v - tt) / sc : v * sc + tt;
        return;
    }
    const float d = a.dout[i], gl = a.dls ? a.dls[i] : 0.f;
    if (!a.reverse) {                                          // v = zb
        const float xr = (v - tt) / sc;
        a.out[i] = xr;
        a.g_ls[i] = d * xr * sc + gl;
        a.g_t[i] = d;
        a.din[i] = d * sc;
    } else {                                                   // v = xb (the block's output), the returned log_s was -ls
        const float zb = v * sc + tt;
        a.out[i] = zb;
        a.g_ls[i] = -(d * zb / sc + gl) + d * tt / sc;
        a.g_t[i] = -d / sc;
        a.din[i] = d / sc;
    }
}

// logdet[b] = sum_k coef * T * logdetW_k + sum_k sum_tiles partial[k][b][tile]      (waveglow.py:175 / :202)
__global__ void logdet_finalize_kernel(const float *__restrict__ lu, int ostride, int n_flows, float coef_T,
                                       const float *__restrict__ partial, int ntile, int B, float *__restrict__ logdet)
{
    // one wave per batch item: the lanes split the n_flows * ntile partial sums (a serial walk by one thread was 26 us per call)
    const int b = blockIdx.x, lane = threadIdx.x;
    float s = 0.f;
    for (int e = lane; e < n_flows * ntile; e += 64) {
        const int k = e / ntile, i = e - k * ntile;
        s += partial[((size_t)k * B + b) * ntile + i];
    }
    for (int k = lane; k < n_flows; k += 64) s += coef_T * lu[(size_t)k * ostride + 2 * WG_MAXC * WG_MAXC];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_down(s, o, 64);
    if (lane == 0) logdet[b] = s;
}
__global__ void scalar_logdet_kernel(const float *lu, float coef_T, float *out) { out[0] = coef_T * lu[2 * WG_MAXC * WG_MAXC]; }

// ------------------------------------------------------------------------------------------------
// mel upsampler: depthwise ConvTranspose1d (waveglow.py:126-130), cropped to T (waveglow.py:157)
//   y[c][j] = bias[c] + sum_i h[c][i] * w[c][j + pad - stride*i]
// ------------------------------------------------------------------------------------------------
__global__ void upsample_fwd_kernel(const float *__restrict__ h, const float *__restrict__ w, const float *__restrict__ bias,
                                    PRef Y, float *__restrict__ yplain, Geo g, int C, int F, int K, int S, int Pd)
{
    const int j = blockIdx.x * blockDim.x + threadIdx.x, c = blockIdx.y, b = blockIdx.z;
    if (j >= g.T) return;
    float s = bias ? bias[c] : 0.f;
    int ihi = (j + Pd) / S;
    if (ihi > F - 1) ihi = F - 1;
    int ilo = (j + Pd - K + 1 + S - 1);
    ilo = ilo <= 0 ? 0 : ilo / S;
    const float *hc = h + ((size_t)b * C + c) * F;
    for (int i = ilo; i <= ihi; ++i) s += hc[i] * w[c * K + (j + Pd - S * i)];
    if (Y.p) *paddr(Y, g, b, c, j) = s;
    if (yplain) yplain[((size_t)b * C + c) * g.T + j] = s;
}

// one block per mel channel: dbias, dw (then weight-norm backward), optional dh.  Every sum keeps four independent accumulators per
// thread (the block's time is its chain of load round trips: 80 blocks for WaveGlow, 311 us as single chains); all sums are reduced in a
// fixed order (bitwise reproducible).  Dynamic LDS: [K] dw + [256] scratch + [256] tap partials.
__global__ __launch_bounds__(256) void upsample_bwd_kernel(const float *__restrict__ h, const float *__restrict__ w, PRef dY, Geo g,
                                                           int C, int F, int K, int S, int Pd,
                                                           const float *__restrict__ gparam, const float *__restrict__ v,
                                                           float *__restrict__ dbias, float *__restrict__ dg,
                                                           float *__restrict__ dv, float *__restrict__ dh)
{
    extern __shared__ float sm[];
    float *dw = sm, *red = sm + K, *tp = sm + K + 256;
    const int c = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    auto block_sum = [&](float x) {                            // every thread gets the sum over the block
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) x += __shfl_xor(x, o, 64);
        __syncthreads();
        if (lane == 0) red[wave] = x;
        __syncthreads();
        return (red[0] + red[1]) + (red[2] + red[3]);
    };
    // dbias
    {
        float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
        for (int b = 0; b < g.B; ++b) {
            const float *row = paddr(dY, g, b, c, 0);
            int t = tid;
            for (; t + 768 < g.T; t += 1024) { s0 += row[t]; s1 += row[t + 256]; s2 += row[t + 512]; s3 += row[t + 768]; }
            for (; t < g.T; t += 256) s0 += row[t];
        }
        const float sb = block_sum((s0 + s1) + (s2 + s3));
        if (tid == 0 && dbias) dbias[c] = sb;
    }
    // dw[kk] = sum_{b,i} h[b,c,i] * dY[b,c,S*i+kk-Pd].  Two shapes occur: many taps over few frames (WaveGlow: K = 65, F = 63)
    // -> a thread per (tap, share of the batch items); few taps over many frames (WSRGlow: K = 3, F = 512 per item) -> the block
    // reduces four taps at a time.
    if (K >= 32) {
        const int nbg = max(1, 256 / K), bg = nbg > 1 ? tid / K : 0;
        for (int kk = nbg > 1 ? tid % K : tid; kk < K && bg < nbg; kk += 256) {       // (one pass unless K > 256)
            float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
            for (int b = bg; b < g.B; b += nbg) {
                const float *hr = h + ((size_t)b * C + c) * F;
                const float *row = paddr(dY, g, b, c, 0);
                auto term = [&](int i) { const int j = S * i + kk - Pd; return (i < F && j >= 0 && j < g.T) ? hr[i] * row[j] : 0.f; };
                for (int i = 0; i < F; i += 4) { s0 += term(i); s1 += term(i + 1); s2 += term(i + 2); s3 += term(i + 3); }
            }
            if (nbg > 1) tp[bg * K + kk] = (s0 + s1) + (s2 + s3);
            else dw[kk] = (s0 + s1) + (s2 + s3);
        }
        __syncthreads();
        if (nbg > 1 && tid < K) {
            float s = 0.f;
            for (int q = 0; q < nbg; ++q) s += tp[q * K + tid];
            dw[tid] = s;
        }
    } else {
        for (int k0 = 0; k0 < K; k0 += 4) {
            float s[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll 2
            for (int e = tid; e < g.B * F; e += 256) {
                const int b = e / F, i = e - b * F;
                const float hv = h[((size_t)b * C + c) * F + i];
                const float *row = paddr(dY, g, b, c, 0);
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int j = S * i + k0 + q - Pd;
                    if (k0 + q < K && j >= 0 && j < g.T) s[q] += hv * row[j];
                }
            }
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float t = block_sum(s[q]);
                if (tid == 0 && k0 + q < K) dw[k0 + q] = t;
            }
        }
    }
    __syncthreads();
    if (dh)
        for (int e = tid; e < g.B * F; e += 256) {
            const int b = e / F, i = e % F;
            float s = 0.f;
            for (int kk = 0; kk < K; ++kk) {
                const int j = S * i + kk - Pd;
                if (j >= 0 && j < g.T) s += *paddr(dY, g, b, c, j) * w[c * K + kk];
            }
            dh[((size_t)b * C + c) * F + i] = s;
        }
    // weight-norm backward of row c (utils.py:14-16): serial over K (K ~ 65)
    if (tid == 0 && (dv || dg)) {
        if (gparam) {
            float ss = 0.f, dot = 0.f;
            for (int kk = 0; kk < K; ++kk) { const float vv = v[c * K + kk]; ss += vv * vv; dot += dw[kk] * vv; }
            const float nrm = sqrtf(ss);
            if (dg) dg[c] = dot / nrm;
            const float aa = gparam[c] / nrm, bq = dot / ss;
            if (dv)
                for (int kk = 0; kk < K; ++kk) dv[c * K + kk] = aa * (dw[kk] - v[c * K + kk] * bq);
        } else if (dv) {
            for (int kk = 0; kk < K; ++kk) dv[c * K + kk] = dw[kk];
        }
    }
}

// ------------------------------------------------------------------------------------------------
// NLL loss (model/loss.py:10-15)
// ------------------------------------------------------------------------------------------------
// Two launches: WG_NLL_BLK blocks per batch item reduce (sum z, sum z^2) of their slice into `part`, one block combines the
// partials in double.  Besides the loss the second kernel produces the four scalars the reference logs every step
// (model/lightning.py:58-64): logdet.sum() / z.numel(), z.mean(), z.std() (unbiased, torch's default) and the loss.
#define WG_NLL_BLK 8
__global__ __launch_bounds__(256) void nll_partial_kernel(const float *__restrict__ z, int N, float *__restrict__ part)
{
    __shared__ float red[2][4];
    const int b = blockIdx.y, k = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int per = (N + WG_NLL_BLK - 1) / WG_NLL_BLK, n0 = k * per, n1 = min(N, n0 + per);
    const float *zb = z + (size_t)b * N;
    float s1 = 0.f, s2 = 0.f;
    for (int n = n0 + tid; n < n1; n += 256) { const float v = zb[n]; s1 += v; s2 = fmaf(v, v, s2); }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { s1 += __shfl_down(s1, o, 64); s2 += __shfl_down(s2, o, 64); }
    if (lane == 0) { red[0][wave] = s1; red[1][wave] = s2; }
    __syncthreads();
    if (tid == 0) {
        float *o = part + ((size_t)b * WG_NLL_BLK + k) * 2;
        o[0] = (red[0][0] + red[0][1]) + (red[0][2] + red[0][3]);
        o[1] = (red[1][0] + red[1][1]) + (red[1][2] + red[1][3]);
    }
}
// metrics (nullable) = [ sum_b logdet_b / (B N), mean(z), std(z), loss ]
__global__ __launch_bounds__(256) void nll_finish_kernel(const float *__restrict__ part, const float *__restrict__ logdet, int B, int N,
                                                         float inv_sigma2, int elementwise_mean, float *__restrict__ loss,
                                                         float *__restrict__ metrics)
{
    __shared__ double red[4][256];
    const int tid = threadIdx.x;
    double s1 = 0.0, s2 = 0.0, ld = 0.0, item = 0.0;
    for (int b = tid; b < B; b += 256) {
        double q1 = 0.0, q2 = 0.0;
        for (int k = 0; k < WG_NLL_BLK; ++k) { q1 += part[((size_t)b * WG_NLL_BLK + k) * 2]; q2 += part[((size_t)b * WG_NLL_BLK + k) * 2 + 1]; }
        s1 += q1; s2 += q2; ld += logdet[b];
        item += 0.5 * q2 * (double)inv_sigma2 - (double)logdet[b];                 // loss.py:11
    }
    red[0][tid] = s1; red[1][tid] = s2; red[2][tid] = ld; red[3][tid] = item;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if (tid < o) {
#pragma unroll
            for (int q = 0; q < 4; ++q) red[q][tid] += red[q][tid + o];
        }
        __syncthreads();
    }
    if (tid == 0) {
        const double n = (double)B * (double)N;
        double l = red[3][0] / (double)B;                                          // :12
        if (elementwise_mean) l /= (double)N;                                      // :13-14
        loss[0] = (float)l;
        if (metrics) {
            const double mean = red[0][0] / n;
            const double var = n > 1.0 ? fmax(red[1][0] - red[0][0] * mean, 0.0) / (n - 1.0) : __builtin_nan("");
            metrics[0] = (float)(red[2][0] / n);
            metrics[1] = (float)mean;
            metrics[2] = (float)sqrt(var);
            metrics[3] = (float)l;
        }
    }
}
__global__ void nll_loss_bwd_kernel(const float *__restrict__ z, int B, int N, float inv_sigma2, int elementwise_mean,
                                    const float *__restrict__ dloss, float *__restrict__ dz, float *__restrict__ dlogdet)
{
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    float sc = (dloss ? dloss[0] : 1.f) / (float)B;
    if (elementwise_mean) sc /= (float)N;
    if (i < (size_t)B * N) dz[i] = z[i] * inv_sigma2 * sc;
    if (i < (size_t)B) dlogdet[i] = -sc;
}

// ------------------------------------------------------------------------------------------------
// Weff_l = W_end Wskip_l  (2 ic x Cd per layer).  WN ends with out = W_end (sum_l Wskip_l gate_l) (model/waveglow.py:104-105): the skip
// sum's Cs rows are contracted to 2 ic <= 32 again at once, and so is everything that flows back through it -- dS = W_end^T G has rank
// 2 ic.  With Weff the engine never forms the skip sum or its gradient:
//   forward / inverse   out = sum_l Weff_l gate_l                 (end_affine_kernel<.., FROMG>: one pass over the gate planes)
//   gate backward       Wskip_l^T dS = Weff_l^T G                 (a K segment of 2 ic -> 32 channels instead of Cs; image rows from effN)
//   weight gradients    dWskip_l = W_end^T P_l,  dW_end = sum_l P_l Wskip_l^T   with  P_l = G gate_l^T  (2 ic x Cd: pgate_kernel, wg_thin.h)
// One thread per gate channel j of a layer: effT[(l Cd + j)][m] (the [K][32] layout of endT) and effN[l][m][j] (32 rows of Cd floats, the
// source of the gate backward's image rows), m < 2 ic, zero beyond.  Runs between the row norms and the pack jobs of a weight pack.
// ------------------------------------------------------------------------------------------------
struct EffJob {
    const float *wE;        // end.weight [2 ic][Cs] (plain weight)
    const float *v;         // W_o.weight_v of the layer, first SKIP row: [Cs][Cd]
    const float *scale;     // g / |v| of those rows [Cs]
    float *effT, *effN;     // [Cd][32] rows of the WN's effT ; [32][Cd]
    unsigned short *effA;   // nullable: rows 0-7 as 16x16x32 A fragments for the gate conv's epilogue (wg_gemm16g.h wgg_gate_nb):
                            // [Cd / 32 slices][hi | lo][k-group q][row m][8 bf16], element j of k-group q = channel 4 q + j (j < 4) or 16 + 4 q + j - 4
    int ic2, Cs, Cd;
};
#define WG_EFF_JOBS 64
struct EffArgs {
    int n;
    EffJob job[WG_EFF_JOBS];
};
// block (x, job): 64 gate channels; thread (c = tid & 63, q = tid >> 6) walks quarter q of the Cs skip rows, eight loads in flight (one
// thread per channel walking all Cs rows was a chain of Cs round trips: 190 us per launch for 4 MFLOP); W_end sits in LDS (broadcast
// reads); the four quarters are added in a fixed order
__global__ __launch_bounds__(256) void weff_kernel(const EffArgs a)
{
    const EffJob j = a.job[blockIdx.y];
    if ((int)blockIdx.x * 64 >= j.Cd) return;
    extern __shared__ float weff_lds[];                       // wE [ic2][Cs], then part [3][32][64]
    float *we = weff_lds, *part = weff_lds + j.ic2 * j.Cs;
    const int tid = threadIdx.x, cl = tid & 63, q = tid >> 6, c = blockIdx.x * 64 + cl;
    for (int i = tid; i < j.ic2 * j.Cs; i += 256) we[i] = j.wE[i];
    __syncthreads();
    float acc[32];
#pragma unroll
    for (int m = 0; m < 32; ++m) acc[m] = 0.f;
    const int sq = (j.Cs + 3) / 4, s0 = q * sq, s1 = min(j.Cs, s0 + sq);
    const int cc = min(c, j.Cd - 1);
    for (int s = s0; s < s1; s += 8) {
        float w[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int ss = min(s + u, s1 - 1);
            w[u] = j.scale[ss] * j.v[(size_t)ss * j.Cd + cc];           // the effective skip weight, as the pack jobs form it
            if (s + u >= s1) w[u] = 0.f;
        }
#pragma unroll
        for (int u = 0; u < 8; ++u)
#pragma unroll
            for (int m = 0; m < 32; ++m)
                if (m < j.ic2) acc[m] = fmaf(we[m * j.Cs + min(s + u, s1 - 1)], w[u], acc[m]);
    }
    if (q)
#pragma unroll
        for (int m = 0; m < 32; ++m) part[((q - 1) * 32 + m) * 64 + cl] = acc[m];
    __syncthreads();
    if (q == 0 && c < j.Cd) {
#pragma unroll
        for (int m = 0; m < 32; ++m) {
            const float x = m < j.ic2 ? (acc[m] + part[m * 64 + cl]) + (part[(32 + m) * 64 + cl] + part[(64 + m) * 64 + cl]) : 0.f;
            j.effT[(size_t)c * 32 + m] = x;
            j.effN[(size_t)m * j.Cd + c] = x;
            if (j.effA && m < 8) {
                const int sl = c >> 5, cc = c & 31, qq = (cc & 15) >> 2, jj = (cc & 3) + (cc >= 16 ? 4 : 0);
                unsigned hh, ll;
                split2(x, 0.f, hh, ll);
                unsigned short *e = j.effA + ((size_t)(sl * 2) * 32 + qq * 8 + m) * 8 + jj;
                e[0] = (unsigned short)(hh & 0xffffu);
                e[32 * 8] = (unsigned short)(ll & 0xffffu);
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// WN.start folded into the first layer's dilated conv (start_fold_shape in wgflow.hip).  h_0 = W_start xa has rank ic (2-4 channels of
// the flow) and layer 0 convolves it at once (model/waveglow.py:99, 41-43): W_0 * (W_start xa) = (W_0[kt] W_start) * xa, a conv over ic
// channels instead of C = 256.  This kernel forms the composed weight in the layout of a conv weight, w0x[o][j][kt] =
// sum_c sW[o] vW[o][c][kt] . sS[c] vS[c][j] (o < 2 Cd, j < ic; the weight-norm scales of both factors applied), so that the ordinary pack
// jobs image it like W_0 itself.  One thread per (o, j, kt); runs between the row norms and the pack jobs of a weight pack.
// ------------------------------------------------------------------------------------------------
struct FoldJob {
    const float *vW, *sW;   // W_0.weight_v [2 Cd][C][radix], g / |v| per row
    const float *vS, *sS;   // start.weight_v [C][ic], g / |v| per row
    float *out;             // [2 Cd][ic][radix]
    int M, C, ic, radix;
};
#define WG_FOLD_JOBS 32
struct FoldArgs {
    int n;
    FoldJob job[WG_FOLD_JOBS];
};
__global__ __launch_bounds__(256) void start_fold_kernel(const FoldArgs a)
{
    const FoldJob j = a.job[blockIdx.y];
    const int e = blockIdx.x * 256 + threadIdx.x, per = j.ic * j.radix;
    if (e >= j.M * per) return;
    const int o = e / per, r = e - o * per, jj = r / j.radix, kt = r - jj * j.radix;
    const float *w = j.vW + (size_t)o * j.C * j.radix + kt;
    float acc[4] = {0.f, 0.f, 0.f, 0.f};                      // four interleaved partial sums: fixed order, four loads in flight
    for (int c = 0; c < j.C; c += 4) {
#pragma unroll
        for (int u = 0; u < 4; ++u)
            if (c + u < j.C) acc[u] = fmaf(w[(size_t)(c + u) * j.radix], j.sS[c + u] * j.vS[(size_t)(c + u) * j.ic + jj], acc[u]);
    }
    j.out[e] = j.sW[o] * ((acc[0] + acc[1]) + (acc[2] + acc[3]));
}

// ------------------------------------------------------------------------------------------------
// weight packing: effective weights (weight norm, utils.py:14-16) laid out k-major for the MFMA kernels
// ------------------------------------------------------------------------------------------------
struct NormJob {
    const float *g, *v;   // g may be NULL (plain weight): scale = 1
    float *scale;
    int rows, cols;
};
#define WG_JOBS 40
struct NormArgs {
    int n;
    NormJob job[WG_JOBS];
};
// one wave per row
__global__ void rownorm_kernel(const NormArgs a)
{
    const NormJob j = a.job[blockIdx.y];
    const int row = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (row >= j.rows) return;
    if (!j.g) { if (lane == 0) j.scale[row] = 1.f; return; }
    float ss = 0.f;
    for (int c = lane; c < j.cols; c += 64) { const float x = j.v[(size_t)row * j.cols + c]; ss += x * x; }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) ss += __shfl_down(ss, o, 64);
    if (lane == 0) j.scale[row] = j.g[row] / sqrtf(ss);
}

// dst[k][m] (ld = ldd, Kp x Mp region, zero filled) = scale[o] * src[o*so + i*si + off]
//   mode 0 ("T"): o = perm(m), i = k      (valid: m < no, k < ni)
//   mode 1 ("N"): o = k,       i = m      (valid: k < no, m < ni)
//   perm (gate interleave, only mode 0 with half > 0): 64-blocks [32 tanh rows | 32 sigmoid rows]
struct PackJob {
    float *dst;
    const float *src, *scale;
    int ldd, Kp, Mp;
    int mode, no, ni, half;
    int so, si, off;
    // fused weight image (packimg_kernel, wg_gemm16.h): the job's rows are chunks [chunk0, chunk0 + ceil(Kp / 32)) of the split bf16
    // image of its matrix (nchunks chunks in all; hi, then lo); dst may then be nullptr (no fp32 copy wanted).  img == nullptr: fp32 only.
    unsigned short *img;
    int chunk0, nchunks;
};
struct PackArgs {
    int n;
    PackJob job[WG_JOBS];
};
__device__ __forceinline__ void pack_job_plain(const PackJob &j)
{
    if (j.mode == 0) {
        // the transposing form through a 32 (k) x 64 (m) LDS tile: consecutive lanes read consecutive k of one source row (12 bytes
        // apart for a k = 3 conv weight) and write consecutive m.  (Element by element in dst order every lane read its own source
        // row: 19 us per launch, 31 launches per training step.)
        __shared__ float tile[32][65];
        const int tid = threadIdx.x;
        const int tk = (j.Kp + 31) / 32, tm = (j.Mp + 63) / 64;
        for (int tl = blockIdx.x; tl < tk * tm; tl += gridDim.x) {
            const int k0 = (tl / tm) * 32, m0 = (tl % tm) * 64;
            __syncthreads();
            const int kk = tid & 31, k = k0 + kk;
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const int mm = (tid >> 5) + 8 * q, m = m0 + mm;
                int o = m;
                if (j.half > 0) {
                    const int qq = m >> 6, r = m & 63;
                    o = r < 32 ? qq * 32 + r : j.half + qq * 32 + (r - 32);
                    if ((qq * 32 + (r & 31)) >= j.half) o = -1;
                }
                float val = 0.f;
                if (o >= 0 && o < j.no && k < j.ni && m < j.Mp) val = j.scale[o] * j.src[(size_t)o * j.so + (size_t)k * j.si + j.off];
                tile[kk][mm] = val;
            }
            __syncthreads();
            const int m = m0 + (tid & 63);
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const int kw = (tid >> 6) + 4 * q;
                if (k0 + kw < j.Kp && m < j.Mp) j.dst[(size_t)(k0 + kw) * j.ldd + m] = tile[kw][tid & 63];
            }
        }
        return;
    }
    const size_t total = (size_t)j.Kp * j.Mp;
    for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (size_t)gridDim.x * blockDim.x) {
        const int k = (int)(e / j.Mp), m = (int)(e % j.Mp);
        float val = 0.f;
        if (j.mode == 0) {
            int o = m;
            if (j.half > 0) {
                const int q = m >> 6, r = m & 63;
                o = r < 32 ? q * 32 + r : j.half + q * 32 + (r - 32);
                if ((q * 32 + (r & 31)) >= j.half) o = -1;
            }
            if (o >= 0 && o < j.no && k < j.ni) val = j.scale[o] * j.src[(size_t)o * j.so + (size_t)k * j.si + j.off];
        } else {
            if (k < j.no && m < j.ni) val = j.scale[k] * j.src[(size_t)k * j.so + (size_t)m * j.si + j.off];
        }
        j.dst[(size_t)k * j.ldd + m] = val;
    }
}
__global__ __launch_bounds__(256) void pack_kernel(const PackArgs a)
{
    pack_job_plain(a.job[blockIdx.y]);
}

// ------------------------------------------------------------------------------------------------
// gradient finalisation: sum the split-K slabs of one weight tensor and apply the weight-norm backward
//   dw[o][i][r] = sum_s slab[s][o][col0 + r*cr + i*ci]
//   dg[o] = <dw,v>/||v|| ; dv = g/||v|| (dw - v <dw,v>/||v||^2)      (g == NULL: dv = dw)
// optionally adds extra[o*I*R + e] * extra_scale  (the W^-T * dlogdet * T term of efficient_modules.py:242)
// ------------------------------------------------------------------------------------------------
// slab[0][e] = sum_s slab[s][e]: pre-reduction of the split-K slabs with the whole GPU when the weight tensor has few rows (the
// finalize kernel runs one block per row, which is 128 blocks for WaveFlow's 128-row weights reading 512 slabs each)
__global__ __launch_bounds__(256) void slab_reduce_kernel(float *__restrict__ slab, int nsplit, size_t n)
{
    // 64 float4 columns x 4 slab lanes per block; a lane walks slabs lane, lane + 4, ... with four loads in flight
    __shared__ float4 part[4][64];
    const int tid = threadIdx.x, c = tid & 63, lane = tid >> 6;
    const size_t e = ((size_t)blockIdx.x * 64 + c) * 4;
    float4 a0 = {0.f, 0.f, 0.f, 0.f}, a1 = a0, a2 = a0, a3 = a0;
    if (e < n) {
        int s = lane;
        for (; s + 12 < nsplit; s += 16) {
            const float4 v0 = *reinterpret_cast<const float4 *>(slab + (size_t)s * n + e);
            const float4 v1 = *reinterpret_cast<const float4 *>(slab + (size_t)(s + 4) * n + e);
            const float4 v2 = *reinterpret_cast<const float4 *>(slab + (size_t)(s + 8) * n + e);
            const float4 v3 = *reinterpret_cast<const float4 *>(slab + (size_t)(s + 12) * n + e);
            a0.x += v0.x; a0.y += v0.y; a0.z += v0.z; a0.w += v0.w;
            a1.x += v1.x; a1.y += v1.y; a1.z += v1.z; a1.w += v1.w;
            a2.x += v2.x; a2.y += v2.y; a2.z += v2.z; a2.w += v2.w;
            a3.x += v3.x; a3.y += v3.y; a3.z += v3.z; a3.w += v3.w;
        }
        for (; s < nsplit; s += 4) {
            const float4 v0 = *reinterpret_cast<const float4 *>(slab + (size_t)s * n + e);
            a0.x += v0.x; a0.y += v0.y; a0.z += v0.z; a0.w += v0.w;
        }
    }
    float4 t;
    t.x = (a0.x + a1.x) + (a2.x + a3.x); t.y = (a0.y + a1.y) + (a2.y + a3.y);
    t.z = (a0.z + a1.z) + (a2.z + a3.z); t.w = (a0.w + a1.w) + (a2.w + a3.w);
    part[lane][c] = t;
    __syncthreads();                              // every slab has been read before slab 0 is overwritten (by this block's columns only)
    if (lane == 0 && e < n) {
        const float4 p1 = part[1][c], p2 = part[2][c], p3 = part[3][c];
        t.x = (t.x + p1.x) + (p2.x + p3.x); t.y = (t.y + p1.y) + (p2.y + p3.y);
        t.z = (t.z + p1.z) + (p2.z + p3.z); t.w = (t.w + p1.w) + (p2.w + p3.w);
        *reinterpret_cast<float4 *>(slab + e) = t;
    }
}

struct FinJob {
    const float *slab;
    int nsplit;
    size_t sstride;     // Mp*Np
    int ldn;            // Np
    int row0;           // first slab row of this tensor
    int rows, I, R;     // tensor is [rows][I][R]
    int col0, ci, cr;
    const float *g, *v;
    float *dg, *dv;
    const float *extra; // nullable, [rows][I*R] accessed transposed: extra[e*rows + o]
    const float *extra_scale_src; // device vector summed over n_extra entries
    int n_extra;
    float extra_mul;
};
#define WG_FIN_MAXCOLS 8192
#define WG_FIN_REGS 16        // columns a thread of the long-row path keeps in registers: rows of up to 4 096 columns
__device__ __forceinline__ void finalize_row(const FinJob &j, int o, float *dw, float (*red)[256])
{
    const int tid = threadIdx.x;
    const int cols = j.I * j.R;
    float esc = 0.f;
    if (j.extra) {
        for (int i = 0; i < j.n_extra; ++i) esc += j.extra_scale_src[i];
        esc *= j.extra_mul;
    }
    // pass 1: sum the split-K slabs walking the slab row in MEMORY order (coalesced), scatter into tensor order in LDS.
    // slab column of tensor element (i, r) is col0 + r*cr + i*ci; ci == 1 for every caller.
    const float *row = j.slab + (size_t)(j.row0 + o) * j.ldn + j.col0;
    if (j.R == 1 && j.ci == 1 && !j.extra && j.g && j.dv && cols > 1024 && cols <= 256 * WG_FIN_REGS) {
        // Long rows of a weight-normed 1x1 conv (WSRGlow's conditioning: 4 096 rows of 3 659 columns): a thread keeps its columns of dw
        // and v in registers -- every load of the row is issued before the first is used, v is read once, nothing goes through the LDS
        // row buffer.  (As three passes of 15 dependent trips through LDS the batch ran at 1.8 TB/s: 211 us per WN.)
        float dwr[WG_FIN_REGS], vr[WG_FIN_REGS];
        const float *vrow = j.v + (size_t)o * cols;
#pragma unroll
        for (int q = 0; q < WG_FIN_REGS; ++q) {
            const int e = tid + 256 * q;
            float s0 = 0.f, s1 = 0.f;
            if (e < cols) {
                int sp = 0;
                for (; sp + 1 < j.nsplit; sp += 2) { s0 += row[(size_t)sp * j.sstride + e]; s1 += row[(size_t)(sp + 1) * j.sstride + e]; }
                if (sp < j.nsplit) s0 += row[(size_t)sp * j.sstride + e];
            }
            dwr[q] = s0 + s1;
            vr[q] = e < cols ? vrow[e] : 0.f;
        }
        float dot = 0.f, ss = 0.f;
#pragma unroll
        for (int q = 0; q < WG_FIN_REGS; ++q) { dot += dwr[q] * vr[q]; ss += vr[q] * vr[q]; }
        red[0][tid] = dot; red[1][tid] = ss;
        __syncthreads();
        for (int q = 128; q > 0; q >>= 1) {
            if (tid < q) { red[0][tid] += red[0][tid + q]; red[1][tid] += red[1][tid + q]; }
            __syncthreads();
        }
        dot = red[0][0]; ss = red[1][0];
        const float nrm = sqrtf(ss);
        if (tid == 0 && j.dg) j.dg[o] = dot / nrm;
        const float aa = j.g[o] / nrm, bq = dot / ss;
        float *dvrow = j.dv + (size_t)o * cols;
#pragma unroll
        for (int q = 0; q < WG_FIN_REGS; ++q) {
            const int e = tid + 256 * q;
            if (e < cols) dvrow[e] = aa * (dwr[q] - vr[q] * bq);
        }
        return;
    }
    for (int r = 0; r < j.R; ++r)
        for (int i = tid; i < j.I; i += 256) {
            const size_t off = (size_t)r * j.cr + (size_t)i * j.ci;
            float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
            int sp = 0;
            for (; sp + 3 < j.nsplit; sp += 4) {
                s0 += row[(size_t)sp * j.sstride + off];
                s1 += row[(size_t)(sp + 1) * j.sstride + off];
                s2 += row[(size_t)(sp + 2) * j.sstride + off];
                s3 += row[(size_t)(sp + 3) * j.sstride + off];
            }
            for (; sp < j.nsplit; ++sp) s0 += row[(size_t)sp * j.sstride + off];
            dw[i * j.R + r] = (s0 + s1) + (s2 + s3);
        }
    __syncthreads();
    float dot = 0.f, ss = 0.f;
    for (int e = tid; e < cols; e += 256) {
        float s = dw[e];
        if (j.extra) { s += j.extra[(size_t)e * j.rows + o] * esc; dw[e] = s; }
        if (j.g) { const float vv = j.v[(size_t)o * cols + e]; dot += s * vv; ss += vv * vv; }
    }
    if (!j.g) {
        if (j.dv)
            for (int e = tid; e < cols; e += 256) j.dv[(size_t)o * cols + e] = dw[e];
        return;
    }
    red[0][tid] = dot; red[1][tid] = ss;
    __syncthreads();
    for (int q = 128; q > 0; q >>= 1) {
        if (tid < q) { red[0][tid] += red[0][tid + q]; red[1][tid] += red[1][tid + q]; }
        __syncthreads();
    }
    dot = red[0][0]; ss = red[1][0];
    const float nrm = sqrtf(ss);
    if (tid == 0 && j.dg) j.dg[o] = dot / nrm;
    if (!j.dv) return;                                   // weight_v frozen, weight_g trainable: only dg is wanted
    const float aa = j.g[o] / nrm, bq = dot / ss;
    for (int e = tid; e < cols; e += 256) j.dv[(size_t)o * cols + e] = aa * (dw[e] - j.v[(size_t)o * cols + e] * bq);
}
__global__ __launch_bounds__(256) void finalize_kernel(const FinJob j)
{
    __shared__ float dw[WG_FIN_MAXCOLS];
    __shared__ float red[2][256];
    finalize_row(j, blockIdx.x, dw, red);
}
// Several tensors in one launch (one block per weight row of every job): a WN layer's backward ends in three of these small
// launches, each too short to fill the GPU; the weight gradients of a whole WN are finalised together instead (FinQueue).
#define WG_FIN_JOBS 32     // a WN of 8 layers queues 26 jobs (end, 8 x (W, V, W_o), start): one launch
struct FinBatch {
    int n;
    int start[WG_FIN_JOBS + 1];     // prefix sums of the jobs' row counts: the grid is exactly their total
    FinJob job[WG_FIN_JOBS];
};
static_assert(sizeof(FinBatch) + 16 <= 4096, "kernel arguments are limited to 4 KB");
__global__ __launch_bounds__(256) void finalize_batch_kernel(const FinBatch b)
{
    __shared__ float dw[WG_FIN_MAXCOLS];
    __shared__ float red[2][256];
    int ji = 0;
    for (int q = 1; q < b.n; ++q)
        if ((int)blockIdx.x >= b.start[q]) ji = q;
    finalize_row(b.job[ji], (int)blockIdx.x - b.start[ji], dw, red);
}

// The same with ONE WAVE per weight row, four rows per block (dynamic LDS: 4 x maxcols floats): a row is 0.3-7 KB of slab per split,
// so a block per row was 11 784 blocks of a few microseconds of dependent latency each for a WN (71 us, bound by block dispatch and
// by four resident blocks per CU); a wave needs no block barrier and no 32 KB static row buffer.
__device__ __forceinline__ void finalize_row_wave(const FinJob &j, int o, float *dw)
{
    const int lane = threadIdx.x & 63;
    const int cols = j.I * j.R;
    float esc = 0.f;
    if (j.extra) {
        for (int i = 0; i < j.n_extra; ++i) esc += j.extra_scale_src[i];
        esc *= j.extra_mul;
    }
    const float *row = j.slab + (size_t)(j.row0 + o) * j.ldn + j.col0;
    for (int r = 0; r < j.R; ++r)
        for (int i = lane; i < j.I; i += 64) {
            const size_t off = (size_t)r * j.cr + (size_t)i * j.ci;
            float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
            int sp = 0;
            for (; sp + 3 < j.nsplit; sp += 4) {
                s0 += row[(size_t)sp * j.sstride + off];
                s1 += row[(size_t)(sp + 1) * j.sstride + off];
                s2 += row[(size_t)(sp + 2) * j.sstride + off];
                s3 += row[(size_t)(sp + 3) * j.sstride + off];
            }
            for (; sp < j.nsplit; ++sp) s0 += row[(size_t)sp * j.sstride + off];
            dw[i * j.R + r] = (s0 + s1) + (s2 + s3);
        }
    __builtin_amdgcn_wave_barrier();                     // (one wave: its LDS accesses execute in order)
    float dot = 0.f, ss = 0.f;
    for (int e = lane; e < cols; e += 64) {
        float s = dw[e];
        if (j.extra) { s += j.extra[(size_t)e * j.rows + o] * esc; dw[e] = s; }
        if (j.g) { const float vv = j.v[(size_t)o * cols + e]; dot += s * vv; ss += vv * vv; }
    }
    if (!j.g) {
        if (j.dv)
            for (int e = lane; e < cols; e += 64) j.dv[(size_t)o * cols + e] = dw[e];
        return;
    }
#pragma unroll
    for (int q = 32; q > 0; q >>= 1) { dot += __shfl_xor(dot, q, 64); ss += __shfl_xor(ss, q, 64); }
    const float nrm = sqrtf(ss);
    if (lane == 0 && j.dg) j.dg[o] = dot / nrm;
    if (!j.dv) return;
    const float aa = j.g[o] / nrm, bq = dot / ss;
    for (int e = lane; e < cols; e += 64) j.dv[(size_t)o * cols + e] = aa * (dw[e] - j.v[(size_t)o * cols + e] * bq);
}
__global__ __launch_bounds__(256) void finalize_batch_wave_kernel(const FinBatch b, int maxcols)
{
    extern __shared__ float dwall[];
    const int gid = (int)blockIdx.x * 4 + (int)(threadIdx.x >> 6);
    if (gid >= b.start[b.n]) return;
    int ji = 0;
    for (int q = 1; q < b.n; ++q)
        if (gid >= b.start[q]) ji = q;
    finalize_row_wave(b.job[ji], gid - b.start[ji], dwall + (size_t)(threadIdx.x >> 6) * maxcols);
}

// InvConv1x1Func.backward tail (efficient_modules.py:276-277): dW = -W^-T dw W^-T - W^-T * dlogdet * T,
// dw = sum of the split-K slabs.  c <= WG_MAXC, one block.
struct InvRevFinArgs {
    const float *slab;
    int nsplit;
    size_t sstride;
    int ldn, c;
    const float *Winv, *dlogdet;
    float T;
    float *dW;
};
__global__ __launch_bounds__(256) void invconv_rev_finalize_kernel(const InvRevFinArgs a)
{
    __shared__ float dw[WG_MAXC * WG_MAXC], tmp[WG_MAXC * WG_MAXC];
    const int c = a.c, tid = threadIdx.x;
    for (int e = tid; e < c * c; e += 256) {
        const int i = e / c, j = e % c;
        float s = 0.f;
        for (int sp = 0; sp < a.nsplit; ++sp) s += a.slab[sp * a.sstride + (size_t)i * a.ldn + j];
        dw[e] = s;
    }
    __syncthreads();
    for (int e = tid; e < c * c; e += 256) {     // tmp = W^-T dw
        const int i = e / c, j = e % c;
        float s = 0.f;
        for (int k = 0; k < c; ++k) s += a.Winv[k * c + i] * dw[k * c + j];
        tmp[e] = s;
    }
    __syncthreads();
    const float gl = a.dlogdet[0] * a.T;
    for (int e = tid; e < c * c; e += 256) {     // -(tmp W^-T) - W^-T gl
        const int i = e / c, j = e % c;
        float s = 0.f;
        for (int k = 0; k < c; ++k) s += tmp[i * c + k] * a.Winv[j * c + k];
        a.dW[e] = -s - a.Winv[j * c + i] * gl;
    }
}

// ------------------------------------------------------------------------------------------------
// Adam update of one contiguous fp32 range (the optimizer the reference instantiates from its configs: torch.optim.Adam,
// model/lightning.py:41-44; amsgrad = false, maximize = false).  Same operation order as torch's single-tensor path:
//   g += wd * p ; m = lerp(m, g, 1 - b1) ; v = v * b2 + (1 - b2) * g * g ; p -= step_size * m / (sqrt(v) / sqrt(bc2) + eps)
// with step_size = lr / (1 - b1^t), bc2 = 1 - b2^t computed by the host in double.  HBM-bound: 16 B read + 12 B written per element.
// ------------------------------------------------------------------------------------------------
struct AdamArgs {
    float *p, *m, *v;
    const float *g;
    size_t n;
    float one_minus_b1, b2, one_minus_b2, eps, wd, step_size, inv_bc2_sqrt;
};
__device__ __forceinline__ void adam_one(float &p, float g, float &m, float &v, const AdamArgs &a)
{
    if (a.wd != 0.f) g = fmaf(a.wd, p, g);
    m = m + a.one_minus_b1 * (g - m);
    v = fmaf(a.one_minus_b2 * g, g, v * a.b2);
    const float denom = sqrtf(v) * a.inv_bc2_sqrt + a.eps;
    p = p - a.step_size * (m / denom);
}
__global__ __launch_bounds__(256) void adam_kernel(const AdamArgs a)
{
    const size_t n4 = a.n >> 2;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
        float4 p = reinterpret_cast<float4 *>(a.p)[i], m = reinterpret_cast<float4 *>(a.m)[i], v = reinterpret_cast<float4 *>(a.v)[i];
        const float4 g = reinterpret_cast<const float4 *>(a.g)[i];
        adam_one(p.x, g.x, m.x, v.x, a);
        adam_one(p.y, g.y, m.y, v.y, a);
        adam_one(p.z, g.z, m.z, v.z, a);
        adam_one(p.w, g.w, m.w, v.w, a);
        reinterpret_cast<float4 *>(a.p)[i] = p;
        reinterpret_cast<float4 *>(a.m)[i] = m;
        reinterpret_cast<float4 *>(a.v)[i] = v;
    }
    if (blockIdx.x == 0 && threadIdx.x < (a.n & 3)) {          // tail
        const size_t i = (n4 << 2) + threadIdx.x;
        adam_one(a.p[i], a.g[i], a.m[i], a.v[i], a);
    }
}

// ------------------------------------------------------------------------------------------------
// Stand-alone NonCausalLayer (model/waveglow.py:18-46; wg_layer_apply): the two conv weights materialised (old-style weight norm over all dims
// but 0 when g is given, utils.py:14-16) into the k-major fp32 matrices the exact-fp32 conv kernel takes.  One block per weight row.
//   kind 0: W [2 Cd][C][radix]  -> Acat[k = tap * C + c][m'], m' = the gate interleave of the row (64-blocks [32 tanh | 32 sigmoid]),
//           followed by an identity block: Acat[radix * C + j][m'(j)] = 1, so that the caller's conditioning y[2 Cd] is one more K segment
//   kind 1: W_o [R][Cd][1]      -> WoT[k = cd][m = row]
// ------------------------------------------------------------------------------------------------
struct LayerPackArgs {
    const float *g, *v;      // g nullable: plain conv weight
    float *dst;              // k-major, leading dimension ld (zero-filled by the caller)
    int rows, fan, ld;       // rows of the weight, elements per row, leading dimension of dst
    int kind, C, Cd, radix;
};
__global__ __launch_bounds__(256) void layer_pack_kernel(const LayerPackArgs a)
{
    __shared__ float red[256];
    const int row = blockIdx.x, tid = threadIdx.x;
    const float *vr = a.v + (size_t)row * a.fan;
    float ss = 0.f;
    for (int e = tid; e < a.fan; e += 256) ss += vr[e] * vr[e];
    red[tid] = ss;
    __syncthreads();
    for (int q = 128; q > 0; q >>= 1) {
        if (tid < q) red[tid] += red[tid + q];
        __syncthreads();
    }
    const float scale = a.g ? a.g[row] / sqrtf(red[0]) : 1.0f;
    if (a.kind == 0) {
        const int ch = row < a.Cd ? row : row - a.Cd;
        const int m = (ch >> 5) * 64 + (row < a.Cd ? 0 : 32) + (ch & 31);
        for (int e = tid; e < a.fan; e += 256) {              // v[row][c][tap]
            const int c = e / a.radix, tap = e - c * a.radix;
            a.dst[(size_t)(tap * a.C + c) * a.ld + m] = vr[e] * scale;
        }
        if (tid == 0) a.dst[(size_t)(a.radix * a.C + row) * a.ld + m] = 1.0f;
    } else if (a.kind == 1) {
        for (int e = tid; e < a.fan; e += 256) a.dst[(size_t)e * a.ld + row] = vr[e] * scale;
    } else if (a.kind == 2) {
        // the dilated conv's weight for its data gradient (wg_layer_backward): K = (tap, output row of the conv), M = input channel
        for (int e = tid; e < a.fan; e += 256) {              // v[row][c][tap]
            const int c = e / a.radix, tap = e - c * a.radix;
            a.dst[(size_t)(tap * a.rows + row) * a.ld + c] = vr[e] * scale;
        }
    } else {
        // W_o for the gate's gradient: K = W_o's output row, M = dilation channel -- the weight as it is, normalised
        for (int e = tid; e < a.fan; e += 256) a.dst[(size_t)row * a.ld + e] = vr[e] * scale;
    }
}
// x[items][C][H][T] (the layout of a Conv2d activation) <-> planes with one plane row per (item, height row)
__global__ void import2d_kernel(const float *__restrict__ src, PRef X, Geo g, int C)
{
    const int t = blockIdx.x * blockDim.x + threadIdx.x, c = blockIdx.y, row = blockIdx.z;
    if (t >= g.T) return;
    const int b = row / g.rows, h = row - b * g.rows;
    *paddr(X, g, row, c, t) = src[(((size_t)b * C + c) * g.rows + h) * g.T + t];
}
__global__ void export2d_kernel(PRef X, float *__restrict__ dst, Geo g, int C)
{
    const int t = blockIdx.x * blockDim.x + threadIdx.x, c = blockIdx.y, row = blockIdx.z;
    if (t >= g.T) return;
    const int b = row / g.rows, h = row - b * g.rows;
    dst[(((size_t)b * C + c) * g.rows + h) * g.T + t] = *paddr(X, g, row, c, t);
}

