// wg_thin.h -- the two THIN products at the ends of WN's backward, each as ONE pass over its wide operand (precision 2).
//
// WN.start is a 1x1 conv from the coupling's ic <= 16 input channels to C residual channels, WN.end one from the Cs skip channels to
// 2 ic outputs (model/waveglow.py:61-67,98-105).  Their backward is two products each -- the weight gradient and the data gradient --
// whose one wide operand (dh_0 resp. skip: C x B T values, 49 MB at the headline shape) is the whole cost and whose other side is 4-8
// rows.  As 128 x 128-tile MFMA launches they ran at 3 % tile occupancy with a 256-way split-K behind them:
//     start: wgrad16s_kernel<1> 48 us + slab_reduce 8 us + convgemm16q<0> 17 us     end: wgrad16_kernel 24 + slab_reduce 7 + to_splane 4 + convgemm16q<0> 19
// Here a workgroup streams 64-column tiles of the wide operand ONCE, coalesced, and produces both results from it on the vector ALU in
// fp32 (the operands are the fp32 values themselves -- hi + lo of an S-plane is exact to 2^-17 -- so these are at least as exact as
// the three-product MFMA form they replace):
//   * the data gradient column by column (a lane owns one time step; the few weights sit in LDS and every read of them is a wave-wide broadcast);
//   * the weight gradient through an LDS tile [channel][time] (pitch 65 floats: conflict-free both ways): thread c then walks the 64
//     time steps of ITS channel against the thin operand's rows, which every lane reads from the same LDS address (broadcast).
// A workgroup keeps its weight-gradient sums in registers over all its tiles and writes ONE partial; thin_fold_kernel adds the partials
// in a fixed order (reproducible bit for bit) and run_finalize sees a single slab.
#pragma once
#include "wg_gemm16s.h"

#define WGTH_TB 64                 // time steps per tile
#define WGTH_LDT 65                // LDS pitch of a channel row
#define WGTH_THREADS 256
#define WGTH_MAXROWS 2             // channel rows per thread: C, Cs <= 512
#define WGTH_BATCH 8                // start: channel groups (2 x 16 bytes per lane each) a wave has in flight
#define WGTH_BATCH2 16              // end: 4-row groups (16 bytes per lane each) a wave has in flight

#if !defined(WGTH_DBG)
#define WGTH_DBG 0                  // timing bisection only: 1 no reduction, 2 no weight-gradient walk, 4 no tile / data-gradient work, 8 no loads
#endif
// out[e] = sum_p part[p][e] (np partials of n floats, n a multiple of 32): a block takes 8 columns of 16 bytes, thread (column, lane l of 32)
// adds partials l, l + 32, ... (all its loads in flight), thread (column, 0) then adds the 32 lanes' sums in lane order -- a fixed order,
// so the result is reproducible.  (Folding inside the producing launch -- the last workgroup to finish, found by an atomic ticket, one
// or two levels -- was measured at 18-20 us of tail on a 20 us kernel, or 150-290 us when ONE workgroup walked all the partials: every
// partial is a round trip to memory.  This launch is ~3 us.)
__global__ __launch_bounds__(256) void thin_fold_kernel(const float *__restrict__ part, int np, int n, float *__restrict__ out)
{
    __shared__ f32x4 sums[32][8];
    const int col = threadIdx.x & 7, l = threadIdx.x >> 3;
    const int e = ((int)blockIdx.x * 8 + col) * 4;
    f32x4 s = {0.f, 0.f, 0.f, 0.f};
    for (int p0 = l; p0 < np; p0 += 32 * 8) {
        f32x4 v[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const int p = p0 + 32 * q;
            v[q] = f32x4{0.f, 0.f, 0.f, 0.f};
            if (p < np) v[q] = *reinterpret_cast<const f32x4 *>(part + (size_t)p * n + e);
        }
#pragma unroll
        for (int q = 0; q < 8; ++q) s += v[q];
    }
    sums[l][col] = s;
    __syncthreads();
    if (l == 0) {
        f32x4 t = sums[0][col];
        for (int q = 1; q < 32; ++q) t += sums[q][col];
        *reinterpret_cast<f32x4 *>(out + e) = t;
    }
}
// floats of the partial buffer for a grid of G workgroups: partials, result
__host__ __device__ inline size_t wgth_part_floats(int G, int n) { return (size_t)(G + 1) * n; }

// ------------------------------------------------------------------------------------------------
// start:  dW[c][j] = sum_{b,t} dh[c][t] xa[j][t]        dxa[j][t] += sum_c W[c][j] dh[c][t]            (model/waveglow.py:98, backward)
// ------------------------------------------------------------------------------------------------
struct ThinStartArgs {
    SRef dh;             // S-plane of dh_0, C channels
    PRef X, dX;          // xa = X channels [ch0, ch0 + ic); dX likewise (accumulated into)
    const float *W;      // fp32 k-major effective weights [C][ldw]: W[c][j]
    int ldw, C, ic;
    Geo g;
    int tiles;           // B * Tt / 64
    float *part;         // [gridDim.x][C * ICP]
};
template <int ICP>
__global__ __launch_bounds__(WGTH_THREADS) void thin_start_kernel(const ThinStartArgs a)
{
    extern __shared__ __attribute__((aligned(16))) float wgth_smem[];
    float *tile = wgth_smem;                                   // [C][65]
    float *xs = tile + (size_t)a.C * WGTH_LDT;                 // [64][ICP]
    float *axs = xs + WGTH_TB * ICP;                           // [4][ICP][64]: the four waves' shares of dxa
    float *Ws = axs + 4 * ICP * 64;                            // [C][ICP]: the weights (every read is wave-uniform: an LDS broadcast)
    const int tid = threadIdx.x, tl = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const Geo g = a.g;
    const int ncg = a.C >> 3, tpb = g.Tt / WGTH_TB;
    float acc[WGTH_MAXROWS][ICP];
#pragma unroll
    for (int r = 0; r < WGTH_MAXROWS; ++r)
#pragma unroll
        for (int j = 0; j < ICP; ++j) acc[r][j] = 0.f;
    for (int e = tid; e < a.C * ICP; e += WGTH_THREADS) Ws[e] = a.W[(size_t)(e / ICP) * a.ldw + (e % ICP)];
    __syncthreads();
    for (int tix = (int)blockIdx.x; tix < a.tiles; tix += (int)gridDim.x) {
        const int b = tix / tpb, t0 = (tix - b * tpb) * WGTH_TB, t = t0 + tl;
        const bool live = t < g.T;
        // ---- phase 1: wave wv takes channel groups wv, wv + 4, ...: LDS tile + its share of dxa ----
        float ax[ICP];
#pragma unroll
        for (int j = 0; j < ICP; ++j) ax[j] = 0.f;
        for (int cg0 = wv; cg0 < ncg; cg0 += 4 * WGTH_BATCH) {   // all of a batch's loads are issued before the first is used
            u32x4 h[WGTH_BATCH], l[WGTH_BATCH];
#pragma unroll
            for (int q = 0; q < WGTH_BATCH; ++q) {
                const int cg = cg0 + 4 * q;
                h[q] = u32x4{0u, 0u, 0u, 0u}; l[q] = h[q];
                if (live && cg < ncg && !(WGTH_DBG & 8)) {
                    const size_t i = s_index(a.dh, g, b, cg * 8, t);
                    h[q] = __builtin_nontemporal_load(reinterpret_cast<const u32x4 *>(a.dh.hi + i));
                    l[q] = __builtin_nontemporal_load(reinterpret_cast<const u32x4 *>(a.dh.hi + a.dh.lo_off + i));
                }
            }
#pragma unroll
            for (int q = 0; q < WGTH_BATCH; ++q) {
                const int cg = cg0 + 4 * q;
                if (cg < ncg && !(WGTH_DBG & 4)) {
                    float v[8];
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        v[2 * e] = __uint_as_float(h[q][e] << 16) + __uint_as_float(l[q][e] << 16);
                        v[2 * e + 1] = __uint_as_float(h[q][e] & 0xffff0000u) + __uint_as_float(l[q][e] & 0xffff0000u);
                    }
                    const float *wrow = Ws + cg * 8 * ICP;
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        tile[(cg * 8 + e) * WGTH_LDT + tl] = v[e];
#pragma unroll
                        for (int j = 0; j < ICP; ++j) ax[j] += wrow[e * ICP + j] * v[e];
                    }
                }
            }
        }
#pragma unroll
        for (int j = 0; j < ICP; ++j) axs[(wv * ICP + j) * 64 + tl] = ax[j];
        for (int e = tid; e < WGTH_TB * ICP; e += WGTH_THREADS) {        // xa tile: e -> (j, time)
            const int j = e >> 6, tt = e & 63;
            xs[tt * ICP + j] = (j < a.ic && t0 + tt < g.T) ? *paddr(a.X, g, b, j, t0 + tt) : 0.f;
        }
        __syncthreads();
        for (int e = tid; e < WGTH_TB * ICP; e += WGTH_THREADS) {        // dxa: the four shares, then into the gradient plane
            const int j = e >> 6, tt = e & 63;
            if (j < a.ic && t0 + tt < g.T) {
                float *p = paddr(a.dX, g, b, j, t0 + tt);
                *p += (axs[(0 * ICP + j) * 64 + tt] + axs[(1 * ICP + j) * 64 + tt]) + (axs[(2 * ICP + j) * 64 + tt] + axs[(3 * ICP + j) * 64 + tt]);
            }
        }
        // ---- phase 2: thread c walks the 64 time steps of its channel ----
#pragma unroll
        for (int r = 0; r < WGTH_MAXROWS; ++r) {
            const int c = tid + r * WGTH_THREADS;
            if (c < a.C && !(WGTH_DBG & 2)) {
                const float *row = tile + (size_t)c * WGTH_LDT;
#pragma unroll 8
                for (int tt = 0; tt < WGTH_TB; ++tt) {
                    const float d = row[tt];
#pragma unroll
                    for (int j = 0; j < ICP; ++j) acc[r][j] += d * xs[tt * ICP + j];
                }
            }
        }
        __syncthreads();
    }
    const int n = a.C * ICP;
    float *mine = a.part + (size_t)blockIdx.x * n;
#pragma unroll
    for (int r = 0; r < WGTH_MAXROWS; ++r) {
        const int c = tid + r * WGTH_THREADS;
        if (c < a.C)
#pragma unroll
            for (int j = 0; j < ICP; ++j) mine[c * ICP + j] = acc[r][j];
    }
}

// ------------------------------------------------------------------------------------------------
// end:  dW[k][m] = sum_{b,t} G[k][t] skip[m][t]        dS[m][t] = sum_k W[k][m] G[k][t]  -> S-plane       (model/waveglow.py:104, backward)
// ------------------------------------------------------------------------------------------------
struct ThinEndArgs {
    PRef G;              // gradient of the WN output: K2 = 2 ic rows
    PRef skip;           // Cs channels, fp32
    SRef dS;             // S-plane out, Cs channels
    const float *W;      // fp32 k-major [K2][ldw]: W[k][m]
    int ldw, Cs, K2;
    Geo g;
    int tiles;
    float *part;         // [gridDim.x][K2P * Cs]
};
template <int K2P>
__global__ __launch_bounds__(WGTH_THREADS) void thin_end_kernel(const ThinEndArgs a)
{
    extern __shared__ __attribute__((aligned(16))) float wgth_smem[];
    float *tile = wgth_smem;                                   // [Cs][65]
    float *gs = tile + (size_t)a.Cs * WGTH_LDT;                // [64][K2P]
    float *Ws = gs + WGTH_TB * K2P;                            // [K2P][Cs]: the weights (every read is wave-uniform: an LDS broadcast)
    const int tid = threadIdx.x, tl = tid & 63, lane = tl;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const Geo g = a.g;
    const int ncg = a.Cs >> 3, tpb = g.Tt / WGTH_TB;
    float acc[WGTH_MAXROWS][K2P];
#pragma unroll
    for (int r = 0; r < WGTH_MAXROWS; ++r)
#pragma unroll
        for (int k = 0; k < K2P; ++k) acc[r][k] = 0.f;
    for (int e = tid; e < K2P * a.Cs; e += WGTH_THREADS) {
        const int k = e / a.Cs, m = e - k * a.Cs;
        Ws[e] = k < a.K2 ? a.W[(size_t)k * a.ldw + m] : 0.f;
    }
    __syncthreads();
    for (int tix = (int)blockIdx.x; tix < a.tiles; tix += (int)gridDim.x) {
        const int b = tix / tpb, t0 = (tix - b * tpb) * WGTH_TB, t = t0 + tl;
        const bool live = t < g.T;
        // ---- G column of this lane (requested first), then the skip tile into LDS: a wave load covers 4 channel rows x 64 time steps ----
        float gk[K2P];
#pragma unroll
        for (int k = 0; k < K2P; ++k) gk[k] = (live && k < a.K2) ? *paddr(a.G, g, b, k, t) : 0.f;
        {
            const int r4 = lane >> 4, q4 = (lane & 15) * 4;
            for (int mb = wv * 4; mb < a.Cs; mb += 16 * WGTH_BATCH2) {   // all of a batch's loads are issued before the first is used
                f32x4 v[WGTH_BATCH2];
#pragma unroll
                for (int q = 0; q < WGTH_BATCH2; ++q) {
                    const int m = mb + 16 * q + r4;
                    v[q] = f32x4{0.f, 0.f, 0.f, 0.f};
                    if (m < a.Cs && !(WGTH_DBG & 8)) v[q] = __builtin_nontemporal_load(reinterpret_cast<const f32x4 *>(paddr(a.skip, g, b, m, t0 + q4)));
                }
#pragma unroll
                for (int q = 0; q < WGTH_BATCH2; ++q) {
                    const int m = mb + 16 * q + r4;
                    if (m < a.Cs && !(WGTH_DBG & 4)) {
                        float *dst = tile + (size_t)m * WGTH_LDT + q4;
                        dst[0] = t0 + q4 + 0 < g.T ? v[q][0] : 0.f; dst[1] = t0 + q4 + 1 < g.T ? v[q][1] : 0.f;
                        dst[2] = t0 + q4 + 2 < g.T ? v[q][2] : 0.f; dst[3] = t0 + q4 + 3 < g.T ? v[q][3] : 0.f;
                    }
                }
            }
        }
        // ---- the data gradient: wave wv takes channel groups wv, wv + 4, ... ----
        if (wv == 0) {
#pragma unroll
            for (int k = 0; k < K2P; ++k) gs[tl * K2P + k] = gk[k];
        }
        for (int cg = wv; cg < ncg && !(WGTH_DBG & 4); cg += 4) {
            float v[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = 0.f;
#pragma unroll
            for (int k = 0; k < K2P; ++k) {
                const float *wk = Ws + k * a.Cs + cg * 8;
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] += wk[e] * gk[k];
            }
            if (live) {
                u32x4 h, l;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    unsigned hh, ll;
                    split2(v[2 * e], v[2 * e + 1], hh, ll);
                    h[e] = hh; l[e] = ll;
                }
                const size_t i = s_index(a.dS, g, b, cg * 8, t);
                *reinterpret_cast<u32x4 *>(a.dS.hi + i) = h;
                *reinterpret_cast<u32x4 *>(a.dS.hi + a.dS.lo_off + i) = l;
            }
        }
        __syncthreads();
        // ---- the weight gradient: thread m walks the 64 time steps of its skip channel ----
#pragma unroll
        for (int r = 0; r < WGTH_MAXROWS; ++r) {
            const int m = tid + r * WGTH_THREADS;
            if (m < a.Cs && !(WGTH_DBG & 2)) {
                const float *row = tile + (size_t)m * WGTH_LDT;
#pragma unroll 8
                for (int tt = 0; tt < WGTH_TB; ++tt) {
                    const float s = row[tt];
#pragma unroll
                    for (int k = 0; k < K2P; ++k) acc[r][k] += s * gs[tt * K2P + k];
                }
            }
        }
        __syncthreads();
    }
    const int n = K2P * a.Cs;
    float *mine = a.part + (size_t)blockIdx.x * n;
#pragma unroll
    for (int r = 0; r < WGTH_MAXROWS; ++r) {
        const int m = tid + r * WGTH_THREADS;
        if (m < a.Cs)
#pragma unroll
            for (int k = 0; k < K2P; ++k) mine[(size_t)k * a.Cs + m] = acc[r][k];
    }
}

// ------------------------------------------------------------------------------------------------
// WN.start forward (model/waveglow.py:99; WN2D.start, waveflow.py:119): h_0 = W_start xa with at most 16 input channels.  As an MFMA conv it was
// TWO launches per WN pass -- xa converted to an S-plane (its K padded from 2-4 channels to 16), then a 128-row-tile product of which
// a fraction of one chunk is real work -- in front of every flow's layer chain (forward, recompute, synthesis).  Here a thread owns one
// time step and 8 output channels: fp32 FMAs on the input values themselves (at least as exact as the split product), the fp32 plane
// (where the chain still has one) and the S-plane unit written straight from registers.
// ------------------------------------------------------------------------------------------------
struct StartFwdArgs {
    PRef X;              // xa = X channels [ch0, ch0 + ic)
    const float *W;      // fp32 effective weights [C][ldw]: W[c][j]
    int ldw, C, ic;
    PRef H;              // fp32 output plane (p == nullptr: none)
    SRef HS;             // S-plane output
    SRef XS;             // hi != nullptr: xa itself as an S-plane of XS.Cp channels (zero beyond ic) -- the first layer's conv reads it when
                         // WN.start is folded into its weight (wgflow.hip start_fold_on); written by the blocks of the first Cp / 8 channel groups
    Geo g;
    int row_sel1;        // Geo::rows > 0: r + 1 = blockIdx.z is the item and the launch covers its height row r
};
__global__ __launch_bounds__(256) void start_fwd_kernel(const StartFwdArgs a)
{
    __shared__ float w[8][16];
    const Geo g = a.g;
    const int tid = threadIdx.x, t = blockIdx.x * 256 + tid, cg = blockIdx.y;
    const int b = a.row_sel1 ? (int)blockIdx.z * g.rows + a.row_sel1 - 1 : (int)blockIdx.z;
    if (tid < 8 * a.ic) {
        const int e = tid / a.ic, j = tid - e * a.ic;
        w[e][j] = cg * 8 + e < a.C ? a.W[(size_t)(cg * 8 + e) * a.ldw + j] : 0.f;
    }
    __syncthreads();
    if (t >= g.T) return;
    float o[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    float xk[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};     // channels 8 cg .. 8 cg + 7 of xa (zero beyond ic)
    for (int j = 0; j < a.ic; ++j) {
        const float xv = *paddr(a.X, g, b, j, t);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            o[e] = fmaf(w[e][j], xv, o[e]);
            if (j == cg * 8 + e) xk[e] = xv;
        }
    }
    if (a.XS.hi && cg * 8 < a.XS.Cp) {
        u32x4 xh, xl;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            unsigned hh, ll;
            split2(xk[2 * e], xk[2 * e + 1], hh, ll);
            xh[e] = hh; xl[e] = ll;
        }
        const size_t ix = s_index(a.XS, g, b, cg * 8, t);
        *reinterpret_cast<u32x4 *>(a.XS.hi + ix) = xh;
        *reinterpret_cast<u32x4 *>(a.XS.hi + a.XS.lo_off + ix) = xl;
    }
    if (a.H.p) {
#pragma unroll
        for (int e = 0; e < 8; ++e)
            if (cg * 8 + e < a.C) *paddr(a.H, g, b, cg * 8 + e, t) = o[e];
    }
    u32x4 h, l;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        unsigned hh, ll;
        split2(o[2 * e], o[2 * e + 1], hh, ll);
        h[e] = hh; l[e] = ll;
    }
    const size_t i = s_index(a.HS, g, b, cg * 8, t);
    *reinterpret_cast<u32x4 *>(a.HS.hi + i) = h;
    *reinterpret_cast<u32x4 *>(a.HS.hi + a.HS.lo_off + i) = l;
}


// ------------------------------------------------------------------------------------------------
// The rank-2ic form of the skip path's gradients (wg_small.h, weff_kernel): with G = (d/d log_s, d/d t) the seed of WN's backward
// (efficient_modules.py:139-144) and dS = W_end^T G,
//     dWskip_l = sum_{b,t} dS (x) gate_l = W_end^T P_l ,    dW_end = sum_{b,t} G (x) S = sum_l P_l Wskip_l^T ,    P_l = sum_{b,t} G (x) gate_l
// so ONE pass over the layers' gate planes (what the skip sum of the recompute pass read, without writing a Cs-row plane, and without
// the 2 x Cs rows dS costs every layer's weight-gradient and gate-backward product) yields every gradient that went through S.
// pgate_kernel: a wave owns 8 gate channels (one 16-byte unit row of a layer's S-plane) and a range of 64-column blocks; lane = time step:
// coalesced unit loads (1 KB per wave instruction), G's rows of that block (coalesced fp32 rows, L2-resident: G is 2 ic x B T), 64 sums
// per lane; at the end a butterfly over the lanes and ONE partial per (range, unit row).  thin_fold_kernel adds the ranges in a fixed order.
// ------------------------------------------------------------------------------------------------
struct PGateArgs {
    const unsigned short *gS[16];  // hi array of layer l's gate S-plane (Cd channels per item)
    size_t g_lo_off;
    PRef G;                        // fp32 plane, rows [0, ic2)
    int Cd, nl, ic2, mrows;        // mrows = ic2 rounded up to 8: rows of P per layer
    Geo g;
    int nblk, per;                 // 64-column blocks in all (B * Tt / 64), per range
    float *part;                   // [gridDim.y][nl][mrows][Cd]
};
// (three workgroups per CU -- __launch_bounds__(256, 3), 168 registers, 12 column ranges -- measured slower: 110.5 against 105-107 us)
__global__ __launch_bounds__(256) void pgate_kernel(const PGateArgs a)
{
    const Geo g = a.g;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int ug = (int)blockIdx.x * 4 + wave;                  // unit row over all layers
    const int upl = a.Cd >> 3, l = ug / upl, cg = ug - l * upl;
    if (l >= a.nl) return;
    const int m0 = (int)blockIdx.z * 8;
    const int tb0 = (int)blockIdx.y * a.per, tb1 = min(a.nblk, tb0 + a.per), bpi = g.Tt / 64;
    const unsigned short *base = a.gS[l] + (size_t)cg * g.P * 8;
    const size_t item = (size_t)upl * g.P * 8;
    float acc[8][8];
#pragma unroll
    for (int m = 0; m < 8; ++m)
#pragma unroll
        for (int e = 0; e < 8; ++e) acc[m][e] = 0.f;
    constexpr int DEPTH = 4;                                     // 64-column blocks in flight per wave
    for (int tb = tb0; tb < tb1; tb += DEPTH) {
        u32x4 vh[DEPTH], vl[DEPTH];
        float gm[DEPTH][8];
        bool ok[DEPTH];
#pragma unroll
        for (int d = 0; d < DEPTH; ++d) {
            const int tbb = min(tb + d, tb1 - 1), b = tbb / bpi, t = (tbb - b * bpi) * 64 + lane;
            ok[d] = tb + d < tb1 && t < g.T;
            const unsigned short *q = base + (size_t)b * item + (size_t)(g.H + t) * 8;
            vh[d] = *reinterpret_cast<const u32x4 *>(q);
            vl[d] = *reinterpret_cast<const u32x4 *>(q + a.g_lo_off);
#pragma unroll
            for (int m = 0; m < 8; ++m) gm[d][m] = (m0 + m < a.ic2) ? *paddr(a.G, g, b, m0 + m, t) : 0.f;
        }
#pragma unroll
        for (int d = 0; d < DEPTH; ++d) {
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const unsigned wh = vh[d][e >> 1], wl = vl[d][e >> 1];
                float x = (e & 1) ? __uint_as_float(wh & 0xffff0000u) + __uint_as_float(wl & 0xffff0000u)
                                  : __uint_as_float(wh << 16) + __uint_as_float(wl << 16);
                x = ok[d] ? x : 0.f;                             // (columns beyond T hold whatever the last pass left: never multiplied)
#pragma unroll
                for (int m = 0; m < 8; ++m) acc[m][e] = fmaf(ok[d] ? gm[d][m] : 0.f, x, acc[m][e]);
            }
        }
    }
#pragma unroll
    for (int m = 0; m < 8; ++m)
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            float s = acc[m][e];
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
            acc[m][e] = s;
        }
    // lane m writes row m0 + m of this unit row: 8 consecutive floats
    float *out = a.part + (((size_t)blockIdx.y * a.nl + l) * a.mrows + m0) * a.Cd + cg * 8;
#pragma unroll
    for (int m = 0; m < 8; ++m)
        if (lane == m) {
            f32x4 lo4 = {acc[m][0], acc[m][1], acc[m][2], acc[m][3]}, hi4 = {acc[m][4], acc[m][5], acc[m][6], acc[m][7]};
            *reinterpret_cast<f32x4 *>(out + (size_t)m * a.Cd) = lo4;
            *reinterpret_cast<f32x4 *>(out + (size_t)m * a.Cd + 4) = hi4;
        }
}

// from P ([nl][mrows][Cd], folded): dWskip_l = W_end^T P_l as [Cs][Cd] matrices (the effective-weight gradient run_finalize takes), and
// dW_end = sum_l P_l Wskip_l^T as [32][Cs] (Wskip_l = scale (x) v: the effective skip rows, as the pack jobs form them)
struct LrFinArgs {
    const float *P, *wE;           // wE: end.weight [ic2][Cs]
    const float *v[16], *scale[16];   // per layer: W_o.weight_v at its first skip row [Cs][Cd], g / |v| of those rows
    float *dWsk, *dWend;           // [nl][Cs][Cd] ; [32][Cs]
    int nl, Cs, Cd, ic2, mrows;
};
__global__ __launch_bounds__(256) void lr_dwsk_kernel(const LrFinArgs a)
{
    const int l = blockIdx.y, e = blockIdx.x * 256 + threadIdx.x;
    if (e >= a.Cs * a.Cd) return;
    const int s = e / a.Cd, k = e - s * a.Cd;
    const float *Pl = a.P + (size_t)l * a.mrows * a.Cd;
    float acc = 0.f;
    for (int m = 0; m < a.ic2; ++m) acc = fmaf(a.wE[(size_t)m * a.Cs + s], Pl[(size_t)m * a.Cd + k], acc);
    a.dWsk[((size_t)l * a.Cs + s) * a.Cd + k] = acc;
}
// one block per skip row s: thread k-slices, 32 sums per thread, a fixed tree over the block
__global__ __launch_bounds__(256) void lr_dwend_kernel(const LrFinArgs a)
{
    __shared__ float red[32][4];
    const int s = blockIdx.x, tid = threadIdx.x;
    float acc[32];
#pragma unroll
    for (int m = 0; m < 32; ++m) acc[m] = 0.f;
    for (int l = 0; l < a.nl; ++l) {
        const float sc = a.scale[l][s];
        const float *vr = a.v[l] + (size_t)s * a.Cd, *Pl = a.P + (size_t)l * a.mrows * a.Cd;
        for (int k = tid; k < a.Cd; k += 256) {
            const float w = sc * vr[k];
#pragma unroll
            for (int m = 0; m < 32; ++m)
                if (m < a.ic2) acc[m] = fmaf(Pl[(size_t)m * a.Cd + k], w, acc[m]);
        }
    }
    // a butterfly over the lanes, then the four waves' sums in wave order: a fixed order
#pragma unroll
    for (int m = 0; m < 32; ++m)
        if (m < a.ic2) {
            float x = acc[m];
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) x += __shfl_xor(x, o, 64);
            if ((tid & 63) == 0) red[m][tid >> 6] = x;
        }
    __syncthreads();
    if (tid < 32) a.dWend[(size_t)tid * a.Cs + s] = tid < a.ic2 ? (red[tid][0] + red[tid][1]) + (red[tid][2] + red[tid][3]) : 0.f;
}
