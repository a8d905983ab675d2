// WaveFlow (model/waveflow.py) pieces that are not the WN2D stack: squeeze / unsqueeze in the [height][time] layout, the dense
// transposed-conv upsampler with LeakyReLU, the autoregressive affine coupling along the height axis fused with WN2D.end, and
// their backward.  All HBM-bound glue; the WN2D convolutions run on the conv / wgrad kernels of wg_gemm16s.h with 3x3 taps
// expressed as 9 shifted K segments over "one height row per plane row" planes (Geo::rows = n_group).
//
// Plane row of (item b, height h) = b * H + h.  X planes have ONE channel (PRef::Cp = 1).
#pragma once
#include <hip/hip_runtime.h>

// x[b][h][t] = audio[b][t * H + h]                                             (waveflow.py:186)
__global__ void wf_squeeze_kernel(const float *__restrict__ audio, PRef X, Geo g, int N)
{
    const int t = blockIdx.x * blockDim.x + threadIdx.x, row = blockIdx.y;
    if (t >= g.T) return;
    const int H = g.rows, b = row / H, h = row - b * H;
    *paddr(X, g, row, 0, t) = audio[(size_t)b * N + (size_t)t * H + h];
}
__global__ void wf_unsqueeze_kernel(PRef X, Geo g, int N, float *__restrict__ audio)
{
    const int t = blockIdx.x * blockDim.x + threadIdx.x, row = blockIdx.y;
    if (t >= g.T) return;
    const int H = g.rows, b = row / H, h = row - b * H;
    audio[(size_t)b * N + (size_t)t * H + h] = *paddr(X, g, row, 0, t);
}
// row `r` of every item from / to the [B][N] layout (the row-by-row inverse)
__global__ void wf_copy_row_kernel(PRef src, PRef dst, Geo g, int r_src, int r_dst)
{
    const int t = blockIdx.x * blockDim.x + threadIdx.x, b = blockIdx.y;
    if (t >= g.T) return;
    *paddr(dst, g, b * g.rows + r_dst, 0, t) = *paddr(src, g, b * g.rows + r_src, 0, t);
}
// z = z.flip(2) along the height axis (waveflow.py:222)
__global__ void wf_flip_kernel(PRef src, PRef dst, Geo g)
{
    const int t = blockIdx.x * blockDim.x + threadIdx.x, row = blockIdx.y;
    if (t >= g.T) return;
    const int H = g.rows, b = row / H, h = row - b * H;
    *paddr(dst, g, b * H + (H - 1 - h), 0, t) = *paddr(src, g, row, 0, t);
}

// ------------------------------------------------------------------------------------------------
// upsampler (waveflow.py:163-169): y = LeakyReLU_0.4(bias + ConvTranspose1d(ReplicationPad1d((0,1))(mel)))[:, :, :W]
//   y[o][j] = bias[o] + sum_c sum_i melpad[c][i] * w[c][o][j + pad - s i],  w[c] = g[c] v[c] / ||v[c]||  (weight norm over dim 0 =
//   the INPUT channel of a transposed conv).  scale[c] = g[c] / ||v[c]|| comes from rownorm_kernel.
// ------------------------------------------------------------------------------------------------
struct WfUpArgs {
    const float *mel, *v, *scale, *bias;
    int M, F, K, s, pad;
    PRef Y;             // [items][Mp][P]
    Geo gi;             // per-item geometry (rows = 0)
};
__global__ void wf_upsample_fwd_kernel(const WfUpArgs a)
{
    const int j = blockIdx.x * blockDim.x + threadIdx.x, o = blockIdx.y, b = blockIdx.z;
    if (j >= a.gi.T) return;
    float acc = a.bias[o];
    const int ihi = min((j + a.pad) / a.s, a.F);                       // padded length F + 1
    int ilo = j + a.pad - a.K + 1;
    ilo = ilo <= 0 ? 0 : (ilo + a.s - 1) / a.s;
    const float *mb = a.mel + (size_t)b * a.M * a.F;
    for (int i = ilo; i <= ihi; ++i) {
        const int k = j + a.pad - a.s * i, isrc = min(i, a.F - 1);
        for (int c = 0; c < a.M; ++c) acc = fmaf(mb[(size_t)c * a.F + isrc] * a.scale[c], a.v[((size_t)c * a.M + o) * a.K + k], acc);
    }
    *paddr(a.Y, a.gi, b, o, j) = acc > 0.f ? acc : 0.4f * acc;
}

// gp[b][o][j] = (sum over the H rows of dYrow[b*H + h][o][j]) * LeakyReLU'(y): the conditioning is broadcast over the height axis
__global__ void wf_rowsum_leaky_kernel(PRef dYrow, Geo g, PRef Y, Geo gi, int M, float *__restrict__ gp)
{
    const int j = blockIdx.x * blockDim.x + threadIdx.x, o = blockIdx.y, b = blockIdx.z;
    if (j >= g.T) return;
    float acc = 0.f;
    for (int h = 0; h < g.rows; ++h) acc += *paddr(dYrow, g, b * g.rows + h, o, j);
    const float y = *paddr(Y, gi, b, o, j);
    gp[((size_t)b * M + o) * g.T + j] = acc * (y > 0.f ? 1.f : 0.4f);
}

// effective-weight gradient dw[c][o*K + k] = sum_b sum_i melpad[b][c][i] gp[b][o][s i + k - pad]  (one block per input channel c),
// dbias[o] = sum gp (block c == 0), dmel (optional, block-per-c as well).
struct WfUpBwdArgs {
    const float *mel, *v, *scale, *gp;
    int B, M, F, K, s, pad, W;
    float *dw, *dbias, *dmel;
};
__global__ __launch_bounds__(256) void wf_upsample_bwd_kernel(const WfUpBwdArgs a)
{
    // grid (M, S): block (c, y) takes every S-th 256-element slice of channel c's work (80 blocks of 256 threads walking 720 weight
    // gradients x batch x frames one after the other took 0.9 ms of a WaveFlow step: a chain of load latencies on a third of the CUs)
    const int c = blockIdx.x, tid = threadIdx.x + 256 * blockIdx.y, nth = 256 * gridDim.y;
    for (int e = tid; e < a.M * a.K; e += nth) {
        const int o = e / a.K, k = e - o * a.K;
        float acc = 0.f;
        for (int b = 0; b < a.B; ++b) {
            const float *mb = a.mel + ((size_t)b * a.M + c) * a.F, *gb = a.gp + ((size_t)b * a.M + o) * a.W;
            // eight frames at a time, their sixteen loads issued together (one load pair per fma, each behind the other, was a chain of
            // B * (F + 1) round trips: 466 us per step at the shipped shape).  A term outside the row is loaded from a clamped, valid
            // address and then NOT added (a select on the product's sum, not a zero factor: 0 * Inf would be NaN): the sum keeps its order.
            for (int i0 = 0; i0 <= a.F; i0 += 8) {
                float mv[8], gv[8];
                bool ok[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const int i = i0 + u, j = a.s * i + k - a.pad;
                    ok[u] = i <= a.F && j >= 0 && j < a.W;
                    mv[u] = mb[min(i, a.F - 1)];
                    gv[u] = gb[min(max(j, 0), a.W - 1)];
                }
#pragma unroll
                for (int u = 0; u < 8; ++u) acc = ok[u] ? fmaf(mv[u], gv[u], acc) : acc;
            }
        }
        a.dw[(size_t)c * a.M * a.K + e] = acc;
    }
    if (a.dbias && blockIdx.y == 0) {
        // dbias[c]: block (c, 0) sums its own channel -- every thread a strided share of the B * W terms, then a fixed tree over the block
        // (80 threads of block 0 walking 3000 terms each, one load behind the other, was 300 us).  The tree below is written for the
        // launch shape wf_upsample_bwd uses: blockDim.x == 256, gridDim.x == M (one block column per channel).
        __shared__ float red[256];
        float acc = 0.f;
        for (int b = 0; b < a.B; ++b)
            for (int j = threadIdx.x; j < a.W; j += 256) acc += a.gp[((size_t)b * a.M + c) * a.W + j];
        red[threadIdx.x] = acc;
        __syncthreads();
        for (int q = 128; q > 0; q >>= 1) {
            if ((int)threadIdx.x < q) red[threadIdx.x] += red[threadIdx.x + q];
            __syncthreads();
        }
        if (threadIdx.x == 0) a.dbias[c] = red[0];
    }
    if (a.dmel)
        for (int e = tid; e < a.B * a.F; e += nth) {
            const int b = e / a.F, i = e - b * a.F;
            float acc = 0.f;
            for (int ii = i; ii <= (i == a.F - 1 ? a.F : i); ++ii)         // the last frame also feeds the replicated one
                for (int o = 0; o < a.M; ++o) {
                    const float *gb = a.gp + ((size_t)b * a.M + o) * a.W;
                    const float *wv = a.v + ((size_t)c * a.M + o) * a.K;
                    for (int k = 0; k < a.K; ++k) {
                        const int j = a.s * ii + k - a.pad;
                        if (j >= 0 && j < a.W) acc = fmaf(wv[k], gb[j], acc);
                    }
                }
            a.dmel[((size_t)b * a.M + c) * a.F + i] = acc * a.scale[c];
        }
}

// ------------------------------------------------------------------------------------------------
// WN2D.end (plain 1x1 conv Cs -> 2, waveflow.py:117,135) fused with the coupling (waveflow.py:196-206):
//   (log_s, t)[r] = W_end S[r] ;  xout[r] = x[r+1] exp(log_s[r]) + t[r] ;  x_next = cat(flip(xout), x[0]) ;  logdet += sum log_s
// grid = (plane rows); the block of WN row r = 0..H-2 of an item writes x_next row H-2-r, the block of row H-1 copies x[0].
// mode 0: forward.  mode 1: backward seed (see below).  mode 2: one row of the inverse: x[r+1] = (z[r+1] - t[r]) / exp(log_s[r]).
// ------------------------------------------------------------------------------------------------
struct WfCoupleArgs {
    const float *endw;      // [2][Cs]
    const float *endb;      // [2]: WN2D(bias=True)'s end.bias, or null
    PRef S;                 // cumulated skip [rows][Cs][P]
    int Cs;
    PRef X, Xn;             // current / next flow state (1 channel)
    PRef dXn, dX;           // mode 1: gradient w.r.t. x_next (in), w.r.t. x (out: the coupling part; WN2D's start conv adds its part later)
    PRef G;                 // mode 1: [rows][Gc][P], channel 0 = d log_s, channel 1 = d t (zero for row H-1)
    const float *dld;       // mode 1: d loss / d logdet [items]
    float *rowsum;          // modes 0, 2: sum of log_s per plane row (mode 2: of -log_s)
    int row_sel;            // mode 2: the WN row being produced (grid.x = items)
    Geo g;
    int mode;
    int noflip;             // use_conv1x1: x_next = cat(x[0], xout) (the 1x1 over the height axis follows) instead of cat(flip(xout), x[0])
    float *raw_ls, *raw_t;  // mode 3 (WN2D.forward on its own, waveflow.py:128-135): plain [items][raw_rows][T] outputs of WN2D.end, no coupling
    int raw_rows;
    // the rank-2 form of the skip path (csrc/wgflow.hip lowrank_on): S does not exist; (log_s, t) of an element = the sum of `nsrc` partial rows
    // of two floats the gate convs left (wg_gemm16q.h wgq_gate_nb), [src][plane row][Tt][2], added in source order
    const float *part;
    int nsrc;
};
// (log_s, t) of (plane row, t) without their biases: W_end . S, or the partial rows
__device__ __forceinline__ void wf_end_out(const WfCoupleArgs &a, const Geo &g, int row, int t, float &ls, float &tt)
{
    if (a.part) {
        const size_t stride = (size_t)g.B * g.Tt * 2;
        const float *q = a.part + ((size_t)row * g.Tt + t) * 2;
        int s = 0;
        for (; s + 8 <= a.nsrc; s += 8) {
            f32x2_t v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = *reinterpret_cast<const f32x2_t *>(q + (size_t)(s + u) * stride);
#pragma unroll
            for (int u = 0; u < 8; ++u) { ls += v[u][0]; tt += v[u][1]; }
        }
        for (; s < a.nsrc; ++s) { const f32x2_t v = *reinterpret_cast<const f32x2_t *>(q + (size_t)s * stride); ls += v[0]; tt += v[1]; }
        return;
    }
    const float *sp = paddr(a.S, g, row, 0, t);
    int c = 0;
    for (; c + 16 <= a.Cs; c += 16) {                          // sixteen channel loads in flight (the same order of fmas as one at a time)
        float sv[16];
#pragma unroll
        for (int u = 0; u < 16; ++u) sv[u] = sp[(size_t)(c + u) * g.P];
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            ls = fmaf(a.endw[c + u], sv[u], ls);
            tt = fmaf(a.endw[a.Cs + c + u], sv[u], tt);
        }
    }
    for (; c < a.Cs; ++c) {
        const float s = sp[(size_t)c * g.P];
        ls = fmaf(a.endw[c], s, ls);
        tt = fmaf(a.endw[a.Cs + c], s, tt);
    }
}
// (a device function: the stage interpreter, wg_stage.h, runs it as one stage; NT threads, `blk` = the block index, row_sel overrides a.row_sel)
template <int NT>
__device__ __forceinline__ void wf_couple_body(const WfCoupleArgs &a, int blk, int row_sel, float *red)
{
    const Geo g = a.g;
    const int H = g.rows, tid = threadIdx.x;
    const int row = a.mode == 2 ? blk * H + row_sel : blk;
    const int b = row / H, r = row - b * H;
    const int x0row = b * H + (a.noflip ? 0 : H - 1);          // where x[0] goes in x_next
    const int orow = b * H + (a.noflip ? r + 1 : H - 2 - r);   // where xout[r] goes
    float lsum = 0.f;
    if (a.mode == 3) {                                         // (log_s, t) of the row as they are
        if (r >= a.raw_rows) return;
        for (int t = tid; t < g.T; t += NT) {
            float ls = a.endb ? a.endb[0] : 0.f, tt = a.endb ? a.endb[1] : 0.f;
            if (a.part) wf_end_out(a, g, row, t, ls, tt);
            else
            for (int c = 0; c < a.Cs; ++c) {
                const float s = *paddr(a.S, g, row, c, t);
                ls = fmaf(a.endw[c], s, ls);
                tt = fmaf(a.endw[a.Cs + c], s, tt);
            }
            const size_t o = ((size_t)b * a.raw_rows + r) * g.T + t;
            a.raw_ls[o] = ls;
            a.raw_t[o] = tt;
        }
        return;
    }
    if (r == H - 1) {                                          // not a WN output row: x_next[x0row] = x[0] and its gradient
        for (int t = tid; t < g.T; t += NT) {
            if (a.mode == 0) *paddr(a.Xn, g, x0row, 0, t) = *paddr(a.X, g, b * H, 0, t);
            if (a.mode == 1) {
                *paddr(a.dX, g, b * H, 0, t) = *paddr(a.dXn, g, x0row, 0, t);
                *paddr(a.G, g, row, 0, t) = 0.f;
                *paddr(a.G, g, row, 1, t) = 0.f;
            }
        }
        if (tid == 0 && a.rowsum && a.mode == 0) a.rowsum[row] = 0.f;
        return;
    }
    for (int t = tid; t < g.T; t += NT) {
        float ls = a.endb ? a.endb[0] : 0.f, tt = a.endb ? a.endb[1] : 0.f;
        wf_end_out(a, g, row, t, ls, tt);
        const float es = expf(ls);
        if (a.mode == 0) {
            *paddr(a.Xn, g, orow, 0, t) = fmaf(*paddr(a.X, g, row + 1, 0, t), es, tt);
            lsum += ls;
        } else if (a.mode == 1) {
            const float gout = *paddr(a.dXn, g, orow, 0, t);
            const float xv = *paddr(a.X, g, row + 1, 0, t);
            *paddr(a.dX, g, row + 1, 0, t) = gout * es;
            *paddr(a.G, g, row, 0, t) = gout * xv * es + a.dld[b];
            *paddr(a.G, g, row, 1, t) = gout;
        } else {
            *paddr(a.Xn, g, row + 1, 0, t) = (*paddr(a.X, g, row + 1, 0, t) - tt) / es;
            lsum -= ls;
        }
    }
    if (a.mode == 1 || !a.rowsum) return;
    red[tid] = lsum;
    __syncthreads();
    for (int q = NT / 2; q > 0; q >>= 1) {
        if (tid < q) red[tid] += red[tid + q];
        __syncthreads();
    }
    if (tid == 0) a.rowsum[row] = red[0];
}
__global__ __launch_bounds__(256) void wf_couple_kernel(const WfCoupleArgs a)
{
    __shared__ float red[256];
    wf_couple_body<256>(a, (int)blockIdx.x, a.row_sel, red);
}
// Mode 2 (one row of the inverse, model/waveflow.py:246-253) as its own launch: ONE block per item -- the sum of log_s over the row stays
// one fixed-order reduction -- of 1024 threads = 256 time steps x 4 quarters of the skip channels, so a thread has Cs / 4 independent loads
// in flight instead of walking all Cs channels one after the other (17.7 us per row step for a 0.7 s utterance, 9 % of the synthesis call).
__global__ __launch_bounds__(1024) void wf_couple_row_kernel(const WfCoupleArgs a)
{
    __shared__ float part[3][256][2];
    __shared__ float red[256];
    const Geo g = a.g;
    const int H = g.rows, tid = threadIdx.x, tq = tid & 255, cq = tid >> 8;
    const int row = (int)blockIdx.x * H + a.row_sel, b = row / H;
    const int c0 = cq * (a.Cs >> 2), c1 = cq == 3 ? a.Cs : c0 + (a.Cs >> 2);
    (void)b;
    float lsum = 0.f;
    for (int t0 = 0; t0 < g.T; t0 += 256) {                    // (every thread takes part in every barrier)
        const int t = t0 + tq;
        float ls = (!cq && a.endb) ? a.endb[0] : 0.f, tt = (!cq && a.endb) ? a.endb[1] : 0.f;
        if (t < g.T) {
#pragma unroll 8
            for (int c = c0; c < c1; ++c) {
                const float s = *paddr(a.S, g, row, c, t);
                ls = fmaf(a.endw[c], s, ls);
                tt = fmaf(a.endw[a.Cs + c], s, tt);
            }
        }
        if (cq) { part[cq - 1][tq][0] = ls; part[cq - 1][tq][1] = tt; }
        __syncthreads();
        if (!cq && t < g.T) {
#pragma unroll
            for (int q = 0; q < 3; ++q) { ls += part[q][tq][0]; tt += part[q][tq][1]; }
            *paddr(a.Xn, g, row + 1, 0, t) = (*paddr(a.X, g, row + 1, 0, t) - tt) / expf(ls);
            lsum -= ls;
        }
        __syncthreads();
    }
    if (!a.rowsum) return;
    if (!cq) red[tq] = lsum;
    __syncthreads();
    for (int q = 128; q > 0; q >>= 1) {
        if (tid < q) red[tid] += red[tid + q];
        __syncthreads();
    }
    if (tid == 0) a.rowsum[row] = red[0];
}
// exact-fp32 mode: out[item][c][t] = sum over the item's height rows of src[item * H + h][c][t] (the conditioning gradient: the conditioning is
// broadcast over the height axis); the S-plane modes use wf_rowsum_s_kernel
__global__ void wf_rowsum_kernel(PRef src, Geo g, PRef out, Geo gi, int nch)
{
    const int t = blockIdx.x * blockDim.x + threadIdx.x, c = blockIdx.y, item = blockIdx.z;
    if (t >= g.T || c >= nch) return;
    float s = 0.f;
    for (int h = 0; h < g.rows; ++h) s += *paddr(src, g, item * g.rows + h, c, t);
    *paddr(out, gi, item, c, t) = s;
}
// logdet[b] = sum over flows and rows of rowsum[k][b*H + h]  (+ coef * logdet W_k of the 1x1 convs: mix != NULL, waveflow.py:206 / :229)
// (fail != NULL and *fail != 0 -- a grid barrier of the row walk gave up, wg_stage.h --: logdet = NaN, so that a broken call is seen)
__global__ void wf_logdet_kernel(const float *__restrict__ rowsum, int nflow, int items, int H, float *__restrict__ logdet,
                                 const float *__restrict__ mix, int mix_stride, float coef, const int *__restrict__ fail = nullptr)
{
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= items) return;
    float s = 0.f;
    for (int k = 0; k < nflow; ++k) {
        float q = 0.f;
        for (int h = 0; h < H; ++h) q += rowsum[((size_t)k * items + b) * H + h];
        s += q;
        if (mix) s += coef * mix[(size_t)k * mix_stride + 2 * H * H];
    }
    if (fail && *fail) s = __builtin_nanf("");
    logdet[b] = s;
}

// ------------------------------------------------------------------------------------------------
// use_conv1x1: InvertibleConv1x1(n_group) over the HEIGHT axis (waveflow.py:179-181,203-206,224-229; efficient_modules.py:37-54).
// Per flow the packed buffer holds [W (H*H) | W^-1 (H*H) | logdet W (NaN if det < 0, as torch.logdet)].
// ------------------------------------------------------------------------------------------------
#define WF_MAXH 128
struct LuBigArgs {
    const float *W[WG_MAX_FLOWS];
    int n, c, ostride;
    float *out;
    unsigned char cs[WG_MAX_FLOWS];     // per-matrix size (0: c) -- WaveGlow's 1x1 weights shrink with the early outputs
    int inv_off, det_off;               // where W^-1 and logdet go inside a matrix's output record (0: c*c and 2*c*c)
};
// one workgroup per matrix: LU with partial pivoting in LDS (rows of c + 1 floats), log|det| and sign, then one thread per column
// of the inverse (forward / back substitution in place in the output)
__global__ __launch_bounds__(256) void lu_big_kernel(const LuBigArgs a)
{
    extern __shared__ float lsm[];
    const int c = a.cs[blockIdx.x] ? (int)a.cs[blockIdx.x] : a.c, ld = c + 1, tid = threadIdx.x;
    float *A = lsm;                                    // [c][c+1]
    int *perm = reinterpret_cast<int *>(lsm + c * ld); // [c]
    __shared__ float redv[256];
    __shared__ int redi[256];
    __shared__ float s_la;
    __shared__ int s_sg;
    const float *W = a.W[blockIdx.x];
    float *o = a.out + (size_t)blockIdx.x * a.ostride, *Wi = o + (a.inv_off ? a.inv_off : c * c);
    for (int e = tid; e < c * c; e += 256) { const float v = W[e]; o[e] = v; A[(e / c) * ld + e % c] = v; }
    for (int i = tid; i < c; i += 256) perm[i] = i;
    if (tid == 0) { s_la = 0.f; s_sg = 1; }
    __syncthreads();
    for (int q = 0; q < c; ++q) {
        float best = -1.f;
        int p = q;
        for (int r = q + tid; r < c; r += 256) { const float v = fabsf(A[r * ld + q]); if (v > best) { best = v; p = r; } }
        redv[tid] = best; redi[tid] = p;
        __syncthreads();
        for (int w = 128; w > 0; w >>= 1) {
            if (tid < w && (redv[tid + w] > redv[tid] || (redv[tid + w] == redv[tid] && redi[tid + w] < redi[tid]))) {
                redv[tid] = redv[tid + w]; redi[tid] = redi[tid + w];
            }
            __syncthreads();
        }
        p = redi[0];
        if (p != q)
            for (int j = tid; j < c; j += 256) { const float t = A[q * ld + j]; A[q * ld + j] = A[p * ld + j]; A[p * ld + j] = t; }
        __syncthreads();
        const float piv = A[q * ld + q];
        if (tid == 0) {
            if (p != q) { const int t = perm[q]; perm[q] = perm[p]; perm[p] = t; s_sg = -s_sg; }
            if (piv < 0.f) s_sg = -s_sg;
            s_la += logf(fabsf(piv));
        }
        for (int r = q + 1 + tid; r < c; r += 256) A[r * ld + q] /= piv;
        __syncthreads();
        const int m = c - q - 1;
        for (int e = tid; e < m * m; e += 256) {
            const int r = q + 1 + e / m, j = q + 1 + e % m;
            A[r * ld + j] = fmaf(-A[r * ld + q], A[q * ld + j], A[r * ld + j]);
        }
        __syncthreads();
    }
    if (tid == 0) o[a.det_off ? a.det_off : 2 * c * c] = s_sg > 0 ? s_la : __builtin_nanf("");
    for (int col = tid; col < c; col += 256) {
        for (int r = 0; r < c; ++r) {                        // L y = P e_col
            float s = perm[r] == col ? 1.f : 0.f;
            for (int j = 0; j < r; ++j) s = fmaf(-A[r * ld + j], Wi[j * c + col], s);
            Wi[r * c + col] = s;
        }
        for (int r = c - 1; r >= 0; --r) {                   // U x = y
            float s = Wi[r * c + col];
            for (int j = r + 1; j < c; ++j) s = fmaf(-A[r * ld + j], Wi[j * c + col], s);
            Wi[r * c + col] = s / A[r * ld + r];
        }
    }
}

// dst[b*H + o][t] = sum_h M[o][h] src[b*H + h][t]   (transpose: M[h][o]); grid (T / 256, H / 8, items), one thread per t and 8 output rows
#define WF_MIX_ROWS 8
__global__ __launch_bounds__(256) void wf_hmix_kernel(PRef src, PRef dst, Geo g, const float *__restrict__ M, int transpose)
{
    __shared__ float w[WF_MIX_ROWS][WF_MAXH];
    const int H = g.rows, o0 = blockIdx.y * WF_MIX_ROWS, b = blockIdx.z, t = blockIdx.x * 256 + threadIdx.x;
    for (int e = threadIdx.x; e < WF_MIX_ROWS * H; e += 256) {
        const int o = o0 + e / H, h = e % H;
        w[e / H][h] = o < H ? (transpose ? M[h * H + o] : M[o * H + h]) : 0.f;
    }
    __syncthreads();
    if (t >= g.T) return;
    float acc[WF_MIX_ROWS];
#pragma unroll
    for (int i = 0; i < WF_MIX_ROWS; ++i) acc[i] = 0.f;
    for (int h = 0; h < H; ++h) {
        const float x = *paddr(src, g, b * H + h, 0, t);
#pragma unroll
        for (int i = 0; i < WF_MIX_ROWS; ++i) acc[i] = fmaf(w[i][h], x, acc[i]);
    }
#pragma unroll
    for (int i = 0; i < WF_MIX_ROWS; ++i)
        if (o0 + i < H) *paddr(dst, g, b * H + o0 + i, 0, t) = acc[i];
}

// Gram partials of the 1x1 weight gradient: part[blk][o][h] = sum over this block's 64 columns of dz[o][t] u[h][t]
// (blk = item * tiles + time tile); both tiles go through LDS, a thread owns H*H / 256 outputs.
__global__ __launch_bounds__(256) void wf_hgram_kernel(PRef dZ, PRef U, Geo g, float *__restrict__ part)
{
    extern __shared__ float gsm[];                     // [2][H][65]
    const int H = g.rows, b = blockIdx.y, t0 = blockIdx.x * 64, tid = threadIdx.x;
    float *dz = gsm, *u = gsm + H * 65;
    for (int e = tid; e < H * 64; e += 256) {
        const int h = e >> 6, tl = e & 63;
        const bool ok = t0 + tl < g.T;
        dz[h * 65 + tl] = ok ? *paddr(dZ, g, b * H + h, 0, t0 + tl) : 0.f;
        u[h * 65 + tl] = ok ? *paddr(U, g, b * H + h, 0, t0 + tl) : 0.f;
    }
    __syncthreads();
    float *out = part + ((size_t)b * gridDim.x + blockIdx.x) * H * H;
    for (int e = tid; e < H * H; e += 256) {
        const int o = e / H, h = e % H;
        float acc = 0.f;
#pragma unroll 8
        for (int tl = 0; tl < 64; ++tl) acc = fmaf(dz[o * 65 + tl], u[h * 65 + tl], acc);
        out[e] = acc;
    }
}
// dW[o][h] = sum_blk part[blk][o][h] + W^-1[h][o] * T * sum_b dlogdet[b]       (efficient_modules.py:240-242)
__global__ __launch_bounds__(256) void wf_hgram_reduce_kernel(const float *__restrict__ part, int nblk, int H, const float *__restrict__ Winv,
                                                              const float *__restrict__ dld, int items, float Tn, float *__restrict__ dW)
{
    const int e = blockIdx.x * 256 + threadIdx.x;
    if (e >= H * H) return;
    float s0 = 0.f, s1 = 0.f;
    int k = 0;
    for (; k + 1 < nblk; k += 2) { s0 += part[(size_t)k * H * H + e]; s1 += part[(size_t)(k + 1) * H * H + e]; }
    if (k < nblk) s0 += part[(size_t)k * H * H + e];
    float gl = 0.f;
    for (int b = 0; b < items; ++b) gl += dld[b];
    const int o = e / H, h = e % H;
    dW[e] = (s0 + s1) + Winv[h * H + o] * gl * Tn;
}

// S-plane row sum over the height axis: out[b][c][t] = sum_h in[b*H + h][c][t] (hi + lo summed in fp32, re-split).  Used for the
// conditioning gradient, which is broadcast over the height axis: V^T (sum_h dxy[h]) instead of sum_h V^T dxy[h].
// Block = 64 time steps x 4 groups of height rows (256 threads): a thread sums its share of the rows with eight rows' loads in flight, the
// four partial sums meet in LDS.  (One thread per time step walking all 64 rows -- a chain of 64 dependent round trips, 188 blocks --
// read its 98 MB at 2.3 TB/s: 42 us per launch, 64 launches per WaveFlow step.)
__global__ __launch_bounds__(256) void wf_rowsum_s_kernel(SRef in, Geo g, SRef out, Geo gi)
{
    __shared__ float part[3][64][9];
    const int tl = threadIdx.x & 63, hq = threadIdx.x >> 6;
    const int t = blockIdx.x * 64 + tl, cg = blockIdx.y, b = blockIdx.z;
    const bool live = t < g.T;
    const int per = (g.rows + 3) >> 2, h0 = hq * per, h1 = min(g.rows, h0 + per);
    float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    for (int hb = h0; hb < h1; hb += 8) {
        u32x4 hi[8], lo[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            hi[q] = u32x4{0u, 0u, 0u, 0u}; lo[q] = u32x4{0u, 0u, 0u, 0u};
            if (live && hb + q < h1) {
                const size_t i = s_index(in, g, b * g.rows + hb + q, cg * 8, t);
                hi[q] = *reinterpret_cast<const u32x4 *>(in.hi + i);
                lo[q] = *reinterpret_cast<const u32x4 *>(in.hi + in.lo_off + i);
            }
        }
#pragma unroll
        for (int q = 0; q < 8; ++q)                            // (a fixed order: four partial sums of 16 rows each, then their sum)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                acc[2 * e] += __uint_as_float(hi[q][e] << 16) + __uint_as_float(lo[q][e] << 16);
                acc[2 * e + 1] += __uint_as_float(hi[q][e] & 0xffff0000u) + __uint_as_float(lo[q][e] & 0xffff0000u);
            }
    }
    if (hq) {
#pragma unroll
        for (int e = 0; e < 8; ++e) part[hq - 1][tl][e] = acc[e];
    }
    __syncthreads();
    if (hq || !live) return;
#pragma unroll
    for (int q = 0; q < 3; ++q)
#pragma unroll
        for (int e = 0; e < 8; ++e) acc[e] += part[q][tl][e];
    u32x4 oh, ol;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        unsigned hh, ll;
        split2(acc[2 * e], acc[2 * e + 1], hh, ll);
        oh[e] = hh; ol[e] = ll;
    }
    const size_t o = s_index(out, gi, b, cg * 8, t);
    *reinterpret_cast<u32x4 *>(out.hi + o) = oh;
    *reinterpret_cast<u32x4 *>(out.hi + out.lo_off + o) = ol;
}

// WN2D's backward on its own (wg_wf_wn_backward): the gradients of the returned (log_s, t), plain [items][rows_in][T], into the seed plane G
// (channel 0 = d log_s, channel 1 = d t; zero for the rows that produced no output) and a zero gradient plane for x -- no coupling here,
// WN2D's start conv adds its part in wn_backward
__global__ void wf_seed_kernel(const float *__restrict__ dls, const float *__restrict__ dt, PRef G, PRef dX, Geo g, int rows_in)
{
    const int t = blockIdx.x * blockDim.x + threadIdx.x, row = blockIdx.y;
    if (t >= g.T) return;
    const int b = row / g.rows, r = row - b * g.rows;
    const size_t o = ((size_t)b * rows_in + r) * g.T + t;
    *paddr(G, g, row, 0, t) = r < rows_in ? dls[o] : 0.f;
    *paddr(G, g, row, 1, t) = r < rows_in ? dt[o] : 0.f;
    *paddr(dX, g, row, 0, t) = 0.f;
}
__global__ void wf_rows_out_kernel(PRef X, Geo g, int rows_out, float *__restrict__ x)
{
    const int t = blockIdx.x * blockDim.x + threadIdx.x, row = blockIdx.y;
    if (t >= g.T) return;
    const int b = row / g.rows, r = row - b * g.rows;
    if (r < rows_out) x[((size_t)b * rows_out + r) * g.T + t] = *paddr(X, g, row, 0, t);
}

// WN2D.forward on its own: x[items][rows_in][T] (plain) into the n_group-row planes; the rows below it are zero (the convs are causal along
// the height axis: they cannot reach the rows above them)
__global__ void wf_rows_in_kernel(const float *__restrict__ x, PRef X, Geo g, int rows_in)
{
    const int t = blockIdx.x * blockDim.x + threadIdx.x, row = blockIdx.y;
    if (t >= g.T) return;
    const int b = row / g.rows, r = row - b * g.rows;
    *paddr(X, g, row, 0, t) = r < rows_in ? x[((size_t)b * rows_in + r) * g.T + t] : 0.f;
}

