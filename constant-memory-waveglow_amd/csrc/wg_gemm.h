// wg_gemm.h -- the two MFMA kernels that carry ~99.9 % of the WaveGlow flow FLOPs (gfx950, wave64).
//
//   convgemm_kernel : out[m][t] = sum_seg sum_c A[k(seg,c)][m] * src_seg[c][t + shift_seg]      ("NN")
//                     dilated k=3 conv + mel conditioning (waveglow.py:42), W_o (waveglow.py:45), and in the
//                     backward pass W_o^T, W^T (dgrad), V^T.  Fused epilogues: gate, residual/skip, gate-backward.
//   wgrad_kernel    : dW[m][n]  = sum_b sum_t A[b][m][t] * Bsrc[b][n][t + shift]                ("NT", split over b,t)
//
// Both run the exact-fp32 matrix instruction v_mfma_f32_32x32x2_f32 (no TF32/xf32 exists on gfx950; parity with
// the reference's fp32 CPU path is 1e-4).  A workgroup is 4 waves (2x2), each wave owns a 64x64 output sub-tile
// (2x2 MFMA tiles, 64 accumulator registers); operands are staged through LDS k-major so that every
// ds_read_b32 of a fragment is 32 consecutive dwords (conflict free), with register-staged prefetch of the
// next chunk behind the MFMAs of the current one.
//
// Activations live in "planes": [B][Cp][P] floats, P = H + Tt + H, data at [H, H+T), everything else zero.
// The zero halo IS the convolution's zero padding, so shifted tile loads need no bounds checks and stay
// inside their row; stores are masked to t < T so the halo stays zero.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

struct Geo {
    int B, T, Tt, H, P;   // Tt = roundup(T,128); P = H + Tt + H
    int rows;             // 0: every plane row is one batch item.  > 0 (WaveFlow): an item is `rows` consecutive plane rows (the height
                          // axis), B = items * rows; taps may then reach other rows of the same item (SSeg::row_off)
};

// reference to channels [ch0, ch0+..) of a plane
struct PRef {
    float *p;
    int Cp;    // channel rows per batch item
    int ch0;
};
__device__ __forceinline__ float *paddr(const PRef &r, const Geo &g, int b, int ch, int t)
{
    return r.p + ((size_t)b * r.Cp + r.ch0 + ch) * g.P + g.H + t;
}

#define WG_TILE 128        // output tile edge (M and N) of both kernels
#define WG_BK 16           // channels per chunk (convgemm)
#define WG_WBK 32          // time steps per chunk (wgrad)
#define WG_MAX_SEG 11     // 3x3 taps + conditioning + the ones segment of a WN2D with biases (WaveFlow); the 1-D WN uses 4 or 5

// ------------------------------------------------------------------------------------------------
// shared inner product: acc[mi][ni] += As[k][wr*64 + mi*32 + r] * Bs[k][wc*64 + ni*32 + c], k < BKK
// As/Bs are k-major with row strides LDA/LDB (floats).
// v_mfma_f32_32x32x2_f32 operand map: lane l holds A[i = l&31][k = l>>5] and B[k = l>>5][j = l&31].
// ------------------------------------------------------------------------------------------------
template <int BKK, int LDA, int LDB>
__device__ __forceinline__ void mma_chunk(const float *__restrict__ As, const float *__restrict__ Bs,
                                          int wr, int wc, int lane, f32x16 (&acc)[2][2])
{
    const int kh = lane >> 5, r = lane & 31;
    const float *ap = As + kh * LDA + wr * 64 + r;
    const float *bp = Bs + kh * LDB + wc * 64 + r;
#pragma unroll
    for (int kk = 0; kk < BKK; kk += 2) {
        const float a0 = ap[kk * LDA], a1 = ap[kk * LDA + 32];
        const float b0 = bp[kk * LDB], b1 = bp[kk * LDB + 32];
        acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[0][0], 0, 0, 0);
        acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc[0][1], 0, 0, 0);
        acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc[1][0], 0, 0, 0);
        acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc[1][1], 0, 0, 0);
    }
}

// gate nonlinearities: hardware exp2/rcp based forms (|err| ~3e-7, well inside the 1e-4 parity budget); the libm
// forms cost ~4x the VALU work, which is visible once the contraction runs on the bf16 pipe.  -DWG_EXACT_GATE restores them.
// The reciprocal is the bare v_rcp_f32 (1 ulp): __frcp_rn expands to the ten-instruction IEEE division sequence, and with 64
// reciprocals per lane and tile that sequence alone was a quarter of the gate conv's epilogue, which is VALU bound (measured:
// 37 us of a 133 us launch were the epilogue's arithmetic, 12 us its stores).
#if !defined(WG_EXACT_GATE)
#define WG_OPT_FASTGATE 1
#endif
__device__ __forceinline__ float wg_sigmoid(float x)
{
#if defined(WG_OPT_FASTGATE)
    return __builtin_amdgcn_rcpf(1.0f + __expf(-x));
#else
    return 1.0f / (1.0f + expf(-x));
#endif
}
__device__ __forceinline__ float wg_tanh(float x)
{
#if defined(WG_OPT_FASTGATE)
    const float e = __expf(-2.0f * fabsf(x));            // in (0,1]: no overflow
    const float t = (1.0f - e) * __builtin_amdgcn_rcpf(1.0f + e);
    return copysignf(t, x);
#else
    return tanhf(x);
#endif
}

// C/D map of the 32x32 tile: col = lane&31, row = (reg&3) + 8*(reg>>2) + 4*(lane>>5)
__device__ __forceinline__ int acc_row(int reg, int lane) { return (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5); }

// ------------------------------------------------------------------------------------------------
// convgemm
// ------------------------------------------------------------------------------------------------
enum {
    EPI_STORE = 0,    // out0[m] = acc (+ aux0[m])
    EPI_GATE = 1,     // packed rows come in 64-blocks [32 tanh | 32 sigmoid]; out0 = gate, out1 = tanh, out2 = sigmoid
    EPI_RESSKIP = 2,  // m < nsplit: out0[m] = acc + aux0[m] ; else out1[m-nsplit] (+)= acc
    EPI_DGATE = 3,    // out0[m] = acc*sf*(1-tw^2) ; out0[nsplit+m] = acc*tw*sf*(1-sf)   (aux0 = tw, aux1 = sf)
    EPI_STORE_FO = 7, // convgemm16q only: EPI_STORE with an fp32 plane as its ONLY output (skip sum, conditioning gradient), hand-issued stores
    EPI_DGATE_SO = 6, // convgemm16q only: EPI_DGATE with S-plane output only; the tanh / sigmoid loads of a row block and its stores hand-issued
    EPI_GATE_SO = 5,  // convgemm16q only: EPI_GATE without the fp32 gate plane (S-plane gate, optional tanh / sigmoid planes), hand-issued stores
    EPI_STORE_SO = 4, // convgemm16q only: EPI_STORE whose output (and accumulate-into input, if any) exist as S-planes ONLY: its own
                      // instantiation, so that the hand-issued stores of its epilogue share the kernel with no compiler-tracked store
};

struct ConvSeg {
    const float *src;   // plane base
    int Cp, ch0, nch;   // rows per item, first row, channels in this segment (multiple of WG_BK)
    int shift;          // time shift of the tap
    int row_off;        // Geo::rows > 0 (WaveFlow's height axis): the tap reads plane row b + row_off; rows outside the tile's own item read as zero
    int per_item;       // Geo::rows > 0: the operand has ONE plane row per item (conditioning broadcast over the height axis)
};

struct ConvGemmArgs {
    const float *A;     // [K][lda], K = sum nch
    int lda, M;         // M = valid output rows
    int nseg;
    ConvSeg seg[WG_MAX_SEG];
    Geo g;
    int epi, nsplit, accumulate;
    PRef out0, out1, out2, aux0, aux1;
    int row_sel1;       // Geo::rows > 0: 0 = blockIdx.z is the plane row; r + 1 = blockIdx.z is the item and the tile is its height row r
};

// fused epilogues shared by the fp32 and the split-precision kernels
template <int EPI>
__device__ __forceinline__ void conv_epilogue(const ConvGemmArgs &a, f32x16 (&acc)[2][2], int t0, int m0, int b, int wr, int wc, int lane)
{
    const Geo g = a.g;
    const int col = lane & 31;
    if (EPI == EPI_GATE) {
        // wave rows: mi = 0 -> tanh pre-activation, mi = 1 -> sigmoid pre-activation of the SAME channels
        const int chb = (m0 >> 1) + wr * 32;   // first channel of this wave's 32
#pragma unroll
        for (int ni = 0; ni < 2; ++ni) {
            const int t = t0 + wc * 64 + ni * 32 + col;
            if (t >= g.T) continue;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int ch = chb + acc_row(r, lane);
                if (2 * ch >= a.M) continue;
                const float tw = wg_tanh(acc[0][ni][r]);
                const float sf = wg_sigmoid(acc[1][ni][r]);
                *paddr(a.out0, g, b, ch, t) = tw * sf;
                if (a.out1.p) {
                    *paddr(a.out1, g, b, ch, t) = tw;
                    *paddr(a.out2, g, b, ch, t) = sf;
                }
            }
        }
    } else {
#pragma unroll
        for (int mi = 0; mi < 2; ++mi)
#pragma unroll
            for (int ni = 0; ni < 2; ++ni) {
                const int t = t0 + wc * 64 + ni * 32 + col;
                if (t >= g.T) continue;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int m = m0 + wr * 64 + mi * 32 + acc_row(r, lane);
                    if (m >= a.M) continue;
                    const float v = acc[mi][ni][r];
                    if (EPI == EPI_STORE) {
                        float o = v;
                        if (a.aux0.p) o += *paddr(a.aux0, g, b, m, t);
                        *paddr(a.out0, g, b, m, t) = o;
                    } else if (EPI == EPI_RESSKIP) {
                        if (m < a.nsplit) {
                            *paddr(a.out0, g, b, m, t) = v + *paddr(a.aux0, g, b, m, t);
                        } else {
                            float *sp = paddr(a.out1, g, b, m - a.nsplit, t);
                            *sp = a.accumulate ? (*sp + v) : v;
                        }
                    } else if (EPI == EPI_DGATE) {
                        const float tw = *paddr(a.aux0, g, b, m, t);
                        const float sf = *paddr(a.aux1, g, b, m, t);
                        *paddr(a.out0, g, b, m, t) = v * sf * (1.0f - tw * tw);
                        *paddr(a.out0, g, b, a.nsplit + m, t) = v * tw * sf * (1.0f - sf);
                    }
                }
            }
    }
}

template <int EPI>
__global__ __launch_bounds__(256) void convgemm_kernel(const ConvGemmArgs a)
{
    __shared__ __attribute__((aligned(16))) float As[2][WG_BK][WG_TILE];
    __shared__ __attribute__((aligned(16))) float Bs[2][WG_BK][WG_TILE];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wr = wave >> 1, wc = wave & 1;
    const Geo g = a.g;
    const int t0 = blockIdx.x * WG_TILE, m0 = blockIdx.y * WG_TILE;
    const int b = a.row_sel1 ? (int)blockIdx.z * g.rows + a.row_sel1 - 1 : (int)blockIdx.z;      // (row_sel1: blockIdx.z is the item, the tile its height row)

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // staging registers
    f32x4 ra[2];
    float rb[8];

    // chunk bookkeeping: chunk index -> (segment, channel offset)
    int nchunks = 0;
    for (int s = 0; s < a.nseg; ++s) nchunks += a.seg[s].nch / WG_BK;

    int cur_seg = 0, cur_c = 0, krow = 0;   // position of the NEXT chunk to load
    auto load_chunk = [&]() {
        // A: rows krow..krow+15, cols m0..m0+127 : 512 float4, 2 per thread
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int row = (tid >> 5) + 8 * j, c4 = tid & 31;
            ra[j] = *reinterpret_cast<const f32x4 *>(a.A + (size_t)(krow + row) * a.lda + m0 + c4 * 4);
        }
        const ConvSeg sg = a.seg[cur_seg];
        // 2-D taps (WaveFlow): another plane row of the same item (zero outside it), or the item's one row of a per-item operand
        int bsrc = b;
        bool rowok = true;
        if (g.rows > 0) {
            const int item = b / g.rows, rr = b - item * g.rows + sg.row_off;
            rowok = rr >= 0 && rr < g.rows;
            bsrc = sg.per_item ? item : (rowok ? b + sg.row_off : b);
        }
        const float *base = sg.src + ((size_t)bsrc * sg.Cp + sg.ch0 + cur_c) * g.P + g.H + t0 + sg.shift;
        if ((sg.shift & 3) == 0) {
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int row = (tid >> 5) + 8 * j, c4 = tid & 31;
                const f32x4 v = *reinterpret_cast<const f32x4 *>(base + (size_t)row * g.P + c4 * 4);
                rb[4 * j + 0] = v[0]; rb[4 * j + 1] = v[1]; rb[4 * j + 2] = v[2]; rb[4 * j + 3] = v[3];
            }
        } else {
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int row = (tid >> 7) + 2 * j, c = tid & 127;
                rb[j] = base[(size_t)row * g.P + c];
            }
        }
        if (!rowok) {
#pragma unroll
            for (int j = 0; j < 8; ++j) rb[j] = 0.f;
        }
        // advance
        krow += WG_BK;
        cur_c += WG_BK;
        if (cur_c >= sg.nch) { cur_c = 0; ++cur_seg; }
    };
    // shift parity of the chunk whose data sits in rb (needed to know its register layout at store time)
    auto store_chunk = [&](int buf, bool aligned) {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int row = (tid >> 5) + 8 * j, c4 = tid & 31;
            *reinterpret_cast<f32x4 *>(&As[buf][row][c4 * 4]) = ra[j];
        }
        if (aligned) {
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int row = (tid >> 5) + 8 * j, c4 = tid & 31;
                f32x4 v;
                v[0] = rb[4 * j + 0]; v[1] = rb[4 * j + 1]; v[2] = rb[4 * j + 2]; v[3] = rb[4 * j + 3];
                *reinterpret_cast<f32x4 *>(&Bs[buf][row][c4 * 4]) = v;
            }
        } else {
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int row = (tid >> 7) + 2 * j, c = tid & 127;
                Bs[buf][row][c] = rb[j];
            }
        }
    };

    bool al = (a.seg[0].shift & 3) == 0;
    load_chunk();
    store_chunk(0, al);
    __syncthreads();
    for (int c = 0; c < nchunks; ++c) {
        const int buf = c & 1;
        bool al_next = false;
        if (c + 1 < nchunks) {
            al_next = (a.seg[cur_seg].shift & 3) == 0;
            load_chunk();
        }
        mma_chunk<WG_BK, WG_TILE, WG_TILE>(&As[buf][0][0], &Bs[buf][0][0], wr, wc, lane, acc);
        if (c + 1 < nchunks) store_chunk(buf ^ 1, al_next);
        __syncthreads();
    }

    conv_epilogue<EPI>(a, acc, t0, m0, b, wr, wc, lane);
}

// ------------------------------------------------------------------------------------------------
// wgrad:  slab[z][m][n] = sum over this block's (b, t-range) of A[b][m][t] * Bsrc[b][n][t + shift]
// M rows come from up to 2 plane segments, N rows from up to WG_MAX_SEG segments (taps / conditioning);
// every segment is padded to a multiple of 32 rows in the (m | n) index space.
// ------------------------------------------------------------------------------------------------
struct WgSeg {
    const float *src;
    int Cp, ch0, nch;   // valid channels (rows beyond are read as zero)
    int shift;
    int blk0;           // first 32-row block of this segment in the index space
    int row_off, per_item;   // as ConvSeg (B operand only)
};

struct WgradArgs {
    int nseg_a, nseg_b;
    WgSeg sa[2];
    WgSeg sb[WG_MAX_SEG];
    Geo g;
    int t_per_split;    // multiple of WG_WBK
    int nts;            // t-splits per batch item
    int b_per_split;    // batch items accumulated inside one block
    float *slab;        // [nsplit][Mp][Np]
    int Mp, Np;
};

// XCD-aware block order for the weight-gradient kernels: the dispatcher deals consecutive workgroup ids round-robin over the 8
// XCDs (private L2s), so the 28 tiles that share one batch item's operands would land on 8 different L2s and every tile would
// be fetched from HBM 4-7 times.  Re-labelling id -> (id % 8) * (n / 8) + id / 8 gives each XCD a contiguous range of logical
// blocks (= whole batch items).  Speed only; falls back to the identity when the grid is not a multiple of 8.
__device__ __forceinline__ void xcd_remap(int &bx, int &by, int &bz)
{
    const int gx = gridDim.x, gy = gridDim.y, n = gx * gy * gridDim.z;
    int id = blockIdx.x + gx * (blockIdx.y + gy * blockIdx.z);
    if ((n & 7) == 0) id = (id & 7) * (n >> 3) + (id >> 3);
    bx = id % gx;
    by = (id / gx) % gy;
    bz = id / (gx * gy);
}

#define WG_WLD 129   // odd LDS row stride: transposed stores are <= 2-way conflicted, reads conflict free

__device__ __forceinline__ const WgSeg &find_seg(const WgSeg *s, int n, int blk)
{
    int i = 0;
#pragma unroll
    for (int j = 1; j < WG_MAX_SEG; ++j)
        if (j < n && blk >= s[j].blk0) i = j;
    return s[i];
}

__global__ __launch_bounds__(256) void wgrad_kernel(const WgradArgs a)
{
    __shared__ float As[2][WG_WBK][WG_WLD];
    __shared__ float Bs[2][WG_WBK][WG_WLD];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wr = wave >> 1, wc = wave & 1;
    int bx, by, zs;
    xcd_remap(bx, by, zs);
    const int n0 = bx * WG_TILE, m0 = by * WG_TILE;
    const int ts = zs % a.nts, bs = zs / a.nts;
    const Geo g = a.g;

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // each thread stages rows (tid>>3) + 32*j (j<4), floats 4*(tid&7) .. +3 of the chunk
    const int lrow = tid >> 3, k4 = (tid & 7) * 4;
    const float *pa[4];
    const float *pb[4];
    bool bal[4];
    int roff[4], pitem[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int ma = m0 + lrow + 32 * j;
        const WgSeg &sa = find_seg(a.sa, a.nseg_a, ma >> 5);
        const int ca = ma - sa.blk0 * 32;
        pa[j] = (ma < a.Mp && ca < sa.nch) ? sa.src + ((size_t)sa.ch0 + ca) * g.P + g.H + k4 : nullptr;
        const int nb = n0 + lrow + 32 * j;
        const WgSeg &sb = find_seg(a.sb, a.nseg_b, nb >> 5);
        const int cb = nb - sb.blk0 * 32;
        pb[j] = (nb < a.Np && cb < sb.nch) ? sb.src + ((size_t)sb.ch0 + cb) * g.P + g.H + sb.shift + k4 : nullptr;
        bal[j] = (sb.shift & 3) == 0;
        roff[j] = sb.row_off; pitem[j] = sb.per_item;
    }
    // per-batch strides of the two operands (rows per item differ per segment -> fold into pointer per j)
    size_t sba[4], sbb[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const WgSeg &sa = find_seg(a.sa, a.nseg_a, (m0 + lrow + 32 * j) >> 5);
        const WgSeg &sb = find_seg(a.sb, a.nseg_b, (n0 + lrow + 32 * j) >> 5);
        sba[j] = (size_t)sa.Cp * g.P;
        sbb[j] = (size_t)sb.Cp * g.P;
    }

    const int t_begin = ts * a.t_per_split;
    int t_end = t_begin + a.t_per_split;
    if (t_end > g.Tt) t_end = g.Tt;
    const int chunks_per_b = (t_end - t_begin + WG_WBK - 1) / WG_WBK;
    const int b_begin = bs * a.b_per_split;
    int b_end = b_begin + a.b_per_split;
    if (b_end > g.B) b_end = g.B;
    const int nchunks = chunks_per_b * (b_end - b_begin);

    f32x4 ra[4], rb[4];
    int lb = b_begin, lt = t_begin;   // position of the next chunk to load
    auto load_chunk = [&]() {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (pa[j]) v = *reinterpret_cast<const f32x4 *>(pa[j] + lb * sba[j] + lt);
            ra[j] = v;
            f32x4 w = {0.f, 0.f, 0.f, 0.f};
            int bsrc = lb;
            bool rowok = true;
            if (g.rows > 0) {                              // 2-D taps: see ConvSeg
                const int item = lb / g.rows, rr = lb - item * g.rows + roff[j];
                rowok = rr >= 0 && rr < g.rows;
                bsrc = pitem[j] ? item : lb + roff[j];
            }
            if (pb[j] && rowok) {
                const float *q = pb[j] + bsrc * sbb[j] + lt;
                if (bal[j]) w = *reinterpret_cast<const f32x4 *>(q);
                else { w[0] = q[0]; w[1] = q[1]; w[2] = q[2]; w[3] = q[3]; }
            }
            rb[j] = w;
        }
        lt += WG_WBK;
        if (lt >= t_end) { lt = t_begin; ++lb; }
    };
    auto store_chunk = [&](int buf) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int row = lrow + 32 * j;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                As[buf][k4 + e][row] = ra[j][e];
                Bs[buf][k4 + e][row] = rb[j][e];
            }
        }
    };

    if (nchunks > 0) {
        load_chunk();
        store_chunk(0);
        __syncthreads();
        for (int c = 0; c < nchunks; ++c) {
            const int buf = c & 1;
            if (c + 1 < nchunks) load_chunk();
            mma_chunk<WG_WBK, WG_WLD, WG_WLD>(&As[buf][0][0], &Bs[buf][0][0], wr, wc, lane, acc);
            if (c + 1 < nchunks) store_chunk(buf ^ 1);
            __syncthreads();
        }
    }

    float *out = a.slab + (size_t)zs * a.Mp * a.Np;
    const int col = lane & 31;
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
        for (int ni = 0; ni < 2; ++ni) {
            const int n = n0 + wc * 64 + ni * 32 + col;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = m0 + wr * 64 + mi * 32 + acc_row(r, lane);
                if (m < a.Mp && n < a.Np) out[(size_t)m * a.Np + n] = acc[mi][ni][r];
            }
        }
}
