// wg_gemm16.h -- split-precision ("bf16x3") variants of the two MFMA kernels.
//
// fp32 operands are split on the fly into hi = bf16(x), lo = bf16(x - hi) and every product is evaluated as
//     a*b ~= a_lo*b_hi + a_hi*b_lo + a_hi*b_hi        (three v_mfma_f32_32x32x16_bf16, fp32 accumulate)
// which drops only the a_lo*b_lo term (2^-16 relative to the product; measured 2e-7 rms of sum|ab| at K = 848, see
// tools/experiments/split_mfma.hip), has fp32 range (bf16 exponent) and runs on the 16x faster bf16 matrix pipe.
// All tensors in HBM stay fp32: activations are converted when a tile is staged into LDS, weights are pre-split
// once per step by wg_pack_weights into chunked [chunk][m][32] hi/lo images.
//
// LDS image of every operand tile: [row][32 k + 8 pad] bf16 (80-byte rows): a fragment of v_mfma_f32_32x32x16_bf16
// (lane l: row l&31, k = 8*(l>>5) .. +7) is one conflict-free ds_read_b128.
#pragma once
#include "wg_gemm.h"
#include "wg_splane.h"
#include <type_traits>

typedef short bf16x8 __attribute__((ext_vector_type(8)));

#define WG16_BK 32                    // k per chunk (two MFMA k-steps)
#define WG16_ROWB 80                  // bytes per LDS row (32 bf16 + 16 B pad)
#define WG16_IMG (WG_TILE * WG16_ROWB)  // one 128-row image

// one k-step of 16: acc[mi][ni] += A(rows wr*64+mi*32..) x B(rows wc*64+ni*32..)
__device__ __forceinline__ void mma16_step(const char *Ahi, const char *Alo, const char *Bhi, const char *Blo,
                                           int ao, int bo, f32x16 (&acc)[2][2])
{
    bf16x8 ah[2], al[2], bh[2], bl[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        ah[i] = *reinterpret_cast<const bf16x8 *>(Ahi + ao + i * 32 * WG16_ROWB);
        al[i] = *reinterpret_cast<const bf16x8 *>(Alo + ao + i * 32 * WG16_ROWB);
        bh[i] = *reinterpret_cast<const bf16x8 *>(Bhi + bo + i * 32 * WG16_ROWB);
        bl[i] = *reinterpret_cast<const bf16x8 *>(Blo + bo + i * 32 * WG16_ROWB);
    }
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
        for (int ni = 0; ni < 2; ++ni) {
            acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[mi], bh[ni], acc[mi][ni], 0, 0, 0);
            acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[mi], bl[ni], acc[mi][ni], 0, 0, 0);
            acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[mi], bh[ni], acc[mi][ni], 0, 0, 0);
        }
}
// a chunk = nk16 (1 or 2) k-steps
__device__ __forceinline__ void mma16_chunk(const char *Ahi, const char *Alo, const char *Bhi, const char *Blo,
                                            int wr, int wc, int lane, int nk16, f32x16 (&acc)[2][2])
{
    const int r = lane & 31, h = lane >> 5;
    const int ao = (wr * 64 + r) * WG16_ROWB + h * 16, bo = (wc * 64 + r) * WG16_ROWB + h * 16;
    mma16_step(Ahi, Alo, Bhi, Blo, ao, bo, acc);
    if (nk16 == 2) mma16_step(Ahi, Alo, Bhi, Blo, ao + 32, bo + 32, acc);
}

// ------------------------------------------------------------------------------------------------
// convgemm16: same contract and epilogues as convgemm_kernel; A comes from the pre-split images
//   img_hi[chunk][lda][32], img_lo = img_hi + nchunks_total*lda*32 (bf16)
// chunks never straddle a segment; a segment whose channel count is 16 mod 32 ends with a half chunk (nk16 = 1).
// ------------------------------------------------------------------------------------------------
struct ConvGemm16Args {
    const unsigned short *img;   // hi image; lo image follows at + img_stride
    size_t img_stride;           // elements between hi and lo images
    ConvGemmArgs c;              // geometry, segments, epilogue (c.A unused)
};

// MT = 64-row wave rows per workgroup: 2 -> 128x128 tile, 4 waves;  4 -> 256x128 tile, 8 waves (the B tile is converted once
// for twice the MFMA work).  LDS: 2 buffers x (MT*64 + MT*64 + 128 + 128) rows x 80 B.
// LDS byte offset of 16-byte piece p of an A image (global order: [128-row block][k-group][row][8 k]; LDS: padded rows)
__device__ __forceinline__ int wg16_a_off(int p)
{
    return ((p >> 9) * 128 + (p & 127)) * WG16_ROWB + ((p >> 7) & 3) * 16;
}

template <int EPI, int MT>
__global__ __launch_bounds__(128 * MT) void convgemm16_kernel(const ConvGemm16Args aa)
{
    constexpr int NT = 128 * MT;                 // threads
    constexpr int AIMG = MT * 64 * WG16_ROWB;    // one A image (hi or lo)
    constexpr int BUF = 2 * AIMG + 2 * WG16_IMG; // A_hi, A_lo, B_hi, B_lo
    constexpr int KPT = 32 / (NT / 128);         // k rows of the B tile converted per thread (16 or 8)
    __shared__ __attribute__((aligned(16))) char smem[2 * BUF];
    const ConvGemmArgs &a = aa.c;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wr = wave >> 1, wc = wave & 1;
    const int t0 = blockIdx.x * WG_TILE, m0 = blockIdx.y * (64 * MT);
    const int b = aa.c.row_sel1 ? (int)blockIdx.z * aa.c.g.rows + aa.c.row_sel1 - 1 : (int)blockIdx.z;      // (row_sel1: see convgemm_kernel)
    const Geo g = a.g;

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    int nchunks = 0;
    for (int s = 0; s < a.nseg; ++s) nchunks += (a.seg[s].nch + WG16_BK - 1) / WG16_BK;

    // staging registers: A 2 x (hi,lo) 16-B pieces, B KPT floats
    u32x4 ra_hi[2], ra_lo[2];
    float rb[KPT];
    int cur_seg = 0, cur_c = 0, chunk = 0;      // next chunk to load
    int nk_loaded = 0;                          // k16 steps of the chunk sitting in the staging registers
    const int bt = tid & 127;                                            // B staging: time column
    const int bk = __builtin_amdgcn_readfirstlane(tid >> 7) * KPT;       // first k row (wave uniform)

    auto load_chunk = [&]() {
        const ConvSeg sg = a.seg[cur_seg];
        const int nvalid = min(WG16_BK, sg.nch - cur_c);
        nk_loaded = nvalid >> 4;
        // A: MT*256 pieces of 16 B per image: piece p -> 128-row block p>>9, k-group (p>>7)&3, row p&127 (see img_kernel)
        const unsigned short *ih = aa.img + ((size_t)chunk * a.lda + m0) * WG16_BK;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int p = tid + NT * j;
            ra_hi[j] = *reinterpret_cast<const u32x4 *>(ih + (size_t)p * 8);
            ra_lo[j] = *reinterpret_cast<const u32x4 *>(ih + aa.img_stride + (size_t)p * 8);
        }
        int bsrc = b;                                                        // 2-D taps (WaveFlow): see ConvSeg
        bool rowok = true;
        if (g.rows > 0) {
            const int item = b / g.rows, rr = b - item * g.rows + sg.row_off;
            rowok = rr >= 0 && rr < g.rows;
            bsrc = sg.per_item ? item : (rowok ? b + sg.row_off : b);
        }
        const float *base = sg.src + ((size_t)bsrc * sg.Cp + sg.ch0 + cur_c + bk) * g.P + g.H + t0 + sg.shift;   // wave uniform
        if (bk < nvalid && rowok) {
#pragma unroll
            for (int j = 0; j < KPT; ++j) rb[j] = base[(size_t)j * g.P + bt];
        } else {
#pragma unroll
            for (int j = 0; j < KPT; ++j) rb[j] = 0.f;
        }
        ++chunk;
        cur_c += WG16_BK;
        if (cur_c >= sg.nch) { cur_c = 0; ++cur_seg; }
    };
    auto store_chunk = [&](int buf) {
        char *sb = smem + buf * BUF;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int p = tid + NT * j;
            const int off = wg16_a_off(p);
            *reinterpret_cast<u32x4 *>(sb + off) = ra_hi[j];
            *reinterpret_cast<u32x4 *>(sb + AIMG + off) = ra_lo[j];
        }
        char *bh = sb + 2 * AIMG + bt * WG16_ROWB + bk * 2, *bl = bh + WG16_IMG;
#pragma unroll
        for (int q = 0; q < KPT / 8; ++q) {
            u32x4 vh, vl;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                unsigned hh, ll;
                split2(rb[8 * q + 2 * e], rb[8 * q + 2 * e + 1], hh, ll);
                vh[e] = hh; vl[e] = ll;
            }
            *reinterpret_cast<u32x4 *>(bh + q * 16) = vh;
            *reinterpret_cast<u32x4 *>(bl + q * 16) = vl;
        }
    };

    load_chunk();
    int nk_cur = nk_loaded;
    store_chunk(0);
    __syncthreads();
    for (int c = 0; c < nchunks; ++c) {
        const int buf = c & 1;
        if (c + 1 < nchunks) load_chunk();
        const char *sb = smem + buf * BUF;
        mma16_chunk(sb, sb + AIMG, sb + 2 * AIMG, sb + 2 * AIMG + WG16_IMG, wr, wc, lane, nk_cur, acc);
        if (c + 1 < nchunks) { store_chunk(buf ^ 1); nk_cur = nk_loaded; }
        __syncthreads();
    }
    conv_epilogue<EPI>(a, acc, t0, m0, b, wr, wc, lane);
}

// ------------------------------------------------------------------------------------------------
// wgrad16: same contract as wgrad_kernel, k = time (32 steps per chunk), both operands converted on the fly
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void wgrad16_kernel(const WgradArgs a)
{
    __shared__ __attribute__((aligned(16))) char smem[2 * 4 * WG16_IMG];
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

    const int lrow = tid >> 3, k4 = (tid & 7) * 4;
    const float *pa[4];
    const float *pb[4];
    bool bal[4];
    size_t sba[4], sbb[4];
    int roff[4], pitem[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int ma = m0 + lrow + 32 * j;
        const WgSeg &sa = find_seg(a.sa, a.nseg_a, ma >> 5);
        const int ca = ma - sa.blk0 * 32;
        pa[j] = (ma < a.Mp && ca < sa.nch) ? sa.src + ((size_t)sa.ch0 + ca) * g.P + g.H + k4 : nullptr;
        sba[j] = (size_t)sa.Cp * g.P;
        const int nb = n0 + lrow + 32 * j;
        const WgSeg &sb = find_seg(a.sb, a.nseg_b, nb >> 5);
        const int cb = nb - sb.blk0 * 32;
        pb[j] = (nb < a.Np && cb < sb.nch) ? sb.src + ((size_t)sb.ch0 + cb) * g.P + g.H + sb.shift + k4 : nullptr;
        sbb[j] = (size_t)sb.Cp * g.P;
        bal[j] = (sb.shift & 3) == 0;
        roff[j] = sb.row_off; pitem[j] = sb.per_item;
    }
    const int t_begin = ts * a.t_per_split;
    int t_end = t_begin + a.t_per_split;
    if (t_end > g.Tt) t_end = g.Tt;
    const int chunks_per_b = (t_end - t_begin + WG16_BK - 1) / WG16_BK;
    const int b_begin = bs * a.b_per_split;
    int b_end = b_begin + a.b_per_split;
    if (b_end > g.B) b_end = g.B;
    const int nchunks = chunks_per_b * (b_end - b_begin);

    f32x4 ra[4], rb[4];
    int lb = b_begin, lt = t_begin;
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
        lt += WG16_BK;
        if (lt >= t_end) { lt = t_begin; ++lb; }
    };
    auto store_chunk = [&](int buf) {
        char *sb = smem + buf * 4 * WG16_IMG;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int off = (lrow + 32 * j) * WG16_ROWB + k4 * 2;
            u32x2 h, l;
            unsigned hh, ll;
            split2(ra[j][0], ra[j][1], hh, ll); h[0] = hh; l[0] = ll;
            split2(ra[j][2], ra[j][3], hh, ll); h[1] = hh; l[1] = ll;
            *reinterpret_cast<u32x2 *>(sb + off) = h;
            *reinterpret_cast<u32x2 *>(sb + WG16_IMG + off) = l;
            split2(rb[j][0], rb[j][1], hh, ll); h[0] = hh; l[0] = ll;
            split2(rb[j][2], rb[j][3], hh, ll); h[1] = hh; l[1] = ll;
            *reinterpret_cast<u32x2 *>(sb + 2 * WG16_IMG + off) = h;
            *reinterpret_cast<u32x2 *>(sb + 3 * WG16_IMG + off) = l;
        }
    };

    if (nchunks > 0) {
        load_chunk();
        store_chunk(0);
        __syncthreads();
        for (int c = 0; c < nchunks; ++c) {
            const int buf = c & 1;
            if (c + 1 < nchunks) load_chunk();
            const char *sb = smem + buf * 4 * WG16_IMG;
            mma16_chunk(sb, sb + WG16_IMG, sb + 2 * WG16_IMG, sb + 3 * WG16_IMG, wr, wc, lane, 2, acc);
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

// ------------------------------------------------------------------------------------------------
// pack + image in one pass (round 3): a pack job whose rows are whole chunks of its matrix's image writes the image straight from
// the 32 (k) x 64 (m) LDS tile it builds anyway -- no fp32 matrix written and read back (and none written at all where nothing reads
// it: PackJob::dst == nullptr).  pack_kernel + img_kernel moved 40 bytes per parameter per layout, this moves 12-20.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void packimg_kernel(const PackArgs a)
{
    const PackJob j = a.job[blockIdx.y];
    if (!j.img) { pack_job_plain(j); return; }
    __shared__ float tile[32][65];
    const int tid = threadIdx.x;
    const int tk = (j.Kp + 31) / 32, tm = (j.Mp + 63) / 64;
    for (int tl = blockIdx.x; tl < tk * tm; tl += gridDim.x) {
        const int kc = tl / tm, k0 = kc * 32, m0 = (tl - kc * tm) * 64;
        __syncthreads();
        // (all eight values of a thread are requested before the first is used: with the load inside the conditional the compiler
        // issued them one round trip at a time)
        float sv[8], vv[8];
        if (j.mode == 0) {                                    // consecutive lanes read consecutive k of one source row (see pack_job_plain)
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
                const bool ok = o >= 0 && o < j.no && k < j.ni && m < j.Mp;
                const int oc = ok ? o : 0, kc = ok ? k : 0;                       // (clamped: element 0 always exists)
                sv[q] = j.scale[oc];
                vv[q] = j.src[(size_t)oc * j.so + (size_t)kc * j.si + j.off];
                if (!ok) sv[q] = 0.f;
            }
#pragma unroll
            for (int q = 0; q < 8; ++q) tile[kk][(tid >> 5) + 8 * q] = sv[q] * vv[q];
        } else {                                              // mode 1: consecutive lanes read consecutive m of source row k
            const int mm = tid & 63, m = m0 + mm;
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const int kk = (tid >> 6) + 4 * q, k = k0 + kk;
                const bool ok = k < j.no && m < j.ni;
                const int kc = ok ? k : 0, mc = ok ? m : 0;
                sv[q] = j.scale[kc];
                vv[q] = j.src[(size_t)kc * j.so + (size_t)mc * j.si + j.off];
                if (!ok) sv[q] = 0.f;
            }
#pragma unroll
            for (int q = 0; q < 8; ++q) tile[(tid >> 6) + 4 * q][mm] = sv[q] * vv[q];
        }
        __syncthreads();
        if (j.dst) {
            const int m = m0 + (tid & 63);
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const int kw = (tid >> 6) + 4 * q;
                if (k0 + kw < j.Kp && m < j.Mp) j.dst[(size_t)(k0 + kw) * j.ldd + m] = tile[kw][tid & 63];
            }
        }
        // image unit: thread -> (m = tid >> 2, 8 k values (tid & 3) * 8 ..): the layout img_kernel writes
        const int m = tid >> 2, k8 = (tid & 3) * 8, mr = m0 + m;
        if (mr < j.ldd) {
            u32x4 vh, vl;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                unsigned hh, ll;
                split2(tile[k8 + 2 * e][m], tile[k8 + 2 * e + 1][m], hh, ll);
                vh[e] = hh; vl[e] = ll;
            }
            const size_t o = ((size_t)(j.chunk0 + kc) * j.ldd + (mr & ~127)) * 32 + (size_t)(k8 >> 3) * 1024 + (size_t)(mr & 127) * 8;
            *reinterpret_cast<u32x4 *>(j.img + o) = vh;
            *reinterpret_cast<u32x4 *>(j.img + (size_t)j.nchunks * j.ldd * 32 + o) = vl;
        }
    }
}

// ------------------------------------------------------------------------------------------------
// pre-split weight images: A32 [K][lda] (k-major fp32, segments of nch rows) -> [chunk][lda][32] hi | lo
// one block per (chunk, 64 columns)
// ------------------------------------------------------------------------------------------------
struct ImgJob {
    const float *A32;
    unsigned short *img;      // hi; lo at + nchunks*lda*32
    int lda, nseg, nchunks;
    int nch[WG_MAX_SEG];
};
#define WG_IMG_JOBS 48
#define WG_IMG_COLS 512     // columns of a chunk one block converts
struct ImgArgs {
    int n;
    int start[WG_IMG_JOBS + 1];     // prefix sums of the jobs' block counts (nchunks * column groups): the grid is exactly their total --
                                    // a (max columns, max chunks, jobs) grid spent most of a launch dispatching blocks that return at once
    ImgJob job[WG_IMG_JOBS];
};
__global__ __launch_bounds__(256) void img_kernel(const ImgArgs a)
{
    __shared__ float tile[32][65];
    int ji = 0;
    for (int q = 1; q < a.n; ++q)
        if ((int)blockIdx.x >= a.start[q]) ji = q;
    const ImgJob j = a.job[ji];
    const int cgroups = (j.lda + WG_IMG_COLS - 1) / WG_IMG_COLS, local = (int)blockIdx.x - a.start[ji];
    const int ci = local / cgroups, cbx = local - ci * cgroups;
    if (ci >= j.nchunks) return;
    // a block walks WG_IMG_COLS / 64 column groups of its chunk: 8 KB per group is too little work per block for the 230 M
    // parameters of WSRGlow (the launch was dispatch bound).  All of a block's loads (64 floats per thread) are issued before the
    // first group is converted: as one load-convert-store round trip per group the kernel moved 0.7 TB/s (72 us per flow, 1.2 % of a
    // training step).
    int seg = 0, c0 = 0, row0 = 0, left = ci;                       // locate chunk ci
    for (seg = 0; seg < j.nseg; ++seg) {
        const int nc = (j.nch[seg] + 31) / 32;
        if (left < nc) { c0 = left * 32; break; }
        left -= nc;
        row0 += j.nch[seg];
    }
    const int nvalid = min(32, j.nch[seg] - c0);
    const int tid = threadIdx.x;
    constexpr int NG = WG_IMG_COLS / 64;
    float v[NG][8];
#pragma unroll
    for (int cbk = 0; cbk < NG; ++cbk) {
        const int mb = (cbx * NG + cbk) * 64;
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const int e = tid + 256 * q, kk = e >> 6, m = e & 63;
            v[cbk][q] = (mb < j.lda && kk < nvalid) ? j.A32[(size_t)(row0 + c0 + kk) * j.lda + mb + m] : 0.f;
        }
    }
#pragma unroll
    for (int cbk = 0; cbk < NG; ++cbk) {
        const int mb = (cbx * NG + cbk) * 64;
        if (mb >= j.lda) return;                                   // (the same for the whole block)
        __syncthreads();
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const int e = tid + 256 * q;
            tile[e >> 6][e & 63] = v[cbk][q];
        }
        __syncthreads();
        // thread -> (m = tid>>2, 8 k values (tid&3)*8..)
        const int m = tid >> 2, k8 = (tid & 3) * 8;
        u32x4 vh, vl;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            unsigned hh, ll;
            split2(tile[k8 + 2 * e][m], tile[k8 + 2 * e + 1][m], hh, ll);
            vh[e] = hh; vl[e] = ll;
        }
        // image layout: per chunk and 128-row block, [k-group 0..3][row 0..127][8 k] -- the conv kernels stage a block with
        // lane-linear 16-byte loads and consecutive lanes must land on consecutive LDS rows (80-byte stride: conflict free), not on
        // the four pieces of one row (2-way conflict, measured as 20 % of the LDS cycles of the row-major layout)
        const int mr = mb + m;
        const size_t o = ((size_t)ci * j.lda + (mr & ~127)) * 32 + (size_t)(k8 >> 3) * 1024 + (size_t)(mr & 127) * 8;
        *reinterpret_cast<u32x4 *>(j.img + o) = vh;
        *reinterpret_cast<u32x4 *>(j.img + (size_t)j.nchunks * j.lda * 32 + o) = vl;
    }
}
