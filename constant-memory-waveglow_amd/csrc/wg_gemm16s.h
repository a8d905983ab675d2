// wg_gemm16s.h -- bf16x3 kernels fed from PRE-SPLIT activation planes ("S-planes"): no conversion work in the main loops.
//
// An S-plane holds a tensor as two bf16 arrays (hi, then lo = x - hi) in a channel-interleaved layout
//     S[b][c/8][p][8]      p = H + t, same pitch P and zero-halo invariant as the fp32 planes,
// i.e. one 16-byte unit = 8 consecutive channels at one time step.  Consequences:
//   * the conv B-tile image [t][k] is filled with plain 16-byte copies and ANY time shift (dilation tap) stays
//     16-byte aligned (a shift moves whole units);
//   * the wgrad operand image [t][c] is filled the same way and read with ds_read_b64_tr_b16 (hardware transpose),
//     which hands each lane 4 consecutive time steps of its channel;
//   * producers write it straight from the MFMA accumulator layout: a lane owns 4 consecutive channels of one time
//     step (8 bytes), a wave store covers 512 contiguous bytes.
// Producers (epilogues below, to_splane_kernel) do the fp32 -> hi/lo split ONCE per element; with the on-the-fly kernels
// of wg_gemm16.h every consumer workgroup repeated it (12x for the dilated conv).
#pragma once
#include "wg_gemm16.h"

// 4 consecutive channels (c multiple of 4) of one time step
// EXPERIMENT -DWG_OPT_NT_S=<mask>: S-plane stores of the conv epilogues with the non-temporal policy (1: store / residual, 2: gate
// backward, 4: gate conv)
#if !defined(WG_OPT_NT_S)
#define WG_OPT_NT_S 0
#endif
template <bool NT = false>
__device__ __forceinline__ void s_store4(const SRef &r, const Geo &g, int b, int c, int t, const float (&v)[4])
{
    u32x2 h, l;
    unsigned hh, ll;
    split2(v[0], v[1], hh, ll); h[0] = hh; l[0] = ll;
    split2(v[2], v[3], hh, ll); h[1] = hh; l[1] = ll;
    const size_t i = s_index(r, g, b, c, t);
    if (NT) {
        __builtin_nontemporal_store(h, reinterpret_cast<u32x2 *>(r.hi + i));
        __builtin_nontemporal_store(l, reinterpret_cast<u32x2 *>(r.hi + r.lo_off + i));
    } else {
        *reinterpret_cast<u32x2 *>(r.hi + i) = h;
        *reinterpret_cast<u32x2 *>(r.hi + r.lo_off + i) = l;
    }
}

// Two 8-byte half units -> one 16-byte unit per lane.  In the MFMA accumulator layout lanes l and l + 32 own channels 4h .. 4h+3
// (h = 0 / 1) of the same time step, i.e. the two halves of ONE S-plane unit, so a direct store is 8 bytes per lane.  Given the
// packed halves of two channel groups q0 (v0) and q1 (v1), v_permlane32_swap exchanges the upper half-wave of one register with the
// lower half-wave of the other: afterwards lanes < 32 hold the whole unit of group q0 and lanes >= 32 the whole unit of group q1,
// and the wave stores 2 x 512 contiguous bytes with one dwordx4 per lane instead of two dwordx2 (half the store instructions).
// Must be executed by ALL lanes (no divergence around it); predicate only the store.
__device__ __forceinline__ u32x4 pair_units(const u32x2 &v0, const u32x2 &v1)
{
    const u32x2 a = __builtin_amdgcn_permlane32_swap(v0[0], v1[0], false, false);
    const u32x2 b = __builtin_amdgcn_permlane32_swap(v0[1], v1[1], false, false);
    u32x4 r;
    r[0] = a[0]; r[1] = b[0]; r[2] = a[1]; r[3] = b[1];
    return r;
}

// fp32 plane channels [ch0, ch0+nvalid) -> S-plane channels [0, Cp_dst) (zero filled beyond nvalid): time step t, channel group cg, plane row b
__device__ __forceinline__ void to_splane_body(const PRef &src, int nvalid, const SRef &dst, const Geo &g, int t, int cg, int b)
{
    if (t >= g.T) return;
    float v[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = (cg * 8 + e < nvalid) ? *paddr(src, g, b, cg * 8 + e, t) : 0.f;
    u32x4 h, l;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        unsigned hh, ll;
        split2(v[2 * e], v[2 * e + 1], hh, ll);
        h[e] = hh; l[e] = ll;
    }
    const size_t i = s_index(dst, g, b, cg * 8, t);
    *reinterpret_cast<u32x4 *>(dst.hi + i) = h;
    *reinterpret_cast<u32x4 *>(dst.hi + dst.lo_off + i) = l;
}
__global__ void to_splane_kernel(PRef src, int nvalid, SRef dst, Geo g)
{
    to_splane_body(src, nvalid, dst, g, blockIdx.x * blockDim.x + threadIdx.x, blockIdx.y, blockIdx.z);
}

// ------------------------------------------------------------------------------------------------
// convgemm16s: A from the pre-split weight images, B from S-planes
// ------------------------------------------------------------------------------------------------
// EXPERIMENT -DWG_OPT_2P=<mask>: drop one cross term of the split product a b ~ a_lo b_hi + a_hi b_lo + a_hi b_hi in a class of products (the
// error table in DESIGN.md section 4b; the default build multiplies all three everywhere).  Bits: 1 gate conv without a_lo b_hi (the
// weights' low half), 2 gate conv without a_hi b_lo (the activations' low half), 4 store / residual / data-gradient convs without the
// weights' low half, 8 gate backward without the weights' low half, 16 weight gradient without the low half of its A operand (the
// gradient planes), 32 weight gradient without the low half of its B operand (the activations).
#if !defined(WG_OPT_2P)
#define WG_OPT_2P 0
#endif
template <int EPI> struct TwoP {
    static constexpr bool no_alo = (EPI == EPI_GATE && (WG_OPT_2P & 1)) || ((EPI == EPI_STORE || EPI == EPI_STORE_SO || EPI == EPI_STORE_FO || EPI == EPI_RESSKIP) && (WG_OPT_2P & 4)) ||
                                   ((EPI == EPI_DGATE || EPI == EPI_DGATE_SO) && (WG_OPT_2P & 8));
    static constexpr bool no_blo = EPI == EPI_GATE && (WG_OPT_2P & 2);
};
struct SSeg {
    const unsigned short *hi;
    size_t lo_off;
    int Cp, ch0;
    int row_off;      // Geo::rows > 0: the tap reads plane row b + row_off; rows outside the tile's own item read as zero
    int per_item;     // Geo::rows > 0: the operand has ONE plane row per item (conditioning broadcast over the height axis)
};
struct ConvGemm16sArgs {
    const unsigned short *img;
    size_t img_stride;
    ConvGemmArgs c;               // geometry, seg[].nch / shift, epilogue operands (seg[].src unused)
    SSeg sseg[WG_MAX_SEG];
    SRef s0;                      // S-plane output (hi == nullptr: none)
    SRef saux;                    // EPI_STORE, convgemm16q only: the accumulate-into input as an S-plane (hi + lo) instead of the fp32 plane aux0
    int ntx, nty, ntz;            // convgemm16w: the tile grid (time tiles, 128-row tiles, plane rows); workgroup w walks tiles w, w + G, ...
    int tap_il, tap_chunks;       // convgemm16q: the first tap_il K segments are the taps of ONE plane (tap_chunks chunks each): the K loop walks
                                  // them interleaved -- channel block 0 of every tap, then block 1, ... -- instead of tap after tap, so that a
                                  // tap's window, which is a neighbouring time tile's centre window, is requested within a few chunks of that
                                  // neighbour's own request and still sits in the XCD's L2 (tap after tap they are 16 chunks = tens of
                                  // microseconds apart: the data-gradient conv fetched every window from HBM again, 388 MB for 147 MB)
    // convgemm16g_kernel / convlayer16g_kernel, EPI_GATE_SO only (wg_gemm16g.h, wgg_gate_nb): with `part` the epilogue also leaves
    // Weff (gate tile) -- the tile's share of WN's `out` = sum_l Weff_l gate_l (wg_small.h weff_kernel) -- as [slot][b][t][8] fp32, slot =
    // 4 x row tile + the wave's row block: the end conv then adds nl x slots rows of 32 bytes per column instead of reading every gate
    // plane again.  eff: Weff_l as MFMA fragments, [M / 64 slices of 32 gate channels][hi | lo][k-group][8 rows][8 bf16] (weff_kernel)
    const float *eff;
    float *part;
    int prow;                     // floats per partial row: 8, or 2 where 2 ic <= 2 (WaveFlow's WN2D: the 16x16x32 kernels only)
    int xcd_items;                // convgemm16q, persistent launches: > 0 = plane rows per XCD (ntz / 8): XCD x (workgroup id & 7) owns the plane
                                  // rows x, x + 8, ... and walks their tiles row by row -- all time tiles of a row are then in flight on ONE
                                  // XCD, so a dilation tap's window (another tile's centre window) and the other row tiles' copy of the same
                                  // columns are hits in that XCD's L2 instead of second and third HBM reads (profiles/r03j: the data-gradient
                                  // conv fetched 388 MB for 147 MB of operands, the gate conv 145 for 67)
};

// ------------------------------------------------------------------------------------------------
// convgemm16p: software-pipelined 128x128 variant (4 waves, two workgroups per CU).
// Every chunk runs two k-steps (half chunks are zero padded in LDS); inside a chunk the fragment reads of step 1 are
// issued between the MFMAs of step 0 and the LDS writes of the NEXT chunk between the MFMAs of step 1, so a wave keeps
// the matrix pipe fed by itself; sched_group_barrier pins that interleave.
// ------------------------------------------------------------------------------------------------
struct Frags16 {
    bf16x8 ah[2], al[2], bh[2], bl[2];
};
__device__ __forceinline__ void read_frags16(Frags16 &f, const char *Ahi, const char *Alo, const char *Bhi, const char *Blo, int ao, int bo)
{
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        f.ah[i] = *reinterpret_cast<const bf16x8 *>(Ahi + ao + i * 32 * WG16_ROWB);
        f.al[i] = *reinterpret_cast<const bf16x8 *>(Alo + ao + i * 32 * WG16_ROWB);
        f.bh[i] = *reinterpret_cast<const bf16x8 *>(Bhi + bo + i * 32 * WG16_ROWB);
        f.bl[i] = *reinterpret_cast<const bf16x8 *>(Blo + bo + i * 32 * WG16_ROWB);
    }
}
__device__ __forceinline__ void mfma12(const Frags16 &f, f32x16 (&acc)[2][2])
{
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
        for (int ni = 0; ni < 2; ++ni) {
            if (!(WG_OPT_2P & 16)) acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f.al[mi], f.bh[ni], acc[mi][ni], 0, 0, 0);     // (mfma12 serves the
            if (!(WG_OPT_2P & 32)) acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f.ah[mi], f.bl[ni], acc[mi][ni], 0, 0, 0);     // weight gradients only)
            acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f.ah[mi], f.bh[ni], acc[mi][ni], 0, 0, 0);
        }
}

// ------------------------------------------------------------------------------------------------
// staging registers of the loader waves (convgemm16q / convgemm16h; protocol: convgemm16w, removed in round 6 -- git show 4c099e9:tools/experiments/wg_gemm16_superseded.h)
// ------------------------------------------------------------------------------------------------
struct Stage8 {
    u32x4 ah[2], al[2], bh[2], bl[2];
};
struct Stage6 {                      // the 64-column tile: one B unit per lane and image
    u32x4 ah[2], al[2], bh[1], bl[1];
};
struct Stage6a {                     // the 64-ROW tile (convgemm16q_kernel<.., M64>): one A unit per lane and image, two B units
    u32x4 ah[1], al[1], bh[2], bl[2];
};
template <int NI> struct StageOf { typedef Stage8 type; };
template <> struct StageOf<1> { typedef Stage6 type; };
#define WG_STAGE_REGS(s) "+v"(s.ah[0]), "+v"(s.ah[1]), "+v"(s.al[0]), "+v"(s.al[1]), "+v"(s.bh[0]), "+v"(s.bh[1]), "+v"(s.bl[0]), "+v"(s.bl[1])
__device__ __forceinline__ void asm_wait_keep8(Stage8 &s) { asm volatile("s_waitcnt vmcnt(8)" : WG_STAGE_REGS(s)::"memory"); }
__device__ __forceinline__ void asm_wait_stage(Stage8 &s) { asm_wait_keep8(s); }
__device__ __forceinline__ void asm_wait_stage(Stage6a &s)
{
    asm volatile("s_waitcnt vmcnt(6)" : "+v"(s.ah[0]), "+v"(s.al[0]), "+v"(s.bh[0]), "+v"(s.bh[1]), "+v"(s.bl[0]), "+v"(s.bl[1])::"memory");
}
__device__ __forceinline__ void asm_wait_stage(Stage6 &s)
{
    asm volatile("s_waitcnt vmcnt(6)" : "+v"(s.ah[0]), "+v"(s.ah[1]), "+v"(s.al[0]), "+v"(s.al[1]), "+v"(s.bh[0]), "+v"(s.bl[0])::"memory");
}
#if defined(WG_DBG_TRACE)      // phase timestamps: [workgroup][16]; read back by wg_dbg_trace_read (tools/experiments/conv_trace.py, wgrad_trace.py)
__device__ unsigned long long wg_dbg_trace[512 * 16];       // 100 MHz wall clock
__device__ unsigned long long wg_dbg_trace_cyc[512 * 16];   // shader cycles (s_memtime): cycles / wall = the clock the chip holds in the phase
#endif
#if defined(WG_DBG_NOBAR)      // timing experiment only (results are garbage): how much of a launch is barrier skew?
#define WG16W_BAR() do { } while (0)
#else
#define WG16W_BAR() __syncthreads()
#endif
#if defined(WG_OPT_MFMA32) || defined(WG_OPT_NO_WSPEC) || defined(WG_OPT_DMA)
#error "the superseded conv kernels (convgemm16w / 16d / 16p) left the tree in round 6: git show 4c099e9:tools/experiments/wg_gemm16_superseded.h"
#endif

// ------------------------------------------------------------------------------------------------
// wgrad16s: weight gradients from S-planes.  dW[m][n] = sum_b sum_t A[b][m][t] * B[b][n][t + shift]
// LDS images are [t (32 rows)][c (128 channels)] bf16 with 320-byte rows, filled by 16-byte unit copies; an MFMA fragment
// (8 consecutive time steps of one channel) is two ds_read_b64_tr_b16 (4x16 hardware transposes), conflict free.
// ------------------------------------------------------------------------------------------------
// The loader / compute split of convgemm16w was tried here too (wgrad16w: 8 waves, asm loads with counted waits, 101 VGPRs, parity
// identical): 157 us against 149 us for this symmetric kernel on the same box -- not kept.
#define WG16_ROWT 320
struct WgSSeg {
    const unsigned short *hi;
    size_t lo_off;
    int Cp, ch0, nch, shift, blk0;
    int row_off, per_item;      // as SSeg (B operand only)
};
// A GROUP of products of one shape in one launch (the weight gradients of all layers of a WN, deferred to the end of its backward: one
// layer alone is 8-28 tiles, i.e. a 64- or 18-way split of the time axis to fill the chip, and every split writes a full-size slab
// that the finalisation reads back -- 130 MB of HBM traffic per layer; eight layers together need a 2- to 8-way split).  The groups
// share every field of WgradSArgs except the operand planes, the tap shifts and the slab; grid z = group * nsplit + split.
#define WG_GRP_MAX 8
struct WgradGrp {
    const unsigned short *a_hi[2];          // nullptr: that segment's rows are zero in this group
    const unsigned short *b_plane[3];       // the (at most three) distinct B planes of a group: the layer input (all its taps), the
    float *slab;                            // conditioning, the ones of a WN with biases; WgradSArgs::b_plane_of says which one a segment reads
    short b_shift[WG_MAX_SEG];              // time shift and plane-row offset of every B segment
    short b_row[WG_MAX_SEG];
};
struct WgradSArgs {
    int nseg_a, nseg_b;
    WgSSeg sa[2];
    WgSSeg sb[WG_MAX_SEG];
    Geo g;
    int cpb, total_chunks, nsplit;      // 32-step chunks per batch item, B * cpb, blocks per tile (each takes an even share of the range)
    float *slab;
    int Mp, Np;
    int ngroups;                        // 0: one product (sa / sb / slab as they are)
    unsigned char b_plane_of[WG_MAX_SEG];
    const unsigned short *zsrc;         // grouped launches: any S-plane (position 0 = zero halo)
    unsigned *sync;                     // grouped launches: one progress counter per (group, split), WG_SYNC_STRIDE apart, zeroed before the
    int sync_n;                         // launch (nullptr: the tiles of a (group, split) run unsynchronised); sync_n = tiles per counter
    WgradGrp grp[WG_GRP_MAX];
};
// Soft lock-step of the workgroups that share operands.  The tiles of one (group, split) read the same A row tiles and B column tiles
// chunk by chunk; left alone they drift apart over the hundreds of chunks of a launch (a CU's first workgroup wins issue arbitration
// over its second one) until the shared lines have left the XCD's L2 and are fetched again: 6.3 GB of HBM traffic per paired launch
// for 2.5 GB of operands (profiles/r02l_hbm_traffic.json).  Every WG_SYNC_EVERY chunks a workgroup adds one to its set's counter and
// waits until all sync_n members have done so.  Performance hint only: the wait is bounded, and a workgroup that runs into the bound
// once (a member that has not been dispatched yet) stops synchronising -- no result depends on the counter, nothing can hang on it.
#define WG_SYNC_STRIDE 32
#if !defined(WG_SYNC_EVERY)
#define WG_SYNC_EVERY 4
#endif
#if !defined(WG_SYNC_SPINS)
#define WG_SYNC_SPINS 256
#endif
typedef short s4v __attribute__((ext_vector_type(4)));
// rows r..r+3 (lo) and r+4..r+7 (hi) of the image, r a multiple of 8 plus the lane's row: the upper four rows are stored rotated
// by 32 bytes inside the 256-byte row payload (see the staging map of wgrad16s_kernel)
template <int PITCH>                      // row pitch in bytes; the row payload is PITCH - 64 bytes
__device__ __forceinline__ bf16x8 tr_frag(const char *img, int rowoff, int col)
{
    typedef __attribute__((address_space(3))) s4v *lds_s4p;
    const s4v lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4p)(img + rowoff + col));
#if defined(WG_OPT_WGRAD_OLDMAP)
    const s4v hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4p)(img + rowoff + 4 * PITCH + col));
#else
    const s4v hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4p)(img + rowoff + 4 * PITCH + ((col + 32) & (PITCH - 65))));
#endif
    bf16x8 r;
    r[0] = lo[0]; r[1] = lo[1]; r[2] = lo[2]; r[3] = lo[3];
    r[4] = hi[0]; r[5] = hi[1]; r[6] = hi[2]; r[7] = hi[3];
    return r;
}
__device__ __forceinline__ int find_sseg_idx(const WgSSeg *s, int n, int blk)
{
    int i = 0;
#pragma unroll
    for (int j = 1; j < WG_MAX_SEG; ++j)
        if (j < n && blk >= s[j].blk0) i = j;
    return i;
}
__device__ __forceinline__ const WgSSeg &find_sseg(const WgSSeg *s, int n, int blk) { return s[find_sseg_idx(s, n, blk)]; }

// MT = 1 (default): 128 x 128 tile, 4 waves, two workgroups per CU.  MT = 2 (-DWG_OPT_WGRAD_TALL): 256 x 128 tile, 8 waves, one
// workgroup per CU; streams 48 KB instead of 64 KB per chunk for the same MFMAs and is still slower (155 vs 137 us).
// the workgroup's work: tile (bx, by) of the product, split index zs (grouped launches: group * nsplit + split)
template <int MT>
__device__ __forceinline__ void wgrad16s_body(const WgradSArgs &a, int bx, int by, int zs)
{
    constexpr int NT = 256 * MT;                // threads
    constexpr int AROW = 256 * MT + 64;         // A image row pitch (bytes): 128 MT channels + pad (pitch = 16 dwords mod 64: four
                                                // consecutive rows of a transposing read fall on disjoint banks)
    constexpr int AIMG = 32 * AROW, BIMG = 32 * WG16_ROWT;
    constexpr int BUF = 2 * AIMG + 2 * BIMG;
    constexpr int NB = 2 / MT;                  // B units per thread and image
    typedef typename StageOf<NB>::type Stage;   // 4 A + 2 NB B loads per chunk
    __shared__ __attribute__((aligned(16))) char smem[2 * BUF];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wr = wave >> 1, wc = wave & 1;
    int grp = 0;
    if (a.ngroups) { grp = zs / a.nsplit; zs -= grp * a.nsplit; }
    const int n0 = bx * WG_TILE, m0 = by * (WG_TILE * MT);
    const Geo g = a.g;

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // staging: unit u = tid + 256*j -> t = (u & 7) + 8 * (u >> 7), channel group cg = (u >> 3) & 15: eight consecutive lanes
    // fetch the eight time steps of one channel group = one whole 128-byte line (the earlier 2 x 2 quad mapping fetched 32-byte
    // pieces, four address-processing passes per line: the operand stream alone took 127 of the launch's 147 us).  In LDS a row is
    // a time step; the eight rows one write pass touches fall on only two bank groups of the 32-bank write port (320-byte pitch =
    // 16 dwords mod 32), a 4-way conflict; rows with (t >> 2) odd are therefore rotated by two units (32 bytes) inside their
    // 256-byte payload -- the transposing fragment read applies the same rotation to its upper four rows (tr_frag) -- which leaves
    // a 2-way write conflict (PMC: a third of this kernel's LDS cycles; profiles/r01m_pmc.json).  It cannot be rotated away: the
    // fragment read needs the four rows of a group on disjoint 16-dword spans of the 64-bank read port, i.e. one rotation per
    // group, and rows t and t+2 of a group are 32 dwords apart.  The conflict-free 2 x 2 quad map (-DWG_OPT_WGRAD_OLDMAP) is
    // still 2 % slower end to end: the LDS pipe is half idle here, the global side is what counts.
    const unsigned short *pa[2], *pb[NB];
    size_t la[2], lb_[NB], sba[2], sbb[NB];
    int loffa[2], loffb[NB], roff[NB], pitem[NB];
#pragma unroll
    for (int j = 0; j < 2; ++j) {                            // A: 32 time steps x 16 MT channel groups = 2 NT units
        const int u = tid + NT * j;
#if defined(WG_OPT_WGRAD_OLDMAP)    // experiment: the 2 x 2 quad map (32-byte global pieces, no LDS write conflicts, no rotation)
        const int tl = 2 * (u >> 5) + (u & 1), cg = (u >> 1) & 15;
        loffa[j] = tl * AROW + cg * 16;
#else
        const int tl = (u & 7) + 8 * (u / (128 * MT)), cg = (u >> 3) & (16 * MT - 1);
        loffa[j] = tl * AROW + ((cg + 2 * ((tl >> 2) & 1)) & (16 * MT - 1)) * 16;
#endif
        const int ma = m0 + 8 * cg;
        const int ia = find_sseg_idx(a.sa, a.nseg_a, ma >> 5);
        const WgSSeg &sa = a.sa[ia];
        const int ca = ma - sa.blk0 * 32;
        const unsigned short *ha = a.ngroups ? a.grp[grp].a_hi[ia] : sa.hi;
        pa[j] = (ma < a.Mp && ca < sa.nch && ha) ? ha + (((size_t)((sa.ch0 + ca) >> 3)) * g.P + g.H + tl) * 8 : nullptr;
        la[j] = sa.lo_off; sba[j] = (size_t)(sa.Cp >> 3) * g.P * 8;
    }
#pragma unroll
    for (int j = 0; j < NB; ++j) {                           // B: 32 x 16 = 512 units
        const int u = tid + NT * j;
#if defined(WG_OPT_WGRAD_OLDMAP)
        const int tl = 2 * (u >> 5) + (u & 1), cg = (u >> 1) & 15;
        loffb[j] = tl * WG16_ROWT + cg * 16;
#else
        const int tl = (u & 7) + 8 * (u >> 7), cg = (u >> 3) & 15;
        loffb[j] = tl * WG16_ROWT + ((cg + 2 * ((tl >> 2) & 1)) & 15) * 16;
#endif
        const int nb = n0 + 8 * cg;
        const int ib = find_sseg_idx(a.sb, a.nseg_b, nb >> 5);
        const WgSSeg &sb = a.sb[ib];
        const int cb = nb - sb.blk0 * 32;
        const unsigned short *hb = a.ngroups ? a.grp[grp].b_plane[a.b_plane_of[ib]] : sb.hi;
        const int bshift = a.ngroups ? (int)a.grp[grp].b_shift[ib] : sb.shift;
        pb[j] = (nb < a.Np && cb < sb.nch && hb) ? hb + (((size_t)((sb.ch0 + cb) >> 3)) * g.P + g.H + bshift + tl) * 8 : nullptr;
        lb_[j] = sb.lo_off; sbb[j] = (size_t)(sb.Cp >> 3) * g.P * 8;
        roff[j] = a.ngroups ? (int)a.grp[grp].b_row[ib] : sb.row_off; pitem[j] = sb.per_item;
    }
    unsigned *sync_ctr = (a.ngroups && a.sync) ? a.sync + (size_t)(grp * a.nsplit + zs) * WG_SYNC_STRIDE : nullptr;
    // this block's share of the flattened (batch item, chunk) range
    const int c_begin = (int)((long)zs * a.total_chunks / a.nsplit), c_end = (int)((long)(zs + 1) * a.total_chunks / a.nsplit);
    const int nchunks = c_end - c_begin;

    // fragment addressing: lane supplies row q = (l&15)>>2 of its 4x16 block, columns 4*(l&3)..; block = rows 8h (+4), cols 16*((l>>4)&1)
    const int fq = (lane & 15) >> 2, fp = lane & 3, fh = lane >> 5, fg = (lane >> 4) & 1;
    const int frowa = (8 * fh + fq) * AROW, frowb = (8 * fh + fq) * WG16_ROWT;
    const int fca = (wr * 64 + 16 * fg + 4 * fp) * 2, fcb = (wc * 64 + 16 * fg + 4 * fp) * 2;
    auto read_step = [&](Frags16 &f, const char *sb, int s) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            f.ah[i] = tr_frag<AROW>(sb, frowa + s * 16 * AROW, fca + i * 64);
            f.al[i] = tr_frag<AROW>(sb + AIMG, frowa + s * 16 * AROW, fca + i * 64);
            f.bh[i] = tr_frag<WG16_ROWT>(sb + 2 * AIMG, frowb + s * 16 * WG16_ROWT, fcb + i * 64);
            f.bl[i] = tr_frag<WG16_ROWT>(sb + 2 * AIMG + BIMG, frowb + s * 16 * WG16_ROWT, fcb + i * 64);
        }
    };
    auto store_stage = [&](const Stage &st, int buf) {
        char *sb = smem + buf * BUF;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            *reinterpret_cast<u32x4 *>(sb + loffa[j]) = st.ah[j];
            *reinterpret_cast<u32x4 *>(sb + AIMG + loffa[j]) = st.al[j];
        }
#pragma unroll
        for (int j = 0; j < NB; ++j) {
            *reinterpret_cast<u32x4 *>(sb + 2 * AIMG + loffb[j]) = st.bh[j];
            *reinterpret_cast<u32x4 *>(sb + 2 * AIMG + BIMG + loffb[j]) = st.bl[j];
        }
    };
    // Two chunks in flight per wave: the loads are issued from inline asm (hipcc's own bookkeeping drains every outstanding load at
    // the loop back edge, which caps a compiler-managed prefetch at half a chunk) and retired by counted waits, as in convgemm16w:
    // the stream alone took 127 of this launch's 147 us at one chunk in flight (64 KB per CU).  Every issue is exactly eight loads
    // in straight-line code: lanes without a source row and chunks past the end read the zero halo (selected pointers, no branch
    // between a load and its wait; tools/check_asm_loads.py covers this kernel too).  (4 + 2 NB loads per issue.)
    const unsigned short *zsrc = a.ngroups ? a.zsrc : a.sa[0].hi;    // plane position 0 of the first operand: always-zero halo
    int lb = c_begin / a.cpb, lt = (c_begin - lb * a.cpb) * WG16_BK, issued = 0;
#define WG_LDP(dst, ptr) asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(dst) : "v"(ptr) : "memory")
    auto issue = [&](Stage &st) {
        const bool live = issued < nchunks;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const unsigned short *qa = (live && pa[j]) ? pa[j] + lb * sba[j] + (size_t)lt * 8 : zsrc;
            const unsigned short *qal = (live && pa[j]) ? qa + la[j] : zsrc;
            WG_LDP(st.ah[j], qa);  WG_LDP(st.al[j], qal);
        }
#pragma unroll
        for (int j = 0; j < NB; ++j) {
            // B operand row: the chunk's own plane row, another row of the same item (2-D taps, zero outside it) or the item's row
            int bsrc = lb;
            bool rowok = true;
            if (g.rows > 0) {
                const int item = lb / g.rows, r = lb - item * g.rows + roff[j];
                rowok = r >= 0 && r < g.rows;
                bsrc = pitem[j] ? item : lb + roff[j];
            }
            const bool bok = live && pb[j] && rowok;
            const unsigned short *qb = bok ? pb[j] + bsrc * sbb[j] + (size_t)lt * 8 : zsrc;
            const unsigned short *qbl = bok ? qb + lb_[j] : zsrc;
            WG_LDP(st.bh[j], qb);  WG_LDP(st.bl[j], qbl);
        }
        if (live) {
            ++issued;
            lt += WG16_BK;
            if (lt >= g.Tt) { lt = 0; ++lb; }
        }
    };
#undef WG_LDP
    if (nchunks > 0) {
        Stage s0, s1;
        issue(s0);                                           // chunk 0
        issue(s1);                                           // chunk 1
        asm_wait_stage(s0);
        store_stage(s0, 0);
        issue(s0);                                           // chunk 2
        __syncthreads();
        // iteration c: multiply chunk c from buffer c & 1; the stage holding chunk c+1 has landed -> write it to the other buffer
        // between the two k-steps, then re-issue that stage for chunk c+3
        auto iter = [&](Stage &st, int c) {
            const char *sb = smem + (c & 1) * BUF;
            Frags16 f0, f1;
            read_step(f0, sb, 0);
            read_step(f1, sb, 1);
            mfma12(f0, acc);
            asm_wait_stage(st);
            store_stage(st, (c & 1) ^ 1);
            issue(st);
            mfma12(f1, acc);
            if (sync_ctr && (c & (WG_SYNC_EVERY - 1)) == WG_SYNC_EVERY - 1 && tid == 0) {
                __hip_atomic_fetch_add(sync_ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                const unsigned want = (unsigned)((c + 1) / WG_SYNC_EVERY) * (unsigned)a.sync_n;
                int spins = 0;
                while (__hip_atomic_load(sync_ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < want) {
                    if (++spins > WG_SYNC_SPINS) { sync_ctr = nullptr; break; }     // (thread 0's copy: it alone uses the pointer)
                    __builtin_amdgcn_s_sleep(8);
                }
            }
            __syncthreads();
        };
        // always in pairs (after the last chunk the spare iteration multiplies a buffer of zero-halo data: adds exact zeros)
        for (int c = 0; c < nchunks; c += 2) {
            iter(s1, c);
            iter(s0, c + 1);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // drain the trailing zero-halo loads before the wave ends
    }
    float *out = (a.ngroups ? a.grp[grp].slab : a.slab) + (size_t)zs * a.Mp * a.Np;
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

template <int MT>
__global__ __launch_bounds__(256 * MT) void wgrad16s_kernel(const WgradSArgs a)
{
    int bx, by, zs;
    xcd_remap(bx, by, zs);
    wgrad16s_body<MT>(a, bx, by, zs);
}

// TWO (grouped) products of different shapes in one 1-D launch: workgroups [0, n0) belong to p[0], the rest to p[1].  The hardware
// hands out workgroups in id order, so the short workgroups of the second product fill the slots the first one leaves (a WN's tap /
// conditioning gradients are 448 workgroups of 756 chunks on 512 slots, its W_o gradients 512 of 189: 945 chunk times back to back,
// 850 dealt out together).
struct WgradPairArgs {
    WgradSArgs p[2];
    int n0, gx[2], gy[2], n[2];              // workgroups of p[0]; tile grid and workgroup count of each product
};
static_assert(sizeof(WgradPairArgs) <= 4096, "kernel arguments are limited to 4 KB");
__global__ __launch_bounds__(256) void wgrad16s_pair_kernel(const WgradPairArgs pp)
{
    const int which = (int)blockIdx.x >= pp.n0;
    int id = (int)blockIdx.x - (which ? pp.n0 : 0);
    const int n = pp.n[which], gx = pp.gx[which], gy = pp.gy[which];
    if ((n & 7) == 0) id = (id & 7) * (n >> 3) + (id >> 3);            // xcd_remap's relabelling, inside this product's range
    wgrad16s_body<1>(pp.p[which], id % gx, (id / gx) % gy, id / (gx * gy));
}
