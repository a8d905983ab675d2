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
// 4 consecutive channels (c multiple of 4) of one time step
__device__ __forceinline__ void s_store4(const SRef &r, const Geo &g, int b, int c, int t, const float (&v)[4])
{
    u32x2 h, l;
    unsigned hh, ll;
    split2(v[0], v[1], hh, ll); h[0] = hh; l[0] = ll;
    split2(v[2], v[3], hh, ll); h[1] = hh; l[1] = ll;
    const size_t i = s_index(r, g, b, c, t);
    *reinterpret_cast<u32x2 *>(r.hi + i) = h;
    *reinterpret_cast<u32x2 *>(r.hi + r.lo_off + i) = l;
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

// fp32 plane channels [ch0, ch0+nvalid) -> S-plane channels [0, Cp_dst) (zero filled beyond nvalid)
__global__ void to_splane_kernel(PRef src, int nvalid, SRef dst, Geo g)
{
    const int t = blockIdx.x * blockDim.x + threadIdx.x, cg = blockIdx.y, b = blockIdx.z;
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

// ------------------------------------------------------------------------------------------------
// epilogues with optional fp32 and S-plane outputs
// ------------------------------------------------------------------------------------------------
// All auxiliary loads of a thread (residual input, skip accumulator, tanh/sigmoid) are issued first, into registers that
// overwrite the accumulators they are combined with, and only then the stores: the epilogue is HBM/latency bound at two
// waves per SIMD, so memory-level parallelism (64-128 loads in flight per lane) is what matters.
// the auxiliary values of the STORE / RESSKIP epilogues (accumulate-into input, residual input, skip accumulator) as the INITIAL
// value of the accumulators: the main loop then adds the products on top and the epilogue has nothing left to load
template <int EPI, int NI = 2>
__device__ __forceinline__ void conv_acc_init(const ConvGemmArgs &a, f32x16 (&acc)[2][NI], int t0, int m0, int b, int wr, int wc, int lane)
{
    // address = wave-uniform base of the 32-row block (64-bit, scalar registers) + a 32-bit per-lane offset: one VGPR per load
    // instead of a 64-bit pointer pair (the preload is the register peak of the store / residual+skip instantiations)
    const Geo g = a.g;
    const int col = lane & 31;
    const unsigned rowoff = (unsigned)(4 * (lane >> 5)) * (unsigned)g.P;          // acc_row's lane part
#pragma unroll
    for (int mi = 0; mi < 2; ++mi) {
        const int mb = m0 + wr * 64 + mi * 32;                                    // first row of this 32-row block (wave uniform)
        const float *base = nullptr;
        if (EPI == EPI_STORE) base = a.aux0.p ? paddr(a.aux0, g, b, mb, t0) : nullptr;
        else if (EPI == EPI_RESSKIP)                                              // nsplit is a multiple of 32: a block lies on one side
            base = mb < a.nsplit ? paddr(a.aux0, g, b, mb, t0) : (a.accumulate ? paddr(a.out1, g, b, mb - a.nsplit, t0) : nullptr);
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) {
            const int tl = wc * (32 * NI) + ni * 32 + col;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int rl = (r & 3) + 8 * (r >> 2);                            // acc_row's register part
                const unsigned off = rowoff + (unsigned)rl * (unsigned)g.P + (unsigned)tl;
                float x = 0.f;
                if (t0 + tl < g.T && mb + rl + 4 * (lane >> 5) < a.M && base) x = base[off];
                acc[mi][ni][r] = x;
            }
        }
    }
}

// NI = 32-column accumulator blocks per wave (2: the 128-column tile; 1: the 64-column tile of the small-grid launches)
template <int EPI, bool PRE = false, int NI = 2>
__device__ __forceinline__ void conv_epilogue_s(const ConvGemmArgs &a, const SRef &s0, f32x16 (&acc)[2][NI], int t0, int m0, int b,
                                                int wr, int wc, int lane)
{
    const Geo g = a.g;
    const int col = lane & 31, h = lane >> 5;
    if (EPI == EPI_GATE) {
        // This epilogue is VALU bound (measured: its arithmetic, not its stores, was a third of the launch), so it is written for
        // the VALU: every address is a wave-uniform 64-bit base + a 32-bit per-lane offset, the null checks of the optional planes
        // are hoisted out of the element loops, and the tanh / sigmoid chains of EIGHT outputs are straight-line code the
        // scheduler can interleave (one chain alone is a string of dependent quarter-rate v_exp / v_rcp).
        const int chb = (m0 >> 1) + wr * 32;                 // first gate channel of this wave (multiple of 32)
        float *b0 = a.out0.p ? paddr(a.out0, g, b, chb, t0) : nullptr;
        float *b1 = a.out1.p ? paddr(a.out1, g, b, chb, t0) : nullptr;
        float *b2 = a.out1.p ? paddr(a.out2, g, b, chb, t0) : nullptr;
        unsigned short *sh = s0.hi + s_index(s0, g, b, chb, t0);          // unit of (channel group of chb, column t0)
        unsigned short *sl = sh + s0.lo_off;
        const unsigned tl = (unsigned)(wc * (32 * NI) + col);
        const unsigned lane_off = (unsigned)(4 * h) * (unsigned)g.P + tl;
        const unsigned s_lane = tl * 8u + (unsigned)(4 * h);              // element offset inside the unit row
        const unsigned s_grp = (unsigned)g.P * 8u;                        // next channel group
#if defined(WG_OPT_SWAP_STORE)
        if (2 * chb >= a.M) return;                          // (2 Cd is a multiple of 64: a wave's 32 gate channels are all valid or none)
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) {
            const int t = t0 + wc * (32 * NI) + ni * 32 + col;
            const bool tok = t < g.T;
#pragma unroll
            for (int qq = 0; qq < 4; qq += 2) {
                float tw[8], sf[8], gv[8];
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    tw[i] = wg_tanh(acc[0][ni][4 * qq + i]);
                    sf[i] = wg_sigmoid(acc[1][ni][4 * qq + i]);
                    gv[i] = tw[i] * sf[i];
                }
                if (tok) {
#pragma unroll
                    for (int u = 0; u < 2; ++u) {
                        const unsigned off = lane_off + (unsigned)(8 * (qq + u)) * (unsigned)g.P + (unsigned)(ni * 32);
                        if (b0) {
#pragma unroll
                            for (int e = 0; e < 4; ++e) b0[off + (unsigned)e * (unsigned)g.P] = gv[4 * u + e];
                        }
                        if (b1) {
#pragma unroll
                            for (int e = 0; e < 4; ++e) {
                                b1[off + (unsigned)e * (unsigned)g.P] = tw[4 * u + e];
                                b2[off + (unsigned)e * (unsigned)g.P] = sf[4 * u + e];
                            }
                        }
                    }
                }
                // S-plane: the halves of groups qq and qq + 1 paired into whole 16-byte units (all lanes take part in the swap)
                u32x2 h0, l0, h1, l1;
                unsigned hh, ll;
                split2(gv[0], gv[1], hh, ll); h0[0] = hh; l0[0] = ll;
                split2(gv[2], gv[3], hh, ll); h0[1] = hh; l0[1] = ll;
                split2(gv[4], gv[5], hh, ll); h1[0] = hh; l1[0] = ll;
                split2(gv[6], gv[7], hh, ll); h1[1] = hh; l1[1] = ll;
                const u32x4 uh = pair_units(h0, h1), ul = pair_units(l0, l1);
                if (tok) {
                    const unsigned so = tl * 8u + (unsigned)(qq + h) * s_grp + (unsigned)(ni * 32 * 8);
                    *reinterpret_cast<u32x4 *>(sh + so) = uh;
                    *reinterpret_cast<u32x4 *>(sl + so) = ul;
                }
                __builtin_amdgcn_sched_barrier(0);           // eight outputs at a time: keeps the epilogue inside 128 VGPRs
            }
        }
#else
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) {
            const int t = t0 + wc * (32 * NI) + ni * 32 + col;
            if (t >= g.T) continue;
#pragma unroll
            for (int qq = 0; qq < 4; qq += 2) {
                float tw[8], sf[8], gv[8];
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    tw[i] = wg_tanh(acc[0][ni][4 * qq + i]);
                    sf[i] = wg_sigmoid(acc[1][ni][4 * qq + i]);
                    gv[i] = tw[i] * sf[i];
                }
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    const int q = qq + u;
                    if (2 * (chb + 8 * q + 4 * h) >= a.M) continue;
                    const unsigned off = lane_off + (unsigned)(8 * q) * (unsigned)g.P + (unsigned)(ni * 32);
#if !defined(WG_DBG_GATE_NOSTORE)
                    if (b0) {
#pragma unroll
                        for (int e = 0; e < 4; ++e) b0[off + (unsigned)e * (unsigned)g.P] = gv[4 * u + e];
                    }
                    if (b1) {
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            b1[off + (unsigned)e * (unsigned)g.P] = tw[4 * u + e];
                            b2[off + (unsigned)e * (unsigned)g.P] = sf[4 * u + e];
                        }
                    }
                    u32x2 vh, vl;
                    unsigned hh, ll;
                    split2(gv[4 * u], gv[4 * u + 1], hh, ll); vh[0] = hh; vl[0] = ll;
                    split2(gv[4 * u + 2], gv[4 * u + 3], hh, ll); vh[1] = hh; vl[1] = ll;
                    const unsigned so = s_lane + (unsigned)q * s_grp + (unsigned)(ni * 32 * 8);
                    *reinterpret_cast<u32x2 *>(sh + so) = vh;
                    *reinterpret_cast<u32x2 *>(sl + so) = vl;
#else
                    if (a.M == 12345 + q && tw[4 * u] + sf[4 * u + 1] + gv[4 * u + 2] + gv[4 * u + 3] == 3.f) b2[off] = gv[4 * u];
#endif
                }
                __builtin_amdgcn_sched_barrier(0);           // eight outputs at a time: keeps the epilogue inside 128 VGPRs
            }
        }
#endif
        return;
    }
    // ---- phase 1: loads ----
    f32x16 ax[2][NI];                // aux0 / skip accumulator
    f32x16 ay[2][NI];                // aux1 (DGATE only)
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) {
            const int t = t0 + wc * (32 * NI) + ni * 32 + col;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = m0 + wr * 64 + mi * 32 + acc_row(r, lane);
                float x = 0.f, y = 0.f;
                if (!PRE && t < g.T && m < a.M) {
                    if (EPI == EPI_STORE) {
                        if (a.aux0.p) x = *paddr(a.aux0, g, b, m, t);
                    } else if (EPI == EPI_RESSKIP) {
                        if (m < a.nsplit) x = *paddr(a.aux0, g, b, m, t);
                        else if (a.accumulate) x = *paddr(a.out1, g, b, m - a.nsplit, t);
                    } else if (EPI == EPI_DGATE) {
                        // wave-uniform 64-bit base of the 32-row block + one 32-bit per-lane offset shared by both planes: the 128
                        // loads in flight cost one address VGPR each, not a pointer pair (the allocator spilled those)
                        const int mb = m0 + wr * 64 + mi * 32;
                        const unsigned off = (unsigned)(acc_row(r, lane)) * (unsigned)g.P + (unsigned)(wc * (32 * NI) + ni * 32 + col);
                        x = paddr(a.aux0, g, b, mb, t0)[off];
                        y = paddr(a.aux1, g, b, mb, t0)[off];
                    }
                }
                ax[mi][ni][r] = x;
                if (EPI == EPI_DGATE) ay[mi][ni][r] = y;
            }
        }
    // ---- phase 2: combine + stores ----
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) {
            const int t = t0 + wc * (32 * NI) + ni * 32 + col;
            if (t >= g.T) continue;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int m = m0 + wr * 64 + mi * 32 + 8 * q + 4 * h;
                if (m >= a.M) continue;
                float o[4];
                if (EPI == EPI_STORE) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        o[e] = acc[mi][ni][4 * q + e] + ax[mi][ni][4 * q + e];
                        if (m + e < a.M && a.out0.p) *paddr(a.out0, g, b, m + e, t) = o[e];
                    }
                    if (s0.hi) s_store4(s0, g, b, m, t, o);
                } else if (EPI == EPI_RESSKIP) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) o[e] = acc[mi][ni][4 * q + e] + ax[mi][ni][4 * q + e];
                    if (m < a.nsplit) {
#pragma unroll
                        for (int e = 0; e < 4; ++e) *paddr(a.out0, g, b, m + e, t) = o[e];
                        if (s0.hi) s_store4(s0, g, b, m, t, o);
                    } else {
#pragma unroll
                        for (int e = 0; e < 4; ++e) *paddr(a.out1, g, b, m + e - a.nsplit, t) = o[e];
                    }
                } else if (EPI == EPI_DGATE) {
                    float o2[4];
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float v = acc[mi][ni][4 * q + e];
                        const float tw = ax[mi][ni][4 * q + e], sf = ay[mi][ni][4 * q + e];
                        o[e] = v * sf * (1.0f - tw * tw);
                        o2[e] = v * tw * sf * (1.0f - sf);
                        if (a.out0.p) {
                            *paddr(a.out0, g, b, m + e, t) = o[e];
                            *paddr(a.out0, g, b, a.nsplit + m + e, t) = o2[e];
                        }
                    }
                    s_store4(s0, g, b, m, t, o);
                    s_store4(s0, g, b, a.nsplit + m, t, o2);
                }
            }
        }
}

// ------------------------------------------------------------------------------------------------
// convgemm16s: A from the pre-split weight images, B from S-planes
// ------------------------------------------------------------------------------------------------
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
            acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f.al[mi], f.bh[ni], acc[mi][ni], 0, 0, 0);
            acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f.ah[mi], f.bl[ni], acc[mi][ni], 0, 0, 0);
            acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f.ah[mi], f.bh[ni], acc[mi][ni], 0, 0, 0);
        }
}

// Measured and dropped (all within +-10 % of this kernel, several slower): a 256x128 / 8-wave tile; LDS-DMA staging
// (global_load_lds with every 5th lane landing in the row pad); inline-asm loads with hand-counted s_waitcnt running one and
// two FULL chunks ahead (hipcc's own waitcnt bookkeeping collapses to vmcnt(0) across the loop back edge, so compiler-managed
// prefetch is only ~half a chunk deep -- but deeper prefetch bought nothing, i.e. L2 latency is not the limiter); and an
// LDS-free, barrier-free variant in which every wave streams its own fragments straight into registers (both operand formats
// are fragment shaped in memory) -- correct, 15 % slower.  Ablations: MFMA+LDS alone 92 us, operand staging alone 104 us (83 us
// when every load hits L1), together 140 us per launch of the dilated conv: the per-CU vector-memory -> VGPR -> LDS path
// (~31 B/clk/CU sustained) is as long as the matrix work and overlaps it poorly.  See DESIGN.md section 4.
template <int EPI>
__global__ __launch_bounds__(256) void convgemm16p_kernel(const ConvGemm16sArgs aa)
{
    constexpr int MT = 2;
    constexpr int NT = 128 * MT;
    constexpr int AIMG = MT * 64 * WG16_ROWB;
    constexpr int BUF = 2 * AIMG + 2 * WG16_IMG;
    constexpr int UPT = 512 / NT;                // B units per thread and image
    __shared__ __attribute__((aligned(16))) char smem[2 * BUF];
    const ConvGemmArgs &a = aa.c;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wr = wave >> 1, wc = wave & 1;
    const int t0 = blockIdx.x * WG_TILE, m0 = blockIdx.y * (64 * MT), b = blockIdx.z;
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

    constexpr int NSET = 1;       // staging register sets (global loads run one chunk ahead of the MFMAs)
    u32x4 ra_hi[NSET][2], ra_lo[NSET][2], rb_hi[NSET][UPT], rb_lo[NSET][UPT];
    int cur_seg = 0, cur_c = 0, chunk = 0;
    const int bt = tid & 127, cg0 = tid >> 7;          // B units: MT=2: (cg0, bt) and (cg0 + 2, bt); MT=4: (cg0, bt), cg0 = 0..3

    auto load_chunk = [&](auto SET) {
        constexpr int S = decltype(SET)::value;
        const int nch = a.seg[cur_seg].nch, shift = a.seg[cur_seg].shift;
        const SSeg ss = aa.sseg[cur_seg];
        const int nvalid = min(WG16_BK, nch - cur_c);
        const unsigned short *ih = aa.img + ((size_t)chunk * a.lda + m0) * WG16_BK;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int p = tid + NT * j;
            ra_hi[S][j] = *reinterpret_cast<const u32x4 *>(ih + (size_t)p * 8);
            ra_lo[S][j] = *reinterpret_cast<const u32x4 *>(ih + aa.img_stride + (size_t)p * 8);
        }
        const unsigned short *p0 = ss.hi + (((size_t)b * (ss.Cp >> 3) + ((ss.ch0 + cur_c) >> 3) + cg0) * g.P + g.H + t0 + shift + bt) * 8;
        if (UPT == 2) {
            rb_hi[S][0] = *reinterpret_cast<const u32x4 *>(p0);       // cg0 in {0,1}: always valid (chunks hold >= 16 channels)
            rb_lo[S][0] = *reinterpret_cast<const u32x4 *>(p0 + ss.lo_off);
            u32x4 vh = {0u, 0u, 0u, 0u}, vl = {0u, 0u, 0u, 0u};
            if (nvalid > 16) {
                const unsigned short *p1 = p0 + (size_t)2 * g.P * 8;
                vh = *reinterpret_cast<const u32x4 *>(p1);
                vl = *reinterpret_cast<const u32x4 *>(p1 + ss.lo_off);
            }
            rb_hi[S][UPT - 1] = vh; rb_lo[S][UPT - 1] = vl;
        } else {
            u32x4 vh = {0u, 0u, 0u, 0u}, vl = {0u, 0u, 0u, 0u};
            if (cg0 * 8 < nvalid) {
                vh = *reinterpret_cast<const u32x4 *>(p0);
                vl = *reinterpret_cast<const u32x4 *>(p0 + ss.lo_off);
            }
            rb_hi[S][0] = vh; rb_lo[S][0] = vl;
        }
        ++chunk;
        cur_c += WG16_BK;
        if (cur_c >= nch) { cur_c = 0; ++cur_seg; }
    };
    auto store_chunk = [&](auto SET, int buf) {
        constexpr int S = decltype(SET)::value;
        char *sb = smem + buf * BUF;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int p = tid + NT * j;
            const int off = wg16_a_off(p);
            *reinterpret_cast<u32x4 *>(sb + off) = ra_hi[S][j];
            *reinterpret_cast<u32x4 *>(sb + AIMG + off) = ra_lo[S][j];
        }
#pragma unroll
        for (int j = 0; j < UPT; ++j) {
            char *q = sb + 2 * AIMG + bt * WG16_ROWB + (cg0 + 2 * j) * 16;
            *reinterpret_cast<u32x4 *>(q) = rb_hi[S][j];
            *reinterpret_cast<u32x4 *>(q + WG16_IMG) = rb_lo[S][j];
        }
    };

    const int r = lane & 31, h = lane >> 5;
    const int ao = (wr * 64 + r) * WG16_ROWB + h * 16, bo = (wc * 64 + r) * WG16_ROWB + h * 16;
    using S0 = std::integral_constant<int, 0>;
    using S1 = std::integral_constant<int, NSET - 1>;

    // one steady-state iteration: multiply chunk c (buffer c&1) while chunk c+1 goes registers -> LDS and (DEPTH2) chunk c+2
    // goes global -> registers.  LSET: register set loaded in this iteration, WSET: set written to LDS.
    auto iter = [&](auto LSET, auto WSET, int c, bool do_load) {
        const char *sb = smem + (c & 1) * BUF;
        Frags16 f0, f1;
        if (do_load) load_chunk(LSET);
        read_frags16(f0, sb, sb + AIMG, sb + 2 * AIMG, sb + 2 * AIMG + WG16_IMG, ao, bo);
        read_frags16(f1, sb, sb + AIMG, sb + 2 * AIMG, sb + 2 * AIMG + WG16_IMG, ao + 32, bo + 32);
        mfma12(f0, acc);
        store_chunk(WSET, (c & 1) ^ 1);
        mfma12(f1, acc);
        // pin the interleave: loads, step-0 fragments, then {3 MFMA, 2 DS reads} x4, {3 MFMA, 2 DS writes} x4
        __builtin_amdgcn_sched_group_barrier(0x020, 4 + 2 * UPT, 0);
        __builtin_amdgcn_sched_group_barrier(0x100, 8, 0);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            __builtin_amdgcn_sched_group_barrier(0x008, 3, 0);
            __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
        }
        // LDS writes of the next chunk: 4 (A) + 2*UPT (B) per thread, two per MFMA group
        __builtin_amdgcn_sched_group_barrier(0x008, 3, 0);
        __builtin_amdgcn_sched_group_barrier(0x200, 2, 0);
        __builtin_amdgcn_sched_group_barrier(0x008, 3, 0);
        __builtin_amdgcn_sched_group_barrier(0x200, 2, 0);
        __builtin_amdgcn_sched_group_barrier(0x008, 3, 0);
        __builtin_amdgcn_sched_group_barrier(0x200, 2, 0);
        __builtin_amdgcn_sched_group_barrier(0x008, 3, 0);
        if (UPT == 2) __builtin_amdgcn_sched_group_barrier(0x200, 2, 0);
        __syncthreads();
    };

    load_chunk(S0{});
    store_chunk(S0{}, 0);
    __syncthreads();
    for (int c = 0; c + 1 < nchunks; ++c) iter(S0{}, S0{}, c, true);
    {
        const char *sb = smem + ((nchunks - 1) & 1) * BUF;
        Frags16 f0, f1;
        read_frags16(f0, sb, sb + AIMG, sb + 2 * AIMG, sb + 2 * AIMG + WG16_IMG, ao, bo);
        read_frags16(f1, sb, sb + AIMG, sb + 2 * AIMG, sb + 2 * AIMG + WG16_IMG, ao + 32, bo + 32);
        mfma12(f0, acc);
        mfma12(f1, acc);
    }
    conv_epilogue_s<EPI>(a, aa.s0, acc, t0, m0, b, wr, wc, lane);
}

// ------------------------------------------------------------------------------------------------
// convgemm16w: wave-specialised variant.  A workgroup is 8 waves on one CU: waves 0-3 only multiply (ds_read + MFMA on the
// 128x128 tile, one per SIMD), waves 4-7 only move operands (global -> registers -> LDS, two chunks in flight, counted
// waits).  The two roles meet at one barrier per chunk.  The per-CU vector-memory -> VGPR -> LDS path and the matrix pipe are
// both busy for about the same time per chunk; in the symmetric kernels every wave alternates between the two and the
// phases overlap poorly, here the overlap is structural.
// Where the time of this kernel goes (gate conv, 27 chunks, 1536 tiles; rocprofv3 PMC: clock 2.06 GHz, matrix pipe 45-53 % busy).
// Timing builds -DWG_DBG_NOLOAD / NOMFMA / NOEPI / GATE_NOSTORE; LDS and MFMA issue rates from tools/experiments/lds_probe.hip and
// mfma_probe.hip; phase stamps per workgroup from -DWG_DBG_TRACE + tools/experiments/conv_trace.py:
//   main loops only (no epilogue)                                     85 us   (loaders alone 70 us, compute waves alone 75 us,
//                                                                              MFMA time 58 us)
//   + the epilogue                                                   125-133 us
//   * the loaders' 70 us is the L2 -> CU path itself: 1.36 GB of operands per launch at the 66-76 GB/s per CU that path delivers;
//     with the matrix pipe active a CU sustains ~46 GB/s;
//   * ds_read_b128 costs 4 cycles (256 B/clk), ds_write_b128 8; the LDS pipe is ~35 % busy -- an earlier note here called the LDS
//     port co-critical with the matrix pipe; the counters do not support that.  What the counters did show: 20 % of the LDS
//     cycles were bank conflicts of the four-lanes-per-row staging write of the weight images (fixed: wg16_a_off);
//   * chained MFMAs issue at the full rate (32 cycles each, any number of accumulators in rotation);
//   * the compute waves lost ~4 LDS round trips per chunk to the compiler's schedule of the plain loop at 128 VGPRs (fixed: the
//     register pipeline in the kernel body); removing every main-loop barrier saves 4 %;
//   * the epilogue of the gate conv is VALU bound (its arithmetic ~35 us, its stores ~10 us per launch): __frcp_rn was the IEEE
//     division sequence, every store carried 64-bit pointer arithmetic (both fixed); while a workgroup is in its epilogue its
//     operand stream pauses (both LDS buffers full), and the stream is what bounds the main loop;
//   * the 256 workgroups dispatched first win issue arbitration against their CU mates: 26-29 us per tile against 55 us, the late
//     half finishes alone at 24 us per tile.  s_setprio on the late half equalises the tiles (38 us) but is neutral to slower
//     in a training step; a start-up stagger, a 4:2 tile split between the halves and dynamic variants are all neutral or slower.
// Register note: the store and residual+skip instantiations take their auxiliary values (accumulate-into input, residual input,
// skip accumulator) as the INITIAL value of the accumulators (conv_acc_init): no epilogue loads, 128 VGPRs, two workgroups per CU
// (store/dgrad conv 143 -> 135 us, residual+skip 95 -> 93.5 us; -DWG_OPT_NO_ACCINIT restores the epilogue loads).  The gate
// backward multiplies by its two auxiliary tensors, holds all 128 values per lane at once (220 VGPRs) and runs one workgroup per
// CU -- deliberately: splitting the epilogue per 32x32 block and capping the kernel at 128 VGPRs restores two
// workgroups per CU but leaves 16-32 loads in flight per lane, and these launches are bound by their epilogue's HBM traffic
// (residual+skip 92 -> 102 us, gate backward 87 -> 119 us).  A 4-stage (4 chunks in flight) loader for launches with fewer
// tiles than CUs (single-utterance synthesis) was also measured: 2.85 -> 2.70 MHz, not kept.
// Where the residual+skip launch (K = 256, 8 chunks) spends its 89-97 us: main loop alone 39 us, + the accumulator-init loads 29 us,
// + the stores 29 us -- additive.  The HBM pattern is not the issue (a copy kernel with the same lane -> element mapping moves
// the planes at 6.6 TB/s, the same as a float4 row-contiguous one; tools/experiments/plane_copy_probe.hip), and the phases do not
// add up because identical workgroups run in lockstep either: delaying every second workgroup of a CU by a quarter to a full tile
// time only adds the delay.  Every phase is bound by the memory system (HBM for the planes, L2 -> CU for the operands).
// A 256(M) x 128(T) workgroup tile at ONE workgroup per CU (4 compute waves with 128x64 tiles = 0.75x the LDS and L1 bytes per
// MFMA, fragments double-buffered in the 256-register budget, barrier between the two k-steps, same asm loaders with 12 loads per
// lane) was built and is bit-identical in results: 145 us for the gate conv against 128 us (compute waves alone 124 us, loaders
// alone 101 us).  With a single workgroup on the CU nothing runs under a tile's epilogue (tanh/sigmoid + 64-192 KB of stores) or
// its first loads; two co-resident workgroups hide exactly that, and two workgroups cap a wave at 128 registers, i.e. at the
// 64x64 wave tile used here.
// The persistent form (workgroups walking tiles w, w + G, ...; the loaders treat all their tiles as one chunk stream) first lost
// (the tile loop around the epilogue cost registers: 177-239 VGPRs -> one workgroup per CU) and is now the default for the store,
// gate and residual+skip instantiations: with the accumulator preload on 32-bit offsets, opaque per-tile lane copies against
// hoisting and the epilogue kept out of the loop-carried state they stay at 126 VGPRs (store/dgrad -9 %, residual+skip -10 %).
// The barrier of chunk c between its two k-steps (fragments of chunk c+1 fetched under the MFMAs of k-step 1) first needed 132
// VGPRs (one workgroup per CU: 148 us); with A0 reloaded in place and only A1 / B double-buffered it fits 125 and is what the
// kernel body does now.
// ------------------------------------------------------------------------------------------------
struct Stage8 {
    u32x4 ah[2], al[2], bh[2], bl[2];
};
struct Stage6 {                      // the 64-column tile: one B unit per lane and image
    u32x4 ah[2], al[2], bh[1], bl[1];
};
template <int NI> struct StageOf { typedef Stage8 type; };
template <> struct StageOf<1> { typedef Stage6 type; };
#define WG_STAGE_REGS(s) "+v"(s.ah[0]), "+v"(s.ah[1]), "+v"(s.al[0]), "+v"(s.al[1]), "+v"(s.bh[0]), "+v"(s.bh[1]), "+v"(s.bl[0]), "+v"(s.bl[1])
__device__ __forceinline__ void asm_wait_keep8(Stage8 &s) { asm volatile("s_waitcnt vmcnt(8)" : WG_STAGE_REGS(s)::"memory"); }
__device__ __forceinline__ void asm_wait_stage(Stage8 &s) { asm_wait_keep8(s); }
__device__ __forceinline__ void asm_wait_stage(Stage6 &s)
{
    asm volatile("s_waitcnt vmcnt(6)" : "+v"(s.ah[0]), "+v"(s.ah[1]), "+v"(s.al[0]), "+v"(s.al[1]), "+v"(s.bh[0]), "+v"(s.bl[0])::"memory");
}
template <int NI> struct FragsW {
    bf16x8 ah[2], al[2], bh[NI], bl[NI];
};
template <int NI>
__device__ __forceinline__ void read_frags_w(FragsW<NI> &f, const char *Ahi, const char *Alo, const char *Bhi, const char *Blo, int ao, int bo)
{
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        f.ah[i] = *reinterpret_cast<const bf16x8 *>(Ahi + ao + i * 32 * WG16_ROWB);
        f.al[i] = *reinterpret_cast<const bf16x8 *>(Alo + ao + i * 32 * WG16_ROWB);
    }
#pragma unroll
    for (int i = 0; i < NI; ++i) {
        f.bh[i] = *reinterpret_cast<const bf16x8 *>(Bhi + bo + i * 32 * WG16_ROWB);
        f.bl[i] = *reinterpret_cast<const bf16x8 *>(Blo + bo + i * 32 * WG16_ROWB);
    }
}
template <int NI>
__device__ __forceinline__ void mfma_w(const FragsW<NI> &f, f32x16 (&acc)[2][NI])
{
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) {
            acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f.al[mi], f.bh[ni], acc[mi][ni], 0, 0, 0);
            acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f.ah[mi], f.bl[ni], acc[mi][ni], 0, 0, 0);
            acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f.ah[mi], f.bh[ni], acc[mi][ni], 0, 0, 0);
        }
}

// NI = 2: 128 x 128 tile (the training shapes).  NI = 1: 128 x 64 tile for launches that would otherwise leave most of the chip idle
// (single-utterance synthesis, WSRGlow's 512-step segments, WaveFlow's row-by-row inverse): twice the workgroups, half the MFMAs
// per chunk and wave, one B unit per loader lane and image (6 loads per chunk).
#if defined(WG_DBG_TRACE)      // phase timestamps of the gate conv (100 MHz wall clock): [workgroup][16] = start, first barrier, then per
                               // tile: main loop done, epilogue done.  Read back by wg_dbg_trace_read (tools/experiments/conv_trace.py).
__device__ unsigned long long wg_dbg_trace[512 * 16];       // 100 MHz wall clock
__device__ unsigned long long wg_dbg_trace_cyc[512 * 16];   // shader cycles (s_memtime): cycles / wall = the clock the chip holds in the phase
#define WG_TRACE(slot) do { if (EPI == EPI_GATE && NI == 2 && lane == 0 && wave == 0 && (slot) < 16) { \
        wg_dbg_trace[blockIdx.x * 16 + (slot)] = wall_clock64(); wg_dbg_trace_cyc[blockIdx.x * 16 + (slot)] = clock64(); } } while (0)
#else
#define WG_TRACE(slot) do { } while (0)
#endif
#if defined(WG_DBG_NOBAR)      // timing experiment only (results are garbage): how much of a launch is barrier skew?
#define WG16W_BAR() do { } while (0)
#else
#define WG16W_BAR() __syncthreads()
#endif
template <int EPI, int NI>
__global__ __launch_bounds__(512) void convgemm16w_kernel(const ConvGemm16sArgs aa)
{
    typedef typename StageOf<NI>::type Stage;
    constexpr int AIMG = WG16_IMG;                            // 128 rows x 80 B
    constexpr int BIMG = 64 * NI * WG16_ROWB;
    constexpr int BUF = 2 * AIMG + 2 * BIMG;
    constexpr int TT = 64 * NI;                               // columns per tile
    __shared__ __attribute__((aligned(16))) char smem[2 * BUF];
    const ConvGemmArgs &a = aa.c;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const Geo g = a.g;
    int nchunks = 0;
    for (int s = 0; s < a.nseg; ++s) nchunks += (a.seg[s].nch + WG16_BK - 1) / WG16_BK;
    // Persistent over tiles: workgroup w of G takes tiles w, w + G, ... in the order a 3-D grid would have dispatched them (time
    // tile fastest, so the row tiles of one (plane row, time tile) stay 16 ids apart = on one XCD and share their B operand in its
    // L2).  The loaders treat the chunks of all their tiles as ONE stream: while the compute waves are in the epilogue of tile
    // i, the first chunks of tile i+1 are already being staged, and the epilogue's stores drain under the next main loop instead of
    // in a chip-wide burst at the end of every workgroup (measured before: 48 of 133 us of the gate launch were that burst).
    // (the gate backward keeps one workgroup per tile on a 3-D grid: its epilogue holds 128 auxiliary loads in flight next to the
    // accumulators, a tile loop around that costs registers it does not have, and its launches are bound by those loads)
    constexpr bool PERSIST = EPI != EPI_DGATE;
    const int ntiles = aa.ntx * aa.nty * aa.ntz, G = (int)gridDim.x;
    const int mine = PERSIST ? (ntiles - 1 - (int)blockIdx.x) / G + 1 : 1;   // host guarantees G <= ntiles
    const int total = mine * nchunks;                                  // this workgroup's chunk stream
    auto tile_at = [&](int k, int &t0, int &m0, int &b) {
        if constexpr (PERSIST) {
            const int id = (int)blockIdx.x + k * G;
            const int tx = id % aa.ntx, q = id / aa.ntx, ty = q % aa.nty, tz = q / aa.nty;
            t0 = tx * TT; m0 = ty * WG_TILE;
            b = a.row_sel1 ? tz * g.rows + a.row_sel1 - 1 : tz;
        } else {
            t0 = blockIdx.x * TT; m0 = blockIdx.y * WG_TILE;
            b = a.row_sel1 ? (int)blockIdx.z * g.rows + a.row_sel1 - 1 : (int)blockIdx.z;
        }
    };

    if (wave >= 4) {
        // ------------------------------- loader waves -------------------------------
        const int lt = tid - 256;
        const int bt = NI == 2 ? (lt & 127) : (lt & 63), cg0 = NI == 2 ? (lt >> 7) : (lt >> 6);     // B unit: position, k-group
        int cur_seg = 0, cur_c = 0, chunk = 0;                // position inside the current tile
        int gchunk = 0, tk = 0, t0, m0, b;                    // position in the stream; tile being loaded
        tile_at(0, t0, m0, b);
#if defined(WG_OPT_ROT)
        // Every tile's K loop starts at a workgroup-dependent chunk and wraps around: the 64 workgroups of an XCD would otherwise
        // sweep the SAME weight-image lines in step (every workgroup reads chunk c of the A image at about the same time).
        const int rot = (int)(((unsigned)blockIdx.x >> 3) * 5u % (unsigned)nchunks);
        int rot_seg = 0, rot_c = 0, cnt = 0;
        for (int i = 0; i < rot; ++i) {
            rot_c += WG16_BK;
            if (rot_c >= a.seg[rot_seg].nch) { rot_c = 0; ++rot_seg; }
        }
        cur_seg = rot_seg; cur_c = rot_c; chunk = rot;
#endif
        const unsigned voff_a = (unsigned)lt * 16u;
        const int a_off0 = wg16_a_off(lt);                    // consecutive lanes -> consecutive 80-byte LDS rows: conflict-free staging
        const unsigned voff_b = (unsigned)((cg0 * g.P + bt) * 16);
#if defined(WG_DBG_NOLOAD)     // timing experiment only: the loaders write whatever their staging registers hold
#define WG_LD(dst, base, voff) asm volatile("" : "=v"(dst) : "v"(voff), "s"(base))
#else
#define WG_LD(dst, base, voff) asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(dst) : "v"(voff), "s"(base) : "memory")
#endif
        const unsigned short *zsrc = aa.sseg[0].hi;           // plane position 0 of the first operand: always-zero halo
        // Every call issues exactly 4 + 2 NI loads in straight-line code: past the last chunk, and for the missing half of a
        // 16-channel chunk, base and offset are SELECTED to the zero halo.  No branch may sit between an asm load and its counted
        // wait -- the compiler treats an asm output as valid at once and is free to copy it on a branch arm before the data has
        // landed (tools/check_asm_loads.py walks the ISA for exactly that).
        auto issue = [&](Stage &st) {
            const bool live = gchunk < total;
            const int sg = min(cur_seg, a.nseg - 1);
            const int nch = a.seg[sg].nch, shift = a.seg[sg].shift;
            const SSeg ss = aa.sseg[sg];
            // source plane row of this segment: the tile's own row, another row of the same item (2-D taps) or the item's single row
            int bsrc = b;
            bool rowok = true;
            if (g.rows > 0) {
                const int item = b / g.rows, r = b - item * g.rows + ss.row_off;
                rowok = r >= 0 && r < g.rows;
                bsrc = ss.per_item ? item : b + ss.row_off;
            }
#if defined(WG_DBG_HALFB)      // timing experiment only (garbage results): the second half of every B chunk is not fetched (-25 % operand bytes)
            const bool blive = live && rowok, full = false;
#else
            const bool blive = live && rowok, full = blive && (nch - cur_c > 16);
#endif
            const unsigned short *ih = aa.img + ((size_t)chunk * a.lda + m0) * WG16_BK, *il = ih + aa.img_stride;
            const unsigned short *row0 = ss.hi + ((size_t)bsrc * (ss.Cp >> 3) + ((ss.ch0 + cur_c) >> 3)) * g.P * 8;   // p = 0: zero halo
#if defined(WG_DBG_HALFA)      // timing experiment only (garbage results): half of every A chunk is not fetched (-25 % operand bytes)
            const unsigned short *pa0 = live ? ih : zsrc, *pa1 = zsrc;
            const unsigned short *pl0 = live ? il : zsrc, *pl1 = zsrc;
#else
            const unsigned short *pa0 = live ? ih : zsrc, *pa1 = live ? ih + 2048 : zsrc;
            const unsigned short *pl0 = live ? il : zsrc, *pl1 = live ? il + 2048 : zsrc;
#endif
            const unsigned va = live ? voff_a : 0u;
            WG_LD(st.ah[0], pa0, va);   WG_LD(st.ah[1], pa1, va);
            WG_LD(st.al[0], pl0, va);   WG_LD(st.al[1], pl1, va);
            if constexpr (NI == 2) {
                const unsigned short *b0 = row0 + (size_t)(g.H + t0 + shift) * 8, *b0l = b0 + ss.lo_off;
                const unsigned short *b1 = b0 + (size_t)2 * g.P * 8, *b1l = b1 + ss.lo_off;
                const unsigned short *pb0 = blive ? b0 : zsrc, *pb0l = blive ? b0l : zsrc;
                const unsigned short *pb1 = full ? b1 : zsrc, *pb1l = full ? b1l : zsrc;
                const unsigned vb = blive ? voff_b : 0u, vb1 = full ? voff_b : 0u;
                WG_LD(st.bh[0], pb0, vb);   WG_LD(st.bl[0], pb0l, vb);
                WG_LD(st.bh[1], pb1, vb1);  WG_LD(st.bl[1], pb1l, vb1);
            } else {
                // one unit per lane: k-groups 2 and 3 of a 16-channel chunk do not exist -> those LANES read the zero halo
                // (offset 0 from the plane row's position 0), the base stays uniform
                const unsigned short *pb = blive ? row0 : zsrc, *pbl = blive ? row0 + ss.lo_off : zsrc;
                const bool lane_ok = blive && (cg0 < 2 || full);
                const unsigned vb = lane_ok ? voff_b + (unsigned)((g.H + t0 + shift) * 16) : 0u;
                WG_LD(st.bh[0], pb, vb);    WG_LD(st.bl[0], pbl, vb);
            }
            if (live) {
                ++gchunk;
                ++chunk;
                cur_c += WG16_BK;
                if (cur_c >= nch) { cur_c = 0; ++cur_seg; }
#if defined(WG_OPT_ROT)
                if (chunk == nchunks) { chunk = 0; cur_seg = 0; cur_c = 0; }     // wrap around inside the tile
                if (++cnt == nchunks) {                       // next tile of this workgroup: back to the rotated start
                    cnt = 0; chunk = rot; cur_seg = rot_seg; cur_c = rot_c;
                    tk = min(tk + 1, mine - 1);
                    tile_at(tk, t0, m0, b);
                }
#else
                if (chunk == nchunks) {                       // next tile of this workgroup (past the last one: never loaded from)
                    chunk = 0; cur_seg = 0; cur_c = 0;
                    tk = min(tk + 1, mine - 1);
                    tile_at(tk, t0, m0, b);
                }
#endif
            }
        };
#undef WG_LD
        auto write = [&](const Stage &st, int buf) {
            char *sb = smem + buf * BUF;
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int off = a_off0 + 32 * j;               // piece lt + 256 j -> row lt & 127, k-group (lt >> 7) + 2 j (wg16_a_off)
                *reinterpret_cast<u32x4 *>(sb + off) = st.ah[j];
                *reinterpret_cast<u32x4 *>(sb + AIMG + off) = st.al[j];
            }
#pragma unroll
            for (int j = 0; j < NI; ++j) {
                char *q = sb + 2 * AIMG + bt * WG16_ROWB + (cg0 + 2 * j) * 16;
                *reinterpret_cast<u32x4 *>(q) = st.bh[j];
                *reinterpret_cast<u32x4 *>(q + BIMG) = st.bl[j];
            }
        };
        Stage s0, s1;
        issue(s0);                                           // chunk 0
        issue(s1);                                           // chunk 1
        asm_wait_stage(s0);
        write(s0, 0);
        issue(s0);                                           // chunk 2
        WG16W_BAR();                                     // buffer 0 ready
        // iteration c: compute waves multiply buffer c&1; we write chunk c+1 (landed) into the other buffer and issue chunk c+3
        auto iter = [&](Stage &st, int c) {
            asm_wait_stage(st);
            write(st, (c & 1) ^ 1);
            issue(st);
            WG16W_BAR();
        };
        // always in pairs (an even chunk count ends with one spare write of zero-halo data into the idle buffer, and the compute
        // waves take one matching extra barrier): the loop body stays branch-free between loads and waits
        for (int c = 0; c + 1 < total; c += 2) {
            iter(s1, c);
            iter(s0, c + 1);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // drain the trailing zero-halo loads before the wave ends
        return;
    }
    // ------------------------------- compute waves -------------------------------
    const int wr = wave >> 1, wc = wave & 1;
    f32x16 acc[2][NI];
    constexpr bool PRE = (EPI == EPI_STORE || EPI == EPI_RESSKIP)
#if defined(WG_OPT_NO_ACCINIT)
                         && false
#endif
        ;
    const int r = lane & 31, h = lane >> 5;
    const int ao = (wr * 64 + r) * WG16_ROWB + h * 16, bo = (wc * 32 * NI + r) * WG16_ROWB + h * 16;
    // Register-pipelined k-steps.  A chunk is two k-steps of 16; a k-step is two groups of 3 NI MFMAs: G0 = rows 0-31 of the wave
    // tile (fragments A0) and G1 = rows 32-63 (A1), both against the step's B fragments.  The fragments of step s+1 are fetched
    // under the MFMAs of step s: B and A1 into a second register set at the start of the step, A0 into its own registers as soon
    // as G0 is issued.  64 accumulators + 14 fragment quads = 120 VGPRs: two workgroups per CU keep fitting, and no MFMA group
    // starts by waiting for an LDS round trip (the compiler's own schedule of the plain loop, short of registers, exposed about
    // four per chunk).  The barrier of chunk c sits between its k-steps: by then every fragment of chunk c is in registers, so
    // the loaders may refill that buffer, and chunk c+1 (written during the first k-step) may be read.
#define WG16W_SB() __builtin_amdgcn_sched_barrier(0)
    bf16x8 a0h, a0l, a1h0, a1l0, a1h1, a1l1, bh0[NI], bl0[NI], bh1[NI], bl1[NI];
    auto rd = [&](const char *q) { return *reinterpret_cast<const bf16x8 *>(q); };
    auto grp = [&](const bf16x8 &xh, const bf16x8 &xl, const bf16x8 (&yh)[NI], const bf16x8 (&yl)[NI], f32x16 (&d)[NI]) {
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) {
            d[ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xl, yh[ni], d[ni], 0, 0, 0);
            d[ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xh, yl[ni], d[ni], 0, 0, 0);
            d[ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xh, yh[ni], d[ni], 0, 0, 0);
        }
    };
    int gc = 0;                                              // chunk index in this workgroup's stream; its buffer is gc & 1
    auto do_tile = [&](int k) {
        int t0, m0, b;
        tile_at(k, t0, m0, b);
        int ln = lane;
        // opaque per tile + nothing scheduled across: the per-lane addressing of the accumulator preload and of the epilogue is
        // recomputed per tile instead of living across the main loop, and the preload of tile k+1 is not hoisted above the
        // epilogue of tile k (that doubled the accumulators and cost the second workgroup per CU)
        if (PERSIST) {
            asm volatile("" : "+v"(ln)::"memory");
            WG16W_SB();
        }
        if (PRE) {
            conv_acc_init<EPI, NI>(a, acc, t0, m0, b, wr, wc, ln);
        } else {
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < NI; ++j)
#pragma unroll
                    for (int q = 0; q < 16; ++q) acc[i][j][q] = 0.f;
        }
        if (k == 0) { WG_TRACE(0); WG16W_BAR(); WG_TRACE(1); } // buffer 0 ready (later tiles: published by the previous chunk's barrier)
        {
            const char *pa = smem + (gc & 1) * BUF + ao, *pb = smem + (gc & 1) * BUF + 2 * AIMG + bo;
#pragma unroll
            for (int i = 0; i < NI; ++i) { bh0[i] = rd(pb + i * 32 * WG16_ROWB); bl0[i] = rd(pb + BIMG + i * 32 * WG16_ROWB); }
            a1h0 = rd(pa + 32 * WG16_ROWB); a1l0 = rd(pa + AIMG + 32 * WG16_ROWB);
            a0h = rd(pa); a0l = rd(pa + AIMG);
        }
#if defined(WG_DBG_NOMFMA)     // timing experiment only: the compute waves just keep the barrier protocol
        for (int c = 0; c < nchunks; ++c, ++gc)
            if (gc + 1 < total || !(total & 1)) WG16W_BAR();
        if (false)
#endif
        for (int c = 0; c < nchunks; ++c, ++gc) {
            const char *pa = smem + (gc & 1) * BUF + ao, *pb = smem + (gc & 1) * BUF + 2 * AIMG + bo;
            const char *na = smem + ((gc & 1) ^ 1) * BUF + ao, *nb = smem + ((gc & 1) ^ 1) * BUF + 2 * AIMG + bo;
            // ---- k-step 0 (fragments *0), fetching k-step 1 of this chunk (fragments *1) ----
#pragma unroll
            for (int i = 0; i < NI; ++i) { bh1[i] = rd(pb + 32 + i * 32 * WG16_ROWB); bl1[i] = rd(pb + 32 + BIMG + i * 32 * WG16_ROWB); }
            a1h1 = rd(pa + 32 + 32 * WG16_ROWB); a1l1 = rd(pa + 32 + AIMG + 32 * WG16_ROWB);
            WG16W_SB();
            grp(a0h, a0l, bh0, bl0, acc[0]);
            WG16W_SB();
            a0h = rd(pa + 32); a0l = rd(pa + 32 + AIMG);
            WG16W_SB();
            grp(a1h0, a1l0, bh0, bl0, acc[1]);
            WG16W_SB();
            if (gc + 1 < total || !(total & 1)) WG16W_BAR();     // matches the loaders' barrier of iteration gc (pairs: see there)
            // ---- k-step 1 (fragments *1), fetching k-step 0 of the next chunk (fragments *0) ----
            // (unconditional: after a tile's last chunk these read LDS that nothing uses -- a branch here would make the compiler
            // drain every outstanding read at the join; the next tile starts with its own fetch)
#pragma unroll
            for (int i = 0; i < NI; ++i) { bh0[i] = rd(nb + i * 32 * WG16_ROWB); bl0[i] = rd(nb + BIMG + i * 32 * WG16_ROWB); }
            a1h0 = rd(na + 32 * WG16_ROWB); a1l0 = rd(na + AIMG + 32 * WG16_ROWB);
            WG16W_SB();
            grp(a0h, a0l, bh1, bl1, acc[0]);
            WG16W_SB();
            a0h = rd(na); a0l = rd(na + AIMG);
            WG16W_SB();
            grp(a1h1, a1l1, bh1, bl1, acc[1]);
            WG16W_SB();
        }
        WG_TRACE(2 + 2 * k);
#if defined(WG_DBG_NOEPI)      // timing experiment only: one store per lane keeps the accumulators alive
        if (acc[0][0][0] + acc[1][0][0] + acc[0][NI - 1][5] + acc[1][NI - 1][7] == 12345.f) a.out0.p[lane] = 1.f;
#else
        int le = lane;
        if (PERSIST) asm volatile("" : "+v"(le)::"memory");   // (a second opaque copy: the epilogue's per-lane offsets are computed here, per
                                                              // tile, neither shared with the preload nor hoisted out of the tile loop)
        conv_epilogue_s<EPI, PRE, NI>(a, aa.s0, acc, t0, m0, b, wr, wc, le);
#endif
        WG_TRACE(3 + 2 * k);
        if (PERSIST) WG16W_SB();
    };
    if constexpr (PERSIST) {
        for (int k = 0; k < mine; ++k) do_tile(k);
    } else {
        do_tile(0);
    }
#undef WG16W_SB
}

// ------------------------------------------------------------------------------------------------
// convgemm16d: LDS-DMA loader ring + register-pipelined compute waves.  Six waves per workgroup, two workgroups per CU:
//   waves 0-3  multiply; fragments of the next k-step are fetched under the twelve MFMAs of the current one (the barrier of
//              chunk c sits between its two k-steps, so a wave never starts a chunk by waiting for LDS);
//   wave 4     streams the A images (weights hi|lo) of the next chunk global -> LDS with global_load_lds_dwordx4,
//   wave 5     the B images (S-planes hi|lo): no staging VGPRs, no ds_write, nothing of the copy passes through a SIMD's
//              register file while its matrix pipe works.
// An LDS-DMA writes lane-linear (base + lane * 16 B), so the images are unpadded 64-byte rows and the bank-conflict-free
// order is an XOR swizzle applied on BOTH sides: lane l of a DMA fetches k-group (l & 3) ^ ((row >> 2) & 3) of its row, and a
// fragment read of k-group q of row r goes to slot q ^ ((r >> 2) & 3).
// Ordering: a loader wave waits vmcnt(0) for its own DMAs, then takes the barrier; a compute wave reads only after that
// barrier (cdna_hip_programming.md section 5, "Read a staged buffer one phase AFTER the wait that retires it").
// MEASURED (opt-in build -DWG_OPT_DMA, parity identical): 232 us per launch of the dilated conv against 129-135 us for
// convgemm16w.  With two 32 KB buffers per workgroup a DMA can only run ONE chunk ahead (its target is free only once the
// compute waves hold the previous chunk in registers) and a chunk lasts ~1.5 us while a load under this kernel's own L2 traffic
// (~10 TB/s aggregate) takes 2-3 us; convgemm16w's staging registers are the extra ~128 KB per CU of buffering that hides it.
// A ring deep enough for DMA (>= 3 chunks ahead) does not fit 2 x 80 KB of LDS.  Kept as the reference point for that trade.
// ------------------------------------------------------------------------------------------------
#define WG16D_IMG (128 * 64)
typedef __attribute__((address_space(3))) void wg_lds_void;
typedef const __attribute__((address_space(1))) void wg_glb_void;
__device__ __forceinline__ void dma16(const void *src, char *lds_dst)
{
    __builtin_amdgcn_global_load_lds((wg_glb_void *)src, (wg_lds_void *)lds_dst, 16, 0, 0);
}
__device__ __forceinline__ void read_frags16d(Frags16 &f, const char *buf, int ao, int bo)
{
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        f.ah[i] = *reinterpret_cast<const bf16x8 *>(buf + ao + i * 2048);
        f.al[i] = *reinterpret_cast<const bf16x8 *>(buf + WG16D_IMG + ao + i * 2048);
        f.bh[i] = *reinterpret_cast<const bf16x8 *>(buf + 2 * WG16D_IMG + bo + i * 2048);
        f.bl[i] = *reinterpret_cast<const bf16x8 *>(buf + 3 * WG16D_IMG + bo + i * 2048);
    }
}

template <int EPI>
__global__ __launch_bounds__(384, 3) void convgemm16d_kernel(const ConvGemm16sArgs aa)
{
    constexpr int IMG = WG16D_IMG, BUF = 4 * IMG;
    __shared__ __attribute__((aligned(1024))) char smem[2 * BUF];
    const ConvGemmArgs &a = aa.c;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int t0 = blockIdx.x * WG_TILE, m0 = blockIdx.y * WG_TILE, b = blockIdx.z;
    const Geo g = a.g;
    int nchunks = 0;
    for (int s = 0; s < a.nseg; ++s) nchunks += (a.seg[s].nch + WG16_BK - 1) / WG16_BK;

    if (wave >= 4) {
        // ------------------------------- loader waves -------------------------------
        const int rsub = lane >> 2, kg = (lane & 3) ^ ((lane >> 4) & 3);     // row inside a 16-row piece, source k-group
        int cur_seg = 0, cur_c = 0, chunk = 0;
        auto issue = [&](int buf) {
            char *dst = smem + buf * BUF;
            if (wave == 4) {
                const char *src = reinterpret_cast<const char *>(aa.img + ((size_t)chunk * a.lda + m0) * WG16_BK) + kg * 2048 + rsub * 16;
                const char *srcl = src + aa.img_stride * 2;
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    dma16(src + j * 256, dst + j * 1024);           // 16 rows further: 256 B in the k-group-major image, 1 KB in LDS
                    dma16(srcl + j * 256, dst + IMG + j * 1024);
                }
            } else {
                const int nch = a.seg[cur_seg].nch, shift = a.seg[cur_seg].shift;
                const SSeg ss = aa.sseg[cur_seg];
                const bool real = kg < 2 || nch - cur_c > 16;
                const unsigned short *row0 = ss.hi + ((size_t)b * (ss.Cp >> 3) + ((ss.ch0 + cur_c) >> 3)) * g.P * 8;   // p = 0: zero halo
                const unsigned short *ph = real ? row0 + ((size_t)kg * g.P + (g.H + t0 + shift) + rsub) * 8 : row0 + rsub * 8;
                const unsigned short *pl = ph + ss.lo_off;
                const int step = real ? 16 * 8 : 0;
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    dma16(ph + j * step, dst + 2 * IMG + j * 1024);
                    dma16(pl + j * step, dst + 3 * IMG + j * 1024);
                }
            }
            ++chunk;
            cur_c += WG16_BK;
            if (cur_c >= a.seg[cur_seg].nch) { cur_c = 0; ++cur_seg; }
        };
        issue(0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();                        // buffer 0 ready
        for (int c = 0; c < nchunks; ++c) {
            if (c + 1 < nchunks) issue((c + 1) & 1);         // the compute waves hold chunk c-1 in registers since barrier c-1
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();                    // barrier c: chunk c+1 has landed
        }
        return;
    }
    // ------------------------------- compute waves -------------------------------
    const int wr = wave >> 1, wc = wave & 1;
    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int q = 0; q < 16; ++q) acc[i][j][q] = 0.f;
    const int r = lane & 31, h = lane >> 5, swz = (r >> 2) & 3;
    const int ao0 = (wr * 64 + r) * 64 + ((h ^ swz) << 4), ao1 = ao0 ^ 32;
    const int bo0 = (wc * 64 + r) * 64 + ((h ^ swz) << 4), bo1 = bo0 ^ 32;
    __syncthreads();                                         // buffer 0 ready
    Frags16 f0, f1;
    read_frags16d(f0, smem, ao0, bo0);
    for (int c = 0; c < nchunks; ++c) {
        const char *sb = smem + (c & 1) * BUF, *sn = smem + ((c & 1) ^ 1) * BUF;
        read_frags16d(f1, sb, ao1, bo1);
        mfma12(f0, acc);
        __builtin_amdgcn_sched_barrier(0);                   // keep the twelve MFMAs between the reads of f1 and the barrier's lgkmcnt(0)
        __syncthreads();                                     // barrier c: both k-steps of chunk c are in registers; chunk c+1 has landed
        __builtin_amdgcn_sched_barrier(0);
        if (c + 1 < nchunks) read_frags16d(f0, sn, ao0, bo0);
        mfma12(f1, acc);
    }
    conv_epilogue_s<EPI>(a, aa.s0, acc, t0, m0, b, wr, wc, lane);
}

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
    const unsigned short *b_plane[2];       // the (at most two) distinct B planes of a group: the layer input (all its taps) and the
    float *slab;                            // conditioning; WgradSArgs::b_plane_of says which one a segment reads
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
