// wg_gemm16_superseded.h -- three conv kernels that the default build no longer uses, kept for A/B builds only:
//   convgemm16w_kernel (-DWG_OPT_MFMA32): the wave-specialised kernel on v_mfma_f32_32x32x16_bf16, the default of round 1;
//   convgemm16p_kernel (-DWG_OPT_NO_WSPEC): its symmetric predecessor (4 waves staging AND multiplying);
//   convgemm16d_kernel (-DWG_OPT_DMA): the LDS-DMA loader ring.
// Included by csrc/wg_gemm16s.h only when one of those switches is set (this file relies on the definitions in front of the include).
// The product kernels are convgemm16q_kernel / convgemm16h_kernel (csrc/wg_gemm16q.h, wg_gemm16h.h); their comments refer to the
// protocol descriptions below.
#pragma once

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
#if defined(WG_DBG_TRACE)
#define WG_TRACE(slot) do { if (EPI == EPI_GATE && NI == 2 && lane == 0 && wave == 0 && (slot) < 16) { \
        wg_dbg_trace[blockIdx.x * 16 + (slot)] = wall_clock64(); wg_dbg_trace_cyc[blockIdx.x * 16 + (slot)] = clock64(); } } while (0)
#else
#define WG_TRACE(slot) do { } while (0)
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
