// wg_gemm16h.h -- the conv kernel for launches that cannot fill the chip: single-utterance synthesis (the gate conv of a 0.7 s utterance
// is 4 x 32 tiles of 128 x 64), WaveFlow's row-by-row inverse.
//
// What bounds such a launch (in-kernel stamps, tools/experiments/infer_trace.py): the 27-chunk main loop of the 128 x 64 form runs at the
// full 2.36 GHz with the matrix pipe 35 % busy; a chunk takes 460 ns because ONE CU takes in its operands (24 KB per chunk, 3/4 of them
// L2 hits) at 52-58 GB/s -- the per-CU rate of MI355X_MICROARCH.md's gather table -- while half of the CUs have no workgroup.  Twice the
// bytes in flight per CU did not change that rate (a two-compute-group form that split K inside the workgroup was built and measured:
// 830 ns per two chunks), so the bytes per CU have to go down: this kernel uses 64 x 64 tiles, i.e. twice the workgroups (one per CU for
// the utterance above) that each stream 16 KB per chunk -- the minimum of (rows + columns) at that tile count.
//
// Structure: convgemm16q's (wg_gemm16q.h) with the roles re-cut for the small tile.  4 compute waves, each the whole 64 rows x 16 columns
// (the gate epilogue pairs rows r and r + 32 of a 64-row block, so a wave owns complete gate channels): 12 MFMAs per chunk and wave; 4
// loader waves, ONE 16-byte piece of each of the four images (A hi, A lo, B hi, B lo) per lane and chunk, WG16H_DEPTH chunks in flight in
// registers, two LDS buffers.  One barrier per chunk: behind it every wave has the chunk's fragments in registers (__syncthreads waits for
// the LDS reads) and the next chunk is staged; the compute waves re-load each A fragment from the next buffer as soon as its three MFMAs
// are issued.  One tile per workgroup.
#pragma once
#include "wg_gemm16q.h"

#ifndef WG16H_DEPTH
#define WG16H_DEPTH 3
#endif
struct Stage4 {
    u32x4 ah, al, bh, bl;
};
__device__ __forceinline__ void asm_wait_stage_h(Stage4 &s)      // all but the newest WG16H_DEPTH - 1 stages have landed
{
    static_assert(WG16H_DEPTH >= 2 && WG16H_DEPTH <= 4, "");
    if (WG16H_DEPTH == 2) asm volatile("s_waitcnt vmcnt(4)" : "+v"(s.ah), "+v"(s.al), "+v"(s.bh), "+v"(s.bl)::"memory");
    if (WG16H_DEPTH == 3) asm volatile("s_waitcnt vmcnt(8)" : "+v"(s.ah), "+v"(s.al), "+v"(s.bh), "+v"(s.bl)::"memory");
    if (WG16H_DEPTH == 4) asm volatile("s_waitcnt vmcnt(12)" : "+v"(s.ah), "+v"(s.al), "+v"(s.bh), "+v"(s.bl)::"memory");
}
#if defined(WG_DBG_TRACE) && defined(WG_DBG_TRACE_SMALL)
#define WGH_TRACE(slot) do { if (EPI == EPI_GATE && lane == 0 && wave == 0) { \
        wg_dbg_trace[blockIdx.x * 16 + (slot)] = wall_clock64(); wg_dbg_trace_cyc[blockIdx.x * 16 + (slot)] = clock64(); } } while (0)
#else
#define WGH_TRACE(slot) do { } while (0)
#endif

// A wave-uniform pointer into scalar registers, for an asm load's "s" operand.  IN_MEMORY = false (the launched kernel: the argument
// block is in the kernarg segment, every address is scalar arithmetic already): nothing to do.  IN_MEMORY = true (the stage interpreter,
// wg_stage.h: the block is read from LDS, the arithmetic is per lane): v_readfirstlane, and FIVE wait states behind it -- a VALU write of
// an SGPR must be that far ahead of a vector-memory instruction that uses it as its address, and the compiler's hazard recogniser does
// not look into the asm statement that holds the load (without the s_nop the loads went to stale bases: memory access faults).
template <bool IN_MEMORY>
__device__ __forceinline__ const unsigned short *wg_uniform_ptr(const unsigned short *q)
{
    if (!IN_MEMORY) return q;
    const unsigned long long v = (unsigned long long)q;
    unsigned lo32, hi32;
    asm volatile("v_readfirstlane_b32 %0, %2\n\tv_readfirstlane_b32 %1, %3\n\ts_nop 4" : "=s"(lo32), "=s"(hi32) : "v"((unsigned)v), "v"((unsigned)(v >> 32)));
    return (const unsigned short *)(((unsigned long long)hi32 << 32) | lo32);
}
#define WG16H_SMEM (2 * (4 * 64 * WG16Q_ROWB))              // two chunk buffers of A hi, A lo, B hi, B lo (4 KB each)
// The tile `id` of the launch's grid: ntx (64-column tiles) x nty (64-ROW tiles) x ntz, time tile fastest.  A device function so that
// the stage interpreter (wg_stage.h) runs the same code for one stage of a recorded launch sequence; row_sel1 overrides the argument
// block's value there (the interpreter walks the height rows of WaveFlow's inverse with ONE recorded program).
// Accumulator init and epilogue of the 64 x 64-tile kernel with HAND-ISSUED memory instructions (EPI_STORE / EPI_RESSKIP; see the note on
// drained vmcnt in wg_gemm16q.h: the compiler's form drained 14-15 times per tile, and these launches are chains of latencies --
// single-utterance synthesis, WaveFlow's row steps).  A wave owns 64 rows x 16 columns: scalar base per 16-row block, the lane's four
// row offsets (fp32 planes) or its unit offset (S-planes) constant, ONE explicit wait behind all loads.
template <int OFF>
__device__ __forceinline__ void wgh_ld4(float &v, const float *base, unsigned voff)
{
    asm volatile("global_load_dword %0, %1, %2 offset:%3" : "=v"(v) : "v"(voff), "s"(base), "n"(OFF) : "memory");
}
__device__ __forceinline__ void wgh_ld8(u32x2 &v, const unsigned short *base, unsigned voff)
{
    asm volatile("global_load_dwordx2 %0, %1, %2" : "=v"(v) : "v"(voff), "s"(base) : "memory");
}
template <int EPI, bool IN_MEMORY>
__device__ __forceinline__ void conv_acc_init_a(const ConvGemmArgs &a, const SRef &saux, const void *safe, f32x4 (&acc)[4][1], int t0, int m0, int b,
                                                int wc, int lane)
{
    // Every load is issued UNCONDITIONALLY (a lane or a block without a value reads `safe`, any readable address, at offset 0) and the
    // zero is selected behind the wait: a load under a branch lets the compiler merge its destination with the other path's value --
    // a register copy -- BEFORE the hand-written wait (tools/check_asm_loads.py finds exactly that).
    const Geo g = a.g;
    const int col = lane & 15, rq = lane >> 4, tc = t0 + wc * 16, t = tc + col;
    if (EPI == EPI_STORE && saux.hi) {
        const unsigned vo_s = (unsigned)(((rq >> 1) * g.P + col) * 16 + 8 * (rq & 1));
        u32x2 h[4], l[4];
        bool ok[4];
#pragma unroll
        for (int mb = 0; mb < 4; ++mb) {
            const int mbase = m0 + mb * 16;
            const bool blk = mbase < a.M;                        // (wave uniform)
            ok[mb] = blk && t < g.T && mbase + 4 * rq < a.M;
            const unsigned short *hb = blk ? saux.hi + s_index(saux, g, b, mbase, tc) : reinterpret_cast<const unsigned short *>(safe);
            const unsigned short *lb = blk ? hb + saux.lo_off : hb;
            const unsigned vo = ok[mb] ? vo_s : 0u;
            wgh_ld8(h[mb], wg_uniform_ptr<IN_MEMORY>(hb), vo); wgh_ld8(l[mb], wg_uniform_ptr<IN_MEMORY>(lb), vo);
        }
        asm volatile("s_waitcnt vmcnt(0)" : "+v"(h[0]), "+v"(h[1]), "+v"(h[2]), "+v"(h[3]), "+v"(l[0]), "+v"(l[1]), "+v"(l[2]), "+v"(l[3])::"memory");
#pragma unroll
        for (int mb = 0; mb < 4; ++mb) {
            const unsigned h0 = ok[mb] ? h[mb][0] : 0u, h1 = ok[mb] ? h[mb][1] : 0u, l0 = ok[mb] ? l[mb][0] : 0u, l1 = ok[mb] ? l[mb][1] : 0u;
            acc[mb][0][0] = __uint_as_float(h0 << 16) + __uint_as_float(l0 << 16);
            acc[mb][0][1] = __uint_as_float(h0 & 0xffff0000u) + __uint_as_float(l0 & 0xffff0000u);
            acc[mb][0][2] = __uint_as_float(h1 << 16) + __uint_as_float(l1 << 16);
            acc[mb][0][3] = __uint_as_float(h1 & 0xffff0000u) + __uint_as_float(l1 & 0xffff0000u);
        }
        return;
    }
    unsigned vo[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) vo[e] = (unsigned)(((4 * rq + e) * g.P + col) * 4);
    float x[4][4];
    bool ok[4][4];
#pragma unroll
    for (int mb = 0; mb < 4; ++mb) {
        const int mbase = m0 + mb * 16;                          // (wave uniform)
        const float *base = nullptr;
        if (EPI == EPI_STORE) base = a.aux0.p ? paddr(a.aux0, g, b, mbase, tc) : nullptr;
        else base = mbase < a.nsplit ? paddr(a.aux0, g, b, mbase, tc) : (a.accumulate ? paddr(a.out1, g, b, mbase - a.nsplit, tc) : nullptr);
        const bool blk = base != nullptr && mbase < a.M;
        const float *bp = reinterpret_cast<const float *>(wg_uniform_ptr<IN_MEMORY>(reinterpret_cast<const unsigned short *>(blk ? base : reinterpret_cast<const float *>(safe))));
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            ok[mb][e] = blk && t < g.T && mbase + 4 * rq + e < a.M;
            wgh_ld4<0>(x[mb][e], bp, ok[mb][e] ? vo[e] : 0u);
        }
    }
    asm volatile("s_waitcnt vmcnt(0)"
                 : "+v"(x[0][0]), "+v"(x[0][1]), "+v"(x[0][2]), "+v"(x[0][3]), "+v"(x[1][0]), "+v"(x[1][1]), "+v"(x[1][2]), "+v"(x[1][3]),
                   "+v"(x[2][0]), "+v"(x[2][1]), "+v"(x[2][2]), "+v"(x[2][3]), "+v"(x[3][0]), "+v"(x[3][1]), "+v"(x[3][2]), "+v"(x[3][3])::"memory");
#pragma unroll
    for (int mb = 0; mb < 4; ++mb)
#pragma unroll
        for (int e = 0; e < 4; ++e) acc[mb][0][e] = ok[mb][e] ? x[mb][e] : 0.f;
}
template <int EPI, bool IN_MEMORY>
__device__ __forceinline__ void conv_epilogue_a(const ConvGemmArgs &a, const SRef &s0, f32x4 (&acc)[4][1], int t0, int m0, int b, int wc, int lane)
{
    const Geo g = a.g;
    const int col = lane & 15, rq = lane >> 4, tc = t0 + wc * 16, t = tc + col;
    const unsigned vo_s = (unsigned)(((rq >> 1) * g.P + col) * 16 + 8 * (rq & 1));
    unsigned vo[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) vo[e] = (unsigned)(((4 * rq + e) * g.P + col) * 4);
#pragma unroll
    for (int mb = 0; mb < 4; ++mb) {
        const int mbase = m0 + mb * 16;
        if (mbase >= a.M) continue;
        const bool res = EPI == EPI_STORE || mbase < a.nsplit;
        const PRef &dst = res ? a.out0 : a.out1;
        if (dst.p) {
            const float *base = reinterpret_cast<const float *>(wg_uniform_ptr<IN_MEMORY>(reinterpret_cast<const unsigned short *>(paddr(dst, g, b, res ? mbase : mbase - a.nsplit, tc))));
#pragma unroll
            for (int e = 0; e < 4; ++e)
                if (t < g.T && mbase + 4 * rq + e < a.M) wgq_st4<0>(base, vo[e], acc[mb][0][e]);
        }
        if (res && s0.hi) {
            u32x2 ph, pl;
            unsigned hh, ll;
            split2(acc[mb][0][0], acc[mb][0][1], hh, ll); ph[0] = hh; pl[0] = ll;
            split2(acc[mb][0][2], acc[mb][0][3], hh, ll); ph[1] = hh; pl[1] = ll;
            const unsigned short *hb = wg_uniform_ptr<IN_MEMORY>(s0.hi + s_index(s0, g, b, mbase, tc));
            if (t < g.T && mbase + 4 * rq < a.M) { wgq_st8<0>(hb, vo_s, ph); wgq_st8<0>(wg_uniform_ptr<IN_MEMORY>(hb + s0.lo_off), vo_s, pl); }
        }
    }
}

template <int EPI, bool IN_MEMORY = false>
__device__ __forceinline__ void convgemm16h_body(const ConvGemm16sArgs &aa, int id, int row_sel1, char *smem)
{
    constexpr int D = WG16H_DEPTH;
    constexpr int AIMG = 64 * WG16Q_ROWB, BIMG = 64 * WG16Q_ROWB;       // 4 KB each
    constexpr int BUF = 2 * AIMG + 2 * BIMG;
    constexpr int TT = 64;
    static_assert(2 * BUF == WG16H_SMEM, "");
    const ConvGemmArgs &a = aa.c;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const Geo g = a.g;
    int nchunks = 0;
    for (int s = 0; s < a.nseg; ++s) nchunks += (a.seg[s].nch + WG16_BK - 1) / WG16_BK;
    const int nbar = (nchunks + D - 1) / D * D;               // barriers after the first: the loaders' iterations come in groups of D
    const int tx = id % aa.ntx, q = id / aa.ntx, ty = q % aa.nty, tz = q / aa.nty;
    const int t0 = tx * TT, m0 = ty * 64;
    const int b = row_sel1 ? tz * g.rows + row_sel1 - 1 : tz;
    if (m0 >= a.M) return;                                    // (M is padded to 128 rows in the image: the upper half tile may be empty)

    if (wave >= 4) {
        // ------------------------------- loader waves -------------------------------
        const int lt = tid - 256;
        const int r = lt & 63, kq = lt >> 6;                  // this lane's piece of every image: row / column r, k-group kq
        const int l_off = wg16q_off(r, kq);
        // A image of a chunk: [128-row block][k-group][row][8]: the 64-row half of block m0 / 128
        const unsigned voff_a = (unsigned)((kq * 128 + (m0 & 64) + r) * 16);
        const unsigned voff_b = (unsigned)((kq * g.P + r) * 16);
        int cur_seg = 0, cur_c = 0, chunk = 0;
#if defined(WG_DBG_NOLOAD)
#define WG_LD(dst, base, voff) asm volatile("" : "=v"(dst) : "v"(voff), "s"(base))
#else
#define WG_LD(dst, base, voff) asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(dst) : "v"(voff), "s"(base) : "memory")
#endif
        const unsigned short *zsrc = aa.sseg[0].hi;           // plane position 0 of the first operand: always-zero halo
        auto issue = [&](Stage4 &st) {                        // exactly 4 loads in straight-line code (tools/check_asm_loads.py)
            const bool live = chunk < nchunks;
            const int sg = min(cur_seg, a.nseg - 1);
            const int nch = a.seg[sg].nch, shift = a.seg[sg].shift;
            const SSeg ss = aa.sseg[sg];
            int bsrc = b;
            bool rowok = true;
            if (g.rows > 0) {
                const int item = b / g.rows, rr = b - item * g.rows + ss.row_off;
                rowok = rr >= 0 && rr < g.rows;
                bsrc = ss.per_item ? item : b + ss.row_off;
            }
            const bool blive = live && rowok, full = blive && (nch - cur_c > 16);
            const unsigned short *ih = aa.img + ((size_t)chunk * a.lda + (m0 & ~127)) * WG16_BK, *il = ih + aa.img_stride;
            const unsigned short *row0 = ss.hi + ((size_t)bsrc * (ss.Cp >> 3) + ((ss.ch0 + cur_c) >> 3)) * g.P * 8;
            const unsigned short *pa = wg_uniform_ptr<IN_MEMORY>(live ? ih : zsrc), *pl = wg_uniform_ptr<IN_MEMORY>(live ? il : zsrc);
            const unsigned va = live ? voff_a : 0u;
            WG_LD(st.ah, pa, va);   WG_LD(st.al, pl, va);
            const unsigned short *pb = wg_uniform_ptr<IN_MEMORY>(blive ? row0 : zsrc), *pbl = wg_uniform_ptr<IN_MEMORY>(blive ? row0 + ss.lo_off : zsrc);
            const bool lane_ok = blive && (kq < 2 || full);
            const unsigned vb = lane_ok ? voff_b + (unsigned)((g.H + t0 + shift) * 16) : 0u;
            WG_LD(st.bh, pb, vb);   WG_LD(st.bl, pbl, vb);
            if (live) {
                ++chunk;
                cur_c += WG16_BK;
                if (cur_c >= nch) { cur_c = 0; ++cur_seg; }
            }
        };
#undef WG_LD
        auto write = [&](const Stage4 &st, int buf) {
            char *sb = smem + buf * BUF + l_off;
            *reinterpret_cast<u32x4 *>(sb) = st.ah;
            *reinterpret_cast<u32x4 *>(sb + AIMG) = st.al;
            *reinterpret_cast<u32x4 *>(sb + 2 * AIMG) = st.bh;
            *reinterpret_cast<u32x4 *>(sb + 2 * AIMG + BIMG) = st.bl;
        };
        Stage4 st[D];
#pragma unroll
        for (int i = 0; i < D; ++i) issue(st[i]);             // chunks 0 .. D-1
        asm_wait_stage_h(st[0]);
        write(st[0], 0);
        issue(st[0]);                                         // chunk D
        WG16W_BAR();                                          // buffer 0 ready
        // iteration c (between barrier c and barrier c + 1; the compute waves multiply chunk c): the stage holding chunk c + 1 has
        // landed -> into the buffer chunk c - 1 was read from (those reads completed before barrier c), then chunk c + 1 + D is requested
        for (int c = 0; c < nbar; c += D) {
#pragma unroll
            for (int i = 0; i < D; ++i) {
                Stage4 &s = st[(i + 1) % D];
                asm_wait_stage_h(s);
                write(s, (c + i + 1) & 1);
                issue(s);
                WG16W_BAR();
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // drain the trailing zero-halo loads before the wave ends
        return;
    }
    // ------------------------------- compute waves -------------------------------
    const int wc = wave;                                      // 16-column block of the tile
    f32x4 acc[4][1];
    WGH_TRACE(8);
    const int r16 = lane & 15, kg = lane >> 4;
    const int ao = wg16q_off(r16, kg), bo = wg16q_off(wc * 16 + r16, kg);
    struct Frags { bf16x8 ah[4], al[4], bh, bl; };
    auto rd = [&](const char *p) { return *reinterpret_cast<const bf16x8 *>(p); };
    auto fetch = [&](Frags &f, const char *sb) {
#pragma unroll
        for (int i = 0; i < 4; ++i) { f.ah[i] = rd(sb + ao + i * 1024); f.al[i] = rd(sb + AIMG + ao + i * 1024); }
        f.bh = rd(sb + 2 * AIMG + bo); f.bl = rd(sb + 2 * AIMG + BIMG + bo);
    };
    if (EPI == EPI_STORE || EPI == EPI_RESSKIP) {
#if !defined(WG_OPT_NO_EPI_BATCH)
        conv_acc_init_a<EPI, IN_MEMORY>(a, aa.saux, aa.img, acc, t0, m0, b, wc, lane);
#else
        conv_acc_init_q<EPI, 1>(a, aa.saux, acc, t0, m0, b, 0, wc, lane);
#endif
    } else {
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int e = 0; e < 4; ++e) acc[i][0][e] = 0.f;
    }
    WGH_TRACE(0);
    WG16W_BAR();                                              // buffer 0 ready
    WGH_TRACE(1);
    Frags f0, f1;
    fetch(f0, smem);
    // one chunk: barrier (chunk c is in registers everywhere, chunk c + 1 is staged), request chunk c + 1's fragments into the other
    // register set, multiply chunk c: the three products of a block are dependent, so the 12 MFMAs go product by product over the
    // four blocks.  (behind the last chunk the request reads a buffer nothing uses)
    auto step = [&](const Frags &f, Frags &fn, int c) {
        __builtin_amdgcn_sched_barrier(0);
        WG16W_BAR();
        __builtin_amdgcn_sched_barrier(0);
        fetch(fn, smem + ((c & 1) ^ 1) * BUF);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int mb = 0; mb < 4; ++mb) if (!TwoP<EPI>::no_alo) acc[mb][0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(f.al[mb], f.bh, acc[mb][0], 0, 0, 0);
#pragma unroll
        for (int mb = 0; mb < 4; ++mb) if (!TwoP<EPI>::no_blo) acc[mb][0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(f.ah[mb], f.bl, acc[mb][0], 0, 0, 0);
#pragma unroll
        for (int mb = 0; mb < 4; ++mb) acc[mb][0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(f.ah[mb], f.bh, acc[mb][0], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
    };
    int c = 0;
    for (; c + 1 < nchunks; c += 2) {
        step(f0, f1, c);
        step(f1, f0, c + 1);
    }
    if (c < nchunks) step(f0, f1, c);
    for (int c = nchunks; c < nbar; ++c) WG16W_BAR();         // the loaders' spare iterations
    WGH_TRACE(2);
#if !defined(WG_OPT_NO_EPI_BATCH)
    if (EPI == EPI_STORE || EPI == EPI_RESSKIP) conv_epilogue_a<EPI, IN_MEMORY>(a, aa.s0, acc, t0, m0, b, wc, lane);
    else
#endif
        conv_epilogue_q<EPI, 1>(a, aa.s0, acc, t0, m0, b, 0, wc, lane);
    WGH_TRACE(3);
}

template <int EPI>
__global__ __launch_bounds__(512) void convgemm16h_kernel(const ConvGemm16sArgs aa)
{
    __shared__ __attribute__((aligned(16))) char smem[WG16H_SMEM];
    convgemm16h_body<EPI>(aa, (int)blockIdx.x, aa.c.row_sel1, smem);
}
