// wg_layer16h.h -- ONE launch per WN layer for launches that cannot fill the chip (single-utterance synthesis, WaveFlow's row steps):
//     xy = W (*) h + V y  ->  gate = tanh(xy[:Cd]) * sigmoid(xy[Cd:])  ->  o = W_o gate  ->  h' = h + o[:C],  skip (+)= o[C:]
// (model/waveglow.py:41-46).  Until round 4 a layer was two launches of convgemm16h_kernel (wg_gemm16h.h): the gate conv, then the
// residual / skip conv.  In a chain of ~250 dependent launches of 6-14 us each, a launch's fixed part -- dispatch, the first operands'
// trip from HBM, the accumulate-into tile's round trip, the drain of its stores -- is what there is to win: the residual / skip conv spent
// 8.7 us on 2.8 us of operand stream (profiles/r04a_infer_gaps.txt).
//
// Why the gate goes through memory and not through LDS here.  A workgroup that keeps a column tile's whole gate (all Cd channels) in
// LDS has to multiply all 2 Cd rows of the layer's weight image itself: for one 0.7 s utterance that is 32 workgroups (of 256 CUs)
// each streaming the layer's 2.3 MB of weights through ONE CU at the 50-60 GB/s a CU takes in -- ~40 us per layer against 23 us for
// the two launches (WaveFlow's 64-channel form of exactly that kernel was built and measured in round 2: 111 against 99 ms per call).
// These launches are bound by what a CU takes in, so the layer's weight rows must stay spread over all CUs -- 64 x 64 tiles, eight
// workgroups per column tile -- and the W_o product, whose K range is ALL gate channels of the column tile, needs the other seven
// workgroups' gates.  So: each workgroup writes its 32 gate channels x 64 columns as S-plane units with WRITE-THROUGH (sc1) stores,
// the eight arrive on a counter of their column tile, and each then runs its 64-row tile of the W_o product with the gate operand
// read by sc1 loads (MI355X_MICROARCH.md, "Workgroup dispatch, XCD placement & inter-workgroup visibility", table row 3: agent-scope
// atomic add by one lane of each storing workgroup behind every storing wave's vmcnt(0) wait and a workgroup barrier; consumer: sc1
// load poll, a workgroup barrier between the poll and EVERY load of the bytes; dwordx2 stores that write whole 128-byte lines; dwordx4
// sc1 loads).  The eight workgroups of a column tile have equal blockIdx.x % 8, i.e. they share an XCD under the round-robin
// placement, which makes the hand-off cheaper -- speed only, nothing depends on it.
//
// Grid: 8 * ceil(ncol / 8) * nty workgroups, id -> (column tile = id % 8 + 8 * (id / 8 / nty), row tile = (id / 8) % nty).  All
// workgroups of a column tile are consecutive slots of one XCD, and a workgroup waits only for members of its own set, so the launch
// completes under any in-order dispatch; the host only uses it for grids of at most two workgroups per CU (every one resident).
// The two counters of a column tile (arrivals, departures; a 128-byte line of their own) are left at zero by the set's last departing
// workgroup.
#pragma once
#include "wg_gemm16h.h"

struct ConvLayer16hArgs {
    ConvGemm16sArgs gate;      // phase A, EPI_GATE: s0 = the gate's S-plane (no fp32 gate, no tanh / sigmoid planes on this path)
    ConvGemm16sArgs wo;        // phase B, EPI_RESSKIP (or EPI_STORE): its one K segment is the gate's S-plane
    unsigned *sync;            // [ncol][WGL_SYNC_STRIDE]: word 0 arrivals, word 1 departures of a column tile, zero before the launch and after it
    int ncol, ntx, nty;        // column tiles = ntx * ntz; 64-row tile slots per column tile = max over the two phases
    int wo_epi;                // EPI_RESSKIP or EPI_STORE
};

// WGL_VARIANT (debug A/B builds of the hand-off): 0 = sc1 stores + sc1 loads (default); 1 = + an agent acquire (buffer_inv sc1) behind the
// poll; 2 = sc0 sc1 (system scope) stores and loads; 3 = plain stores + agent release, agent acquire behind the poll, plain loads
#if !defined(WGL_VARIANT)
#define WGL_VARIANT 0
#endif
// A column tile's two counters sit on a 128-byte line of their own: the sets of different column tiles run on different XCDs, whose L2s are
// not coherent with each other -- counters of two XCDs in one line (the first version: adjacent words) gave lost arrivals and early exits
// of the poll as soon as the sets were not in lock step (a cold first call), i.e. stale gates
#define WGL_SYNC_STRIDE 32
#define WGL_SPIN_MAX (1 << 22)     // polls of the arrival counter before a consumer gives up (and poisons its outputs)
template <int OFF>
__device__ __forceinline__ void wgl_st8_sc1(const unsigned short *base, unsigned voff, const u32x2 &v)
{
#if WGL_VARIANT == 2
    asm volatile("global_store_dwordx2 %0, %1, %2 offset:%3 sc0 sc1" ::"v"(voff), "v"(v), "s"(base), "n"(OFF) : "memory");
#elif WGL_VARIANT == 3
    asm volatile("global_store_dwordx2 %0, %1, %2 offset:%3" ::"v"(voff), "v"(v), "s"(base), "n"(OFF) : "memory");
#else
    asm volatile("global_store_dwordx2 %0, %1, %2 offset:%3 sc1" ::"v"(voff), "v"(v), "s"(base), "n"(OFF) : "memory");
#endif
}

// One phase of the layer on the 64 x 64 tile (t0, m0) of plane row b: convgemm16h_body's loop (wg_gemm16h.h) with
//   GATE_OUT: the gate epilogue storing S-plane units write-through (sc1) and draining them (phase A);
//   B_SC1:    the B operand (activations) fetched by sc1 loads (phase B: the bytes other workgroups of this launch have just written);
//   poison:   phase B after a hand-off that timed out: the outputs become NaN instead of silently wrong numbers.
//   arrive / need / s_bad (B_SC1 only): the hand-off itself happens INSIDE this phase, as late as possible -- the loader waves first request
//             the weight halves (A images) of their first WG16H_DEPTH chunks and the compute waves their accumulate-into tile, none of
//             which depend on the gate; only then one lane polls the column tile's arrival counter, the workgroup meets at a barrier
//             (between the poll and EVERY load of the handed-off bytes) and the B halves are requested.  A poll that gives up (s_bad)
//             turns the outputs into NaN instead of silently wrong numbers.
template <int EPI, bool GATE_OUT, bool B_SC1>
__device__ __forceinline__ void layer16h_phase(const ConvGemm16sArgs &aa, int t0, int m0, int b, char *smem, const unsigned *arrive, unsigned need,
                                               int *s_bad)
{
    constexpr int D = WG16H_DEPTH;
    constexpr int AIMG = 64 * WG16Q_ROWB, BIMG = 64 * WG16Q_ROWB;
    constexpr int BUF = 2 * AIMG + 2 * BIMG;
    const ConvGemmArgs &a = aa.c;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const Geo g = a.g;
    int nchunks = 0;
    for (int s = 0; s < a.nseg; ++s) nchunks += (a.seg[s].nch + WG16_BK - 1) / WG16_BK;
    const int nbar = (nchunks + D - 1) / D * D;

    if (wave >= 4) {
        // ------------------------------- loader waves -------------------------------
        const int lt = tid - 256;
        const int r = lt & 63, kq = lt >> 6;
        const int l_off = wg16q_off(r, kq);
        const unsigned voff_a = (unsigned)((kq * 128 + (m0 & 64) + r) * 16);
        const unsigned voff_b = (unsigned)((kq * g.P + r) * 16);
        struct Cur { int seg, c, chunk; };
        Cur ca = {0, 0, 0}, cb = {0, 0, 0};                   // the A halves and the B halves of a stage are requested by their own cursors
#define WGL_LDA(dst, base, voff) asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(dst) : "v"(voff), "s"(base) : "memory")
#define WGL_LDB(dst, base, voff)                                                                                           \
    do {                                                                                                                   \
        if (B_SC1 && WGL_VARIANT == 2) asm volatile("global_load_dwordx4 %0, %1, %2 sc0 sc1" : "=v"(dst) : "v"(voff), "s"(base) : "memory"); \
        else if (B_SC1 && WGL_VARIANT != 3) asm volatile("global_load_dwordx4 %0, %1, %2 sc1" : "=v"(dst) : "v"(voff), "s"(base) : "memory");       \
        else asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(dst) : "v"(voff), "s"(base) : "memory");                 \
    } while (0)
        const unsigned short *zsrc = aa.sseg[0].hi;           // plane position 0 of the first operand: always-zero halo
        auto advance = [&](Cur &q) {
            if (q.chunk < nchunks) {
                ++q.chunk;
                q.c += WG16_BK;
                if (q.c >= a.seg[min(q.seg, a.nseg - 1)].nch) { q.c = 0; ++q.seg; }
            }
        };
        auto issueA = [&](Stage4 &st) {                       // exactly 2 loads in straight-line code (tools/check_asm_loads.py)
            const bool live = ca.chunk < nchunks;
            const unsigned short *ih = aa.img + ((size_t)ca.chunk * a.lda + (m0 & ~127)) * WG16_BK, *il = ih + aa.img_stride;
            const unsigned short *pa = live ? ih : zsrc, *pl = live ? il : zsrc;
            const unsigned va = live ? voff_a : 0u;
            WGL_LDA(st.ah, pa, va);   WGL_LDA(st.al, pl, va);
            advance(ca);
        };
        auto issueB = [&](Stage4 &st) {                       // exactly 2 loads in straight-line code
            const bool live = cb.chunk < nchunks;
            const int sg = min(cb.seg, a.nseg - 1);
            const int nch = a.seg[sg].nch, shift = a.seg[sg].shift;
            const SSeg ss = aa.sseg[sg];
            int bsrc = b;
            bool rowok = true;
            if (g.rows > 0) {
                const int item = b / g.rows, rr = b - item * g.rows + ss.row_off;
                rowok = rr >= 0 && rr < g.rows;
                bsrc = ss.per_item ? item : b + ss.row_off;
            }
            const bool blive = live && rowok, full = blive && (nch - cb.c > 16);
            const unsigned short *row0 = ss.hi + ((size_t)bsrc * (ss.Cp >> 3) + ((ss.ch0 + cb.c) >> 3)) * g.P * 8;
            const unsigned short *pb = blive ? row0 : zsrc, *pbl = blive ? row0 + ss.lo_off : zsrc;
            const bool lane_ok = blive && (kq < 2 || full);
            const unsigned vb = lane_ok ? voff_b + (unsigned)((g.H + t0 + shift) * 16) : 0u;
            WGL_LDB(st.bh, pb, vb);   WGL_LDB(st.bl, pbl, vb);
            advance(cb);
        };
        auto issue = [&](Stage4 &st) { issueA(st); issueB(st); };
#undef WGL_LDA
#undef WGL_LDB
        auto write = [&](const Stage4 &st, int buf) {
            char *sb = smem + buf * BUF + l_off;
            *reinterpret_cast<u32x4 *>(sb) = st.ah;
            *reinterpret_cast<u32x4 *>(sb + AIMG) = st.al;
            *reinterpret_cast<u32x4 *>(sb + 2 * AIMG) = st.bh;
            *reinterpret_cast<u32x4 *>(sb + 2 * AIMG + BIMG) = st.bl;
        };
        Stage4 st[D];
        if constexpr (B_SC1) {
            static_assert(D == 3, "the counted waits of the split prologue below are written for three stages");
            // in flight after the prologue, oldest first: A0 A0 A1 A1 A2 A2 | B0 B0 B1 B1 B2 B2
#pragma unroll
            for (int i = 0; i < D; ++i) issueA(st[i]);
            if (tid == 256) {                                 // one lane polls (sc1 load); bounded: a broken hand-off shows as NaN, not as a hang
                int spins = 0;
                while (__hip_atomic_load(arrive, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < need && spins < WGL_SPIN_MAX) {
                    __builtin_amdgcn_s_sleep(1);
                    ++spins;
                }
                *s_bad = spins >= WGL_SPIN_MAX;
#if WGL_VARIANT == 1 || WGL_VARIANT == 3
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
            }
            __syncthreads();                                  // between the poll and EVERY load of the handed-off bytes
#pragma unroll
            for (int i = 0; i < D; ++i) issueB(st[i]);
            asm volatile("s_waitcnt vmcnt(4)" : "+v"(st[0].ah), "+v"(st[0].al), "+v"(st[0].bh), "+v"(st[0].bl)::"memory");   // all but B1 B1 B2 B2
            write(st[0], 0);
            issue(st[0]);                                     // chunk D: in flight B1 B1 B2 B2 A3 A3 B3 B3
            WG16W_BAR();                                      // buffer 0 ready
            asm volatile("s_waitcnt vmcnt(6)" : "+v"(st[1].ah), "+v"(st[1].al), "+v"(st[1].bh), "+v"(st[1].bl)::"memory");   // B1 B1 have landed
            write(st[1], 1);
            issue(st[1]);                                     // in flight B2 B2 | chunk 3 | chunk 4: the steady state's vmcnt(8) from here on
            WG16W_BAR();
            // (stage indices must stay compile-time constants: a run-time index sends the stage registers through scratch memory --
            // copied right behind their loads, before the data has landed; tools/check_asm_loads.py finds exactly that)
            asm_wait_stage_h(st[2]); write(st[2], 0); issue(st[2]); WG16W_BAR();       // iteration 1
            asm_wait_stage_h(st[0]); write(st[0], 1); issue(st[0]); WG16W_BAR();       // iteration 2
            for (int c = D; c < nbar; c += D) {
#pragma unroll
                for (int i = 0; i < D; ++i) {
                    Stage4 &s = st[(i + 1) % D];
                    asm_wait_stage_h(s);
                    write(s, (c + i + 1) & 1);
                    issue(s);
                    WG16W_BAR();
                }
            }
        } else {
#pragma unroll
            for (int i = 0; i < D; ++i) issue(st[i]);
            asm_wait_stage_h(st[0]);
            write(st[0], 0);
            issue(st[0]);
            WG16W_BAR();                                      // buffer 0 ready
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
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // drain the trailing zero-halo loads
        return;
    }
    // ------------------------------- compute waves -------------------------------
    const int wc = wave;
    f32x4 acc[4][1];
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
        conv_acc_init_a<EPI, false>(a, aa.saux, aa.img, acc, t0, m0, b, wc, lane);      // (requested and landed before the hand-off is waited for)
        if constexpr (B_SC1) {
            __syncthreads();                                  // the loader waves' poll barrier
            if (*s_bad) {
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int e = 0; e < 4; ++e) acc[i][0][e] = __builtin_nanf("");
            }
        }
    } else {
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int e = 0; e < 4; ++e) acc[i][0][e] = 0.f;
    }
    WG16W_BAR();                                              // buffer 0 ready
    Frags f0, f1;
    fetch(f0, smem);
    auto step = [&](const Frags &f, Frags &fn, int c) {
        __builtin_amdgcn_sched_barrier(0);
        WG16W_BAR();
        __builtin_amdgcn_sched_barrier(0);
        fetch(fn, smem + ((c & 1) ^ 1) * BUF);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int mb = 0; mb < 4; ++mb) acc[mb][0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(f.al[mb], f.bh, acc[mb][0], 0, 0, 0);
#pragma unroll
        for (int mb = 0; mb < 4; ++mb) acc[mb][0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(f.ah[mb], f.bl, acc[mb][0], 0, 0, 0);
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
    for (int cc = nchunks; cc < nbar; ++cc) WG16W_BAR();      // the loaders' spare iterations
    if constexpr (GATE_OUT) {
        // rows 0-31 of the tile are the tanh halves, rows 32-63 the sigmoid halves of the same 32 gate channels (pack_kernel's 64-row
        // interleave): channel chb + mbp*16 + 4 rq + e pairs acc[mbp] with acc[mbp + 2].  A wave instruction stores 2 unit rows x 256
        // contiguous bytes: whole 128-byte lines.
        const int col = lane & 15, rq = lane >> 4, tc = t0 + wc * 16, t = tc + col;
        const int chb = m0 >> 1;
        const unsigned vo_s = (unsigned)(((rq >> 1) * g.P + col) * 16 + 8 * (rq & 1));
#pragma unroll
        for (int mbp = 0; mbp < 2; ++mbp) {
            float gv[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) gv[e] = wg_tanh(acc[mbp][0][e]) * wg_sigmoid(acc[mbp + 2][0][e]);
            u32x2 vh, vl;
            unsigned hh, ll;
            split2(gv[0], gv[1], hh, ll); vh[0] = hh; vl[0] = ll;
            split2(gv[2], gv[3], hh, ll); vh[1] = hh; vl[1] = ll;
            const unsigned short *hb = aa.s0.hi + s_index(aa.s0, g, b, chb + mbp * 16, tc);
            if (t < g.T) { wgl_st8_sc1<0>(hb, vo_s, vh); wgl_st8_sc1<0>(hb + aa.s0.lo_off, vo_s, vl); }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // every storing wave drains its write-through stores before the arrival
    } else {
        conv_epilogue_a<EPI, false>(a, aa.s0, acc, t0, m0, b, wc, lane);
    }
}

template <int EPIB>
__global__ __launch_bounds__(512) void convlayer16h_kernel(const ConvLayer16hArgs la)
{
    __shared__ __attribute__((aligned(16))) char smem[WG16H_SMEM];
    __shared__ int s_bad;
    const int tid = threadIdx.x;
    const int id = (int)blockIdx.x, xcd = id & 7, slot = id >> 3;
    const int ct = xcd + 8 * (slot / la.nty), ty = slot - (slot / la.nty) * la.nty;
    if (ct >= la.ncol) return;
    const int tx = ct % la.ntx, tz = ct / la.ntx;
    const int t0 = tx * 64, m0 = ty * 64;
    const Geo &g = la.gate.c.g;
    const int b = la.gate.c.row_sel1 ? tz * g.rows + la.gate.c.row_sel1 - 1 : tz;
    const bool doA = m0 < la.gate.c.M, doB = m0 < la.wo.c.M;
    if (doA) layer16h_phase<EPI_GATE, true, false>(la.gate, t0, m0, b, smem, nullptr, 0u, nullptr);
    __syncthreads();                                          // every storing wave has drained its stores
    unsigned *arrive = la.sync + (size_t)ct * WGL_SYNC_STRIDE, *depart = arrive + 1;
#if WGL_VARIANT == 3
    if (tid == 0 && doA) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
#endif
    if (tid == 0 && doA) __hip_atomic_fetch_add(arrive, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (!doB) return;
    layer16h_phase<EPIB, false, true>(la.wo, t0, m0, b, smem, arrive, (unsigned)((la.gate.c.M + 63) / 64), &s_bad);
    if (tid == 0) {                                           // the set's last departure leaves both counters at zero for the next launch
        const unsigned nb = (unsigned)((la.wo.c.M + 63) / 64);
        if (__hip_atomic_fetch_add(depart, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == nb - 1) {
            __hip_atomic_store(arrive, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(depart, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}
