// wg_gemm16hk.h -- EXPERIMENT, not part of the build: convgemm16h with 2 or 4 k-steps per LDS buffer and barrier.
// Built and measured in round 3 (parity green): single-utterance synthesis 2.95 -> 3.16 (KS = 2) / 3.22 ms (KS = 4) per call, WaveFlow's
// inverse 99.3 -> 106 / 114 ms.  A launch of the small-tile conv is NOT a chain of chunk hand-overs: it lasts as long as ONE CU needs to take
// in its operands (~46-58 GB/s per CU whatever the chunk size), and larger chunks only add the latency of the first one.
// To try it again: append this file to csrc/wg_gemm16h.h and dispatch to convgemm16hk_kernel<EPI, KS> in run_convgemm's small-tile branch.

// ------------------------------------------------------------------------------------------------
// convgemm16hk: the same tile with KS k-steps (KS x 32 k) per LDS buffer and barrier.
// A launch of convgemm16h is a chain of chunk hand-overs (stage wait -> LDS write -> barrier -> fragment read): 352 ns per 32-k chunk
// with the matrix pipe a third busy (tools/experiments/infer_trace.py), and the launches that take this kernel are exactly the ones
// that are as long as that chain (single-utterance synthesis, WaveFlow's row steps: 21-27 chunks).  Fewer, larger chunks: one barrier
// per KS k-steps; the loaders move 4 KS pieces per lane and chunk, the compute waves fetch k-step j + 1 of the SAME buffer while they
// multiply k-step j and cross to the other buffer only behind the chunk's one barrier.
// ------------------------------------------------------------------------------------------------
template <int KS> struct StageHK {
    u32x4 ah[KS], al[KS], bh[KS], bl[KS];
};
__device__ __forceinline__ void asm_wait_stage_hk(StageHK<2> &s)      // D = 3: all but the newest 2 stages (16 loads) have landed
{
    asm volatile("s_waitcnt vmcnt(16)" : "+v"(s.ah[0]), "+v"(s.ah[1]), "+v"(s.al[0]), "+v"(s.al[1]), "+v"(s.bh[0]), "+v"(s.bh[1]), "+v"(s.bl[0]),
                 "+v"(s.bl[1])::"memory");
}
__device__ __forceinline__ void asm_wait_stage_hk(StageHK<4> &s)      // D = 2: all but the newest stage (16 loads) have landed
{
    asm volatile("s_waitcnt vmcnt(16)" : "+v"(s.ah[0]), "+v"(s.ah[1]), "+v"(s.ah[2]), "+v"(s.ah[3]), "+v"(s.al[0]), "+v"(s.al[1]), "+v"(s.al[2]),
                 "+v"(s.al[3]), "+v"(s.bh[0]), "+v"(s.bh[1]), "+v"(s.bh[2]), "+v"(s.bh[3]), "+v"(s.bl[0]), "+v"(s.bl[1]), "+v"(s.bl[2]),
                 "+v"(s.bl[3])::"memory");
}
template <int EPI, int KS>
__global__ __launch_bounds__(512) void convgemm16hk_kernel(const ConvGemm16sArgs aa)
{
    static_assert(KS == 2 || KS == 4, "");
    constexpr int D = KS == 4 ? 2 : 3;                        // stages (chunks) in flight per loader lane
    constexpr int IMG = 64 * WG16Q_ROWB;                      // 4 KB: one image of one k-step
    constexpr int KBUF = 4 * IMG;                             // A hi, A lo, B hi, B lo of one k-step
    constexpr int BUF = KS * KBUF;
    constexpr int TT = 64;
    __shared__ __attribute__((aligned(16))) char smem[2 * BUF];
    const ConvGemmArgs &a = aa.c;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const Geo g = a.g;
    int nk = 0;                                               // k-steps of 32
    for (int s = 0; s < a.nseg; ++s) nk += (a.seg[s].nch + WG16_BK - 1) / WG16_BK;
    const int nchunks = (nk + KS - 1) / KS;
    const int nbar = (nchunks + D - 1) / D * D;
    const int id = (int)blockIdx.x;
    const int tx = id % aa.ntx, q = id / aa.ntx, ty = q % aa.nty, tz = q / aa.nty;
    const int t0 = tx * TT, m0 = ty * 64;
    const int b = a.row_sel1 ? tz * g.rows + a.row_sel1 - 1 : tz;
    if (m0 >= a.M) return;

    if (wave >= 4) {
        // ------------------------------- loader waves -------------------------------
        const int lt = tid - 256;
        const int r = lt & 63, kq = lt >> 6;
        const int l_off = wg16q_off(r, kq);
        const unsigned voff_a = (unsigned)((kq * 128 + (m0 & 64) + r) * 16);
        const unsigned voff_b = (unsigned)((kq * g.P + r) * 16);
        int cur_seg = 0, cur_c = 0, kstep = 0;
#if defined(WG_DBG_NOLOAD)
#define WG_LD(dst, base, voff) asm volatile("" : "=v"(dst) : "v"(voff), "s"(base))
#else
#define WG_LD(dst, base, voff) asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(dst) : "v"(voff), "s"(base) : "memory")
#endif
        const unsigned short *zsrc = aa.sseg[0].hi;
        auto issue = [&](StageHK<KS> &st) {                   // exactly 4 KS loads in straight-line code (tools/check_asm_loads.py)
#pragma unroll
            for (int j = 0; j < KS; ++j) {
                const bool live = kstep < nk;
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
                const unsigned short *ih = aa.img + ((size_t)kstep * a.lda + (m0 & ~127)) * WG16_BK, *il = ih + aa.img_stride;
                const unsigned short *row0 = ss.hi + ((size_t)bsrc * (ss.Cp >> 3) + ((ss.ch0 + cur_c) >> 3)) * g.P * 8;
                const unsigned short *pa = live ? ih : zsrc, *pl = live ? il : zsrc;
                const unsigned va = live ? voff_a : 0u;
                WG_LD(st.ah[j], pa, va);   WG_LD(st.al[j], pl, va);
                const unsigned short *pb = blive ? row0 : zsrc, *pbl = blive ? row0 + ss.lo_off : zsrc;
                const bool lane_ok = blive && (kq < 2 || full);
                const unsigned vb = lane_ok ? voff_b + (unsigned)((g.H + t0 + shift) * 16) : 0u;
                WG_LD(st.bh[j], pb, vb);   WG_LD(st.bl[j], pbl, vb);
                if (live) {
                    ++kstep;
                    cur_c += WG16_BK;
                    if (cur_c >= nch) { cur_c = 0; ++cur_seg; }
                }
            }
        };
#undef WG_LD
        auto write = [&](const StageHK<KS> &st, int buf) {
#pragma unroll
            for (int j = 0; j < KS; ++j) {
                char *sb = smem + buf * BUF + j * KBUF + l_off;
                *reinterpret_cast<u32x4 *>(sb) = st.ah[j];
                *reinterpret_cast<u32x4 *>(sb + IMG) = st.al[j];
                *reinterpret_cast<u32x4 *>(sb + 2 * IMG) = st.bh[j];
                *reinterpret_cast<u32x4 *>(sb + 3 * IMG) = st.bl[j];
            }
        };
        StageHK<KS> st[D];
#pragma unroll
        for (int i = 0; i < D; ++i) issue(st[i]);
        asm_wait_stage_hk(st[0]);
        write(st[0], 0);
        issue(st[0]);
        WG16W_BAR();                                          // chunk 0 ready
        for (int c = 0; c < nbar; c += D) {
#pragma unroll
            for (int i = 0; i < D; ++i) {
                StageHK<KS> &s = st[(i + 1) % D];
                asm_wait_stage_hk(s);
                write(s, (c + i + 1) & 1);
                issue(s);
                WG16W_BAR();
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
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
        for (int i = 0; i < 4; ++i) { f.ah[i] = rd(sb + ao + i * 1024); f.al[i] = rd(sb + IMG + ao + i * 1024); }
        f.bh = rd(sb + 2 * IMG + bo); f.bl = rd(sb + 3 * IMG + bo);
    };
    if (EPI == EPI_STORE || EPI == EPI_RESSKIP) {
        conv_acc_init_q<EPI, 1>(a, aa.saux, acc, t0, m0, b, 0, wc, lane);
    } else {
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int e = 0; e < 4; ++e) acc[i][0][e] = 0.f;
    }
    WG16W_BAR();                                              // chunk 0 ready
    Frags f0, f1;
    fetch(f0, smem);
    auto mul = [&](const Frags &f) {
#pragma unroll
        for (int mb = 0; mb < 4; ++mb) if (!TwoP<EPI>::no_alo) acc[mb][0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(f.al[mb], f.bh, acc[mb][0], 0, 0, 0);
#pragma unroll
        for (int mb = 0; mb < 4; ++mb) if (!TwoP<EPI>::no_blo) acc[mb][0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(f.ah[mb], f.bl, acc[mb][0], 0, 0, 0);
#pragma unroll
        for (int mb = 0; mb < 4; ++mb) acc[mb][0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(f.ah[mb], f.bh, acc[mb][0], 0, 0, 0);
    };
    // chunk c: its k-steps 0 .. KS-1 alternate the two fragment sets (KS is even: every chunk starts on f0); k-step j + 1 is requested from
    // the same buffer before k-step j is multiplied; the last k-step's request crosses to the other buffer behind the chunk's barrier
    for (int c = 0; c < nchunks; ++c) {
        const char *cur = smem + (c & 1) * BUF, *nxt = smem + ((c & 1) ^ 1) * BUF;
#pragma unroll
        for (int j = 0; j < KS; ++j) {
            Frags &f = (j & 1) ? f1 : f0, &fn = (j & 1) ? f0 : f1;
            __builtin_amdgcn_sched_barrier(0);
            if (j == KS - 1) {
                WG16W_BAR();                                  // chunk c is in registers everywhere, chunk c + 1 is staged
                __builtin_amdgcn_sched_barrier(0);
                fetch(fn, nxt);
            } else {
                fetch(fn, cur + (j + 1) * KBUF);
            }
            __builtin_amdgcn_sched_barrier(0);
            mul(f);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    for (int c = nchunks; c < nbar; ++c) WG16W_BAR();         // the loaders' spare iterations
    conv_epilogue_q<EPI, 1>(a, aa.s0, acc, t0, m0, b, 0, wc, lane);
}
