// wg_wgrad16t.h -- the weight gradients of a WN as ONE launch of one workgroup per CU ("t": every operand fragment is a transposing
// LDS read).  dW[m][n] = sum_b sum_t A[b][m][t] * B[b][n][t + shift], operands pre-split bf16 hi / lo S-planes, three
// v_mfma_f32_16x16x32_bf16 per fragment pair (a_lo b_hi + a_hi b_lo + a_hi b_hi), fp32 accumulate, split-K slabs for finalize_*.
//
// What was wrong with wgrad16s_pair_kernel (profiles/r02l_*: 1.36 ms per WN, 22 % of the training step):
//   * two 4-wave workgroups per CU on 128 x 128 tiles.  A CU's first workgroup wins issue arbitration over its second, so the 28 tiles
//     of a (layer, split) -- which read the same A row tiles and B column tiles chunk by chunk -- drift apart over their 756 chunks
//     until the shared lines have left the XCD's L2: 5.3-6.3 GB of HBM traffic per launch for 2.5 GB of operands.  (Holding them
//     together with a progress counter per set does bring the traffic down to 2.4 GB -- measured, profiles/r03b_wgrad_sync.txt -- but a
//     rendezvous every few chunks costs more than the bytes: 1.36 -> 2.25 ms.)
//   * v_mfma_f32_32x32x16_bf16 (the conv kernels gained from the 16x16x32 shape: same FLOPs per cycle, less energy on a chip whose
//     clock is set by its power limit, wg_gemm16q.h) and a [t][c] LDS image with 320-byte rows whose staging write is a 2-way bank
//     conflict (a third of the kernel's LDS cycles).
//
// This kernel:
//   * ONE 12-wave workgroup per CU -- 8 compute waves (4 x 2, a 64 x 64 tile each) + 4 loader waves -- on a 256 x 128 tile: 48 KB of
//     L2 -> LDS stream per chunk for two 128 x 128 tiles instead of 64 KB, and no second workgroup to lose arbitration: the CUs of a set
//     run the same instruction stream on their own CU, which keeps them in step without any cross-workgroup synchronisation (measured:
//     2.7 GB of HBM traffic per launch, profiles/r03d_hbm_traffic.json).
//   * the grid is exactly the chip: 256 workgroups, workgroup id -> (XCD = id & 7, slot = id >> 3).  The host plans PHASES
//     (WgtPhase, plan_wgt in wgflow.hip): in a phase a product's K range [k0, k1) is cut into `splits` parts and the tiles of one
//     (layer, part) -- a SET: they share operands -- take consecutive slots of ONE XCD, so that set's operand lines are fetched into
//     that XCD's L2 once.  At the headline shape: phase A = the tap / conditioning gradients (14 tiles per layer) as 2 parts on slots
//     0..27 of XCD `layer`, next to the first half of K of that layer's W_o gradient (4 tiles) on slots 28..31; phase B = the second
//     half of K of the W_o gradients as 8 parts of 4 tiles over all 32 slots.  Every CU gets 768 + 96 chunks: the whole launch is one
//     round with no tail (the 128 x 128 form dealt 448 long and 512 short workgroups onto 512 slots).
//   * LDS image [t (32 rows)][c (128 channels)] with UNPADDED 256-byte rows; the 16-byte unit (8 channels of one time step) ch of
//     row t sits at position ch ^ F(t), F(t) = ((t >> 3) & 1) << 3 | (t & 3) << 1 | ((t >> 2) & 1).  Under that swizzle BOTH the
//     operand fetch (ds_read_b64_tr_b16: the 16-lane group g reads rows 8g + {0..3}, then 8g + 4 + {0..3}, of a 16-channel column
//     block) and the loaders' staging write (eight consecutive lanes hold eight consecutive time steps of one channel group = one
//     128-byte line of the S-plane) are bank-conflict free (tools/experiments/lds_tr_layout_check.py; PMC: 0 conflict cycles).
//   * a ring of THREE chunk buffers in LDS and NO workgroup barrier in the chunk loop: loaders and compute waves hand chunks over
//     through per-wave progress words in LDS (see the loader loop).  With a barrier per chunk the launch needed 2 470-2 570 cycles per
//     chunk for the ~1 300-1 540 cycles its MFMAs take (profiles/r03c_wgrad_phases.txt).
//   * the loader waves' addresses are scalar (an SGPR base per unit that the scalar unit advances, one constant lane offset): a loader
//     wave shares its SIMD with two compute waves, and every vector-ALU instruction it issues is an MFMA issue slot lost.
#pragma once
#include "wg_gemm16q.h"
#include <type_traits>

#define WGT_PH_MAX 8
#if !defined(WGT_POLL_SLEEP)
#define WGT_POLL_SLEEP 2                                   // the loaders look at the compute waves' progress every 64 x this many cycles
#endif
#define WGT_SPIN_CAP (1 << 22)                             // polls of a hand-over word before a wave gives up (about a second) and the result is poisoned
#if !defined(WGT_BAR_GROUP)
#define WGT_BAR_GROUP 3                                    // the chunk's barrier sits in front of this MFMA group
#endif
#define WGT_IMG (32 * 256)                                 // one [32 t][128 c] bf16 image
#define WGT_BUF (6 * WGT_IMG)                              // A: 2 row halves x (hi, lo); B: hi, lo  = 48 KB
struct WgtPhase {
    int prod;                  // which of the two grouped products
    int off, per_xcd;          // slots [off, off + per_xcd) of every XCD work on this phase
    int splits, k0, k1;        // chunk range [k0, k1) of the flattened (batch item, 32-step chunk) axis, cut into `splits` parts
    int slab0;                 // part s of this phase writes slab slab0 + s of its (product, layer)
    int tn, tiles, ngroups;    // of the product (copied here so that nothing indexes the kernel arguments with a run-time value):
                               // 128-column tiles per row of tiles, tiles per set, layers
    // A product whose row of tiles is wider than an XCD has slots (WSRGlow: 35 column tiles x 2 row tiles per layer) is cut into `nsub`
    // column SUB-SETS of `tns` tile columns per layer (`tiles` = row tiles x tns then; they still share the A operand, and each its own
    // B columns), and a phase may start at set `set0` of the enumeration (layer, sub-set, part): the sets then take several ROUNDS of
    // the chip, one phase entry each (plan_wgt_rounds in wgflow.hip).  nsub = 1, tns = tn, set0 = 0: the whole row of tiles, as above.
    int nsub, tns, set0;
};
struct WgtArgs {
    WgradSArgs p[2];           // segments, planes and slabs of the two products (as wgrad16s_pair_kernel); p[].nsplit = slabs per layer
    WgtPhase ph[WGT_PH_MAX];
    int nph;
    int nvalid[2];             // columns that exist (a multiple of 16); blocks beyond are not multiplied
};
static_assert(sizeof(WgtArgs) <= 4096, "kernel arguments are limited to 4 KB");

__device__ __forceinline__ int wgt_F(int t) { return (((t >> 3) & 1) << 3) | ((t & 3) << 1) | ((t >> 2) & 1); }
__device__ __forceinline__ int wgt_off(int t, int ch) { return t * 256 + ((ch ^ wgt_F(t)) << 4); }

struct WgtItem {
    int prod, grp, m0, n0, cb, ce, slab;
};
// the work of slot (xcd, r) in phase `ph` (false: none)
__device__ __forceinline__ bool wgt_item(const __attribute__((address_space(4))) WgtArgs *aa, int ph, int xcd, int r, WgtItem &it)
{
    const __attribute__((address_space(4))) WgtPhase &p = aa->ph[ph];                           // (ph is a compile-time constant at every call: no dynamic index into the arguments)
    if (r < p.off || r >= p.off + p.per_xcd) return false;
    const int local = xcd * p.per_xcd + (r - p.off);
    const int T = p.tiles, tns = p.tns, nsub = p.nsub;
    const int set = p.set0 + local / T, tile = local % T;
    const int unit = set / p.splits, split = set - unit * p.splits;
    const int grp = unit / nsub, sub = unit - grp * nsub;
    if (grp >= p.ngroups) return false;
    const int nt = sub * tns + tile % tns;                  // (a ragged last sub-set leaves slots idle)
    if (nt >= p.tn) return false;
    const int len = p.k1 - p.k0;
    it.prod = p.prod; it.grp = grp;
    it.m0 = (tile / tns) * 256; it.n0 = nt * WG_TILE;
    it.cb = p.k0 + (int)((long)split * len / p.splits); it.ce = p.k0 + (int)((long)(split + 1) * len / p.splits);
    it.slab = p.slab0 + split;
    return it.ce > it.cb;
}

typedef __attribute__((address_space(3))) s4v *wgt_lds_s4p;
// one MFMA operand (8 consecutive time steps of the lane's channel): two 4 x 16 transposing reads
__device__ __forceinline__ bf16x8 wgt_frag(const char *p1, const char *p2)
{
#if defined(WG_DBG_B128READ)   // timing experiment only (garbage results): one 16-byte read per fragment instead of two transposing 8-byte reads
    (void)p2;
    return *reinterpret_cast<const bf16x8 *>((const char *)((size_t)p1 & ~(size_t)15));
#endif
    const s4v lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((wgt_lds_s4p)p1);
    const s4v hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((wgt_lds_s4p)p2);
    bf16x8 r;
    r[0] = lo[0]; r[1] = lo[1]; r[2] = lo[2]; r[3] = lo[3];
    r[4] = hi[0]; r[5] = hi[1]; r[6] = hi[2]; r[7] = hi[3];
    return r;
}

struct WgtStage {                                          // one chunk of a loader lane: six units, hi and lo
    u32x4 h[6], l[6];
};
__device__ __forceinline__ void wgt_wait_stage(WgtStage &s)  // all but the newest 12 loads have landed: stage s is complete
{
    asm volatile("s_waitcnt vmcnt(12)"
                 : "+v"(s.h[0]), "+v"(s.h[1]), "+v"(s.h[2]), "+v"(s.h[3]), "+v"(s.h[4]), "+v"(s.h[5]), "+v"(s.l[0]), "+v"(s.l[1]), "+v"(s.l[2]),
                   "+v"(s.l[3]), "+v"(s.l[4]), "+v"(s.l[5])::"memory");
}
__device__ __forceinline__ void wgt_drain(WgtStage &s)       // every load has landed
{
    asm volatile("s_waitcnt vmcnt(0)"
                 : "+v"(s.h[0]), "+v"(s.h[1]), "+v"(s.h[2]), "+v"(s.h[3]), "+v"(s.h[4]), "+v"(s.h[5]), "+v"(s.l[0]), "+v"(s.l[1]), "+v"(s.l[2]),
                   "+v"(s.l[3]), "+v"(s.l[4]), "+v"(s.l[5])::"memory");
}

// (the arguments are read through the kernarg segment pointer: indexing the by-value parameter with a run-time product / layer number
// makes the compiler copy all of it to scratch and index it there)
typedef const __attribute__((address_space(4))) WgtArgs *wgt_kargs_p;
typedef const __attribute__((address_space(4))) WgradSArgs *wgt_sargs_p;
__global__ __launch_bounds__(768) void wgrad16t_kernel(const WgtArgs aa_by_value)
{
    (void)aa_by_value;
    const wgt_kargs_p aa = (wgt_kargs_p)__builtin_amdgcn_kernarg_segment_ptr();
    __shared__ __attribute__((aligned(16))) char smem[3 * WGT_BUF];
    __shared__ int s_items[WGT_PH_MAX][8];
    __shared__ int s_nitems, s_total;
    __shared__ volatile int s_fail;                           // a wave gave up waiting for a hand-over: the slabs are poisoned with NaN
    __shared__ __attribute__((aligned(16))) unsigned s_ready[4], s_done[8];                // chunks staged, per loader wave / chunks read, per compute wave
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int xcd = (int)blockIdx.x & 7, slot = (int)blockIdx.x >> 3;
    if (tid == 0) {
        int n = 0, total = 0;
#pragma unroll
        for (int ph = 0; ph < WGT_PH_MAX; ++ph) {
            WgtItem it;
            if (ph < aa->nph && wgt_item(aa, ph, xcd, slot, it)) {
                s_items[n][0] = it.prod; s_items[n][1] = it.grp; s_items[n][2] = it.m0; s_items[n][3] = it.n0;
                s_items[n][4] = it.cb; s_items[n][5] = it.ce; s_items[n][6] = it.slab;
                total += it.ce - it.cb;
                ++n;
            }
        }
        s_nitems = n; s_total = total; s_fail = 0;
        for (int i = 0; i < 4; ++i) s_ready[i] = 0;
        for (int i = 0; i < 8; ++i) s_done[i] = 0;
    }
    __syncthreads();
    const int nitems = __builtin_amdgcn_readfirstlane(s_nitems), total = __builtin_amdgcn_readfirstlane(s_total);
    if (total == 0) return;
    const int gP = aa->p[0].g.P, gH = aa->p[0].g.H, gTt = aa->p[0].g.Tt;
    const int cpb = aa->p[0].cpb;

    if (wave >= 8) {
        // ------------------------------- loader waves (4: twelve 16-byte units per lane and chunk) -------------------------------
        // 8 compute + 4 loader waves = 3 waves per SIMD = a 170-register budget: at 4 per SIMD (128 registers) the compute waves' 64
        // accumulators + 48 fragment registers + the swizzled addresses of the transposing reads spilled inside the chunk loop.
        //
        // A loader wave shares its SIMD with two compute waves, and every vector-ALU instruction it issues takes issue cycles from
        // their MFMAs.  Per-lane 64-bit source pointers (a multiply-add, an add and four selects per load) cost the launch a third of
        // its time -- the chunk took 2 470 cycles with the loads removed but the address arithmetic left in, 1 530 with the loader waves
        // gone (profiles/r03c_wgrad_phases.txt).  So the addresses are SCALAR: a wave fetches, per unit, 2 channel groups x 32 time steps
        // = 16 channels of one 32-channel block, which lie in ONE operand segment; its source is an SGPR pointer that the scalar unit
        // advances chunk by chunk, plus one 32-bit lane offset that never changes.
#if defined(WG_DBG_NOLOADER)   // timing experiment only: the compute waves alone (with their barriers)
        return;
#endif
        const int lw = wave - 8;
        const int tl = lane & 31, half = lane >> 5;           // this lane's time step inside a chunk; which of the wave's two channel groups
        const unsigned voff = (unsigned)((half * gP + tl) * 16);
        const int dst0 = wgt_off(tl, 2 * lw + half), dst1 = wgt_off(tl, 2 * lw + half + 8);      // unit positions inside a [32][128] image
        const unsigned short *zsrc = aa->p[0].zsrc;           // plane position 0: always-zero halo
        // unit u = 0..5: A half 0 (channel groups 2 lw + {0, 1}, then + 8), A half 1 (same), B (same).  All wave uniform:
        const unsigned short *cur[6];                         // hi plane at the chunk being requested (channel group 2 lw (+ 8), time step 0 of the chunk)
        long lo_[6], sb_[6];                                  // hi -> lo plane, one batch item further (elements)
        bool ok[6];
        int it = 0, c = 0, ce = 0, ti = 0, gchunk = 0;        // item, chunk in K space, its time position
        auto uni = [](const unsigned short *q) {              // (keeps a pointer in scalar registers)
            const unsigned long long v = (unsigned long long)q;
            const unsigned lo32 = __builtin_amdgcn_readfirstlane((unsigned)v), hi32 = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
            return (const unsigned short *)(((unsigned long long)hi32 << 32) | lo32);
        };
        auto setup = [&](int k) {
            // (wave-uniform values read back from LDS: readfirstlane keeps them in scalar registers, so that the kernel arguments they
            // index are fetched with scalar loads)
            const int prod = __builtin_amdgcn_readfirstlane(s_items[k][0]), grp = __builtin_amdgcn_readfirstlane(s_items[k][1]);
            const int m0 = __builtin_amdgcn_readfirstlane(s_items[k][2]), n0 = __builtin_amdgcn_readfirstlane(s_items[k][3]);
            const wgt_sargs_p a = &aa->p[prod];
            c = __builtin_amdgcn_readfirstlane(s_items[k][4]); ce = __builtin_amdgcn_readfirstlane(s_items[k][5]);
            const int bi = c / cpb;
            ti = (c - bi * cpb) * WG16_BK;
            // (segments are selected by an unrolled scan with constant indices: a run-time index into the kernel arguments would
            // make the compiler keep a copy of all of them in scratch)
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int ma = m0 + 128 * (u >> 1) + 64 * (u & 1) + 16 * lw;      // first of the wave's 16 rows of this unit
                int blk0 = 0, nch = 0, ch0 = 0, Cp = 0;
                long lo = 0;
                const unsigned short *ha = nullptr;
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    if (j < a->nseg_a && (ma >> 5) >= a->sa[j].blk0) {
                        blk0 = a->sa[j].blk0; nch = a->sa[j].nch; ch0 = a->sa[j].ch0; Cp = a->sa[j].Cp; lo = (long)a->sa[j].lo_off;
                        ha = a->grp[grp].a_hi[j];
                    }
                const int ca = ma - blk0 * 32;                // (segment sizes are multiples of 16 on this path: 16 rows are valid or none)
                ok[u] = ma < a->Mp && ca < nch && ha != nullptr;
                sb_[u] = (long)(Cp >> 3) * gP * 8; lo_[u] = lo;
                cur[u] = uni(ok[u] ? ha + ((long)((ch0 + ca) >> 3) * gP + gH) * 8 + bi * sb_[u] + (long)ti * 8 : zsrc);
            }
#pragma unroll
            for (int u = 4; u < 6; ++u) {
                const int nb = n0 + 64 * (u & 1) + 16 * lw;
                int blk0 = 0, nch = 0, ch0 = 0, Cp = 0, bshift = 0;
                long lo = 0;
                const unsigned short *hb = nullptr;
#pragma unroll
                for (int j = 0; j < WG_MAX_SEG; ++j)
                    if (j < a->nseg_b && (nb >> 5) >= a->sb[j].blk0) {
                        blk0 = a->sb[j].blk0; nch = a->sb[j].nch; ch0 = a->sb[j].ch0; Cp = a->sb[j].Cp; lo = (long)a->sb[j].lo_off;
                        hb = a->b_plane_of[j] == 0 ? a->grp[grp].b_plane[0] : a->b_plane_of[j] == 1 ? a->grp[grp].b_plane[1] : a->grp[grp].b_plane[2];
                        bshift = (int)a->grp[grp].b_shift[j];
                    }
                const int cb = nb - blk0 * 32;
                ok[u] = nb < a->Np && cb < nch && hb != nullptr;
                sb_[u] = (long)(Cp >> 3) * gP * 8; lo_[u] = lo;
                cur[u] = uni(ok[u] ? hb + ((long)((ch0 + cb) >> 3) * gP + gH + bshift) * 8 + bi * sb_[u] + (long)ti * 8 : zsrc);
            }
        };
        setup(0);
#if defined(WG_DBG_NOLOAD)     // timing experiment only: the loaders write whatever their staging registers hold
#define WGT_LDP(dst, base, vo) asm volatile("" : "=v"(dst) : "v"(vo), "s"(base))
#else
#define WGT_LDP(dst, base, vo) asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(dst) : "v"(vo), "s"(base) : "memory")
#endif
        // exactly twelve loads in straight-line code (tools/check_asm_loads.py): units without a source and chunks past the end read the
        // zero halo (selected base, lane offset 0), no branch between a load and its wait
        auto issue = [&](WgtStage &st) {
            const bool live = gchunk < total;
#pragma unroll
            for (int u = 0; u < 6; ++u) {
                const bool on = live && ok[u];
                const unsigned short *q = on ? cur[u] : zsrc, *ql = on ? cur[u] + lo_[u] : zsrc;
                const unsigned vo = on ? voff : 0u;
                WGT_LDP(st.h[u], q, vo);
                WGT_LDP(st.l[u], ql, vo);
            }
            if (live) {
                ++gchunk;
                ti += WG16_BK;
                const bool wrap = ti >= gTt;                  // next batch item: one plane further, back to its first time step
                if (wrap) ti = 0;
#pragma unroll
                for (int u = 0; u < 6; ++u) cur[u] += wrap ? sb_[u] - (long)(gTt - WG16_BK) * 8 : (long)WG16_BK * 8;
                if (++c == ce && it + 1 < nitems) setup(++it);
            }
        };
#undef WGT_LDP
        auto write = [&](const WgtStage &st, int buf) {
#if defined(WG_DBG_NOWRITE)    // timing experiment only
            return;
#endif
            char *sb = smem + buf * WGT_BUF;
#pragma unroll
            for (int u = 0; u < 6; ++u) {
                // images: A hi half 0, A hi half 1, A lo half 0, A lo half 1, B hi, B lo
                char *ih = sb + (u < 4 ? (u >> 1) * WGT_IMG : 4 * WGT_IMG) + ((u & 1) ? dst1 : dst0);
                *reinterpret_cast<u32x4 *>(ih) = st.h[u];
                *reinterpret_cast<u32x4 *>(ih + (u < 4 ? 2 * WGT_IMG : WGT_IMG)) = st.l[u];
            }
        };
        // LDS ring of THREE chunk buffers and NO workgroup barrier in the chunk loop.  Chunk j lives in buffer j % 3.  The hand-over is
        // a progress word per wave in LDS: a loader wave stores j + 1 into s_ready[its index] behind its twelve staging writes of
        // chunk j; a compute wave stores j + 1 into s_done[its index] behind its last fragment request from chunk j.  Chunk j is complete
        // when min(s_ready) > j; buffer j % 3 may be refilled (with chunk j + 3) when min(s_done) > j.  The LDS executes a wave's
        // operations in order, so a progress store is ordered behind the data operations it stands for without any wait.  Everybody only
        // ever waits for what it needs: the compute waves never wait for each other, and the loaders run up to two chunks ahead in LDS
        // plus two in their register stages.
        // Why: with one barrier per chunk (every wave of the workgroup at the same point of every chunk) the launch took 2 470-2 570 cycles
        // per chunk for 1 536 cycles of MFMAs -- 1 890 with the compute waves alone and their barrier, 1 529 without barriers: phase-locked
        // at a barrier the two compute waves of a SIMD wait for their fragments at the same moments instead of filling each other's gaps,
        // and every wave waits for the slowest of twelve (profiles/r03c_wgrad_phases.txt).
        // (WGT_ORDER: a compiler-level fence -- no instruction -- on either side of every progress word access, so that the data reads /
        // writes the word stands for cannot be moved across it by any pass; the hardware side is the in-order LDS)
#define WGT_ORDER() __atomic_signal_fence(__ATOMIC_SEQ_CST)
        auto wait_done = [&](unsigned want) {                 // every compute wave has read chunk want - 1
            for (int spins = 0;;) {
                const unsigned v = __hip_atomic_load(&s_done[lane & 7], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                unsigned m = __builtin_amdgcn_readlane(v, 0);
#pragma unroll
                for (int i = 1; i < 8; ++i) m = min(m, (unsigned)__builtin_amdgcn_readlane(v, i));
                if (m >= want) break;
                __builtin_amdgcn_s_sleep(WGT_POLL_SLEEP);
                if (++spins > WGT_SPIN_CAP) { s_fail = 1; break; }      // (a hand-over that never comes is a bug: poison the result, do not hang the GPU)
            }
            WGT_ORDER();
        };
        WgtStage s0, s1;
        issue(s0);                                            // chunk 0
        issue(s1);                                            // chunk 1
        int wb = 0, j = 0;                                    // chunk being staged and its buffer
        auto iter = [&](WgtStage &st) {                       // chunk j (landed in `st`) -> LDS, chunk j + 2 requested
            wgt_wait_stage(st);
            if (j >= 3) wait_done((unsigned)(j - 2));
            write(st, wb);
            WGT_ORDER();
            __hip_atomic_store(&s_ready[lw], (unsigned)(j + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            wb = wb == 2 ? 0 : wb + 1;
            ++j;
            issue(st);
        };
        for (int cc = 0; cc + 1 < total; cc += 2) {           // (the two stages strictly alternate on every path: tools/check_asm_loads.py)
            iter(s0);
            iter(s1);
        }
        if (total & 1) iter(s0);
        // (the trailing loads -- zero-halo data nobody reads -- are retired HERE, with both stages named: a register the compiler
        // believes dead it would hand out again while the load that targets it is still in flight)
        wgt_drain(s0);
        wgt_drain(s1);
        return;
    }
    // ------------------------------- compute waves: 4 (M) x 2 (N), a 64 x 64 tile each -------------------------------
#if defined(WG_DBG_HALFWAVES)  // timing experiment only: ONE compute wave per SIMD (how fast is a wave's instruction stream alone?)
    if (wave >= 4) return;
#endif
#if defined(WGT_COMPUTE_PRIO)
    __builtin_amdgcn_s_setprio(WGT_COMPUTE_PRIO);            // experiment: the compute waves' instructions ahead of the loaders'
#endif
    const int wm = wave >> 1, wc = wave & 1;
    const int fg = lane >> 4, fq = (lane & 15) >> 2, fp = lane & 3;
    const int t1 = 8 * fg + fq, t2 = t1 + 4;
    // byte offsets of the lane's two reads for column block 0 of its wave tile; block b: offset ^ (b << 5) (the swizzle is an XOR on
    // the unit index, a block is two units)
    const int ao1 = (wm >> 1) * WGT_IMG + wgt_off(t1, 8 * (wm & 1) + (fp >> 1)) + 8 * (fp & 1);
    const int ao2 = (wm >> 1) * WGT_IMG + wgt_off(t2, 8 * (wm & 1) + (fp >> 1)) + 8 * (fp & 1);
    const int bo1 = 4 * WGT_IMG + wgt_off(t1, 8 * wc + (fp >> 1)) + 8 * (fp & 1);
    const int bo2 = 4 * WGT_IMG + wgt_off(t2, 8 * wc + (fp >> 1)) + 8 * (fp & 1);
#if defined(WGT_NO_SB)         // experiment: the compiler's own order of the chunk loop
#define WGT_SB() do { } while (0)
#else
#define WGT_SB() __builtin_amdgcn_sched_barrier(0)
#endif
#if defined(WG_DBG_TRACE)      // phase stamps (slots 8..15 of the conv kernels' trace arrays): start, first barrier, then per item main loop / slab done
#define WGT_TRACE(slot) do { if (lane == 0 && wave == 0 && (slot) < 8) { \
        wg_dbg_trace[blockIdx.x * 16 + 8 + (slot)] = wall_clock64(); wg_dbg_trace_cyc[blockIdx.x * 16 + 8 + (slot)] = clock64(); } } while (0)
#else
#define WGT_TRACE(slot) do { } while (0)
#endif
    WGT_TRACE(0);
    f32x4 acc[4][4];
    // fragments: the four A row blocks of the chunk (hi, lo) stay resident, B is double-buffered one group (column block) ahead
    bf16x8 ah[4], al[4], bh[2], bl[2];
    auto rdA = [&](const char *buf, int mb, int lo) { return wgt_frag(buf + lo * 2 * WGT_IMG + (ao1 ^ (mb << 5)), buf + lo * 2 * WGT_IMG + (ao2 ^ (mb << 5))); };
    auto rdB = [&](const char *buf, int nb, int lo) { return wgt_frag(buf + lo * WGT_IMG + (bo1 ^ (nb << 5)), buf + lo * WGT_IMG + (bo2 ^ (nb << 5))); };
    int gc = 0, bcur = 0;                                    // chunk index in this workgroup's stream; its buffer (gc % 3)
    // one chunk: 4 groups (column blocks) of 12 MFMAs.  Behind the first MFMAs of group nb the wave requests the B fragments of the next
    // group (group 3: block 0 of the NEXT chunk, whose buffer has been complete since the previous barrier); in group 3 every A
    // fragment is re-fetched from the next chunk as soon as its last MFMAs are issued (convgemm16q_kernel's register budget: a second
    // set of A fragments, fetched a whole chunk ahead, spilled at three waves per SIMD).
    auto wait_ready = [&](unsigned want) {                   // all four loader waves have staged chunk want - 1
#if defined(WG_DBG_NOLOADER) || defined(WG_DBG_NOWAITREADY)
        return;
#endif
        for (int spins = 0;;) {
            const unsigned v = __hip_atomic_load(&s_ready[lane & 3], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            const unsigned m = min(min((unsigned)__builtin_amdgcn_readlane(v, 0), (unsigned)__builtin_amdgcn_readlane(v, 1)),
                                   min((unsigned)__builtin_amdgcn_readlane(v, 2), (unsigned)__builtin_amdgcn_readlane(v, 3)));
            if (m >= want) break;
            __builtin_amdgcn_s_sleep(1);
            if (++spins > WGT_SPIN_CAP) { s_fail = 1; break; }
        }
        WGT_ORDER();
    };
    auto chunk = [&](auto full_tag, int nbv) {
        constexpr bool FULL = decltype(full_tag)::value;
        const int bnxt = bcur == 2 ? 0 : bcur + 1;
        const char *cur = smem + bcur * WGT_BUF, *nxt = smem + bnxt * WGT_BUF;
#pragma unroll
        for (int nb = 0; nb < 4; ++nb) {
            const int cb = nb & 1, nx = cb ^ 1;
#if !defined(WG_DBG_NOHANDOVER)
            if (nb == 3 && gc + 1 < total) {
                WGT_SB();
                wait_ready((unsigned)(gc + 2));          // the next chunk is in LDS (it normally has been for a while)
            }
#endif
            WGT_SB();
#pragma unroll
            for (int mb = 0; mb < 4; ++mb) {
#if defined(WG_DBG_NOMFMA)     // timing experiment only: fragments are fetched, nothing is multiplied
                if (false) {
#else
                if (FULL || nb < nbv) {
#endif
                    if (!(WG_OPT_2P & 16)) acc[mb][nb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al[mb], bh[cb], acc[mb][nb], 0, 0, 0);
                    if (!(WG_OPT_2P & 32)) acc[mb][nb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[mb], bl[cb], acc[mb][nb], 0, 0, 0);
                    acc[mb][nb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[mb], bh[cb], acc[mb][nb], 0, 0, 0);
                }
                // the next group's B fragments in two halves behind the first and the second row block's MFMAs: fewer LDS instructions in a
                // row between two MFMAs of this wave (a wave alone on its SIMD issues an MFMA every 23.5 instead of every 13.6 cycles)
                if (mb == 0) {
                    WGT_SB();
                    bh[nx] = nb == 3 ? rdB(nxt, 0, 0) : rdB(cur, nb + 1, 0);
                    WGT_SB();
                }
                if (mb == 1) {
                    WGT_SB();
                    bl[nx] = nb == 3 ? rdB(nxt, 0, 1) : rdB(cur, nb + 1, 1);
#if !defined(WG_DBG_NOHANDOVER)
                    if (nb == 2) {
                        WGT_ORDER();
                        __hip_atomic_store(&s_done[wave], (unsigned)(gc + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    }
#endif
                    WGT_SB();
                }
                if (nb == 3) {
                    WGT_SB();
                    ah[mb] = rdA(nxt, mb, 0); al[mb] = rdA(nxt, mb, 1);
                    WGT_SB();
                }
            }
            WGT_SB();
        }
        bcur = bnxt;
        ++gc;
    };
    for (int k = 0; k < nitems; ++k) {
        const int prod = __builtin_amdgcn_readfirstlane(s_items[k][0]), grp = __builtin_amdgcn_readfirstlane(s_items[k][1]);
        const int m0 = __builtin_amdgcn_readfirstlane(s_items[k][2]), n0 = __builtin_amdgcn_readfirstlane(s_items[k][3]);
        const int nch = __builtin_amdgcn_readfirstlane(s_items[k][5] - s_items[k][4]);
        const wgt_sargs_p a = &aa->p[prod];
        // column blocks of this wave that exist (wave uniform)
        const int nbv = __builtin_amdgcn_readfirstlane(max(0, min(4, ((prod ? aa->nvalid[1] : aa->nvalid[0]) - (n0 + wc * 64) + 15) >> 4)));
        (void)nbv;
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int q = 0; q < 4; ++q) acc[i][j][q] = 0.f;
        if (k == 0) {
            wait_ready(1u);                                  // chunk 0 is in LDS
            WGT_TRACE(1);
#pragma unroll
            for (int i = 0; i < 4; ++i) { ah[i] = rdA(smem, i, 0); al[i] = rdA(smem, i, 1); }
            bh[0] = rdB(smem, 0, 0); bl[0] = rdB(smem, 0, 1);
        }
        for (int c = 0; c < nch; ++c) chunk(std::true_type{}, 4);
        WGT_TRACE(2 + 2 * k);
        // the item's slab: block (mb, nb): rows mb * 16 + 4 (lane >> 4) + e, column nb * 16 + (lane & 15)
        float *out = a->grp[grp].slab + (size_t)__builtin_amdgcn_readfirstlane(s_items[k][6]) * a->Mp * a->Np;
        const float poison = s_fail ? __builtin_nanf("") : 0.f;
        const int col = lane & 15, rq = lane >> 4;
#pragma unroll
        for (int mb = 0; mb < 4; ++mb)
#pragma unroll
            for (int nb = 0; nb < 4; ++nb) {
                const int n = n0 + wc * 64 + nb * 16 + col;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int m = m0 + wm * 64 + mb * 16 + 4 * rq + e;
                    if (m < a->Mp && n < a->Np) out[(size_t)m * a->Np + n] = acc[mb][nb][e] + poison;
                }
            }
        WGT_TRACE(3 + 2 * k);
        WGT_SB();
    }
#undef WGT_SB
}
