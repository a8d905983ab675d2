// wg_wgrad16t.h -- the weight gradients of a WN as ONE launch of one workgroup per CU ("t": every operand fragment is a transposing
// LDS read).  dW[m][n] = sum_b sum_t A[b][m][t] * B[b][n][t + shift], operands pre-split bf16 hi / lo S-planes, three
// v_mfma_f32_16x16x32_bf16 per fragment pair (a_lo b_hi + a_hi b_lo + a_hi b_hi), fp32 accumulate, split-K slabs for finalize_*.
//
// What was wrong with wgrad16s_pair_kernel (profiles/r02l_*: 1.36 ms per WN, 22 % of the training step):
//   * two 4-wave workgroups per CU on 128 x 128 tiles.  A CU's first workgroup wins issue arbitration over its second, so the 28 tiles
//     of a (layer, split) -- which read the same A row tiles and B column tiles chunk by chunk -- drift apart over their 756 chunks
//     until the shared lines have left the XCD's L2: 5.3-6.3 GB of HBM traffic per launch for 2.5 GB of operands.  (Holding them
//     together with a progress counter per set does bring the traffic down to 2.4 GB -- measured, profiles/r03b_wgrad_sync.txt -- but a
//     barrier every few chunks costs more than the bytes: 1.36 -> 2.25 ms.)
//   * v_mfma_f32_32x32x16_bf16 (the conv kernels gained from the 16x16x32 shape: same FLOPs per cycle, less energy on a chip whose
//     clock is set by its power limit, wg_gemm16q.h) and a [t][c] LDS image with 320-byte rows whose staging write is a 2-way bank
//     conflict (a third of the kernel's LDS cycles).
//
// This kernel:
//   * ONE 12-wave workgroup per CU (8 compute + 4 loader waves, the protocol of convgemm16q_kernel<.., MG = 2>) on a 256 x 128 tile:
//     48 KB of L2 -> LDS stream per chunk for two 128 x 128 tiles instead of 64 KB, and no second workgroup to lose arbitration:
//     the CUs of a set run the same instruction stream on their own CU, which is what keeps them in step.
//   * the grid is exactly the chip: 256 workgroups, workgroup id -> (XCD = id & 7, slot = id >> 3).  The host plans PHASES
//     (WgtPhase): in a phase a product's K range [k0, k1) is cut into `splits` parts and the tiles of one (layer, part) -- a SET: they
//     share operands -- take consecutive slots of ONE XCD, so that set's operand lines are fetched into that XCD's L2 once.  At the
//     headline shape: phase A = the tap / conditioning gradients (14 tiles per layer) as 2 parts on slots 0..27 of XCD `layer`, next
//     to the first half of K of that layer's W_o gradient (4 tiles) on slots 28..31; phase B = the second half of K of the W_o
//     gradients as 8 parts of 4 tiles over all 32 slots.  Every CU gets 768 + 96 chunks: the whole launch is one round with no tail
//     (the 128 x 128 form dealt 448 long and 512 short workgroups onto 512 slots).
//   * LDS image [t (32 rows)][c (128 channels)] with UNPADDED 256-byte rows; the 16-byte unit (8 channels of one time step) ch of
//     row t sits at position ch ^ F(t), F(t) = ((t >> 3) & 1) << 3 | (t & 3) << 1 | ((t >> 2) & 1).  Under that swizzle BOTH the
//     operand fetch (ds_read_b64_tr_b16: the 16-lane group g reads rows 8g + {0..3}, then 8g + 4 + {0..3}, of a 16-channel column
//     block) and the loaders' staging write (eight consecutive lanes hold eight consecutive time steps of one channel group = one
//     128-byte line of the S-plane) are bank-conflict free (tools/experiments/lds_tr_layout_check.py).
//   * the column blocks of a tile beyond the product's last column (the conditioning segment ends at column 848 of 896) are not
//     multiplied: no time on the critical path (the other waves of the workgroup still take a full chunk), but 4 % less MFMA energy.
#pragma once
#include "wg_gemm16q.h"
#include <type_traits>

#define WGT_PH_MAX 4
#define WGT_IMG (32 * 256)                                 // one [32 t][128 c] bf16 image
#define WGT_BUF (6 * WGT_IMG)                              // A: 2 row halves x (hi, lo); B: hi, lo  = 48 KB
struct WgtPhase {
    int prod;                  // which of the two grouped products
    int off, per_xcd;          // slots [off, off + per_xcd) of every XCD work on this phase
    int splits, k0, k1;        // chunk range [k0, k1) of the flattened (batch item, 32-step chunk) axis, cut into `splits` parts
    int slab0;                 // part s of this phase writes slab slab0 + s of its (product, layer)
    int tn, tiles, ngroups;    // of the product (copied here so that nothing indexes the kernel arguments with a run-time value):
};                             // 128-column tiles per row of tiles, tiles per set, layers
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
    const int tn = p.tn, T = p.tiles, ng = p.ngroups;
    const int set = local / T, tile = local - set * T;
    const int grp = set / p.splits, split = set - grp * p.splits;
    if (grp >= ng) return false;
    const int len = p.k1 - p.k0;
    it.prod = p.prod; it.grp = grp;
    it.m0 = (tile / tn) * 256; it.n0 = (tile % tn) * WG_TILE;
    it.cb = p.k0 + (int)((long)split * len / p.splits); it.ce = p.k0 + (int)((long)(split + 1) * len / p.splits);
    it.slab = p.slab0 + split;
    return it.ce > it.cb;
}

typedef __attribute__((address_space(3))) s4v *wgt_lds_s4p;
// one MFMA operand (8 consecutive time steps of the lane's channel): two 4 x 16 transposing reads
__device__ __forceinline__ bf16x8 wgt_frag(const char *p1, const char *p2)
{
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

// (the arguments are read through the kernarg segment pointer: indexing the by-value parameter with a run-time product / layer number
// makes the compiler copy all of it to scratch and index it there)
typedef const __attribute__((address_space(4))) WgtArgs *wgt_kargs_p;
typedef const __attribute__((address_space(4))) WgradSArgs *wgt_sargs_p;
__global__ __launch_bounds__(768) void wgrad16t_kernel(const WgtArgs aa_by_value)
{
    (void)aa_by_value;
    const wgt_kargs_p aa = (wgt_kargs_p)__builtin_amdgcn_kernarg_segment_ptr();
    __shared__ __attribute__((aligned(16))) char smem[2 * WGT_BUF];
    __shared__ int s_items[WGT_PH_MAX][8];
    __shared__ int s_nitems, s_total;
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
        s_nitems = n; s_total = total;
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
        const int lt = tid - 512;
        const int tl = lt & 31, cgl = lt >> 5;                // this lane's time step inside a chunk and channel group (0..7; and + 8)
        const int dst0 = wgt_off(tl, cgl), dst1 = wgt_off(tl, cgl + 8);      // unit positions inside a [32][128] image
        const unsigned short *zsrc = aa->p[0].zsrc;            // plane position 0: always-zero halo
        // unit u = 0..5: A half 0 (channel groups cgl, cgl + 8), A half 1 (same), B (same)
        const unsigned short *pu[6];
        size_t lo_[6], sb_[6];
        int it = 0, c = 0, ce = 0, bi = 0, ti = 0, gchunk = 0;     // item, chunk in K space, its (batch item, time) position
        auto setup = [&](int k) {
            // (wave-uniform values read back from LDS: readfirstlane keeps them in scalar registers, so that the kernel arguments they
            // index are fetched with scalar loads)
            const int prod = __builtin_amdgcn_readfirstlane(s_items[k][0]), grp = __builtin_amdgcn_readfirstlane(s_items[k][1]);
            const int m0 = __builtin_amdgcn_readfirstlane(s_items[k][2]), n0 = __builtin_amdgcn_readfirstlane(s_items[k][3]);
            const wgt_sargs_p a = &aa->p[prod];
            c = __builtin_amdgcn_readfirstlane(s_items[k][4]); ce = __builtin_amdgcn_readfirstlane(s_items[k][5]);
            bi = c / cpb; ti = (c - bi * cpb) * WG16_BK;
            // (segments are selected by an unrolled scan with wave-uniform indices: a lane-dependent index into the kernel arguments
            // would make the compiler keep a copy of all 4 KB of them in scratch)
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int ma = m0 + 128 * (u >> 1) + 8 * (cgl + 8 * (u & 1));
                int blk0 = 0, nch = 0, ch0 = 0, Cp = 0;
                size_t lo = 0;
                const unsigned short *ha = nullptr;
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    if (j < a->nseg_a && (ma >> 5) >= a->sa[j].blk0) {
                        blk0 = a->sa[j].blk0; nch = a->sa[j].nch; ch0 = a->sa[j].ch0; Cp = a->sa[j].Cp; lo = a->sa[j].lo_off;
                        ha = a->grp[grp].a_hi[j];
                    }
                const int ca = ma - blk0 * 32;
                pu[u] = (ma < a->Mp && ca < nch && ha) ? ha + (((size_t)((ch0 + ca) >> 3)) * gP + gH + tl) * 8 : nullptr;
                lo_[u] = lo; sb_[u] = (size_t)(Cp >> 3) * gP * 8;
            }
#pragma unroll
            for (int u = 4; u < 6; ++u) {
                const int nb = n0 + 8 * (cgl + 8 * (u & 1));
                int blk0 = 0, nch = 0, ch0 = 0, Cp = 0, bshift = 0;
                size_t lo = 0;
                const unsigned short *hb = nullptr;
#pragma unroll
                for (int j = 0; j < WG_MAX_SEG; ++j)
                    if (j < a->nseg_b && (nb >> 5) >= a->sb[j].blk0) {
                        blk0 = a->sb[j].blk0; nch = a->sb[j].nch; ch0 = a->sb[j].ch0; Cp = a->sb[j].Cp; lo = a->sb[j].lo_off;
                        hb = a->b_plane_of[j] ? a->grp[grp].b_plane[1] : a->grp[grp].b_plane[0];
                        bshift = (int)a->grp[grp].b_shift[j];
                    }
                const int cb = nb - blk0 * 32;
                pu[u] = (nb < a->Np && cb < nch && hb) ? hb + (((size_t)((ch0 + cb) >> 3)) * gP + gH + bshift + tl) * 8 : nullptr;
                lo_[u] = lo; sb_[u] = (size_t)(Cp >> 3) * gP * 8;
            }
        };
        setup(0);
#define WGT_LDP(dst, ptr) asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(dst) : "v"(ptr) : "memory")
        // exactly twelve loads in straight-line code (tools/check_asm_loads.py): lanes without a source row and chunks past the end
        // read the zero halo through selected pointers, no branch between a load and its wait
        auto issue = [&](WgtStage &st) {
            const bool live = gchunk < total;
            const size_t to = (size_t)ti * 8;
#pragma unroll
            for (int u = 0; u < 6; ++u) {
                const bool ok = live && pu[u];
                const unsigned short *q = ok ? pu[u] + bi * sb_[u] + to : zsrc, *ql = ok ? q + lo_[u] : zsrc;
                WGT_LDP(st.h[u], q);
                WGT_LDP(st.l[u], ql);
            }
            if (live) {
                ++gchunk;
                ti += WG16_BK;
                if (ti >= gTt) { ti = 0; ++bi; }
                if (++c == ce && it + 1 < nitems) setup(++it);
            }
        };
#undef WGT_LDP
        auto write = [&](const WgtStage &st, int buf) {
            char *sb = smem + buf * WGT_BUF;
#pragma unroll
            for (int u = 0; u < 6; ++u) {
                // images: A hi half 0, A hi half 1, A lo half 0, A lo half 1, B hi, B lo
                char *ih = sb + (u < 4 ? (u >> 1) * WGT_IMG : 4 * WGT_IMG) + ((u & 1) ? dst1 : dst0);
                *reinterpret_cast<u32x4 *>(ih) = st.h[u];
                *reinterpret_cast<u32x4 *>(ih + (u < 4 ? 2 * WGT_IMG : WGT_IMG)) = st.l[u];
            }
        };
        WgtStage s0, s1;
        issue(s0);
        issue(s1);
        wgt_wait_stage(s0);
        write(s0, 0);
        issue(s0);
        WG16W_BAR();                                          // buffer 0 ready
        auto iter = [&](WgtStage &st, int cc) {
            wgt_wait_stage(st);
            write(st, (cc & 1) ^ 1);
            issue(st);
            WG16W_BAR();
        };
        for (int cc = 0; cc + 1 < total; cc += 2) {           // always in pairs: see convgemm16w_kernel
            iter(s1, cc);
            iter(s0, cc + 1);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        return;
    }
    // ------------------------------- compute waves: 4 (M) x 2 (N), a 64 x 64 tile each -------------------------------
    const int wm = wave >> 1, wc = wave & 1;
    const int fg = lane >> 4, fq = (lane & 15) >> 2, fp = lane & 3;
    const int t1 = 8 * fg + fq, t2 = t1 + 4;
    // byte offsets of the lane's two reads for column block 0 of its wave tile; block b: offset ^ (b << 5) (the swizzle is an XOR on
    // the unit index, a block is two units)
    const int ao1 = (wm >> 1) * WGT_IMG + wgt_off(t1, 8 * (wm & 1) + (fp >> 1)) + 8 * (fp & 1);
    const int ao2 = (wm >> 1) * WGT_IMG + wgt_off(t2, 8 * (wm & 1) + (fp >> 1)) + 8 * (fp & 1);
    const int bo1 = 4 * WGT_IMG + wgt_off(t1, 8 * wc + (fp >> 1)) + 8 * (fp & 1);
    const int bo2 = 4 * WGT_IMG + wgt_off(t2, 8 * wc + (fp >> 1)) + 8 * (fp & 1);
#define WGT_SB() __builtin_amdgcn_sched_barrier(0)
    f32x4 acc[4][4];
    bf16x8 ah[4], al[4], bh[2], bl[2];
    auto rdA = [&](const char *buf, int mb, int lo) { return wgt_frag(buf + lo * 2 * WGT_IMG + (ao1 ^ (mb << 5)), buf + lo * 2 * WGT_IMG + (ao2 ^ (mb << 5))); };
    auto rdB = [&](const char *buf, int nb, int lo) { return wgt_frag(buf + lo * WGT_IMG + (bo1 ^ (nb << 5)), buf + lo * WGT_IMG + (bo2 ^ (nb << 5))); };
    int gc = 0;                                              // chunk index in this workgroup's stream; its buffer is gc & 1
    for (int k = 0; k < nitems; ++k) {
        const int prod = __builtin_amdgcn_readfirstlane(s_items[k][0]), grp = __builtin_amdgcn_readfirstlane(s_items[k][1]);
        const int m0 = __builtin_amdgcn_readfirstlane(s_items[k][2]), n0 = __builtin_amdgcn_readfirstlane(s_items[k][3]);
        const int nch = __builtin_amdgcn_readfirstlane(s_items[k][5] - s_items[k][4]);
        const wgt_sargs_p a = &aa->p[prod];
        // column blocks of this wave that exist (wave uniform)
        const int nbv = __builtin_amdgcn_readfirstlane(max(0, min(4, ((prod ? aa->nvalid[1] : aa->nvalid[0]) - (n0 + wc * 64) + 15) >> 4)));
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int q = 0; q < 4; ++q) acc[i][j][q] = 0.f;
        if (k == 0) WG16W_BAR();                             // buffer 0 ready (later items: published by the previous chunk's barrier)
        {
            const char *cur = smem + (gc & 1) * WGT_BUF;
#pragma unroll
            for (int i = 0; i < 4; ++i) { ah[i] = rdA(cur, i, 0); al[i] = rdA(cur, i, 1); }
            bh[0] = rdB(cur, 0, 0); bl[0] = rdB(cur, 0, 1);
        }
        // (two instances of the chunk loop: the common one, every column block valid, is branch free)
        auto chunks = [&](auto full_tag) {
            constexpr bool FULL = decltype(full_tag)::value;
            for (int c = 0; c < nch; ++c, ++gc) {
                const char *cur = smem + (gc & 1) * WGT_BUF, *nxt = smem + ((gc & 1) ^ 1) * WGT_BUF;
#pragma unroll
                for (int nb = 0; nb < 4; ++nb) {
                    const int cb = nb & 1, nx = cb ^ 1;
                    if (nb == 3) {
                        // every fragment of this chunk is in registers (the B of this last group was requested a group ago): release
                        // the buffer / publish the next one
                        WGT_SB();
                        if (gc + 1 < total || !(total & 1)) WG16W_BAR();
                    }
                    WGT_SB();
#pragma unroll
                    for (int mb = 0; mb < 4; ++mb) {
                        if (FULL || nb < nbv) {
                            acc[mb][nb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al[mb], bh[cb], acc[mb][nb], 0, 0, 0);
                            acc[mb][nb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[mb], bl[cb], acc[mb][nb], 0, 0, 0);
                            acc[mb][nb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[mb], bh[cb], acc[mb][nb], 0, 0, 0);
                        }
                        if (mb == 0) {
                            // the next group's B is requested BEHIND this group's first MFMAs (convgemm16q_kernel)
                            WGT_SB();
                            if (nb == 3) { bh[nx] = rdB(nxt, 0, 0); bl[nx] = rdB(nxt, 0, 1); }
                            else { bh[nx] = rdB(cur, nb + 1, 0); bl[nx] = rdB(cur, nb + 1, 1); }
                            WGT_SB();
                        }
                        if (nb == 3) {
                            // (unconditional: after an item's last chunk these read LDS that nothing uses; the next item starts with its
                            // own fetch)
                            WGT_SB();
                            ah[mb] = rdA(nxt, mb, 0); al[mb] = rdA(nxt, mb, 1);
                            WGT_SB();
                        }
                    }
                    WGT_SB();
                }
            }
        };
        if (nbv == 4) chunks(std::true_type{});
        else chunks(std::false_type{});
        // the item's slab: block (mb, nb): rows mb * 16 + 4 (lane >> 4) + e, column nb * 16 + (lane & 15)
        float *out = a->grp[grp].slab + (size_t)__builtin_amdgcn_readfirstlane(s_items[k][6]) * a->Mp * a->Np;
        const int col = lane & 15, rq = lane >> 4;
#pragma unroll
        for (int mb = 0; mb < 4; ++mb)
#pragma unroll
            for (int nb = 0; nb < 4; ++nb) {
                const int n = n0 + wc * 64 + nb * 16 + col;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int m = m0 + wm * 64 + mb * 16 + 4 * rq + e;
                    if (m < a->Mp && n < a->Np) out[(size_t)m * a->Np + n] = acc[mb][nb][e];
                }
            }
        WGT_SB();
    }
#undef WGT_SB
}
