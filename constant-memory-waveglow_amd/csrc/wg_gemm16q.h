// wg_gemm16q.h -- the wave-specialised conv kernel on v_mfma_f32_16x16x32_bf16 ("q": a lane owns a quad of 4 rows per 16x16 block).
//
// Why another MFMA shape.  The conv kernels do not run at the clock the roofline assumes: stamped with s_memtime / s_memrealtime
// (-DWG_DBG_TRACE) the main loop of the gate conv holds 1.34-1.42 GHz on random data -- the chip's power management, not the issue
// stream, sets the rate of the matrix pipe.  On this chip the 16x16x32 shape does the same FLOPs per cycle for less energy:
// tools/experiments/shape_probe.hip (a compute wave's chunk loop alone: 64x64 tile per wave, all fragments re-read from LDS, three
// products per fragment pair, random data, 8 waves per CU) measured 2052-2074 TFLOP/s at 2.04-2.09 GHz against 1850 TFLOP/s at
// 1.85 GHz for 32x32x16 -- +11 % for the same cycles per FLOP (MI355X_MICROARCH.md, "DVFS give-back" item 7).
//
// What changes against convgemm16w (wg_gemm16s.h), whose workgroup structure, loader waves, barrier protocol and persistent tile
// walk are kept:
//   * LDS images are UNPADDED 64-byte rows [row][32 k]; the 16-byte unit of k-group q of row r sits at position q ^ F(r),
//     F(r) = {0,2,3,1}[(r >> 1) & 3].  With that one swizzle both access patterns are bank-conflict free (searched exhaustively,
//     tools/experiments/lds_layout_search.py): the 16x16x32 fragment read (lane l: row l & 15, k-group l >> 4) and the loaders'
//     lane-linear staging write (consecutive lanes -> consecutive rows of one k-group).  64 KB of LDS per workgroup instead of 80.
//   * a chunk (32 k) is ONE k-step of 12 NB MFMAs (NB = column blocks of 16 per wave), issued as NB groups: group nb multiplies the
//     four row blocks by column block nb.  Registers: all four A fragments of the chunk (hi, lo: 32 VGPRs) stay resident, B is
//     double-buffered one group ahead (16 VGPRs), 64 accumulators: 112.  The chunk's barrier sits in front of its LAST group: by then
//     every fragment of the chunk is in registers (the loaders may refill the buffer) and the next chunk has been staged; the last
//     group reloads each A fragment from the next buffer as soon as its MFMAs are issued and fetches the next chunk's first B.
//   * accumulator layout: block (mb, nb) of a wave's 64 x 32 NI tile: rows mb*16 + 4*(lane >> 4) + e (e = register 0..3), column
//     nb*16 + (lane & 15).  A lane still owns 4 consecutive channels of one time step, so S-plane stores are unchanged in kind
//     (8 bytes per lane, 2 x 256 contiguous bytes per wave instruction).
#pragma once
#include "wg_gemm16s.h"
#include <type_traits>

typedef float f32x4 __attribute__((ext_vector_type(4)));

#define WG16Q_ROWB 64                                     // bytes per LDS row: 32 bf16, no padding
__device__ __forceinline__ int wg16q_swz(int row) { return (0x1320 >> (4 * ((row >> 1) & 3))) & 3; }     // F(row): nibbles 0,2,3,1
// byte offset of the unit of k-group kg of image row `row`
__device__ __forceinline__ int wg16q_off(int row, int kg) { return row * WG16Q_ROWB + ((kg ^ wg16q_swz(row)) << 4); }

// ------------------------------------------------------------------------------------------------
// epilogues in the 16x16 accumulator layout (same contracts as conv_acc_init / conv_epilogue_s)
// ------------------------------------------------------------------------------------------------
// The saved tanh / sigmoid planes of a layer are private to these kernels (written by the gate conv of the recompute pass or of a
// stored-activation forward, read once by the gate backward): they are kept CHANNEL-INTERLEAVED, [b][c / 4][t][4] fp32, so that the
// four rows a lane owns at one column are ONE 16-byte unit -- a wave's access is 4 x 256 contiguous bytes instead of 16 x 64.  (In the
// planar layout both sides moved 4 bytes per lane and instruction: the gate backward, bound by these bytes, ran at 3.7 TB/s.)
#if !defined(WG_OPT_PLANAR_TS)
#define WG_TS_INTERLEAVED 1
#else
#define WG_TS_INTERLEAVED 0
#endif
__device__ __forceinline__ float *paddr4(const PRef &r, const Geo &g, int b, int ch, int t)     // ch: a multiple of 4
{
    return r.p + (((size_t)b * (r.Cp >> 2) + ((r.ch0 + ch) >> 2)) * g.P + g.H + t) * 4;
}
// S-plane units and the accumulator layout.  A lane of the 16 x 16 MFMA output owns rows 4 rq .. 4 rq + 3 (rq = lane >> 4) of one column:
// HALF a 16-byte unit; lanes l and l + 16 own the two halves of the same unit.  As 8-byte accesses a wave instruction touches every
// 64-byte line twice (once from each 16-lane row): the launches bound by these bytes ran at 3-4 TB/s (the residual conv spent 34 of its
// 38 us on them: tools/experiments/shape_ab.sh).  v_permlane16_swap exchanges the odd 16-lane rows of one register with the even rows of
// another, so two column blocks nb0, nb1 pair up: even rows move the WHOLE unit of column block nb0, odd rows that of nb1 -- one 16-byte
// access per lane instead of two 8-byte ones, 512 contiguous bytes per channel group and instruction.
// MEASURED AND NOT ADOPTED (-DWG_OPT_UNIT16; parity green): the residual conv went from 38.3 to 40.8 us -- the width of these accesses
// is not what makes them slow.
//   store: x = the lane's piece of nb0, y = its piece of nb1  ->  (x', y') = swap(x, y) is the unit the lane stores, in that order;
//   load:  the lane loads a unit (lo8, hi8)                    ->  (x, y) = swap(lo8, hi8) are its pieces of nb0 and nb1.
// Must be executed by ALL lanes (no divergence around it); predicate only the memory access.
__device__ __forceinline__ void swap16_unit(u32x4 &u)      // words (0, 2) and (1, 3) of a unit
{
    const u32x2 r0 = __builtin_amdgcn_permlane16_swap(u[0], u[2], false, false);
    const u32x2 r1 = __builtin_amdgcn_permlane16_swap(u[1], u[3], false, false);
    u[0] = r0[0]; u[2] = r0[1]; u[1] = r1[0]; u[3] = r1[1];
}
template <int EPI, int NB>
__device__ __forceinline__ void conv_acc_init_q(const ConvGemmArgs &a, const SRef &saux, f32x4 (&acc)[4][NB], int t0, int m0, int b, int wr, int wc,
                                                int lane)
{
    const Geo g = a.g;
    const int col = lane & 15, rq = lane >> 4;
#if defined(WG_OPT_UNIT16)
    if constexpr (EPI == EPI_STORE && NB >= 2) {
        if (saux.hi) {
#pragma unroll
            for (int mb = 0; mb < 4; ++mb) {
                const int mu = m0 + wr * 64 + mb * 16 + 8 * (rq >> 1);            // first row of the lane's unit
#pragma unroll
                for (int nb = 0; nb < NB; nb += 2) {
                    const int t = t0 + wc * (16 * NB) + (nb + ((rq & 1))) * 16 + col;   // even rows: column block nb, odd rows: nb + 1
                    u32x4 uh = {0u, 0u, 0u, 0u}, ul = {0u, 0u, 0u, 0u};
                    if (t < g.T && mu < a.M) {
                        const size_t i = s_index(saux, g, b, mu, t);
                        uh = *reinterpret_cast<const u32x4 *>(saux.hi + i);
                        ul = *reinterpret_cast<const u32x4 *>(saux.hi + saux.lo_off + i);
                    }
                    swap16_unit(uh); swap16_unit(ul);
#pragma unroll
                    for (int q = 0; q < 2; ++q) {                                 // q = 0: the piece of nb (words 0, 1), q = 1: of nb + 1 (words 2, 3)
                        acc[mb][nb + q][0] = __uint_as_float(uh[2 * q] << 16) + __uint_as_float(ul[2 * q] << 16);
                        acc[mb][nb + q][1] = __uint_as_float(uh[2 * q] & 0xffff0000u) + __uint_as_float(ul[2 * q] & 0xffff0000u);
                        acc[mb][nb + q][2] = __uint_as_float(uh[2 * q + 1] << 16) + __uint_as_float(ul[2 * q + 1] << 16);
                        acc[mb][nb + q][3] = __uint_as_float(uh[2 * q + 1] & 0xffff0000u) + __uint_as_float(ul[2 * q + 1] & 0xffff0000u);
                    }
                }
            }
            return;
        }
    }
#endif
    if ((EPI == EPI_STORE || EPI == EPI_STORE_SO) && saux.hi) {
        // the value to accumulate into comes as an S-plane: a lane's 4 rows of one column are exactly one half unit (8 bytes) of the hi
        // array and one of the lo array; x = hi + lo (the fp32 plane of such a tensor is then never written nor read)
#pragma unroll
        for (int mb = 0; mb < 4; ++mb) {
            const int m = m0 + wr * 64 + mb * 16 + 4 * rq;
#pragma unroll
            for (int nb = 0; nb < NB; ++nb) {
                const int t = t0 + wc * (16 * NB) + nb * 16 + col;
                u32x2 vh = {0u, 0u}, vl = {0u, 0u};
                if (t < g.T && m < a.M) {
                    const size_t i = s_index(saux, g, b, m, t);
                    vh = *reinterpret_cast<const u32x2 *>(saux.hi + i);
                    vl = *reinterpret_cast<const u32x2 *>(saux.hi + saux.lo_off + i);
                }
                acc[mb][nb][0] = __uint_as_float(vh[0] << 16) + __uint_as_float(vl[0] << 16);
                acc[mb][nb][1] = __uint_as_float(vh[0] & 0xffff0000u) + __uint_as_float(vl[0] & 0xffff0000u);
                acc[mb][nb][2] = __uint_as_float(vh[1] << 16) + __uint_as_float(vl[1] << 16);
                acc[mb][nb][3] = __uint_as_float(vh[1] & 0xffff0000u) + __uint_as_float(vl[1] & 0xffff0000u);
            }
        }
        return;
    }
    if (EPI == EPI_STORE_SO) {                                                   // (no fp32 accumulate-into plane in this form: zero)
#pragma unroll
        for (int mb = 0; mb < 4; ++mb)
#pragma unroll
            for (int nb = 0; nb < NB; ++nb)
#pragma unroll
                for (int e = 0; e < 4; ++e) acc[mb][nb][e] = 0.f;
        return;
    }
#pragma unroll
    for (int mb = 0; mb < 4; ++mb) {
        const int mbase = m0 + wr * 64 + mb * 16;                                 // first row of this 16-row block (wave uniform)
        const float *base = nullptr;
        if (EPI == EPI_STORE || EPI == EPI_STORE_FO) base = a.aux0.p ? paddr(a.aux0, g, b, mbase, t0) : nullptr;
        else if (EPI == EPI_RESSKIP)                                              // nsplit is a multiple of 32: a block lies on one side
            base = mbase < a.nsplit ? paddr(a.aux0, g, b, mbase, t0) : (a.accumulate ? paddr(a.out1, g, b, mbase - a.nsplit, t0) : nullptr);
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) {
            const int tl = wc * (16 * NB) + nb * 16 + col;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const unsigned off = (unsigned)(4 * rq + e) * (unsigned)g.P + (unsigned)tl;
                float x = 0.f;
                if (t0 + tl < g.T && mbase + 4 * rq + e < a.M && base) x = base[off];
                acc[mb][nb][e] = x;
            }
        }
    }
}

// 8-byte stores of one row block's column blocks: base (scalar) + voff (the lane's constant byte offset) + 256 * nb
template <int OFF>
__device__ __forceinline__ void wgq_st8(const unsigned short *base, unsigned voff, const u32x2 &v)
{
#if defined(WG_OPT_ST_SC1)     // experiment: write-through S-plane stores -- fewer dirty lines for the end-of-kernel write-back (the kernel boundary costs
    asm volatile("global_store_dwordx2 %0, %1, %2 offset:%3 sc1" ::"v"(voff), "v"(v), "s"(base), "n"(OFF) : "memory");      // ~1.5 us + dirty bytes / 6 TB/s)
#else
    asm volatile("global_store_dwordx2 %0, %1, %2 offset:%3" ::"v"(voff), "v"(v), "s"(base), "n"(OFF) : "memory");
#endif
}
template <int OFF>
__device__ __forceinline__ void wgq_st16nt(const float *base, unsigned voff, const f32x4 &v)
{
    // (s_nop 1: a store of more than 64 bits needs two wait states before its data registers may be written again, and the compiler's
    // hazard recogniser does not look into asm statements)
    asm volatile("global_store_dwordx4 %0, %1, %2 offset:%3 nt\n\ts_nop 1" ::"v"(voff), "v"(v), "s"(base), "n"(OFF) : "memory");
}
// (the data registers come straight out of an MFMA: a vector-memory instruction that reads the result of a matrix instruction needs up to
// 19 wait states behind it, and the compiler's hazard recogniser does not look into asm statements -- without the s_nop pair the store
// read registers the pipe had not written yet, and two runs of one step differed)
template <int OFF>
__device__ __forceinline__ void wgq_st16p(const float *base, unsigned voff, const f32x4 &v)
{
    asm volatile("s_nop 15\n\ts_nop 3\n\tglobal_store_dwordx4 %0, %1, %2 offset:%3\n\ts_nop 1" ::"v"(voff), "v"(v), "s"(base), "n"(OFF) : "memory");
}
// EPI_GATE_SO, column block NBI of a wave tile: tanh, sigmoid, gate of the lane's 8 channels; stores at immediate offset 256 * NBI.
// pbase (nullptr: none): the wave's share of WN's `out` for this block -- Weff (rows 0-7, the wave's 32 gate channels: A fragments eah / eal)
// times the split gate the lane holds, one more 16x16x32 product (wg_gemm16g.h, wgg_gate_nb) -- goes to pbase + 16 NBI columns of 8 floats
template <int OFF>
__device__ __forceinline__ void wgq_st8p(const float *base, unsigned voff, const f32x2_t &v)      // (as wgq_st16p: the data come out of an MFMA)
{
    asm volatile("s_nop 15\n\ts_nop 3\n\tglobal_store_dwordx2 %0, %1, %2 offset:%3" ::"v"(voff), "v"(v), "s"(base), "n"(OFF) : "memory");
}
template <int NB, int NBI>
__device__ __forceinline__ void wgq_gate_nb(f32x4 (&acc)[4][NB], bool live, const float *const (&bt)[2], const float *const (&bs)[2],
                                            const unsigned short *const (&sh)[2], const unsigned short *const (&sl)[2], bool has_ts,
                                            unsigned vo_t, unsigned vo_s, const float *pbase = nullptr, unsigned vo_p = 0, bool plive = false,
                                            bf16x8 eah = bf16x8{}, bf16x8 eal = bf16x8{}, bool prow2 = false)
{
    float tw[8], sf[8], gv[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        tw[i] = wg_tanh(acc[i >> 2][NBI][i & 3]);
        sf[i] = wg_sigmoid(acc[2 + (i >> 2)][NBI][i & 3]);
        gv[i] = tw[i] * sf[i];
    }
    u32x2 gh[2], gl[2];
#pragma unroll
    for (int mbp = 0; mbp < 2; ++mbp) {
        unsigned hh, ll;
        split2(gv[4 * mbp], gv[4 * mbp + 1], hh, ll); gh[mbp][0] = hh; gl[mbp][0] = ll;
        split2(gv[4 * mbp + 2], gv[4 * mbp + 3], hh, ll); gh[mbp][1] = hh; gl[mbp][1] = ll;
    }
    if (pbase) {
        const u32x4 bh4 = {gh[0][0], gh[0][1], gh[1][0], gh[1][1]}, bl4 = {gl[0][0], gl[0][1], gl[1][0], gl[1][1]};
        const bf16x8 bh = __builtin_bit_cast(bf16x8, bh4), bl = __builtin_bit_cast(bf16x8, bl4);
        f32x4 o = {0.f, 0.f, 0.f, 0.f};
        o = __builtin_amdgcn_mfma_f32_16x16x32_bf16(eal, bh, o, 0, 0, 0);
        o = __builtin_amdgcn_mfma_f32_16x16x32_bf16(eah, bl, o, 0, 0, 0);
        o = __builtin_amdgcn_mfma_f32_16x16x32_bf16(eah, bh, o, 0, 0, 0);
        if (prow2) {                                        // two rows per column (2 ic <= 2): lanes 0-15, 8 bytes each
            const f32x2_t o2 = {o[0], o[1]};
            if (live && plive) wgq_st8p<128 * NBI>(pbase, vo_p, o2);
        } else if (live && plive) wgq_st16p<512 * NBI>(pbase, vo_p, o);      // lanes 0-31: rows 4 rq .. 4 rq + 3 of the lane's column
    }
#pragma unroll
    for (int mbp = 0; mbp < 2; ++mbp) {
        f32x4 vt, vs;
#pragma unroll
        for (int e = 0; e < 4; ++e) { vt[e] = tw[4 * mbp + e]; vs[e] = sf[4 * mbp + e]; }
        const u32x2 vh = gh[mbp], vl = gl[mbp];
        if (live) {
            if (has_ts && bt[mbp]) wgq_st16nt<256 * NBI>(bt[mbp], vo_t, vt);      // (tanh: only where something still reads it, see conv_epilogue_q)
            if (has_ts) wgq_st16nt<256 * NBI>(bs[mbp], vo_t, vs);
            wgq_st8<256 * NBI>(sh[mbp], vo_s, vh);
            wgq_st8<256 * NBI>(sl[mbp], vo_s, vl);
        }
    }
    __builtin_amdgcn_sched_barrier(0);                       // eight outputs at a time
}
template <int OFF>
__device__ __forceinline__ void wgq_st4(const float *base, unsigned voff, float v)
{
    asm volatile("global_store_dword %0, %1, %2 offset:%3" ::"v"(voff), "v"(v), "s"(base), "n"(OFF) : "memory");
}
template <int OFF>
__device__ __forceinline__ void wgq_ld16nt(f32x4 &v, const float *base, unsigned voff)
{
    asm volatile("global_load_dwordx4 %0, %1, %2 offset:%3 nt" : "=v"(v) : "v"(voff), "s"(base), "n"(OFF) : "memory");
}
template <int OFF>
__device__ __forceinline__ void wgq_ld8(u32x2 &v, const unsigned short *base, unsigned voff)
{
    asm volatile("global_load_dwordx2 %0, %1, %2 offset:%3" : "=v"(v) : "v"(voff), "s"(base), "n"(OFF) : "memory");
}
// tanh of a gate element from what the S-plane mode keeps of it: gate = tanh . sigmoid (the S-plane's hi + lo) and the saved sigmoid.
// sigmoid == 0 (the pre-activation underflowed): gate is 0 too and the factor sigmoid (1 - tanh^2) the caller forms is 0 whatever tanh is.
__device__ __forceinline__ float wg_tanh_from_gate(float gate, float sf)
{
    const float t = __fdividef(gate, sf);
    return sf > 0.f ? fminf(fmaxf(t, -1.f), 1.f) : 0.f;
}
template <int NB>
__device__ __forceinline__ void wgq_wait_loads8(u32x2 (&x)[NB], u32x2 (&y)[NB], f32x4 (&z)[NB])
{
    if constexpr (NB == 4)
        asm volatile("s_waitcnt vmcnt(0)" : "+v"(x[0]), "+v"(x[1]), "+v"(x[2]), "+v"(x[3]), "+v"(y[0]), "+v"(y[1]), "+v"(y[2]), "+v"(y[3]),
                                            "+v"(z[0]), "+v"(z[1]), "+v"(z[2]), "+v"(z[3])::"memory");
    else if constexpr (NB == 2)
        asm volatile("s_waitcnt vmcnt(0)" : "+v"(x[0]), "+v"(x[1]), "+v"(y[0]), "+v"(y[1]), "+v"(z[0]), "+v"(z[1])::"memory");
    else
        asm volatile("s_waitcnt vmcnt(0)" : "+v"(x[0]), "+v"(y[0]), "+v"(z[0])::"memory");
}
// every hand-issued load has landed (names the registers so that nothing that reads them moves above the wait)
template <int NB>
__device__ __forceinline__ void wgq_wait_loads(f32x4 (&x)[NB], f32x4 (&y)[NB])
{
    if constexpr (NB == 4)
        asm volatile("s_waitcnt vmcnt(0)" : "+v"(x[0]), "+v"(x[1]), "+v"(x[2]), "+v"(x[3]), "+v"(y[0]), "+v"(y[1]), "+v"(y[2]), "+v"(y[3])::"memory");
    else if constexpr (NB == 2)
        asm volatile("s_waitcnt vmcnt(0)" : "+v"(x[0]), "+v"(x[1]), "+v"(y[0]), "+v"(y[1])::"memory");
    else
        asm volatile("s_waitcnt vmcnt(0)" : "+v"(x[0]), "+v"(y[0])::"memory");
}
template <int NB>
__device__ __forceinline__ void wgq_store_row(const unsigned short *hb, const unsigned short *lb, unsigned vo, const u32x2 (&ph)[NB],
                                              const u32x2 (&pl)[NB], int tw0, int T)
{
    if (tw0 < T) { wgq_st8<0>(hb, vo, ph[0]); wgq_st8<0>(lb, vo, pl[0]); }
    if constexpr (NB > 1) if (tw0 + 16 < T) { wgq_st8<256>(hb, vo, ph[1]); wgq_st8<256>(lb, vo, pl[1]); }
    if constexpr (NB > 2) if (tw0 + 32 < T) { wgq_st8<512>(hb, vo, ph[2]); wgq_st8<512>(lb, vo, pl[2]); }
    if constexpr (NB > 3) if (tw0 + 48 < T) { wgq_st8<768>(hb, vo, ph[3]); wgq_st8<768>(lb, vo, pl[3]); }
}
template <int EPI, int NB>
__device__ __forceinline__ void conv_epilogue_q(const ConvGemmArgs &a, const SRef &s0, f32x4 (&acc)[4][NB], int t0, int m0, int b,
                                                int wr, int wc, int lane, const SRef &saux = SRef{nullptr, 0, 8, 0}, const float *eff = nullptr,
                                                float *part = nullptr, int prow = 8)
{
    const Geo g = a.g;
    const int col = lane & 15, rq = lane >> 4;
    if (EPI == EPI_GATE_SO) {
        // EPI_GATE with every store hand-issued (see EPI_STORE_SO below): scalar bases per 16-channel block, two constant lane offsets
        // (the fp32 tanh / sigmoid planes [c / 4][t][4]: 16 bytes per lane; the gate's S-plane: 8 bytes), column blocks as immediates
        const int chb = (m0 >> 1) + wr * 32;
        if (2 * chb >= a.M) return;
        // the saved planes: sigmoid (out2) whenever the pass keeps them, tanh (out1) only where the gate backward still reads it -- in the
        // S-plane mode it takes tanh = gate / sigmoid from the gate's own S-plane, and out1 is null
        const bool has_ts = a.out2.p != nullptr;
        const int tl0 = wc * (16 * NB);
        const float *bt[2], *bs[2];
        const unsigned short *sh[2], *sl[2];
#pragma unroll
        for (int mbp = 0; mbp < 2; ++mbp) {
            bt[mbp] = (has_ts && a.out1.p) ? paddr4(a.out1, g, b, chb + mbp * 16, t0 + tl0) : nullptr;
            bs[mbp] = has_ts ? paddr4(a.out2, g, b, chb + mbp * 16, t0 + tl0) : nullptr;
            sh[mbp] = s0.hi + s_index(s0, g, b, chb + mbp * 16, t0 + tl0);
            sl[mbp] = sh[mbp] + s0.lo_off;
        }
        const unsigned vo_t = (unsigned)((rq * g.P + col) * 16), vo_s = (unsigned)(((rq >> 1) * g.P + col) * 16 + 8 * (rq & 1));
        const int tw0 = t0 + tl0 + col;
        // the wave's share of WN's `out` (ConvGemm16sArgs::part): its 32 gate channels are slice chb / 32 of the layer's Weff fragments
        // ([slice][hi | lo][k-group][8 rows][16 B], weff_kernel); rows 8-15 of the A operand are zero
        const int slot = chb >> 5;
        bf16x8 eah = bf16x8{}, eal = bf16x8{};
        const float *pbase = nullptr;
        if (part) {
            if (col < 8) {
                const char *ef = reinterpret_cast<const char *>(eff) + slot * 1024 + (rq * 8 + col) * 16;
                eah = *reinterpret_cast<const bf16x8 *>(ef); eal = *reinterpret_cast<const bf16x8 *>(ef + 512);
            }
            pbase = part + (((size_t)slot * g.B + b) * g.Tt + t0 + tl0) * prow;
        }
        const bool prow2 = prow == 2;
        const unsigned vo_p = prow2 ? (unsigned)(col * 8) : (unsigned)(col * 32 + (rq & 1) * 16);
        const bool plive = prow2 ? rq == 0 : rq < 2;
        wgq_gate_nb<NB, 0>(acc, tw0 < g.T, bt, bs, sh, sl, has_ts, vo_t, vo_s, pbase, vo_p, plive, eah, eal, prow2);
        if constexpr (NB > 1) wgq_gate_nb<NB, 1>(acc, tw0 + 16 < g.T, bt, bs, sh, sl, has_ts, vo_t, vo_s, pbase, vo_p, plive, eah, eal, prow2);
        if constexpr (NB > 2) wgq_gate_nb<NB, 2>(acc, tw0 + 32 < g.T, bt, bs, sh, sl, has_ts, vo_t, vo_s, pbase, vo_p, plive, eah, eal, prow2);
        if constexpr (NB > 3) wgq_gate_nb<NB, 3>(acc, tw0 + 48 < g.T, bt, bs, sh, sl, has_ts, vo_t, vo_s, pbase, vo_p, plive, eah, eal, prow2);
        return;
    }
    if (EPI == EPI_GATE) {
        // rows 0-31 of the wave tile are the tanh halves, rows 32-63 the sigmoid halves of the same 32 gate channels (pack_kernel's
        // 64-row interleave): channel chb + mbp*16 + 4 rq + e pairs acc[mbp] with acc[mbp + 2]
        const int chb = (m0 >> 1) + wr * 32;
        if (2 * chb >= a.M) return;                              // (2 Cd is a multiple of 64: a wave's 32 gate channels are all valid or none)
        float *b0 = a.out0.p ? paddr(a.out0, g, b, chb, t0) : nullptr;
        float *b1 = a.out1.p ? paddr(a.out1, g, b, chb, t0) : nullptr;
        float *b2 = a.out2.p ? paddr(a.out2, g, b, chb, t0) : nullptr;
        unsigned short *sh = s0.hi + s_index(s0, g, b, chb, t0);
        unsigned short *sl = sh + s0.lo_off;
        const unsigned s_grp = (unsigned)g.P * 8u;
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) {
            const unsigned tl = (unsigned)(wc * (16 * NB) + nb * 16 + col);
            if (t0 + (int)tl >= g.T) continue;
            float tw[8], sf[8], gv[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                tw[i] = wg_tanh(acc[i >> 2][nb][i & 3]);
                sf[i] = wg_sigmoid(acc[2 + (i >> 2)][nb][i & 3]);
                gv[i] = tw[i] * sf[i];
            }
#pragma unroll
            for (int mbp = 0; mbp < 2; ++mbp) {
                const unsigned off = (unsigned)(mbp * 16 + 4 * rq) * (unsigned)g.P + tl;
                if (b0) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) b0[off + (unsigned)e * (unsigned)g.P] = gv[4 * mbp + e];
                }
                if (b2) {
                    // (non-temporal: the saved tanh / sigmoid planes are read once, by the gate backward of this layer, 15 layer
                    // launches later: kept out of L2's way, -0.45 ms per training step; tanh only where out1 is given)
#if WG_TS_INTERLEAVED
                    const int chq = chb + mbp * 16 + 4 * rq;
                    f32x4 vt, vs;
#pragma unroll
                    for (int e = 0; e < 4; ++e) { vt[e] = tw[4 * mbp + e]; vs[e] = sf[4 * mbp + e]; }
                    if (b1) __builtin_nontemporal_store(vt, reinterpret_cast<f32x4 *>(paddr4(a.out1, g, b, chq, t0 + (int)tl)));
                    __builtin_nontemporal_store(vs, reinterpret_cast<f32x4 *>(paddr4(a.out2, g, b, chq, t0 + (int)tl)));
#else
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        if (b1) __builtin_nontemporal_store(tw[4 * mbp + e], &b1[off + (unsigned)e * (unsigned)g.P]);
                        __builtin_nontemporal_store(sf[4 * mbp + e], &b2[off + (unsigned)e * (unsigned)g.P]);
                    }
#endif
                }
                u32x2 vh, vl;
                unsigned hh, ll;
                split2(gv[4 * mbp], gv[4 * mbp + 1], hh, ll); vh[0] = hh; vl[0] = ll;
                split2(gv[4 * mbp + 2], gv[4 * mbp + 3], hh, ll); vh[1] = hh; vl[1] = ll;
                const unsigned so = (unsigned)(2 * mbp + (rq >> 1)) * s_grp + tl * 8u + (unsigned)(4 * (rq & 1));
                if (WG_OPT_NT_S & 4) {
                    __builtin_nontemporal_store(vh, reinterpret_cast<u32x2 *>(sh + so));
                    __builtin_nontemporal_store(vl, reinterpret_cast<u32x2 *>(sl + so));
                } else {
                    *reinterpret_cast<u32x2 *>(sh + so) = vh;
                    *reinterpret_cast<u32x2 *>(sl + so) = vl;
                }
            }
            __builtin_amdgcn_sched_barrier(0);                   // eight outputs at a time
        }
        return;
    }
    if (EPI == EPI_DGATE_SO) {
        // EPI_DGATE with hand-issued memory instructions.  The compiler's form issued a row block's eight 16-byte tanh / sigmoid loads
        // two at a time, each pair behind a full drain (a fresh 64-bit address pair per load, and this part may read an address register
        // late): five round trips per block, four blocks per tile -- 49 of the launch's 82 us (tools/experiments/shape_ab.sh).  Here a
        // block's loads are scalar base + ONE constant lane offset + immediate: issued back to back, one round trip per block; the
        // S-plane stores likewise (see EPI_STORE_SO).
        const unsigned vo_t = (unsigned)((rq * g.P + col) * 16), vo_s = (unsigned)(((rq >> 1) * g.P + col) * 16 + 8 * (rq & 1));
        const int tl0 = wc * (16 * NB), tw0 = t0 + tl0 + col;
        const bool twg = a.aux0.p == nullptr;                   // tanh from the gate's S-plane (saux) instead of a saved plane
#pragma unroll
        for (int mb = 0; mb < 4; ++mb) {
            const int mbase = m0 + wr * 64 + mb * 16;
            if (mbase >= a.M) continue;                         // (wave uniform; M is a multiple of 16 on this path)
            const float *py = paddr4(a.aux1, g, b, mbase, t0 + tl0);
            const unsigned short *hb = s0.hi + s_index(s0, g, b, mbase, t0 + tl0), *lb = hb + s0.lo_off;
            const unsigned short *hb2 = s0.hi + s_index(s0, g, b, a.nsplit + mbase, t0 + tl0), *lb2 = hb2 + s0.lo_off;
            f32x4 ax[NB], ay[NB];
            // (each form issues, waits and finishes its values INSIDE its branch: registers of loads in flight must not meet at a join --
            // the compiler would copy them there before the wait, tools/check_asm_loads.py)
            if (twg) {
                // tanh is not kept in the S-plane mode: the gate's own S-plane (saux: hi + lo, 2^-17 relative) and the saved sigmoid give
                // tanh = gate / sigmoid -- the same bytes read, 4 bytes per element less written by every gate conv that keeps its planes
                const unsigned short *gb = saux.hi + s_index(saux, g, b, mbase, t0 + tl0), *gbl = gb + saux.lo_off;
                u32x2 gh_[NB], gl_[NB];
                f32x4 sy[NB];
                wgq_ld8<0>(gh_[0], gb, vo_s); wgq_ld8<0>(gl_[0], gbl, vo_s); wgq_ld16nt<0>(sy[0], py, vo_t);
                if constexpr (NB > 1) { wgq_ld8<256>(gh_[1], gb, vo_s); wgq_ld8<256>(gl_[1], gbl, vo_s); wgq_ld16nt<256>(sy[1], py, vo_t); }
                if constexpr (NB > 2) { wgq_ld8<512>(gh_[2], gb, vo_s); wgq_ld8<512>(gl_[2], gbl, vo_s); wgq_ld16nt<512>(sy[2], py, vo_t); }
                if constexpr (NB > 3) { wgq_ld8<768>(gh_[3], gb, vo_s); wgq_ld8<768>(gl_[3], gbl, vo_s); wgq_ld16nt<768>(sy[3], py, vo_t); }
                wgq_wait_loads8<NB>(gh_, gl_, sy);
#pragma unroll
                for (int nb = 0; nb < NB; ++nb) {
                    ay[nb] = sy[nb];
                    ax[nb][0] = wg_tanh_from_gate(__uint_as_float(gh_[nb][0] << 16) + __uint_as_float(gl_[nb][0] << 16), sy[nb][0]);
                    ax[nb][1] = wg_tanh_from_gate(__uint_as_float(gh_[nb][0] & 0xffff0000u) + __uint_as_float(gl_[nb][0] & 0xffff0000u), sy[nb][1]);
                    ax[nb][2] = wg_tanh_from_gate(__uint_as_float(gh_[nb][1] << 16) + __uint_as_float(gl_[nb][1] << 16), sy[nb][2]);
                    ax[nb][3] = wg_tanh_from_gate(__uint_as_float(gh_[nb][1] & 0xffff0000u) + __uint_as_float(gl_[nb][1] & 0xffff0000u), sy[nb][3]);
                }
            } else {
                const float *px = paddr4(a.aux0, g, b, mbase, t0 + tl0);
                f32x4 tx[NB], ty[NB];
                wgq_ld16nt<0>(tx[0], px, vo_t); wgq_ld16nt<0>(ty[0], py, vo_t);
                if constexpr (NB > 1) { wgq_ld16nt<256>(tx[1], px, vo_t); wgq_ld16nt<256>(ty[1], py, vo_t); }
                if constexpr (NB > 2) { wgq_ld16nt<512>(tx[2], px, vo_t); wgq_ld16nt<512>(ty[2], py, vo_t); }
                if constexpr (NB > 3) { wgq_ld16nt<768>(tx[3], px, vo_t); wgq_ld16nt<768>(ty[3], py, vo_t); }
                wgq_wait_loads<NB>(tx, ty);
#pragma unroll
                for (int nb = 0; nb < NB; ++nb) { ax[nb] = tx[nb]; ay[nb] = ty[nb]; }
            }
            u32x2 ph[NB], pl[NB], qh[NB], ql[NB];
#pragma unroll
            for (int nb = 0; nb < NB; ++nb) {
                float o[4], o2[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float v = acc[mb][nb][e], tw = ax[nb][e], sf = ay[nb][e];
                    o[e] = v * sf * (1.0f - tw * tw);
                    o2[e] = v * tw * sf * (1.0f - sf);
                }
                unsigned hh, ll;
                split2(o[0], o[1], hh, ll); ph[nb][0] = hh; pl[nb][0] = ll;
                split2(o[2], o[3], hh, ll); ph[nb][1] = hh; pl[nb][1] = ll;
                split2(o2[0], o2[1], hh, ll); qh[nb][0] = hh; ql[nb][0] = ll;
                split2(o2[2], o2[3], hh, ll); qh[nb][1] = hh; ql[nb][1] = ll;
            }
            wgq_store_row<NB>(hb, lb, vo_s, ph, pl, tw0, g.T);
            wgq_store_row<NB>(hb2, lb2, vo_s, qh, ql, tw0, g.T);
        }
        return;
    }
    if (EPI == EPI_DGATE) {
        // The auxiliary loads (tanh / sigmoid of the forward gate) of one 16-row block at a time, then combine and store it.  (All 128
        // values loaded first took the kernel to 210 VGPRs, i.e. ONE workgroup per CU for a launch bound by these bytes.)
#if defined(WG_OPT_DGATE_ALL)
        constexpr int MBS = 4;
#else
        constexpr int MBS = 1;
#endif
#pragma unroll
        for (int mb0 = 0; mb0 < 4; mb0 += MBS) {
            f32x4 ax[MBS][NB], ay[MBS][NB];
#pragma unroll
            for (int q = 0; q < MBS; ++q) {
                const int mbase = m0 + wr * 64 + (mb0 + q) * 16;
                const float *p0 = paddr(a.aux0, g, b, mbase, t0), *p1 = paddr(a.aux1, g, b, mbase, t0);
#pragma unroll
                for (int nb = 0; nb < NB; ++nb) {
                    const int tl = wc * (16 * NB) + nb * 16 + col;
#if WG_TS_INTERLEAVED
                    // (non-temporal, like the stores that saved them: this is their only use; M is a multiple of 4 here)
                    f32x4 vx = {0.f, 0.f, 0.f, 0.f}, vy = {0.f, 0.f, 0.f, 0.f};
                    if (t0 + tl < g.T && mbase + 4 * rq < a.M) {
                        vy = __builtin_nontemporal_load(reinterpret_cast<const f32x4 *>(paddr4(a.aux1, g, b, mbase + 4 * rq, t0 + tl)));
                        if (a.aux0.p) vx = __builtin_nontemporal_load(reinterpret_cast<const f32x4 *>(paddr4(a.aux0, g, b, mbase + 4 * rq, t0 + tl)));
                        else {                                  // tanh = gate / sigmoid from the gate's S-plane (see EPI_DGATE_SO)
                            const size_t gi = s_index(saux, g, b, mbase + 4 * rq, t0 + tl);
                            const u32x2 uh = *reinterpret_cast<const u32x2 *>(saux.hi + gi), ul = *reinterpret_cast<const u32x2 *>(saux.hi + saux.lo_off + gi);
                            vx[0] = wg_tanh_from_gate(__uint_as_float(uh[0] << 16) + __uint_as_float(ul[0] << 16), vy[0]);
                            vx[1] = wg_tanh_from_gate(__uint_as_float(uh[0] & 0xffff0000u) + __uint_as_float(ul[0] & 0xffff0000u), vy[1]);
                            vx[2] = wg_tanh_from_gate(__uint_as_float(uh[1] << 16) + __uint_as_float(ul[1] << 16), vy[2]);
                            vx[3] = wg_tanh_from_gate(__uint_as_float(uh[1] & 0xffff0000u) + __uint_as_float(ul[1] & 0xffff0000u), vy[3]);
                        }
                    }
                    ax[q][nb] = vx; ay[q][nb] = vy;
                    (void)p0; (void)p1;
#else
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const unsigned off = (unsigned)(4 * rq + e) * (unsigned)g.P + (unsigned)tl;
                        float x = 0.f, y = 0.f;
                        // (non-temporal, like the stores that saved them: this is their only use)
                        if (t0 + tl < g.T && mbase + 4 * rq + e < a.M) { x = __builtin_nontemporal_load(&p0[off]); y = __builtin_nontemporal_load(&p1[off]); }
                        ax[q][nb][e] = x; ay[q][nb][e] = y;
                    }
#endif
                }
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int q = 0; q < MBS; ++q)
#pragma unroll
                for (int nb = 0; nb < NB; ++nb) {
                    const int mb = mb0 + q;
                    const int t = t0 + wc * (16 * NB) + nb * 16 + col, m = m0 + wr * 64 + mb * 16 + 4 * rq;
                    if (t >= g.T || m >= a.M) continue;
                    float o[4], o2[4];
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float v = acc[mb][nb][e], tw = ax[q][nb][e], sf = ay[q][nb][e];
                        o[e] = v * sf * (1.0f - tw * tw);
                        o2[e] = v * tw * sf * (1.0f - sf);
                        if (a.out0.p) {
                            *paddr(a.out0, g, b, m + e, t) = o[e];
                            *paddr(a.out0, g, b, a.nsplit + m + e, t) = o2[e];
                        }
                    }
                    s_store4<(WG_OPT_NT_S & 2) != 0>(s0, g, b, m, t, o);
                    s_store4<(WG_OPT_NT_S & 2) != 0>(s0, g, b, a.nsplit + m, t, o2);
                }
            __builtin_amdgcn_sched_barrier(0);
        }
        return;
    }
    // EPI_STORE / EPI_RESSKIP: the auxiliary values were the accumulators' initial value (conv_acc_init_q): stores only
#if !defined(WG_OPT_NO_EPI_BATCH) && !defined(WG_OPT_UNIT16)
    // S-plane only (the residual stream and its gradient, s_only_chain): every store of the tile issued back to back.  On this part a
    // vector-memory instruction may read its address registers LATE, so the compiler drains vmcnt before anything overwrites a register
    // of a store in flight: with one address computation per store, each store waited for the previous one to leave -- nine full drains
    // per tile (the residual conv spent 16 of its 38 us there, tools/experiments/shape_ab.sh).  Here the stores are hand-issued: scalar
    // base + ONE constant 32-bit lane offset + immediate (column blocks are 256 bytes apart) -- no address register is ever rewritten,
    // nothing drains, and the tile's stores leave while the next tile is already being multiplied.
    if (EPI == EPI_STORE_FO) {                                  // fp32 plane only: four constant lane offsets (the lane's four rows), column blocks 64 bytes apart
        unsigned vo[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) vo[e] = (unsigned)(((4 * rq + e) * g.P + col) * 4);
        const int tw0 = t0 + wc * (16 * NB) + col;
#pragma unroll
        for (int mb = 0; mb < 4; ++mb) {
            const int mbase = m0 + wr * 64 + mb * 16;
            if (mbase >= a.M) continue;
            const float *fb = paddr(a.out0, g, b, mbase, t0 + wc * (16 * NB));
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                if (mbase + 4 * rq + e >= a.M) continue;
                if (tw0 < g.T) wgq_st4<0>(fb, vo[e], acc[mb][0][e]);
                if constexpr (NB > 1) if (tw0 + 16 < g.T) wgq_st4<64>(fb, vo[e], acc[mb][1][e]);
                if constexpr (NB > 2) if (tw0 + 32 < g.T) wgq_st4<128>(fb, vo[e], acc[mb][2][e]);
                if constexpr (NB > 3) if (tw0 + 48 < g.T) wgq_st4<192>(fb, vo[e], acc[mb][3][e]);
            }
        }
        return;
    }
    if (EPI == EPI_STORE_SO) {
        const unsigned vo = (unsigned)(((rq >> 1) * g.P + col) * 16 + 8 * (rq & 1));       // bytes: unit row, time step, half unit
        const int tw0 = t0 + wc * (16 * NB) + col;
#pragma unroll
        for (int mb = 0; mb < 4; ++mb) {
            const int mbase = m0 + wr * 64 + mb * 16;
            if (mbase >= a.M) continue;
            const unsigned short *hb = s0.hi + ((size_t)b * (s0.Cp >> 3) + ((s0.ch0 + mbase) >> 3)) * g.P * 8 + (size_t)(g.H + t0 + wc * (16 * NB)) * 8;
            const unsigned short *lb = hb + s0.lo_off;
            u32x2 ph[NB], pl[NB];
#pragma unroll
            for (int nb = 0; nb < NB; ++nb) {
                unsigned hh, ll;
                split2(acc[mb][nb][0], acc[mb][nb][1], hh, ll); ph[nb][0] = hh; pl[nb][0] = ll;
                split2(acc[mb][nb][2], acc[mb][nb][3], hh, ll); ph[nb][1] = hh; pl[nb][1] = ll;
            }
            if (mbase + 4 * rq < a.M) wgq_store_row<NB>(hb, lb, vo, ph, pl, tw0, g.T);
        }
        return;
    }
#endif
#pragma unroll
    for (int mb = 0; mb < 4; ++mb) {
        const int mbase = m0 + wr * 64 + mb * 16;
        if (mbase >= a.M) continue;
        const bool res = EPI == EPI_STORE || mbase < a.nsplit;                    // RESSKIP: rows below nsplit -> out0 (+ S), the rest -> out1
        const PRef &dst = res ? a.out0 : a.out1;
        float *base = dst.p ? paddr(dst, g, b, res ? mbase : mbase - a.nsplit, t0) : nullptr;
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) {
            const int tl = wc * (16 * NB) + nb * 16 + col, t = t0 + tl, m = mbase + 4 * rq;
            if (t >= g.T || m >= a.M) continue;
            float o[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                o[e] = acc[mb][nb][e];
                if (base && m + e < a.M) base[(unsigned)(4 * rq + e) * (unsigned)g.P + (unsigned)tl] = o[e];
            }
#if defined(WG_OPT_UNIT16)
            if constexpr (NB >= 2) continue;                 // (the S-plane goes out below, two column blocks at a time)
#endif
            if (res && s0.hi) s_store4<(WG_OPT_NT_S & 1) != 0>(s0, g, b, m, t, o);
        }
#if defined(WG_OPT_UNIT16)
        if constexpr (NB >= 2) {
            if (res && s0.hi) {                              // (wave uniform: every lane takes part in the swaps)
                const int mu = mbase + 8 * (rq >> 1);
#pragma unroll
                for (int nb = 0; nb < NB; nb += 2) {
                    u32x4 uh, ul;
                    unsigned hh, ll;
                    split2(acc[mb][nb][0], acc[mb][nb][1], hh, ll); uh[0] = hh; ul[0] = ll;
                    split2(acc[mb][nb][2], acc[mb][nb][3], hh, ll); uh[1] = hh; ul[1] = ll;
                    split2(acc[mb][nb + 1][0], acc[mb][nb + 1][1], hh, ll); uh[2] = hh; ul[2] = ll;
                    split2(acc[mb][nb + 1][2], acc[mb][nb + 1][3], hh, ll); uh[3] = hh; ul[3] = ll;
                    swap16_unit(uh); swap16_unit(ul);
                    const int t = t0 + wc * (16 * NB) + (nb + (rq & 1)) * 16 + col;
                    if (t < g.T && mu < a.M) {
                        const size_t i = s_index(s0, g, b, mu, t);
                        *reinterpret_cast<u32x4 *>(s0.hi + i) = uh;
                        *reinterpret_cast<u32x4 *>(s0.hi + s0.lo_off + i) = ul;
                    }
                }
            }
        }
#endif
    }
}

// ------------------------------------------------------------------------------------------------
// the kernel: 8 waves, waves 0-3 multiply, waves 4-7 move operands (see convgemm16w_kernel for the protocol)
// ------------------------------------------------------------------------------------------------
#if defined(WG_DBG_TRACE)
#if defined(WG_DBG_TRACE_SMALL)                           // stamp the 128 x 64 form (small grids) instead
#define WGQ_TRACE_NI 1
#else
#define WGQ_TRACE_NI 2
#endif
#define WGQ_TRACE(slot) do { if ((EPI == EPI_GATE || EPI == EPI_GATE_SO) && NI == WGQ_TRACE_NI && lane == 0 && wave == 0 && (slot) < 16) { \
        wg_dbg_trace[blockIdx.x * 16 + (slot)] = wall_clock64(); wg_dbg_trace_cyc[blockIdx.x * 16 + (slot)] = clock64(); } } while (0)
#else
#define WGQ_TRACE(slot) do { } while (0)
#endif
// MG = 128-row compute groups per workgroup.  MG = 1: 8 waves (4 compute + 4 loaders), 128 x 64 NI tile, two workgroups per CU.
// MG = 2 (NI = 2 only): 16 waves, ONE workgroup per CU, 256 x 128 tile: the two compute groups share the B image of every chunk,
// i.e. the L2 -> LDS stream carries 48 KB per chunk for two 128 x 128 tiles instead of 64 KB, and there is no second, slower
// co-resident workgroup whose last tiles run alone.
// M64 (NI = 2, MG = 1 only): products with at most 64 rows (WaveFlow's 64-channel WN2D: data-gradient conv, residual / skip products, gate
// backward).  On the 128-row tile half of every MFMA multiplied padding rows: the data-gradient conv ran at 181 TF where the full-height gate
// conv reaches 314 (profiles/r04_wf_*).  The tile becomes 64 rows x 128 columns: the four compute waves sit side by side (64 rows x 32
// columns each: NB = 2), the loader waves fetch ONE A unit per lane, image and chunk instead of two, the A images take half the LDS.
// CG2 (MG = 2, NI = 2 only): the two compute groups sit side by side instead of on top of each other -- a 128 x 256 tile whose groups share
// the A image of every chunk (16 KB of weights + 32 KB of activations per chunk for two 128 x 128 tiles instead of 2 x 32 KB).  For
// products with at most 128 rows, which cannot share a B image between row groups (WaveFlow's gate conv: M = 2 Cd = 128): with two
// 128 x 128 workgroups per CU that launch takes in 39 GB/s per CU, the per-CU limit, at 38 % of the matrix peak (profiles/r04_wf_*).
template <int EPI, int NI, int MG = 1, bool M64 = false, bool CG2 = false>
__global__ __launch_bounds__(512 * MG) void convgemm16q_kernel(const ConvGemm16sArgs aa)
{
    static_assert(MG == 1 || NI == 2, "the two-group workgroup is built for the 128-column tile");
    static_assert(!M64 || (MG == 1 && NI == 2), "the 64-row tile is built for the 128-column tile of one compute group");
    static_assert(!CG2 || (MG == 2 && NI == 2 && !M64), "column groups: the two-group workgroup only");
    typedef typename std::conditional<M64 || CG2, Stage6a, typename StageOf<MG == 2 ? 1 : NI>::type>::type Stage;   // loads per loader lane and chunk: MG = 1: 4 A + 2 NI B; MG = 2: 4 A + 2 B; M64 / CG2: 2 A + 4 B
    constexpr int AIMG = (M64 ? 64 : CG2 ? 128 : 128 * MG) * WG16Q_ROWB;  // 128 MG rows x 64 B (M64: 64 rows; CG2: 128)
    constexpr int BIMG = (CG2 ? 256 : 64 * NI) * WG16Q_ROWB;
    constexpr int BUF = 2 * AIMG + 2 * BIMG;
    constexpr int TT = CG2 ? 256 : 64 * NI;                   // columns per tile
    constexpr int NB = M64 ? NI : 2 * NI;                     // 16-column blocks per wave
    __shared__ __attribute__((aligned(16))) char smem[2 * BUF];
    const ConvGemmArgs &a = aa.c;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const Geo g = a.g;
    int nchunks = 0;
    for (int s = 0; s < a.nseg; ++s) nchunks += (a.seg[s].nch + WG16_BK - 1) / WG16_BK;
    constexpr bool PERSIST = EPI != EPI_DGATE && EPI != EPI_DGATE_SO;                // (the gate backward keeps one workgroup per tile: see convgemm16w_kernel)
    const int ntiles = aa.ntx * aa.nty * aa.ntz, G = (int)gridDim.x;
    // (xcd_items: the workgroup's XCD owns ntz / 8 plane rows = xl tiles, dealt to its G / 8 workgroups; see ConvGemm16sArgs)
    // (xcd_items < 0, "XCD columns": plane rows that do not divide by 8 -- WSRGlow's batch of 12 --: the ntx * ntz column tiles are cut into
    // eight contiguous ranges, XCD x walks range x with the column tile fastest, so that the chunk streams of its 64 workgroup slots
    // share BOTH operands in its L2: a weight block is streamed once per XCD instead of once per column tile.  The conditioning
    // gradient of WSRGlow -- 29 row tiles x 48 column tiles x K = 4096 -- fetched 3.26 GB per launch for 0.25 GB of operands at 6.3 TB/s.)
    const int xper = aa.ntx * aa.nty, xslots = G >> 3, xslot = (int)blockIdx.x >> 3, xid = (int)blockIdx.x & 7;
    const int ncol = aa.ntx * aa.ntz, xc0 = aa.xcd_items < 0 ? xid * ncol / 8 : 0, xcn = aa.xcd_items < 0 ? (xid + 1) * ncol / 8 - xc0 : 0;
    const int xl = aa.xcd_items < 0 ? xcn * aa.nty : aa.xcd_items * xper;
    const int mine = !PERSIST ? 1 : aa.xcd_items ? (xslot < xl ? (xl - 1 - xslot) / xslots + 1 : 0) : (ntiles - 1 - (int)blockIdx.x) / G + 1;
    const int total = mine * nchunks;
    auto tile_at = [&](int k, int &t0, int &m0, int &b) {
        if constexpr (PERSIST) {
            int id = (int)blockIdx.x + k * G;
            if (aa.xcd_items < 0) {                             // local tile -> (row tile, column tile of this XCD's range): column tiles adjacent
                const int local = xslot + k * xslots, ty = local / max(xcn, 1), c = xc0 + local - ty * xcn;
                id = (c / aa.ntx * aa.nty + ty) * aa.ntx + c % aa.ntx;
            } else if (aa.xcd_items) {                          // local tile -> (plane row of this XCD, time tile, row tile): row tiles adjacent
                const int local = xslot + k * xslots, zl = local / xper, rem = local - zl * xper;
                id = (((int)blockIdx.x & 7) + 8 * zl) * xper + (rem % aa.nty) * aa.ntx + rem / aa.nty;
            }
            const int tx = id % aa.ntx, q = id / aa.ntx, ty = q % aa.nty, tz = q / aa.nty;
            t0 = tx * TT; m0 = ty * (M64 ? 64 : CG2 ? WG_TILE : WG_TILE * MG);
            b = a.row_sel1 ? tz * g.rows + a.row_sel1 - 1 : tz;
        } else {
            t0 = blockIdx.x * TT; m0 = blockIdx.y * (M64 ? 64 : CG2 ? WG_TILE : WG_TILE * MG);
            b = a.row_sel1 ? (int)blockIdx.z * g.rows + a.row_sel1 - 1 : (int)blockIdx.z;
        }
    };

    if (wave >= 4 * MG) {
        // ------------------------------- loader waves (as convgemm16w_kernel; only the LDS destination differs) -------------------------------
        const int lt = tid - 256 * MG;
#if defined(WG_OPT_LOADER_PRIO)
        __builtin_amdgcn_s_setprio(WG_OPT_LOADER_PRIO);      // experiment: the loaders' few instructions never queue behind the compute waves
#endif
        // (CG2, 512 loader lanes: B units (column lt & 255, k-groups lt >> 8 and + 2) of the 256-column image)
        const int bt = CG2 ? (lt & 255) : NI == 2 ? (lt & 127) : (lt & 63), cg0 = CG2 ? (lt >> 8) : NI == 2 ? (lt >> 7) : (lt >> 6);     // B unit: position, k-group
        const int nil = aa.tap_il * aa.tap_chunks;            // chunks walked interleaved over the taps (ConvGemm16sArgs::tap_il; 0: none)
        int cur_seg = aa.tap_il, cur_c = 0, chunk = nil, v = 0; // v: position in the tile's walk; (cur_seg, cur_c, chunk): the sequential part behind
        int gchunk = 0, tk = 0, t0, m0, b;
        tile_at(0, t0, m0, b);
        const unsigned voff_a = M64 ? (unsigned)(((lt >> 6) * 128 + (lt & 63)) * 16) : (unsigned)lt * 16u;
        // A pieces of a lane.  MG = 1: pieces lt and lt + 256 of the 128-row block: row lt & 127, k-groups (lt >> 7) and (lt >> 7) + 2.
        // MG = 2 (512 loader lanes): piece lt (row lt & 127, k-group lt >> 7) of BOTH 128-row blocks of the tile.
        // M64: ONE piece per lane: row lt & 63, k-group lt >> 6 of the 64 live rows of the (128-row) image block.
        // CG2: ONE piece per lane of the single 128-row block: row lt & 127, k-group lt >> 7 (0 .. 3).
        const int arow = M64 ? (lt & 63) : (lt & 127), akg = M64 ? (lt >> 6) : (lt >> 7);
        const int a_off[2] = {wg16q_off(arow, akg), MG == 2 ? wg16q_off(128 + arow, akg) : wg16q_off(arow, akg + 2)};
        constexpr int A_NEXT = MG == 2 ? 4096 : 2048;         // elements from a lane's first A piece to its second
        const int b_off[2] = {wg16q_off(bt, cg0), wg16q_off(bt, (cg0 + 2) & 3)};
        const unsigned voff_b = (unsigned)((cg0 * g.P + bt) * 16);
#if defined(WG_DBG_NOLOAD)
#define WG_LD(dst, base, voff) asm volatile("" : "=v"(dst) : "v"(voff), "s"(base))
#else
#define WG_LD(dst, base, voff) asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(dst) : "v"(voff), "s"(base) : "memory")
#endif
        // EXPERIMENT -DWG_OPT_A_POLICY=1 (sc1) / 2 (nt): the WEIGHT images fetched past the CU's L1 -- a workgroup reads its A chunks once,
        // while the taps of a layer re-read overlapping windows of the activation planes a few chunks apart (tap_il): an L1 that is not
        // flushed by the weight stream could serve those
#if defined(WG_OPT_A_POLICY) && WG_OPT_A_POLICY == 1 && !defined(WG_DBG_NOLOAD)
#define WG_LDA(dst, base, voff) asm volatile("global_load_dwordx4 %0, %1, %2 sc1" : "=v"(dst) : "v"(voff), "s"(base) : "memory")
#elif defined(WG_OPT_A_POLICY) && WG_OPT_A_POLICY == 2 && !defined(WG_DBG_NOLOAD)
#define WG_LDA(dst, base, voff) asm volatile("global_load_dwordx4 %0, %1, %2 nt" : "=v"(dst) : "v"(voff), "s"(base) : "memory")
#else
#define WG_LDA(dst, base, voff) WG_LD(dst, base, voff)
#endif
        // (measured: the non-temporal policy -- "nt" -- on the activation stream costs 7 % of a training step: the m-tiles of a column
        // tile and the taps of a layer re-read those lines from L2)
        const unsigned short *zsrc = aa.sseg[0].hi;           // plane position 0 of the first operand: always-zero halo
        auto issue = [&](Stage &st) {                         // exactly 4 + 2 NI loads in straight-line code (tools/check_asm_loads.py)
            const bool live = gchunk < total;
            const bool il = v < nil;
            // (a division per chunk: counters kept by increments instead measured 4 % SLOWER per step -- they took the address chain out of
            // the scalar unit)
            const int sgi = il ? v % aa.tap_il : cur_seg, cbi = il ? v / aa.tap_il : 0;
            const int ci = il ? cbi * WG16_BK : cur_c, chi = il ? sgi * aa.tap_chunks + cbi : chunk;
            const int sg = min(sgi, a.nseg - 1);
            const int nch = a.seg[sg].nch, shift = a.seg[sg].shift;
            const SSeg ss = aa.sseg[sg];
            int bsrc = b;
            bool rowok = true;
            if (g.rows > 0) {
                const int item = b / g.rows, r = b - item * g.rows + ss.row_off;
                rowok = r >= 0 && r < g.rows;
                bsrc = ss.per_item ? item : b + ss.row_off;
            }
#if defined(WG_DBG_TAPB)       // timing experiment only (results are garbage): the B operand of every tap but the first is not fetched -- what a
            const bool blive = live && rowok && !(il && sgi % WG_DBG_TAPB != 0), full = blive && (nch - ci > 16);      // B window held in LDS across the taps could win at most
#else
            const bool blive = live && rowok, full = blive && (nch - ci > 16);
#endif
            const unsigned short *ih = aa.img + ((size_t)chi * a.lda + m0) * WG16_BK, *il_ = ih + aa.img_stride;
            const unsigned short *row0 = ss.hi + ((size_t)bsrc * (ss.Cp >> 3) + ((ss.ch0 + ci) >> 3)) * g.P * 8;
            const unsigned short *pa0 = live ? ih : zsrc, *pa1 = live ? ih + A_NEXT : zsrc;
            const unsigned short *pl0 = live ? il_ : zsrc, *pl1 = live ? il_ + A_NEXT : zsrc;
            const unsigned va = live ? voff_a : 0u;
            if constexpr (M64 || CG2) {
                WG_LDA(st.ah[0], pa0, va);   WG_LDA(st.al[0], pl0, va);
                (void)pa1; (void)pl1;
            } else {
                WG_LDA(st.ah[0], pa0, va);   WG_LDA(st.ah[1], pa1, va);
                WG_LDA(st.al[0], pl0, va);   WG_LDA(st.al[1], pl1, va);
            }
            if constexpr ((NI == 2 && MG == 1) || CG2) {
                const unsigned short *b0 = row0 + (size_t)(g.H + t0 + shift) * 8, *b0l = b0 + ss.lo_off;
                const unsigned short *b1 = b0 + (size_t)2 * g.P * 8, *b1l = b1 + ss.lo_off;
                const unsigned short *pb0 = blive ? b0 : zsrc, *pb0l = blive ? b0l : zsrc;
                const unsigned short *pb1 = full ? b1 : zsrc, *pb1l = full ? b1l : zsrc;
                const unsigned vb = blive ? voff_b : 0u, vb1 = full ? voff_b : 0u;
                WG_LD(st.bh[0], pb0, vb);   WG_LD(st.bl[0], pb0l, vb);
                WG_LD(st.bh[1], pb1, vb1);  WG_LD(st.bl[1], pb1l, vb1);
            } else {
                const unsigned short *pb = blive ? row0 : zsrc, *pbl = blive ? row0 + ss.lo_off : zsrc;
                const bool lane_ok = blive && (cg0 < 2 || full);
                const unsigned vb = lane_ok ? voff_b + (unsigned)((g.H + t0 + shift) * 16) : 0u;
                WG_LD(st.bh[0], pb, vb);    WG_LD(st.bl[0], pbl, vb);
            }
            if (live) {
                ++gchunk;
                if (!il) {
                    ++chunk;
                    cur_c += WG16_BK;
                    if (cur_c >= nch) { cur_c = 0; ++cur_seg; }
                }
                if (++v == nchunks) {
                    v = 0; chunk = nil; cur_seg = aa.tap_il; cur_c = 0;
                    tk = min(tk + 1, mine - 1);
                    tile_at(tk, t0, m0, b);
                }
            }
        };
#undef WG_LD
#undef WG_LDA
        auto write = [&](const Stage &st, int buf) {
            char *sb = smem + buf * BUF;
#pragma unroll
            for (int j = 0; j < ((M64 || CG2) ? 1 : 2); ++j) {
                *reinterpret_cast<u32x4 *>(sb + a_off[j]) = st.ah[j];
                *reinterpret_cast<u32x4 *>(sb + AIMG + a_off[j]) = st.al[j];
            }
#pragma unroll
            for (int j = 0; j < (CG2 ? 2 : MG == 2 ? 1 : NI); ++j) {
                *reinterpret_cast<u32x4 *>(sb + 2 * AIMG + b_off[j]) = st.bh[j];
                *reinterpret_cast<u32x4 *>(sb + 2 * AIMG + BIMG + b_off[j]) = st.bl[j];
            }
        };
        Stage s0, s1;
        issue(s0);
        issue(s1);
        asm_wait_stage(s0);
        write(s0, 0);
        issue(s0);
        WG16W_BAR();                                          // buffer 0 ready
        auto iter = [&](Stage &st, int c) {
            asm_wait_stage(st);
            write(st, (c & 1) ^ 1);
            issue(st);
            WG16W_BAR();
        };
        for (int c = 0; c + 1 < total; c += 2) {              // always in pairs: see convgemm16w_kernel
            iter(s1, c);
            iter(s0, c + 1);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        return;
    }
    // ------------------------------- compute waves -------------------------------
    // M64: the four waves side by side, 64 rows x 32 columns each
    const int grp = wave >> 2, wr = M64 ? 0 : (wave >> 1) & 1, wc = M64 ? (wave & 3) : (wave & 1);       // grp: which 128-row half of the tile (MG = 2)
    f32x4 acc[4][NB];
    const int r16 = lane & 15, kg = lane >> 4;
    // CG2: both groups read the one 128-row A block; group grp multiplies columns grp * 128 .. + 127 of the 256-column B image
    const int ao = wg16q_off((CG2 ? 0 : grp * 128) + wr * 64 + r16, kg), bo = wg16q_off((CG2 ? grp * 128 : 0) + wc * (16 * NB) + r16, kg);      // + 16-row block * 1024
#define WGQ_SB() __builtin_amdgcn_sched_barrier(0)
    bf16x8 ah[4], al[4], bh[2], bl[2];
    auto rd = [&](const char *q) { return *reinterpret_cast<const bf16x8 *>(q); };
    int gc = 0;                                              // chunk index in this workgroup's stream; its buffer is gc & 1
    WGQ_TRACE(8);                                            // (kernel entry)
    auto do_tile = [&](int k) {
        int t0, m0, b;
        tile_at(k, t0, m0, b);
        if (CG2) t0 += grp * 128;                            // (the epilogue's columns; the loaders keep the tile's own t0)
        else m0 += grp * 128;
        int ln = lane;
        if (PERSIST) {
            asm volatile("" : "+v"(ln)::"memory");
            WGQ_SB();
        }
        if (EPI == EPI_STORE || EPI == EPI_STORE_SO || EPI == EPI_STORE_FO || EPI == EPI_RESSKIP) {
            conv_acc_init_q<EPI, NB>(a, aa.saux, acc, t0, m0, b, wr, wc, ln);
        } else {
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < NB; ++j)
#pragma unroll
                    for (int q = 0; q < 4; ++q) acc[i][j][q] = 0.f;
        }
        if (k == 0) { WGQ_TRACE(0); WG16W_BAR(); WGQ_TRACE(1); }     // buffer 0 ready (later tiles: published by the previous chunk's barrier)
        {
            const char *pa = smem + (gc & 1) * BUF + ao, *pb = smem + (gc & 1) * BUF + 2 * AIMG + bo;
#pragma unroll
            for (int i = 0; i < 4; ++i) { ah[i] = rd(pa + i * 1024); al[i] = rd(pa + AIMG + i * 1024); }
            bh[0] = rd(pb); bl[0] = rd(pb + BIMG);
        }
#if defined(WG_DBG_NOMFMA)
        for (int c = 0; c < nchunks; ++c, ++gc)
            if (gc + 1 < total || !(total & 1)) WG16W_BAR();
        if (false)
#endif
        for (int c = 0; c < nchunks; ++c, ++gc) {
            const char *pb = smem + (gc & 1) * BUF + 2 * AIMG + bo;
            const char *na = smem + ((gc & 1) ^ 1) * BUF + ao, *nb_ = smem + ((gc & 1) ^ 1) * BUF + 2 * AIMG + bo;
#pragma unroll
            for (int nb = 0; nb < NB; ++nb) {
                const int cur = nb & 1, nxt = cur ^ 1;
                if (nb == NB - 1) {
                    // every fragment of this chunk is in registers (the B of this last group was requested a group ago): release the
                    // buffer / publish the next one
                    WGQ_SB();
                    if (gc + 1 < total || !(total & 1)) WG16W_BAR();
                }
                WGQ_SB();
#pragma unroll
                for (int mb = 0; mb < 4; ++mb) {
                    if (!TwoP<EPI>::no_alo) acc[mb][nb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al[mb], bh[cur], acc[mb][nb], 0, 0, 0);
                    if (!TwoP<EPI>::no_blo) acc[mb][nb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[mb], bl[cur], acc[mb][nb], 0, 0, 0);
                    acc[mb][nb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[mb], bh[cur], acc[mb][nb], 0, 0, 0);
                    if (mb == 0) {
                        // the next group's B is requested BEHIND this group's first MFMAs: everything those wait for was requested a
                        // group ago, so the wait in front of them is short (requested in front of them, the compiler waits for the
                        // fresh reads as well: one exposed LDS round trip per chunk in the first version of this loop)
                        WGQ_SB();
                        if (nb == NB - 1) { bh[nxt] = rd(nb_); bl[nxt] = rd(nb_ + BIMG); }
                        else { bh[nxt] = rd(pb + (nb + 1) * 1024); bl[nxt] = rd(pb + BIMG + (nb + 1) * 1024); }
                        WGQ_SB();
                    }
                    if (nb == NB - 1) {
                        // (unconditional: after a tile's last chunk these read LDS that nothing uses -- a branch here would make the
                        // compiler drain every outstanding read at the join; the next tile starts with its own fetch)
                        WGQ_SB();
                        ah[mb] = rd(na + mb * 1024); al[mb] = rd(na + AIMG + mb * 1024);
                        WGQ_SB();
                    }
                }
                WGQ_SB();
            }
        }
        WGQ_TRACE(2 + 2 * k);
#if defined(WG_DBG_NOEPI)
        if (acc[0][0][0] + acc[1][0][0] + acc[2][NB - 1][1] + acc[3][NB - 1][3] == 12345.f) a.out0.p[lane] = 1.f;
#else
        int le = lane;
        if (PERSIST) asm volatile("" : "+v"(le)::"memory");
#if defined(WG_OPT_EPI_PRIO)
        __builtin_amdgcn_s_setprio(WG_OPT_EPI_PRIO);         // experiment: the epilogue's VALU / store issue ahead of the co-resident workgroup's waves
#endif
        conv_epilogue_q<EPI, NB>(a, aa.s0, acc, t0, m0, b, wr, wc, le, aa.saux, aa.eff, aa.part, aa.prow);
#if defined(WG_OPT_EPI_PRIO)
        __builtin_amdgcn_s_setprio(0);
#endif
#endif
        WGQ_TRACE(3 + 2 * k);
        if (PERSIST) WGQ_SB();
    };
    if constexpr (PERSIST) {
        for (int k = 0; k < mine; ++k) do_tile(k);
    } else {
        do_tile(0);
    }
#undef WGQ_SB
}
