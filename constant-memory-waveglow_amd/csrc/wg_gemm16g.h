// wg_gemm16g.h -- the conv product on 256 x 192 tiles: eight waves that all multiply, operands by LDS-DMA ("g": global_load_lds).
//
// Why another form (round 5).  convgemm16q_kernel<.., 2, 2> (wg_gemm16q.h) runs the headline gate conv -- M = 512, K = 848, 24 x 2000
// columns (model/waveglow.py:41-46) -- at 117-122 us with the matrix pipe 56 % busy: per 32-deep chunk of its 256 x 128 tile it moves
// 48 KB from L2 into LDS through the registers of eight loader waves (global_load -> VGPR -> ds_write_b128), its eight compute waves
// (64 x 64 each, 128 registers) re-read 128 KB of fragments from LDS, and a barrier joins the two roles (DESIGN.md sections 4a, 4d, 7).
// What the measurements of rounds 2-4 say pays on this power-limited part is less energy and fewer bytes per MFMA, so here:
//   * tile 256 rows x 192 columns, columns FLATTENED over (plane row, time): 24 x 2048 padded columns are exactly 256 column tiles, so
//     the gate conv is 2 tiles per CU and the 256-row products (data-gradient conv, skip product, residual conv) 1 tile per CU, where
//     128-column tiles gave 3 and 1.5 (the half-empty second round of the latter is why they never ran on the 256-row tile);
//     a chunk streams 32 KB of weights + 24 KB of activations for 256 x 192 x 32 MACs: 37 KB per 256 x 128 x 32 against 48 KB;
//   * NO loader waves: every wave multiplies a 64-row x 96-column tile (4 x 2 waves; 96 accumulators, up to 256 registers per wave) and
//     issues seven global_load_lds_dwordx4 per chunk; operands never pass through registers or ds_write;
//   * per chunk a wave reads 8 A + 12 B fragments for 72 MFMAs (0.28 reads per MFMA against 0.33); the CU reads 160 KB of fragments
//     per 256 x 192 chunk = 107 KB per 256 x 128 x 32 against 128 KB;
//   * LDS images are k-group PLANES: [hi | lo][k-group q][row][16 B].  One DMA instruction is 64 lanes x 16 B = 64 consecutive rows of
//     one plane = 1 KB contiguous in LDS AND in memory (the weight image is stored k-group-major per 128-row block, an S-plane is
//     [c / 8][p][8]): no swizzle on either side.  The 16x16x32 fragment read (lane l: row l & 15, k-group l >> 4) is conflict free
//     because ds_read_b128 is served in four groups of 16 lanes -- {0-3, 12-15, 20-27}, ... (MI355X_MICROARCH.md, LDS) -- whose two
//     k-groups read complementary row sets, and the planes are a multiple of 256 B apart;
//   * ring: two A buffers (32 KB each; A of chunk c + 1 is copied into a second register set while chunk c multiplies), three B buffers
//     (24 KB each, read where they are used): 136 KB, + 18 KB of tables (what a chunk of the K walk needs, described ONCE per workgroup,
//     and each wave's byte offsets per chunk, derived once per tile: see WggDesc below).  ONE barrier per chunk (2 304 MFMA cycles per
//     SIMD), the chunk's last column block multiplied behind the next one (the protocol is written out in front of `sblock` below); a
//     chunk's loads are issued two chunks ahead of its use.
// An LDS-DMA write is ordered for a reader only by the issuing wave's vmcnt wait followed by a barrier the reader has passed
// (cdna_hip_programming.md, "Read a staged buffer one phase AFTER the wait that retires it").  Past the stream's end the last chunk is
// fetched again, so the instruction counts the waits rely on never change; tools/check_asm_loads.py walks the ISA of every instantiation
// with the in-order queue of hand-issued instructions as its state and checks what each counted wait finds in flight.
// Measured: gate conv 121.7 -> 105-111 us, data-gradient conv 106 -> 95-97 us, skip sum 142 -> 130 us; the main
// loop holds 2.0-2.1 GHz (1.45-1.53 before) in 3 020-3 100 cycles per chunk, matrix pipe busy 0.61-0.64 (0.56).
#pragma once
#include "wg_gemm16q.h"

#define WGG_BM 256
#define WGG_BN 192
#define WGG_APLANE (WGG_BM * 16)
#define WGG_AIMG (4 * WGG_APLANE)
#define WGG_ABUF (2 * WGG_AIMG)
#define WGG_BPLANE (WGG_BN * 16)
#define WGG_BIMG (4 * WGG_BPLANE)
#define WGG_BBUF (2 * WGG_BIMG)
#define WGG_BBASE (2 * WGG_ABUF)
#define WGG_LDS (2 * WGG_ABUF + 3 * WGG_BBUF)             // 139 264 bytes

typedef __attribute__((address_space(3))) char wgg_lds_char;

// one LDS-DMA instruction: 64 lanes x 16 bytes from sbase + voff (per lane) to LDS bytes [lds_dst, lds_dst + 1024).  M0 carries the
// LDS address; it is compiler-reserved, so it is written and restored inside the statement (cdna_hip_programming.md section 5.7).
__device__ __forceinline__ void wgg_glds16(const void *sbase, unsigned voff, unsigned lds_dst)
{
#if defined(WGG_DBG_NOLOAD)                               // timing build: the address work stays, nothing is fetched (results are garbage)
    asm volatile("" ::"v"(voff), "s"(lds_dst), "s"(sbase) : "memory");
#else
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %3\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(voff), "s"(lds_dst), "s"(sbase)
                 : "memory");
#endif
}

// Split K (xcd_items = 2, ntz = splits): a product whose tiles cannot fill the chip -- WSRGlow's gate conv, M = 512 x 6 144 columns = 64
// tiles, K = 4 432 -- is cut along K into `ntz` parts per tile, one workgroup each; the parts leave their raw accumulators in a slab
// ([part][tile][wave][block][lane] f32x4: every store a contiguous 1 KB) and gate_finish16g_kernel adds them and runs the gate epilogue.
#define WGG_EPI_PART 8
__device__ __forceinline__ void wgg_st16(const float *base, unsigned voff, const f32x4 &v)
{
    asm volatile("global_store_dwordx4 %0, %1, %2\n\ts_nop 1" ::"v"(voff), "v"(v), "s"(base) : "memory");
}
// EPI_GATE_SO, column block NBI of the wave tile: 16 columns of plane row b from time step t0 on (the flattened columns of a tile may
// belong to two plane rows, so every block has its own bases; otherwise as wgq_gate_nb)
// with `part`: the block's 8 x 16 share of WN's `out`, Weff (2 ic <= 8 rows x the wave's 32 gate channels) . gate (32 x 16), goes to
// part[slot][b][t][8].  It is ONE more 16x16x32 product per column block on the split operands the epilogue holds anyway: the lane's
// eight gate values, packed to bf16 for the S-plane store, ARE the B fragment (lane = column l & 15, k-group l >> 4; the channel order
// inside the wave's 32 -- j < 4: 4 q + j, j >= 4: 16 + 4 q + j - 4 -- is a permutation of K, and the Weff fragments eah / eal are packed
// in the same order: weff_kernel), and rows 0-7 of the result lie in lanes 0-31 as four rows per lane: one 16-byte store.  (As 64 FMAs +
// 16 cross-lane adds per block on the vector ALU this cost 4.3 us per tile: 147 against 138 us per layer launch.)
template <int NBI>
__device__ __forceinline__ void wgg_gate_nb(const ConvGemmArgs &a, const SRef &s0, f32x4 (&acc)[4][6], int b, int t0, int chb, int lane,
                                            float *part = nullptr, int slot = 0, bf16x8 eah = bf16x8{}, bf16x8 eal = bf16x8{})
{
    const Geo g = a.g;
    const int col = lane & 15, rq = lane >> 4;
    const bool live = b < g.B && t0 + col < g.T;
    // (the saved planes: sigmoid always when the pass keeps them; tanh only where the gate backward still reads it -- in the S-plane mode it
    // takes tanh = gate / sigmoid from the gate's own S-plane: ConvGemmArgs::out1 is then null)
    const bool has_ts = a.out2.p != nullptr, has_t = a.out1.p != nullptr;
    const unsigned vo_t = (unsigned)((rq * g.P + col) * 16), vo_s = (unsigned)(((rq >> 1) * g.P + col) * 16 + 8 * (rq & 1));
    float tw[8], sf[8], gv[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        tw[i] = wg_tanh(acc[i >> 2][NBI][i & 3]);
        sf[i] = wg_sigmoid(acc[2 + (i >> 2)][NBI][i & 3]);
        gv[i] = tw[i] * sf[i];
    }
    const int bb = min(b, g.B - 1);
    u32x2 gh[2], gl[2];                                       // the gate as packed bf16 pairs: [16-channel block][pair]
#pragma unroll
    for (int mbp = 0; mbp < 2; ++mbp) {
        unsigned hh, ll;
        split2(gv[4 * mbp], gv[4 * mbp + 1], hh, ll); gh[mbp][0] = hh; gl[mbp][0] = ll;
        split2(gv[4 * mbp + 2], gv[4 * mbp + 3], hh, ll); gh[mbp][1] = hh; gl[mbp][1] = ll;
    }
    if (part) {
        const u32x4 bh4 = {gh[0][0], gh[0][1], gh[1][0], gh[1][1]}, bl4 = {gl[0][0], gl[0][1], gl[1][0], gl[1][1]};
        const bf16x8 bh = __builtin_bit_cast(bf16x8, bh4), bl = __builtin_bit_cast(bf16x8, bl4);
        f32x4 o = {0.f, 0.f, 0.f, 0.f};
        o = __builtin_amdgcn_mfma_f32_16x16x32_bf16(eal, bh, o, 0, 0, 0);
        o = __builtin_amdgcn_mfma_f32_16x16x32_bf16(eah, bl, o, 0, 0, 0);
        o = __builtin_amdgcn_mfma_f32_16x16x32_bf16(eah, bh, o, 0, 0, 0);
        const float *pb = part + (((size_t)slot * g.B + bb) * g.Tt + t0) * 8;
        // (o comes straight out of the matrix pipe: up to 19 wait states before a vector-memory instruction may read it, and the hazard
        // recogniser does not look into asm statements)
        asm volatile("s_nop 15\n\ts_nop 3" : "+v"(o));
        if (live && rq < 2) wgg_st16(pb, (unsigned)(col * 32 + rq * 16), o);      // rows 4 rq .. 4 rq + 3 of column col
    }
#if defined(WGG_OPT_UNIT16)
    // A/B (measured, parity green, NOT adopted: 57.10 / 57.15 against 57.16 / 57.19 ms per step, layer launch 148.5 / 147.7 against
    // 147.1 / 147.4 us -- gpurun_out/r06l: the width of the S-plane stores is not what the store drain costs, as round 2 found for the
    // older kernel): the gate's S-plane as 16-byte stores.  Lanes l and l + 16 hold the two halves of one unit; v_permlane16_swap trades the odd
    // 16-lane rows of the first 16-channel block's registers for the even rows of the second's: even rows then hold a whole unit of
    // block 0, odd rows one of block 1 -- two 16-byte stores per column block instead of four 8-byte ones
    {
        u32x4 uh = {gh[0][0], gh[0][1], gh[1][0], gh[1][1]}, ul = {gl[0][0], gl[0][1], gl[1][0], gl[1][1]};
        swap16_unit(uh); swap16_unit(ul);
        const int mbp = rq & 1;                               // the block this lane's unit belongs to
        const unsigned short *sh = s0.hi + s_index(s0, g, bb, chb, t0), *sl = sh + s0.lo_off;
        const unsigned vo_u = (unsigned)(((2 * mbp + (rq >> 1)) * g.P + col) * 16);
        if (live) {
            asm volatile("global_store_dwordx4 %0, %1, %2\n\ts_nop 1" ::"v"(vo_u), "v"(uh), "s"(sh) : "memory");
            asm volatile("global_store_dwordx4 %0, %1, %2\n\ts_nop 1" ::"v"(vo_u), "v"(ul), "s"(sl) : "memory");
        }
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const float *bt = has_t ? paddr4(a.out1, g, bb, chb + q * 16, t0) : nullptr;
            const float *bs = has_ts ? paddr4(a.out2, g, bb, chb + q * 16, t0) : nullptr;
            f32x4 vt, vs;
#pragma unroll
            for (int e = 0; e < 4; ++e) { vt[e] = tw[4 * q + e]; vs[e] = sf[4 * q + e]; }
            if (live && has_t) wgq_st16nt<0>(bt, vo_t, vt);
            if (live && has_ts) wgq_st16nt<0>(bs, vo_t, vs);
        }
    }
#else
#pragma unroll
    for (int mbp = 0; mbp < 2; ++mbp) {
        const float *bt = has_t ? paddr4(a.out1, g, bb, chb + mbp * 16, t0) : nullptr;
        const float *bs = has_ts ? paddr4(a.out2, g, bb, chb + mbp * 16, t0) : nullptr;
        const unsigned short *sh = s0.hi + s_index(s0, g, bb, chb + mbp * 16, t0), *sl = sh + s0.lo_off;
        f32x4 vt, vs;
#pragma unroll
        for (int e = 0; e < 4; ++e) { vt[e] = tw[4 * mbp + e]; vs[e] = sf[4 * mbp + e]; }
        const u32x2 vh = gh[mbp], vl = gl[mbp];
#if defined(WGG_DBG_NOSTORE)                              // timing build: the epilogue's arithmetic without its stores
        if (live && vh[0] == 0x12345u && vl[1] == 0x54321u) {
#else
        if (live) {
#endif
            if (has_t) wgq_st16nt<0>(bt, vo_t, vt);
            if (has_ts) wgq_st16nt<0>(bs, vo_t, vs);
#if defined(WGG_OPT_ST_SC1)                               // experiment: write-through stores (no dirty lines left for the end-of-kernel write-back)
            asm volatile("global_store_dwordx2 %0, %1, %2 sc1" ::"v"(vo_s), "v"(vh), "s"(sh) : "memory");
            asm volatile("global_store_dwordx2 %0, %1, %2 sc1" ::"v"(vo_s), "v"(vl), "s"(sl) : "memory");
#elif defined(WGG_OPT_ST_NT)
            asm volatile("global_store_dwordx2 %0, %1, %2 nt" ::"v"(vo_s), "v"(vh), "s"(sh) : "memory");
            asm volatile("global_store_dwordx2 %0, %1, %2 nt" ::"v"(vo_s), "v"(vl), "s"(sl) : "memory");
#else
            wgq_st8<0>(sh, vo_s, vh);
            wgq_st8<0>(sl, vo_s, vl);
#endif
        }
    }
#endif
    __builtin_amdgcn_sched_barrier(0);
}

// EPI_STORE_SO, column block NBI: the lane's 4 rows of each of the wave's four 16-row blocks as S-plane half units
template <int NBI>
__device__ __forceinline__ void wgg_store_nb(const ConvGemmArgs &a, const SRef &s0, f32x4 (&acc)[4][6], int b, int t0, int mw, int lane)
{
    const Geo g = a.g;
    const int col = lane & 15, rq = lane >> 4;
    const bool live = b < g.B && t0 + col < g.T;
    const int bb = min(b, g.B - 1);
    const unsigned vo = (unsigned)(((rq >> 1) * g.P + col) * 16 + 8 * (rq & 1));
#if defined(WGG_OPT_UNIT16)
#pragma unroll
    for (int mb = 0; mb < 4; mb += 2) {                       // (M is a multiple of 256 here: every row exists)
        unsigned h0, l0, h1, l1, h2, l2, h3, l3;
        split2(acc[mb][NBI][0], acc[mb][NBI][1], h0, l0); split2(acc[mb][NBI][2], acc[mb][NBI][3], h1, l1);
        split2(acc[mb + 1][NBI][0], acc[mb + 1][NBI][1], h2, l2); split2(acc[mb + 1][NBI][2], acc[mb + 1][NBI][3], h3, l3);
        u32x4 uh = {h0, h1, h2, h3}, ul = {l0, l1, l2, l3};
        swap16_unit(uh); swap16_unit(ul);
        const unsigned short *hb = s0.hi + s_index(s0, g, bb, mw + mb * 16, t0), *lb = hb + s0.lo_off;
        const unsigned vo_u = (unsigned)(((2 * (rq & 1) + (rq >> 1)) * g.P + col) * 16);
        if (live) {
            asm volatile("global_store_dwordx4 %0, %1, %2\n\ts_nop 1" ::"v"(vo_u), "v"(uh), "s"(hb) : "memory");
            asm volatile("global_store_dwordx4 %0, %1, %2\n\ts_nop 1" ::"v"(vo_u), "v"(ul), "s"(lb) : "memory");
        }
    }
}
#else
#pragma unroll
    for (int mb = 0; mb < 4; ++mb) {
        const int mbase = mw + mb * 16;
        const unsigned short *hb = s0.hi + s_index(s0, g, bb, mbase, t0), *lb = hb + s0.lo_off;
        u32x2 ph, pl;
        unsigned hh, ll;
        split2(acc[mb][NBI][0], acc[mb][NBI][1], hh, ll); ph[0] = hh; pl[0] = ll;
        split2(acc[mb][NBI][2], acc[mb][NBI][3], hh, ll); ph[1] = hh; pl[1] = ll;
        if (live && mbase + 4 * rq < a.M) { wgq_st8<0>(hb, vo, ph); wgq_st8<0>(lb, vo, pl); }
    }
}
#endif
// EPI_STORE_FO, column block NBI: an fp32 plane as the only output (skip sum): the lane's 4 rows of every 16-row block are four rows of the
// plane, 4 bytes each (a wave instruction covers 4 x 64 contiguous bytes)
template <int NBI>
__device__ __forceinline__ void wgg_store_fo_nb(const ConvGemmArgs &a, f32x4 (&acc)[4][6], int b, int t0, int mw, int lane)
{
    const Geo g = a.g;
    const int col = lane & 15, rq = lane >> 4;
    const bool live = b < g.B && t0 + col < g.T;
    const int bb = min(b, g.B - 1);
    unsigned vo[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) vo[e] = (unsigned)(((4 * rq + e) * g.P + col) * 4);
#pragma unroll
    for (int mb = 0; mb < 4; ++mb) {
        const int mbase = mw + mb * 16;
        const float *fb = paddr(a.out0, g, bb, mbase, t0);
#pragma unroll
        for (int e = 0; e < 4; ++e)
            if (live && mbase + 4 * rq + e < a.M) wgq_st4<0>(fb, vo[e], acc[mb][NBI][e]);
    }
}
// The accumulate-into value of EPI_STORE_SO (an S-plane: x = hi + lo): 48 loads per lane (8 bytes each: the lane's 4 rows of one column as
// half units of the hi and of the lo array).  A product on its own leaves them to the compiler, behind the prologue's DMA (hand-issued IN
// FRONT of the DMA -- scalar bases, one lane offset, one wait -- measured slower: data-gradient conv 99 against 95 us; the whole chip
// reads 50 MB at once and the first chunks queue behind it).  Inside convlayer16g_kernel the residual product's loads are hand-issued by
// the GATE product's tail, right behind its last stores, so that this burst runs under the drain, the barrier and the residual product's
// own prologue instead of in front of its main loop (11-13 us of prologue before).
__device__ __forceinline__ void wgg_ld8(u32x2 &v, const unsigned short *base, unsigned voff)
{
    asm volatile("global_load_dwordx2 %0, %1, %2" : "=v"(v) : "v"(voff), "s"(base) : "memory");
}
// column block NBI: issue
template <int NBI>
__device__ __forceinline__ void wgg_init_issue(const ConvGemmArgs &a, const SRef &saux, u32x2 (&rh)[4][6], u32x2 (&rl)[4][6], int b, int t0, int mw, int lane)
{
    const Geo g = a.g;
    const int col = lane & 15, rq = lane >> 4;
    const int bb = min(b, g.B - 1);                           // (a column block past the last plane row: anything valid, it is never stored)
    const unsigned vo = (unsigned)(((rq >> 1) * g.P + col) * 16 + 8 * (rq & 1));
#pragma unroll
    for (int mb = 0; mb < 4; ++mb) {
        const unsigned short *hb = saux.hi + s_index(saux, g, bb, mw + mb * 16, t0);
        wgg_ld8(rh[mb][NBI], hb, vo);
        wgg_ld8(rl[mb][NBI], hb + saux.lo_off, vo);
    }
}
// every load above has landed: names the registers so that nothing that reads them moves above the wait
__device__ __forceinline__ void wgg_init_tie(u32x2 (&r)[4][6])
{
    asm volatile("" : "+v"(r[0][0]), "+v"(r[0][1]), "+v"(r[0][2]), "+v"(r[0][3]), "+v"(r[0][4]), "+v"(r[0][5]), "+v"(r[1][0]), "+v"(r[1][1]), "+v"(r[1][2]),
                      "+v"(r[1][3]), "+v"(r[1][4]), "+v"(r[1][5]), "+v"(r[2][0]), "+v"(r[2][1]), "+v"(r[2][2]), "+v"(r[2][3]), "+v"(r[2][4]), "+v"(r[2][5]),
                      "+v"(r[3][0]), "+v"(r[3][1]), "+v"(r[3][2]), "+v"(r[3][3]), "+v"(r[3][4]), "+v"(r[3][5])::"memory");
}
__device__ __forceinline__ void wgg_init_convert(f32x4 (&acc)[4][6], const u32x2 (&rh)[4][6], const u32x2 (&rl)[4][6])
{
#pragma unroll
    for (int mb = 0; mb < 4; ++mb)
#pragma unroll
        for (int nb = 0; nb < 6; ++nb) {
            const u32x2 vh = rh[mb][nb], vl = rl[mb][nb];
            acc[mb][nb][0] = __uint_as_float(vh[0] << 16) + __uint_as_float(vl[0] << 16);
            acc[mb][nb][1] = __uint_as_float(vh[0] & 0xffff0000u) + __uint_as_float(vl[0] & 0xffff0000u);
            acc[mb][nb][2] = __uint_as_float(vh[1] << 16) + __uint_as_float(vl[1] << 16);
            acc[mb][nb][3] = __uint_as_float(vh[1] & 0xffff0000u) + __uint_as_float(vl[1] & 0xffff0000u);
        }
}

// What a chunk of the K walk needs to be fetched, independent of the tile: built ONCE per workgroup (thread v describes chunk v) into an
// LDS table, because the walk itself -- interleaved taps, segment boundaries, half chunks: two integer divisions and a dozen 64-bit
// multiply-adds per chunk -- measured 334 scalar instructions per chunk and wave in the first version of this kernel: with all eight waves
// multiplying, nobody hides them (compute-only timing build: 118 us per gate conv for 62 us of MFMA issue).  From it every wave derives,
// once per TILE, the byte offsets of its seven pieces for every chunk (lane v computes chunk v: WggOffs, a second LDS table per wave), so
// that a chunk costs the wave three broadcast LDS reads, seven vector adds and two readfirstlane.
struct WggDesc {
    unsigned long long b_base;    // the chunk's operand plane: address of its hi array
    unsigned a_off;               // bytes from the weight image's first chunk to this chunk's (hi image, row 0)
    unsigned b_off;               // bytes from b_base to the unit (plane row 0, first k-group of the chunk, position H + shift)
    unsigned lo_off;              // bytes from the hi to the lo array of that plane
    unsigned item_stride;         // bytes from one plane row to the next
    unsigned nq;                  // k-groups of the chunk that hold channels (1..4); the others are fetched from the zero halo
    unsigned pad;
};
struct WggOffs { unsigned a[4], b[3], pad; };             // a: bytes from the weight image, b: bytes from the chunk's b_base
#define WGG_MAXCHUNKS 56                                  // (the longest product that runs here: the data-gradient conv, 48 chunks)
#define WGG_TAB WGG_LDS                                   // the tables sit behind the rings
#define WGG_WTAB (WGG_TAB + WGG_MAXCHUNKS * 32)
#define WGG_EFF (WGG_WTAB + 8 * WGG_MAXCHUNKS * 32)       // EPI_GATE_SO with ConvGemm16sArgs::part: Weff of the layer, [8][M / 2] fp32
#define WGG_EFF_BYTES 8192
#define WGG_LDS_ALL (WGG_EFF + WGG_EFF_BYTES)             // 163 584 bytes
static_assert(sizeof(WggDesc) == 32 && sizeof(WggOffs) == 32 && WGG_LDS_ALL <= 160 * 1024, "LDS budget");

// ConvGemm16sArgs as for convgemm16q_kernel, with ntx = column tiles of 192 flattened columns (ceil(B * Tt / 192)), nty = 256-row tiles,
// ntz unused.  EPI_STORE_FO: no accumulate-into plane.  Requires: Geo::rows == 0, no row_sel1, M a multiple of 256, H >= 64 (the zero halo serves as the 1 KB zero source),
// S-plane operands whose hi + lo arrays span less than 4 GB, a weight image (hi + lo) of less than 4 GB, at most WGG_MAXCHUNKS chunks.  Grid: min(tiles, CUs) workgroups of 512
// threads; with a grid that is a multiple of 8, XCD x (workgroup id & 7) owns the column tiles [x ntx / 8, (x + 1) ntx / 8) and all their
// row tiles, row tile fastest: the row tiles of a column tile run side by side on one L2, and a dilation tap's window is a neighbouring
// column tile's centre window on the same XCD.
// xcd_items (reused): 0 = the row tiles of a column tile on neighbouring workgroups of the XCD, in step (above); 1 = a workgroup OWNS its
// column tiles and walks their row tiles one after the other (convlayer16g_kernel below: the residual product that follows needs the whole
// gate of its columns on one CU; the activations of a column tile are then streamed once per row tile instead of being shared in L2).
// One product as a device function: everything from the chunk table to the last tile's stores (every wave returns with all its loads
// and stores done, vmcnt(0)); smem: WGG_LDS_ALL bytes.
// PRE: the accumulate-into loads of this product's FIRST tile are already in flight in (ih, il), issued by the product in front; NEXTI: this
// product's tail issues those of the product `nx` behind it (`nxt`: same column tiles, one row tile) and returns with them in flight
template <int EPI, bool LAYERK = false, bool PRE = false, bool NEXTI = false>
__device__ __forceinline__ void wgg_stream(const ConvGemm16sArgs &aa, char *smem, u32x2 (&ih)[4][6], u32x2 (&il)[4][6], const ConvGemm16sArgs *nxt)
{
    static_assert(EPI == EPI_GATE_SO || EPI == EPI_STORE_SO || EPI == EPI_STORE_FO || EPI == WGG_EPI_PART, "the hand-issued epilogues");
    const ConvGemmArgs &a = aa.c;
    const Geo g = a.g;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const unsigned lds0 = (unsigned)(size_t)(wgg_lds_char *)smem;
    int nchunks = 0;
    for (int s = 0; s < a.nseg; ++s) nchunks += (a.seg[s].nch + WG16_BK - 1) / WG16_BK;
    const int nil = aa.tap_il * aa.tap_chunks;                // chunks walked interleaved over the taps (ConvGemm16sArgs::tap_il)
    const bool splitk = aa.xcd_items == 2;                    // ONE (tile, K part) per workgroup: chunks [k0, k0 + nchunks) of the walk
    int k0 = 0, part = 0;
    const int G = (int)gridDim.x, bid = (int)blockIdx.x;
    const bool xm = (G & 7) == 0;
    const int nx = xm ? 8 : 1, xid = xm ? (bid & 7) : 0, slot = xm ? (bid >> 3) : bid, xslots = xm ? (G >> 3) : G;
    const bool own = aa.xcd_items == 1;
    const int c_lo = xid * aa.ntx / nx, n_cols = (xid + 1) * aa.ntx / nx - c_lo, n_local = n_cols * aa.nty;
    int mine = own ? (slot < n_cols ? ((n_cols - 1 - slot) / xslots + 1) * aa.nty : 0) : (slot < n_local ? (n_local - 1 - slot) / xslots + 1 : 0);
    int sk_tile = 0;                                          // split K: the XCD's (column tile, row tile) this workgroup works on
    if (splitk) {
        const int S = aa.ntz;
        mine = slot < n_local * S ? 1 : 0;
        part = slot % S; sk_tile = slot / S;
        const int kb = (int)((long)part * nchunks / S), ke = (int)((long)(part + 1) * nchunks / S);
        k0 = kb; nchunks = ke - kb;
    }
    const int total = mine * nchunks;
    if (total == 0) return;
    auto tile_at = [&](int k, int &ct, int &m0) __attribute__((always_inline)) {
        const int kk = min(k, mine - 1);
        if (splitk) {
            const int cl = sk_tile / aa.nty;
            ct = c_lo + cl; m0 = (sk_tile - cl * aa.nty) * WGG_BM;
        } else if (own) {
            const int ci = kk / aa.nty;
            ct = c_lo + slot + ci * xslots; m0 = (kk - ci * aa.nty) * WGG_BM;
        } else {
            const int L = slot + kk * xslots, cl = L / aa.nty;
            ct = c_lo + cl; m0 = (L - cl * aa.nty) * WGG_BM;
        }
    };
    const int ncols = g.B * g.Tt;

    if constexpr (EPI == EPI_GATE_SO) {
        if (aa.part) {                                        // Weff of this layer as fragments: 1 KB per 32 gate channels (at most WGG_EFF_BYTES: the launch site checks)
            const int n4 = (a.M >> 6) * 1024 / 16;
            for (int i = tid; i < n4; i += 512)
                reinterpret_cast<u32x4 *>(smem + WGG_EFF)[i] = reinterpret_cast<const u32x4 *>(aa.eff)[i];
        }
    }
    // ---------------------------------------------- the chunk table ----------------------------------------------
    if (tid < nchunks) {
        const int v = k0 + tid;                               // (the walk's chunk this entry describes)
        int sg, ci, chi;
        if (v < nil) {
            sg = v % aa.tap_il;
            const int cbi = v / aa.tap_il;
            ci = cbi * WG16_BK; chi = sg * aa.tap_chunks + cbi;
        } else {
            int c = nil, s = aa.tap_il;
            for (;;) {
                const int n = (a.seg[s].nch + WG16_BK - 1) / WG16_BK;
                if (v < c + n || s + 1 >= a.nseg) break;
                c += n; ++s;
            }
            sg = s; ci = (v - c) * WG16_BK; chi = v;
        }
        const SSeg ss = aa.sseg[sg];
        WggDesc d;
        d.b_base = (unsigned long long)(size_t)ss.hi;
        d.a_off = (unsigned)chi * (unsigned)a.lda * (unsigned)(WG16_BK * 2);
        d.b_off = ((unsigned)((ss.ch0 + ci) >> 3) * (unsigned)g.P + (unsigned)(g.H + a.seg[sg].shift)) * 16u;
        d.lo_off = (unsigned)(ss.lo_off * 2);
        d.item_stride = (unsigned)(ss.Cp >> 3) * (unsigned)g.P * 16u;
        d.nq = (unsigned)min(4, (a.seg[sg].nch - ci + 7) >> 3);
        d.pad = 0;
        *reinterpret_cast<WggDesc *>(smem + WGG_TAB + tid * 32) = d;
    }
    __syncthreads();

    // ---------------------------------------------- the wave's DMA pieces ----------------------------------------------
    // A: instruction i = wave + 8 k (k = 0..3) of the 32 per chunk: image i >> 4 (hi, lo), k-group (i >> 2) & 3, 64-row part i & 3
    // B: instruction i = wave + 8 k (k = 0..2) of the 24: image i / 12, k-group (i % 12) / 3, 64-column piece i % 3
    // MEASURED AND NOT ADOPTED: an uneven split.  Of the two waves of a SIMD the older one (waves 0-3) wins every arbitration, finishes
    // its multiplications first and waits at the chunk's barrier (phase stamps: 800 cycles per chunk on wave 0 against 160 on wave 4)
    // while its partner, alone, cannot keep the matrix pipe busy through its own DMA issue (an LDS-DMA instruction holds its wave for
    // 60-180 cycles).  With waves 0-3 (kept in front by s_setprio 1) issuing 5 + 5, 6 + 4 or 7 + 5 of the 8 A + 6 B instructions a
    // SIMD's pair owes per chunk the waits even out (236 / 164 cycles) and the launch takes 110-114 us against 107.5-109.4 --
    // MI355X_MICROARCH.md, "Two waves per SIMD", item 3: moving work between the two waves of a SIMD is zero- or negative-sum.  (Also
    // slower: the even split with waves 0-3 on the hi and waves 4-7 on the lo images, 114-116 us.)
    unsigned a_src[4], a_dst[4], b_dst[3], b_hl[3], b_q[3];   // a_src: bytes from a chunk's image block (hi, tile row 0) to the piece
    int b_pc[3];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int i = wave + 8 * k, hl = i >> 4, q = (i >> 2) & 3, part = i & 3;
        a_src[k] = 2u * ((unsigned)hl * (unsigned)aa.img_stride + (unsigned)((part >> 1) * (128 * WG16_BK) + (q * 128 + 64 * (part & 1)) * 8));
        a_dst[k] = lds0 + (unsigned)(hl * WGG_AIMG + q * WGG_APLANE + part * 1024);
    }
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const int i = wave + 8 * k;
        b_hl[k] = (unsigned)(i / 12); b_q[k] = (unsigned)((i % 12) / 3); b_pc[k] = i % 3;
        b_dst[k] = lds0 + (unsigned)(WGG_BBASE + (int)b_hl[k] * WGG_BIMG + (int)b_q[k] * WGG_BPLANE + b_pc[k] * 1024);
    }
    const unsigned voff = (unsigned)lane * 16u;
    int gi = 0, ik = 0, iv = 0, ict, im0;
    char *const wtab = smem + WGG_WTAB + wave * (WGG_MAXCHUNKS * 32);
    // per tile: lane v < nchunks writes the wave's seven byte offsets of chunk v (k-groups without channels: offset 0 = position 0 of the
    // plane, H >= 64 columns of zeros = 1 KB).  Written and read by this wave only: the LDS serves a wave's operations in order.
    auto build_table = [&]() __attribute__((always_inline)) {
        unsigned pb[3], pc[3];                                // plane row / byte offset (k-group row + first time step) of the three B pieces
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            int cf = ict * WGG_BN + 64 * b_pc[k];
            if (cf >= ncols) cf = 0;                          // (a piece past the last column: fetch something valid, nothing of it is stored)
            const int b = cf / g.Tt;
            pb[k] = (unsigned)b; pc[k] = (b_q[k] * (unsigned)g.P + (unsigned)(cf - b * g.Tt)) * 16u;
        }
        if (lane < nchunks) {
            const WggDesc d = *reinterpret_cast<const WggDesc *>(smem + WGG_TAB + lane * 32);
            WggOffs o;
#pragma unroll
            for (int k = 0; k < 4; ++k) o.a[k] = d.a_off + a_src[k] + (unsigned)im0 * (unsigned)(WG16_BK * 2);
#pragma unroll
            for (int k = 0; k < 3; ++k) o.b[k] = b_q[k] < d.nq ? d.b_off + b_hl[k] * d.lo_off + pb[k] * d.item_stride + pc[k] : 0u;
            o.pad = 0;
            *reinterpret_cast<WggOffs *>(wtab + lane * 32) = o;
        }
    };
    tile_at(0, ict, im0);
    build_table();
    // the seven instructions of a chunk: desc_request() asks for the chunk's offsets a little ahead, prep() adds the lane's own 16 bytes
    // and steps the stream, fire<k>() issues one -- a phase spreads them over its column blocks.  Past the stream's end the last chunk is
    // fetched again (valid addresses, buffers nobody reads): the instruction counts the waits rely on never change.
    u32x4 e0, e1;
    u32x2 ebs;
    unsigned va[4], vb[3], bd;
    const void *sbB;
    auto desc_request = [&]() __attribute__((always_inline)) {
        e0 = *reinterpret_cast<const u32x4 *>(wtab + iv * 32); e1 = *reinterpret_cast<const u32x4 *>(wtab + iv * 32 + 16);
        ebs = *reinterpret_cast<const u32x2 *>(smem + WGG_TAB + iv * 32);
    };
    auto prep = [&](int bslot) __attribute__((always_inline)) {
        // (the builtin returns int: through unsigned, or the low word's sign bit would fill the high word)
        auto rfl = [](unsigned x) __attribute__((always_inline)) { return (unsigned)__builtin_amdgcn_readfirstlane((int)x); };
        sbB = reinterpret_cast<const void *>((unsigned long long)rfl(ebs[0]) | ((unsigned long long)rfl(ebs[1]) << 32));
#pragma unroll
        for (int k = 0; k < 4; ++k) va[k] = e0[k] + voff;
#pragma unroll
        for (int k = 0; k < 3; ++k) vb[k] = e1[k] + voff;
        bd = (unsigned)bslot * WGG_BBUF;
        if (gi < total) {
            ++gi;
            if (++iv == nchunks) {
                iv = 0; ++ik;
                tile_at(ik, ict, im0);
                build_table();
            }
        }
    };
    auto fire = [&](auto ABUF, auto KC) __attribute__((always_inline)) {
        constexpr int k = decltype(KC)::value, abuf = decltype(ABUF)::value;
        if constexpr (k < 4) wgg_glds16(aa.img, va[k], a_dst[k] + (unsigned)(abuf * WGG_ABUF));
        else wgg_glds16(sbB, vb[k - 4], b_dst[k - 4] + bd);
    };
#define WGG_IC(n) std::integral_constant<int, n>()
    auto issue = [&](auto ABUF, int bslot) __attribute__((always_inline)) {
        desc_request();
        prep(bslot);
        fire(ABUF, WGG_IC(0)); fire(ABUF, WGG_IC(1)); fire(ABUF, WGG_IC(2)); fire(ABUF, WGG_IC(3)); fire(ABUF, WGG_IC(4)); fire(ABUF, WGG_IC(5)); fire(ABUF, WGG_IC(6));
    };

    // ---------------------------------------------- multiply ----------------------------------------------
    const int wr = wave >> 1, wc = wave & 1;                  // 4 x 2 waves: rows 64 wr .., columns 96 wc ..
    const int ao = (lane >> 4) * WGG_APLANE + (64 * wr + (lane & 15)) * 16;           // + 256 per 16-row block, + WGG_AIMG for lo
    const int bo = WGG_BBASE + (lane >> 4) * WGG_BPLANE + (96 * wc + (lane & 15)) * 16;
    f32x4 acc[4][6];
    bf16x8 Ah[2][4], Al[2][4];
    auto rd = [&](const char *q) __attribute__((always_inline)) { return *reinterpret_cast<const bf16x8 *>(q); };
#define WGG_SB() __builtin_amdgcn_sched_barrier(0)
#if defined(WGG_DBG_NOMFMA)                               // timing build: fragments are read, nothing is multiplied
#define WGG_MFMA(A_, B_, C_) asm volatile("" : "+v"(C_) : "v"(A_), "v"(B_))
#else
#define WGG_MFMA(A_, B_, C_) C_ = __builtin_amdgcn_mfma_f32_16x16x32_bf16(A_, B_, C_, 0, 0, 0)
#endif
#if defined(WGG_DBG_NOBAR)                                // timing build: no barriers (results are garbage)
#define WGG_BAR() asm volatile("" ::: "memory")
#else
#define WGG_BAR() do { __builtin_amdgcn_s_barrier(); asm volatile("" ::: "memory"); } while (0)
#endif
    int ck = 0, cc = 0, ct, m0, bs = 0, gc = 0;
    tile_at(0, ct, m0);
    // the wave's six column blocks: plane row and first time step of each (a block of 16 never straddles plane rows: Tt is a multiple of 16)
    int eb[6], et[6];
    auto block_pos = [&]() __attribute__((always_inline)) {
        const int cf0 = ct * WGG_BN + 96 * wc;
        int b = cf0 / g.Tt, t = cf0 - b * g.Tt;
#pragma unroll
        for (int nb = 0; nb < 6; ++nb) {
            eb[nb] = b; et[nb] = t;
            t += 16;
            if (t >= g.Tt) { t = 0; ++b; }
        }
    };
    // the accumulators' initial value: zero, or (EPI_STORE_SO with an accumulate-into S-plane) that plane's tile
    auto acc_start = [&](auto FIRST) __attribute__((always_inline)) {
        if (EPI == EPI_STORE_SO && aa.saux.hi) {
            if constexpr (PRE && decltype(FIRST)::value) {
                // (in flight since the product in front finished, and retired by the prologue's vmcnt(8) above: nothing left to wait for)
                asm volatile("s_waitcnt vmcnt(14)" ::: "memory");
                wgg_init_tie(ih); wgg_init_tie(il);
            } else {
                const int mw = m0 + 64 * wr, col = lane & 15, rq = lane >> 4;
                block_pos();
#pragma unroll
                for (int nb = 0; nb < 6; ++nb)
#pragma unroll
                    for (int mb = 0; mb < 4; ++mb) {
                        const size_t i = s_index(aa.saux, g, min(eb[nb], g.B - 1), mw + mb * 16 + 4 * rq, et[nb] + col);
                        ih[mb][nb] = *reinterpret_cast<const u32x2 *>(aa.saux.hi + i);
                        il[mb][nb] = *reinterpret_cast<const u32x2 *>(aa.saux.hi + aa.saux.lo_off + i);
                    }
            }
            wgg_init_convert(acc, ih, il);
        } else {
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 6; ++j)
#pragma unroll
                    for (int q = 0; q < 4; ++q) acc[i][j][q] = 0.f;
        }
    };
    auto epilogue = [&]() __attribute__((always_inline)) {
#if !defined(WGG_DBG_NOEPI)
        block_pos();
        if constexpr (EPI == EPI_GATE_SO) {
            const int chb = (m0 >> 1) + 32 * wr;                 // the wave's 64 rows = [32 tanh | 32 sigmoid] of 32 gate channels
            // the wave's Weff fragments (hi, lo): slice = (row tile, wave row) of the staged image, [slice][hi | lo][k-group][8 rows][16 B];
            // A rows 8-15 are zero
            const int slot = (m0 >> 8) * 4 + wr;
            bf16x8 eah = bf16x8{}, eal = bf16x8{};
            if (aa.part && (lane & 15) < 8) {
                const char *ef = smem + WGG_EFF + slot * 1024 + ((lane >> 4) * 8 + (lane & 15)) * 16;
                eah = *reinterpret_cast<const bf16x8 *>(ef); eal = *reinterpret_cast<const bf16x8 *>(ef + 512);
            }
            wgg_gate_nb<0>(a, aa.s0, acc, eb[0], et[0], chb, lane, aa.part, slot, eah, eal); wgg_gate_nb<1>(a, aa.s0, acc, eb[1], et[1], chb, lane, aa.part, slot, eah, eal);
            wgg_gate_nb<2>(a, aa.s0, acc, eb[2], et[2], chb, lane, aa.part, slot, eah, eal); wgg_gate_nb<3>(a, aa.s0, acc, eb[3], et[3], chb, lane, aa.part, slot, eah, eal);
            wgg_gate_nb<4>(a, aa.s0, acc, eb[4], et[4], chb, lane, aa.part, slot, eah, eal); wgg_gate_nb<5>(a, aa.s0, acc, eb[5], et[5], chb, lane, aa.part, slot, eah, eal);
        } else if constexpr (EPI == WGG_EPI_PART) {
            // the raw accumulators, block by block, 1 KB per wave and store: slab[(part * tiles + tile) * 8 + wave][mb * 6 + nb][lane]
            const int tile = ct * aa.nty + m0 / WGG_BM, ntiles = aa.ntx * aa.nty;
            const float *sbase = a.out0.p + ((size_t)(part * ntiles + tile) * 8 + wave) * (24 * 64 * 4);
            const unsigned vo = (unsigned)lane * 16u;
#pragma unroll
            for (int mb = 0; mb < 4; ++mb)
#pragma unroll
                for (int nb = 0; nb < 6; ++nb) wgg_st16(sbase + (mb * 6 + nb) * 256, vo, acc[mb][nb]);
        } else if constexpr (EPI == EPI_STORE_FO) {
            const int mw = m0 + 64 * wr;
            wgg_store_fo_nb<0>(a, acc, eb[0], et[0], mw, lane); wgg_store_fo_nb<1>(a, acc, eb[1], et[1], mw, lane);
            wgg_store_fo_nb<2>(a, acc, eb[2], et[2], mw, lane); wgg_store_fo_nb<3>(a, acc, eb[3], et[3], mw, lane);
            wgg_store_fo_nb<4>(a, acc, eb[4], et[4], mw, lane); wgg_store_fo_nb<5>(a, acc, eb[5], et[5], mw, lane);
        } else {
            const int mw = m0 + 64 * wr;
            wgg_store_nb<0>(a, aa.s0, acc, eb[0], et[0], mw, lane); wgg_store_nb<1>(a, aa.s0, acc, eb[1], et[1], mw, lane);
            wgg_store_nb<2>(a, aa.s0, acc, eb[2], et[2], mw, lane); wgg_store_nb<3>(a, aa.s0, acc, eb[3], et[3], mw, lane);
            wgg_store_nb<4>(a, aa.s0, acc, eb[4], et[4], mw, lane); wgg_store_nb<5>(a, aa.s0, acc, eb[5], et[5], mw, lane);
        }
#else
        if (acc[0][0][0] + acc[1][1][1] + acc[2][4][2] + acc[3][5][3] == 12345.f) aa.s0.hi[lane] = 1;
#endif
    };
#if defined(WG_DBG_TRACE)
    // slots 0-7: inside the chunk WGG_TRACE_GC (before the TOP wait, after it, after the barrier, after blocks 0 and 2, after the MID wait,
    // after its barrier, after block 5); 8: kernel entry, 9: first chunk, 10 / 11: tile 0's main loop / epilogue done, 12 / 13: tile 1's
#if !defined(WGG_TRACE_GC)
#define WGG_TRACE_GC 12
#endif
#if defined(WGG_TRACE_LAYER)       // stamp the products of convlayer16g_kernel only (otherwise: those of the plain launches only)
#define WGG_TRACE_ON LAYERK
#else
#define WGG_TRACE_ON (!LAYERK)
#endif
#define WGG_TRACE(slot) do { if (WGG_TRACE_ON && lane == 0 && wave == WGG_TRACE_WAVE) { const int r_ = (EPI == EPI_GATE_SO ? 0 : 256) + blockIdx.x; \
        wg_dbg_trace[r_ * 16 + (slot)] = wall_clock64(); wg_dbg_trace_cyc[r_ * 16 + (slot)] = clock64(); } } while (0)
#define WGG_TRACE_IN(slot) do { if (gc == WGG_TRACE_GC) WGG_TRACE(slot); } while (0)
#if !defined(WGG_TRACE_WAVE)
#define WGG_TRACE_WAVE 0
#endif
#else
#define WGG_TRACE(slot) do { } while (0)
#define WGG_TRACE_IN(slot) do { } while (0)
#endif
    // ------------------------------------------------------------------------------------------------------------------------
    // ONE barrier per chunk, the chunk's last column block multiplied BEHIND the next chunk's barrier (107.6-110.1 us per gate conv
    // against 113.6 for the first form of this kernel, which had a second barrier in the middle of the chunk for A(c + 1)):
    //   TOP(c): wait until the own pieces of A(c + 1) and B(c) have landed (vmcnt(3): all but the three B pieces of chunk c + 1), all own
    //           LDS reads returned (lgkmcnt(0)); barrier.  Now A(c + 1) and B(c) are readable, A(c)'s buffer (in registers since phase
    //           c - 1) and B(c - 1)'s (its last fragments are in registers) are free.
    //   then:   request B(c)'s first fragments; multiply column block 5 of chunk c - 1 from registers (that covers the LDS round trip
    //           which otherwise every wave of the workgroup sits out at once behind the barrier); if chunk c - 1 ended a tile: epilogue;
    //           column blocks 0-4 of chunk c, each requesting the next block's B fragments (three register pairs in rotation), block
    //           n < 4 copying row block n of A(c + 1) into the other A register set, blocks 0-1 issuing the four A pieces of chunk
    //           c + 2, blocks 2-4 its three B pieces.
    // The weights have ~0.9 chunk times to land (L2 hits), the activations ~1.7.
    // ------------------------------------------------------------------------------------------------------------------------
    bf16x8 B3h[3], B3l[3];
    auto mm4 = [&](const bf16x8 (&A_)[4], const bf16x8 &B_, auto NBC) __attribute__((always_inline)) {
        constexpr int nb = decltype(NBC)::value;
#pragma unroll
        for (int mb = 0; mb < 4; ++mb) WGG_MFMA(A_[mb], B_, acc[mb][nb]);
    };
    auto sblock = [&](auto PARC, auto NBC, const char *pbuf, const char *pan) __attribute__((always_inline)) {
        constexpr int PAR = decltype(PARC)::value, nb = decltype(NBC)::value, cur = nb % 3, nxt = (nb + 1) % 3;
        WGG_SB();
        mm4(Al[PAR], B3h[cur], NBC);
        WGG_SB();
        B3h[nxt] = rd(pbuf + (nb + 1) * 256); B3l[nxt] = rd(pbuf + WGG_BIMG + (nb + 1) * 256);
        if constexpr (nb < 4) { Ah[PAR ^ 1][nb] = rd(pan + nb * 256); Al[PAR ^ 1][nb] = rd(pan + WGG_AIMG + nb * 256); }
        WGG_SB();
        if constexpr (nb == 0) { prep(bs >= 1 ? bs - 1 : 2); fire(PARC, WGG_IC(0)); fire(PARC, WGG_IC(1)); }
        if constexpr (nb == 1) { fire(PARC, WGG_IC(2)); fire(PARC, WGG_IC(3)); }
        if constexpr (nb >= 2) fire(PARC, WGG_IC(nb + 2));
        WGG_SB();
        mm4(Ah[PAR], B3l[cur], NBC);
        mm4(Ah[PAR], B3h[cur], NBC);
        WGG_SB();
    };
    // column block 5 of the chunk whose A fragments are register set PAR (its B fragments: pair 2)
    auto last_block = [&](auto PARC) __attribute__((always_inline)) {
        constexpr int PAR = decltype(PARC)::value;
        WGG_SB();
        mm4(Al[PAR], B3h[2], std::integral_constant<int, 5>());
        mm4(Ah[PAR], B3l[2], std::integral_constant<int, 5>());
        mm4(Ah[PAR], B3h[2], std::integral_constant<int, 5>());
        WGG_SB();
    };
    auto tile_done = [&]() __attribute__((always_inline)) {
        if (ck < 2) WGG_TRACE(10 + 2 * ck);
        epilogue();
        if (ck < 2) WGG_TRACE(11 + 2 * ck);
        ++ck;
        tile_at(ck, ct, m0);
    };
    auto phase = [&](auto PARC) __attribute__((always_inline)) {
        constexpr int PAR = decltype(PARC)::value;
        WGG_TRACE_IN(0);
        asm volatile("s_waitcnt vmcnt(3)\n\ts_waitcnt lgkmcnt(0)" ::: "memory");
        WGG_TRACE_IN(1);
        WGG_BAR();
        WGG_TRACE_IN(2);
        const char *pbuf = smem + bs * WGG_BBUF + bo, *pan = smem + (PAR ^ 1) * WGG_ABUF + ao;
        B3h[0] = rd(pbuf); B3l[0] = rd(pbuf + WGG_BIMG);
        desc_request();
        if (gc > 0) {
            last_block(std::integral_constant<int, PAR ^ 1>());
            if (cc == 0) { tile_done(); acc_start(std::false_type()); }
        }
        WGG_TRACE_IN(3);
        sblock(PARC, std::integral_constant<int, 0>(), pbuf, pan);
        sblock(PARC, std::integral_constant<int, 1>(), pbuf, pan);
        sblock(PARC, std::integral_constant<int, 2>(), pbuf, pan);
        WGG_TRACE_IN(4);
        sblock(PARC, std::integral_constant<int, 3>(), pbuf, pan);
        sblock(PARC, std::integral_constant<int, 4>(), pbuf, pan);
        WGG_TRACE_IN(7);
        bs = bs == 2 ? 0 : bs + 1;
        ++gc;
        if (++cc == nchunks) cc = 0;
    };
    WGG_TRACE(8);
    if constexpr (PRE) {
        // Behind another product of the same launch whose OUTPUT is this product's B operand (convlayer16g_kernel): the weight pieces of
        // chunks 0 and 1 go out at once -- they depend on nothing -- then the wave waits for everything older than those eight (vmcnt(8):
        // the stores of the product in front, its trailing fetches, the accumulate-into loads it issued for us), the barrier makes that
        // true of every wave's stores, and only then the activation pieces are requested.  The table builds above and the weights' trip
        // run under the store drain (4-9 us: this part acknowledges a store at memory speed) instead of behind it.
        desc_request();
        prep(0);
        fire(WGG_IC(0), WGG_IC(0)); fire(WGG_IC(0), WGG_IC(1)); fire(WGG_IC(0), WGG_IC(2)); fire(WGG_IC(0), WGG_IC(3));
        const void *sb0 = sbB;
        const unsigned v0[3] = {vb[0], vb[1], vb[2]}, d0 = bd;
        desc_request();
        prep(1);
        fire(WGG_IC(1), WGG_IC(0)); fire(WGG_IC(1), WGG_IC(1)); fire(WGG_IC(1), WGG_IC(2)); fire(WGG_IC(1), WGG_IC(3));
        asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        WGG_BAR();
#pragma unroll
        for (int k = 0; k < 3; ++k) wgg_glds16(sb0, v0[k], b_dst[k] + d0);
        fire(WGG_IC(1), WGG_IC(4)); fire(WGG_IC(1), WGG_IC(5)); fire(WGG_IC(1), WGG_IC(6));
    } else {
        issue(std::integral_constant<int, 0>(), 0);
        issue(std::integral_constant<int, 1>(), 1);
    }
    acc_start(std::true_type());
    asm volatile("s_waitcnt vmcnt(10)" ::: "memory");        // the own pieces of A(0): younger are B(0), A(1), B(1)
    WGG_BAR();
#pragma unroll
    for (int mb = 0; mb < 4; ++mb) { Ah[0][mb] = rd(smem + ao + mb * 256); Al[0][mb] = rd(smem + WGG_AIMG + ao + mb * 256); }
    WGG_TRACE(9);
    while (gc < total) {
        phase(std::integral_constant<int, 0>());
        if (gc < total) phase(std::integral_constant<int, 1>());
    }
    if (total & 1) last_block(std::integral_constant<int, 0>());      // (the last chunk's A fragments: register set (total - 1) & 1)
    else last_block(std::integral_constant<int, 1>());
    tile_done();
    if constexpr (NEXTI) {
        // the accumulate-into tile of the next product's first tile (the workgroup's first column tile, one row tile: its geometry is
        // this product's), 48 loads per lane right behind the last stores; return with them in flight
        ck = 0;
        tile_at(0, ct, m0);
        block_pos();
        const int mw = 64 * wr;
        wgg_init_issue<0>(nxt->c, nxt->saux, ih, il, eb[0], et[0], mw, lane); wgg_init_issue<1>(nxt->c, nxt->saux, ih, il, eb[1], et[1], mw, lane);
        wgg_init_issue<2>(nxt->c, nxt->saux, ih, il, eb[2], et[2], mw, lane); wgg_init_issue<3>(nxt->c, nxt->saux, ih, il, eb[3], et[3], mw, lane);
        wgg_init_issue<4>(nxt->c, nxt->saux, ih, il, eb[4], et[4], mw, lane); wgg_init_issue<5>(nxt->c, nxt->saux, ih, il, eb[5], et[5], mw, lane);
        // (no wait here: the next product waits for the stores, the trailing fetches and these loads behind its own first weight pieces;
        // a trailing fetch lands in a ring slot before anything the same wave requests into it later -- the queue is in order)
    } else {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // the trailing fetches must not land in another workgroup's LDS
    }
    WGG_TRACE(14);                                           // (every store of the wave is done)
#undef WGG_SB
#undef WGG_MFMA
#undef WGG_BAR
#undef WGG_IC
#undef WGG_TRACE
#undef WGG_TRACE_ON
#undef WGG_TRACE_IN
}

template <int EPI>
__global__ __launch_bounds__(512) void convgemm16g_kernel(const ConvGemm16sArgs aa)
{
    __shared__ __attribute__((aligned(1024))) char smem[WGG_LDS_ALL];
    u32x2 ih[4][6], il[4][6];
    wgg_stream<EPI>(aa, smem, ih, il, nullptr);
}

// ------------------------------------------------------------------------------------------------------------------------------------
// A WN layer's gate conv AND the residual product behind it in one launch (model/waveglow.py:41-46: conv -> gate -> W_o -> res + x).
// Rounds 2-4 built this three times with the gate crossing WORKGROUPS inside the launch (counters, write-through stores, polls) and
// measured each form no faster than two launches.  Here nothing crosses: with 256 x 192 tiles over flattened
// columns a workgroup owns whole column tiles (p[0].xcd_items = 1), computes BOTH 256-row tiles of the gate for them one after the other
// and stores them as usual; when its own stores have reached L2 (every wave's vmcnt(0), a barrier) the same workgroup streams that gate
// back in as the B operand of the residual product h_{i+1} = h_i + W_res gate (p[1]: K = the gate's channels, 8 chunks; the accumulate-into
// value is h_i's S-plane) for the same columns.  What it saves is a launch per layer -- 161 per training step at the headline shape, each
// ~35 us for 12 us of main loop: dispatch gap, prologue, a cold L2 -- for 8 more chunks and one more epilogue on a workgroup that is
// running anyway.  What it costs: the two row tiles of a column tile no longer run side by side on two CUs sharing the activations in
// L2, so the activations are fetched once per row tile.
// ------------------------------------------------------------------------------------------------------------------------------------
struct ConvLayer16gArgs {
    ConvGemm16sArgs p[2];         // [0] the gate conv (EPI_GATE_SO, nty row tiles), [1] the residual product (EPI_STORE_SO, one row tile); both xcd_items = 1
};
static_assert(sizeof(ConvLayer16gArgs) <= 4096, "kernel arguments are limited to 4 KB");
__global__ __launch_bounds__(512) void convlayer16g_kernel(const ConvLayer16gArgs la)
{
    __shared__ __attribute__((aligned(1024))) char smem[WGG_LDS_ALL];
    u32x2 ih[4][6], il[4][6];                                 // the residual product's accumulate-into tile, in flight across the seam
    wgg_stream<EPI_GATE_SO, true, false, true>(la.p[0], smem, ih, il, &la.p[1]);
    // nobody may still be reading the rings or the tables when the next product's prologue overwrites them (the gate tiles' stores are
    // waited for inside the residual product, behind its first weight pieces; this CU's L1 never held those lines).  (The tile positions the gate
    // product's tail used for the loads are the residual product's own: same column tiles, same workgroup order, row tile 0.)
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    wgg_stream<EPI_STORE_SO, true, true, false>(la.p[1], smem, ih, il, nullptr);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");         // (nothing the gate product's tail issued outlives the wave on any path)
}

// the K parts of a split product added up and gated: one workgroup per (tile, column block of the wave tiles), thread = (wave, lane) as in
// the product; the epilogue is the product's own (wgg_gate_nb)
__global__ __launch_bounds__(512) void gate_finish16g_kernel(const ConvGemm16sArgs aa, const float *slab, int splits)
{
    const ConvGemmArgs &a = aa.c;
    const Geo g = a.g;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int tile = (int)blockIdx.x / 6, nb = (int)blockIdx.x - tile * 6, ntiles = aa.ntx * aa.nty;
    const int ct = tile / aa.nty, m0 = (tile - ct * aa.nty) * WGG_BM, wr = wave >> 1, wc = wave & 1;
    f32x4 acc[4][6];
#pragma unroll
    for (int mb = 0; mb < 4; ++mb) {
        f32x4 x = {0.f, 0.f, 0.f, 0.f};
        for (int p = 0; p < splits; ++p) {
            const f32x4 v = *reinterpret_cast<const f32x4 *>(slab + (((size_t)(p * ntiles + tile) * 8 + wave) * 24 + (mb * 6 + nb)) * 256 + lane * 4);
            x += v;
        }
        acc[mb][0] = x;
    }
    const int cf = ct * WGG_BN + 96 * wc + 16 * nb, b = cf / g.Tt, t0 = cf - b * g.Tt;
    wgg_gate_nb<0>(a, aa.s0, acc, b, t0, (m0 >> 1) + 32 * wr, lane);
}

