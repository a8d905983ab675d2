// wg_gemm16m.h -- the 64 x 64-tile conv kernel with MANY chunks in flight ("m"): launches that cannot fill the chip (single-utterance
// synthesis, WaveFlow's row-by-row inverse), where a launch is as long as one CU needs to take in its tile's operands.
//
// convgemm16h_kernel (wg_gemm16h.h) moves them through the registers of four loader waves, three chunks (48 KB) in flight: 352 ns per
// 16 KB chunk = 46 GB/s per CU (profiles/r02d_infer_trace.json), 14.4 us per gate conv of a 0.7 s utterance (27 chunks) -- flat since
// round 2.  Here the operands go by LDS-DMA (global_load_lds_dwordx4) into a ring of EIGHT chunk buffers (128 KB of the CU's 160 KB
// LDS; a workgroup owns its CU anyway), seven chunks in flight, issued by the four waves that also multiply (one per SIMD: each the
// whole 64 rows x 16 columns, as convgemm16h's compute waves, so the epilogues are that kernel's).  Images are k-group planes
// [A hi | A lo | B hi | B lo][k-group][64 rows][16 B] as in wg_gemm16g.h: a DMA instruction is one plane = 1 KB contiguous on both sides,
// wave w moves k-group w of all four images.  Per chunk: wait for the own pieces of chunk c (vmcnt(24): all but the six younger
// chunks), barrier, ten fragment reads, the four pieces of chunk c + 7 into the buffer chunk c - 1 was read from, twelve MFMAs.  What a
// chunk needs is described once per workgroup (thread v: chunk v) in an LDS table: the K walk costs the waves nothing.
// Requires H >= 64 (the zero halo is the 1 KB zero source).
//
// MEASURED AND NOT ADOPTED (round 5, gpurun_out/r05o.txt): parity green (55 synthesis / block tests), and slower than convgemm16h_kernel:
// 2.70 against 2.46-2.52 ms per 0.7 s WaveGlow utterance (5.96 against 6.4-6.6 MHz), WaveFlow's row-by-row synthesis 96.6 against 86.7 ms.
// Seven chunks in flight instead of three buy nothing -- round 2 had measured the same with register stages (a CU takes its operands in
// at 46-58 GB/s whatever the depth) -- and one multiplying wave per SIMD pays the barrier and the LDS round trip of every chunk in the
// open.  To build it again: include this file behind wg_gemm16h.h in csrc/wgflow.hip and launch convgemm16m_kernel<EPI> with 256 threads
// where run_convgemm takes convgemm16h_kernel.
#pragma once
#include "wg_gemm16g.h"
#include "wg_gemm16h.h"

#define WGM_RING 8
#define WGM_SLOT (16 * 1024)
#define WGM_MAXCHUNKS 160
#define WGM_TAB (WGM_RING * WGM_SLOT)
#define WGM_LDS (WGM_TAB + WGM_MAXCHUNKS * 32)
struct WgmDesc {
    unsigned long long a;        // address of the chunk's weight image (hi), k-group 0, the tile's first row
    unsigned long long b;        // address of the S-plane unit (the tile's plane row, first k-group of the chunk, time step t0 + shift), hi array; 0: reads as zero
    unsigned long long lo_off;   // bytes from the hi to the lo array of that plane
    unsigned nq, pad;            // k-groups of the chunk that hold channels
};
static_assert(sizeof(WgmDesc) == 32, "");

template <int EPI, bool IN_MEMORY = false>
__device__ __forceinline__ void convgemm16m_body(const ConvGemm16sArgs &aa, int id, int row_sel1, char *smem)
{
    constexpr int R = WGM_RING;
    const ConvGemmArgs &a = aa.c;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const Geo g = a.g;
    const unsigned lds0 = (unsigned)(size_t)(wgg_lds_char *)smem;
    int nchunks = 0;
    for (int s = 0; s < a.nseg; ++s) nchunks += (a.seg[s].nch + WG16_BK - 1) / WG16_BK;
    const int tx = id % aa.ntx, q_ = id / aa.ntx, ty = q_ % aa.nty, tz = q_ / aa.nty;
    const int t0 = tx * 64, m0 = ty * 64;
    const int b = row_sel1 ? tz * g.rows + row_sel1 - 1 : tz;
    if (m0 >= a.M) return;                                    // (M is padded to 128 rows in the image: the upper half tile may be empty)
    // ------------------------------- the chunk table -------------------------------
    if (tid < nchunks) {
        int s = 0, c = 0;
        for (;;) {
            const int n = (a.seg[s].nch + WG16_BK - 1) / WG16_BK;
            if (tid < c + n || s + 1 >= a.nseg) break;
            c += n; ++s;
        }
        const int ci = (tid - c) * WG16_BK;
        const SSeg ss = aa.sseg[s];
        int bsrc = b;
        bool rowok = true;
        if (g.rows > 0) {
            const int item = b / g.rows, rr = b - item * g.rows + ss.row_off;
            rowok = rr >= 0 && rr < g.rows;
            bsrc = ss.per_item ? item : b + ss.row_off;
        }
        WgmDesc d;
        d.a = (unsigned long long)(size_t)(aa.img + ((size_t)tid * a.lda + (m0 & ~127)) * WG16_BK + (size_t)(m0 & 64) * 8);
        d.b = rowok ? (unsigned long long)(size_t)(ss.hi + (((size_t)bsrc * (ss.Cp >> 3) + ((ss.ch0 + ci) >> 3)) * g.P + (size_t)(g.H + t0 + a.seg[s].shift)) * 8) : 0ull;
        d.lo_off = (unsigned long long)ss.lo_off * 2;
        d.nq = (unsigned)min(4, (a.seg[s].nch - ci + 7) >> 3);
        d.pad = 0;
        *reinterpret_cast<WgmDesc *>(smem + WGM_TAB + tid * 32) = d;
    }
    __syncthreads();
    // ------------------------------- DMA: wave w moves k-group w of A hi, A lo, B hi, B lo -------------------------------
    const unsigned voff = (unsigned)lane * 16u;
    const unsigned long long zsrc = (unsigned long long)(size_t)aa.sseg[0].hi;     // plane position 0: H >= 64 columns of zeros = 1 KB
    const unsigned long long a_lo = (unsigned long long)aa.img_stride * 2, a_q = (unsigned long long)wave * (128 * 16), b_q = (unsigned long long)wave * g.P * 16;
    auto rfl = [](unsigned x) __attribute__((always_inline)) { return (unsigned)__builtin_amdgcn_readfirstlane((int)x); };
    int ic = 0;                                               // next chunk to issue (past the end: the last one again, into buffers nobody reads)
    auto issue = [&]() __attribute__((always_inline)) {
        const char *e = smem + WGM_TAB + min(ic, nchunks - 1) * 32;
        const u32x4 d0 = *reinterpret_cast<const u32x4 *>(e), d1 = *reinterpret_cast<const u32x4 *>(e + 16);
        const unsigned long long pa = ((unsigned long long)rfl(d0[1]) << 32 | rfl(d0[0])) + a_q;
        const unsigned long long pb0 = (unsigned long long)rfl(d0[3]) << 32 | rfl(d0[2]);
        const unsigned long long lo = (unsigned long long)rfl(d1[1]) << 32 | rfl(d1[0]);
        const bool ok = pb0 != 0 && (unsigned)wave < rfl(d1[2]);
        const unsigned long long pb = ok ? pb0 + b_q : zsrc, pbl = ok ? pb0 + b_q + lo : zsrc;
        const unsigned dst = lds0 + (unsigned)((ic % R) * WGM_SLOT) + (unsigned)wave * 1024u;
        wgg_glds16(reinterpret_cast<const void *>(pa), voff, dst);
        wgg_glds16(reinterpret_cast<const void *>(pa + a_lo), voff, dst + 4096u);
        wgg_glds16(reinterpret_cast<const void *>(pb), voff, dst + 8192u);
        wgg_glds16(reinterpret_cast<const void *>(pbl), voff, dst + 12288u);
        ++ic;
    };
#pragma unroll 1
    for (int i = 0; i < R - 1; ++i) issue();
    // ------------------------------- multiply: the wave's 64 rows x 16 columns -------------------------------
    const int wc = wave;
    f32x4 acc[4][1];
    if (EPI == EPI_STORE || EPI == EPI_RESSKIP) {
        conv_acc_init_a<EPI, IN_MEMORY>(a, aa.saux, aa.img, acc, t0, m0, b, wc, lane);       // (hand-issued loads behind one vmcnt(0): the prologue's pieces land with them)
    } else {
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int e = 0; e < 4; ++e) acc[i][0][e] = 0.f;
    }
    const int ao = (lane >> 4) * 1024 + (lane & 15) * 16, bo = 8192 + (lane >> 4) * 1024 + (wc * 16 + (lane & 15)) * 16;
    auto rd = [&](const char *p) __attribute__((always_inline)) { return *reinterpret_cast<const bf16x8 *>(p); };
#pragma unroll 1
    for (int c = 0; c < nchunks; ++c) {
        // the own pieces of chunk c have landed (younger: the six chunks behind it), every own LDS read has returned
        asm volatile("s_waitcnt vmcnt(24)\n\ts_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        const char *sb = smem + (c % R) * WGM_SLOT;
        bf16x8 ah[4], al[4], bh, bl;
#pragma unroll
        for (int i = 0; i < 4; ++i) { ah[i] = rd(sb + ao + i * 256); al[i] = rd(sb + 4096 + ao + i * 256); }
        bh = rd(sb + bo); bl = rd(sb + 4096 + bo);
        __builtin_amdgcn_sched_barrier(0);
        issue();                                              // chunk c + 7 -> the buffer of chunk c - 1 (everybody's reads of it returned before the barrier)
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int mb = 0; mb < 4; ++mb) acc[mb][0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al[mb], bh, acc[mb][0], 0, 0, 0);
#pragma unroll
        for (int mb = 0; mb < 4; ++mb) acc[mb][0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[mb], bl, acc[mb][0], 0, 0, 0);
#pragma unroll
        for (int mb = 0; mb < 4; ++mb) acc[mb][0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[mb], bh, acc[mb][0], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
    }
    if (EPI == EPI_STORE || EPI == EPI_RESSKIP) conv_epilogue_a<EPI, IN_MEMORY>(a, aa.s0, acc, t0, m0, b, wc, lane);
    else conv_epilogue_q<EPI, 1>(a, aa.s0, acc, t0, m0, b, 0, wc, lane);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");         // the trailing fetches must not land in another workgroup's LDS
}

template <int EPI>
__global__ __launch_bounds__(256) void convgemm16m_kernel(const ConvGemm16sArgs aa)
{
    __shared__ __attribute__((aligned(1024))) char smem[WGM_LDS];
    convgemm16m_body<EPI>(aa, (int)blockIdx.x, aa.c.row_sel1, smem);
}
