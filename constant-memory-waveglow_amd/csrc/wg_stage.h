// wg_stage.h -- a recorded launch sequence executed by ONE persistent kernel ("stage interpreter").
//
// WaveFlow's inverse (model/waveflow.py:230-254, reverse_mode_forward) is 63 dependent row steps per flow, each 19 small launches (the
// start conv, eight layers of gate conv + residual/skip conv on 64 x 64 tiles, WN.end fused with the coupling): ~9 600 launches of
// ~10 us per call, every one of them far too small to fill the chip -- the call is a chain of launch latencies (163 kHz for a 0.7 s
// utterance).  A row step's launches do not change from row to row except for the row index, so the host RECORDS them once per flow
// (Ctx::rec in wgflow.hip: the same wn_forward / wf_couple code that would launch them fills a list of argument blocks instead) and
// wf_rowsteps_kernel walks rows x stages on the device: a workgroup takes the tiles of the current stage it is dealt, then all
// workgroups meet at a grid barrier (an atomic counter; release / acquire fences make the planes written before it visible across
// XCDs) -- 2-4 us instead of a kernel boundary plus launch latency.  The tile code is the launched kernels' own (convgemm16h_body,
// to_splane_body, wf_couple_body): results are bit-identical to the launch-by-launch path.
//
// Co-residency: the barrier needs every workgroup of the grid running, so the grid is capped at one workgroup per CU (and at the widest
// stage's tile count); a stage with more tiles than workgroups is walked in rounds.  A barrier that is not released within
// WGS_SPIN_CAP polls raises a flag the host checks -- the workgroup goes on (wrong results, reported) instead of hanging the GPU.
#pragma once
#include "wg_gemm16h.h"
#include "wg_wf.h"

#define WGS_CONV_STORE 0
#define WGS_CONV_GATE 1
#define WGS_CONV_RESSKIP 2
#define WGS_TOSPLANE 3
#define WGS_WFCOUPLE 4
#define WGS_SPIN_CAP (1 << 24)
#define WGS_THREADS 512

struct WgsToSplane {
    PRef src;
    SRef dst;
    Geo g;
    int nvalid, cgs, items;            // channel groups of the destination; items (the stage converts plane row item * rows + row of each)
};
struct WgStage {
    int kind, nblocks;
    int pad[2];
    union U {
        ConvGemm16sArgs conv;
        WgsToSplane tsp;
        WfCoupleArgs cpl;
        __host__ __device__ U() {}
    } u;
};
// the program goes to device memory through kernel arguments (stream ordered, no host buffer has to outlive the call): four stages a launch
#define WGS_PER_STORE 4
struct WgsStoreArgs {
    WgStage st[WGS_PER_STORE];
};
static_assert(sizeof(WgsStoreArgs) + 16 <= 4096, "kernel arguments are limited to 4 KB");
__global__ void wgs_store_kernel(const WgsStoreArgs a, int n, WgStage *dst)
{
    const int nw = (int)(sizeof(WgStage) / sizeof(int));
    for (int s = 0; s < n; ++s) {
        const int *src = reinterpret_cast<const int *>(&a.st[s]);
        int *d = reinterpret_cast<int *>(dst + s);
        for (int i = threadIdx.x; i < nw; i += blockDim.x) d[i] = src[i];
    }
}

// every workgroup of the grid has arrived `target / gridDim.x` times; what they wrote before is visible after
__device__ __forceinline__ void wgs_grid_barrier(unsigned *ctr, unsigned target, int *fail)
{
    __syncthreads();
    if (threadIdx.x == 0) {
        __threadfence();                                      // release: this workgroup's stores reach memory (other XCDs' L2s do not snoop)
        __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        int spins = 0;
        while (__hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
            __builtin_amdgcn_s_sleep(1);
            if (++spins > WGS_SPIN_CAP) { *fail = 1; break; }
        }
        __threadfence();                                      // acquire: drop stale lines before the next stage reads the others' planes
    }
    __syncthreads();
}

// rows [0, nrows) x stages [0, nstages): conv stages run with row_sel1 = row + 1, the coupling with row_sel = row, the S-plane
// conversion on plane row item * g.rows + row
__global__ __launch_bounds__(WGS_THREADS) void wf_rowsteps_kernel(const WgStage *prog, int nstages, int nrows, unsigned *bar, int *fail)
{
    __shared__ __attribute__((aligned(16))) char smem[WG16H_SMEM];
    __shared__ __attribute__((aligned(16))) int s_stage[(sizeof(WgStage) + 3) / 4];
    __shared__ float red[WGS_THREADS];
    const int tid = threadIdx.x;
    const unsigned G = gridDim.x;
    unsigned arrived = 0;
    for (int row = 0; row < nrows; ++row)
        for (int s = 0; s < nstages; ++s) {
            {                                                  // the stage's argument block: global memory -> LDS (uniform reads from there)
                const int *src = reinterpret_cast<const int *>(prog + s);
                for (int i = tid; i < (int)(sizeof(WgStage) / 4); i += WGS_THREADS) s_stage[i] = src[i];
            }
            __syncthreads();
            const WgStage &st = *reinterpret_cast<const WgStage *>(s_stage);
            const int kind = __builtin_amdgcn_readfirstlane(st.kind), nb = __builtin_amdgcn_readfirstlane(st.nblocks);
            for (int blk = (int)blockIdx.x; blk < nb; blk += (int)G) {
#if defined(WG_DBG_ROWWALK_ONLY)   // bisecting aid: only this stage of row 0 runs
                if (s != WG_DBG_ROWWALK_ONLY || row != 0) continue;
#endif
#if defined(WG_DBG_ROWWALK)    // bisecting aid: bit 0 skips the conv stages, bit 1 the S-plane conversion, bit 2 the coupling
                if (((WG_DBG_ROWWALK & 1) && kind <= WGS_CONV_RESSKIP) || ((WG_DBG_ROWWALK & 2) && kind == WGS_TOSPLANE) ||
                    ((WG_DBG_ROWWALK & 4) && kind == WGS_WFCOUPLE) || ((WG_DBG_ROWWALK & 8) && kind == WGS_CONV_STORE) ||
                    ((WG_DBG_ROWWALK & 16) && kind == WGS_CONV_GATE) || ((WG_DBG_ROWWALK & 32) && kind == WGS_CONV_RESSKIP)) continue;
#endif
#if defined(WG_DBG_ROWWALK_GLOBALARGS)
                const ConvGemm16sArgs &cv = prog[s].u.conv;
#else
                const ConvGemm16sArgs &cv = st.u.conv;
#endif
                if (kind == WGS_CONV_STORE) convgemm16h_body<EPI_STORE, true>(cv, blk, row + 1, smem);
                else if (kind == WGS_CONV_GATE) convgemm16h_body<EPI_GATE, true>(cv, blk, row + 1, smem);
                else if (kind == WGS_CONV_RESSKIP) convgemm16h_body<EPI_RESSKIP, true>(cv, blk, row + 1, smem);
                else if (kind == WGS_TOSPLANE) {
                    // blk -> (time block of WGS_THREADS columns, channel group, item)
                    const WgsToSplane &a = st.u.tsp;
                    const int tb = (a.g.T + WGS_THREADS - 1) / WGS_THREADS;
                    const int bx = blk % tb, q = blk / tb, cg = q % a.cgs, item = q / a.cgs;
                    to_splane_body(a.src, a.nvalid, a.dst, a.g, bx * WGS_THREADS + tid, cg, item * a.g.rows + row);
                } else if (kind == WGS_WFCOUPLE) wf_couple_body<WGS_THREADS>(st.u.cpl, blk, row, red);
                __syncthreads();                              // (the LDS buffers are free again)
            }
            arrived += G;
            wgs_grid_barrier(bar, arrived, fail);
        }
}
