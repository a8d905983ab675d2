// wgflow.hip -- host side of libwgflow.so: the C ABI of include/wgflow.h on top of the kernels in
// wg_gemm.h / wg_small.h.  Every function only enqueues kernels on the caller's stream; all device memory
// comes from the caller (packed-weight buffer + workspace), carved by the deterministic layouts below.
#include "../../include/wgflow.h"
#include "wg_small.h"
#include "wg_gemm16.h"
#include "wg_gemm16s.h"
#include "wg_gemm16q.h"
#include "wg_gemm16g.h"
#include "wg_gemm16h.h"
#include "wg_wgrad16t.h"
#include "wg_wsr.h"
#include "wg_wf.h"
#include "wg_mel.h"
#include "wg_stage.h"
#include "wg_layer16h.h"
#include "wg_layer16q.h"
#include "wg_thin.h"
#include "wg_probe.h"

#include <algorithm>
#include <atomic>
#include <mutex>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <vector>

namespace {

inline int rup(int x, int m) { return (x + m - 1) / m * m; }
inline size_t rupz(size_t x, size_t m) { return (x + m - 1) / m * m; }

// ------------------------------------------------------------------------------------------------
// developer switches from the environment
// ------------------------------------------------------------------------------------------------
// Read ONCE per process (first use) and again only when the caller asks (wg_reload_env: what a test or an A/B run calls after it has
// changed a variable).  getenv() on every launch raced with setenv / os.environ writes from other threads of the process (undefined
// behaviour in libc) and put several scans of the environment in front of each of the ~250 launches of a synthesis call.
struct EnvSw {
    bool g192 = true;            // WG_G192=0: the conv products never take the 256 x 192-tile kernel of wg_gemm16g.h
    bool g192_splitk = true;     // WG_G192_SPLITK=0: a gate conv whose tiles cannot fill the chip is not cut along K
    bool g192_own = false;       // WG_G192_OWN=1: experiment, column ownership for stand-alone products
    bool layer_fusion = false;   // WG_LAYER_FUSION=1: the one-launch layer for small grids (wg_layer16h.h), opt-in
    bool layer_fusion_big = false;   // WG_LAYER_FUSION_BIG=1: the one-launch layer on 256 x 128 tiles (wg_layer16q.h), opt-in
    bool layer_g = true;         // WG_LAYER_G=0: gate conv and residual product as two launches
    bool inv_seam = false;       // WG_INV_SEAM=1: end conv + affine + inverse 1x1 + next start conv as one launch, opt-in
    bool lowrank = true;         // WG_LOWRANK=0: the skip sum and its gradient are formed as planes again (lowrank_on below)
    bool lowrank_all = false;    // WG_LOWRANK=2: also where it was measured slower (2 ic > 8: WSRGlow) -- tests of those instantiations
    bool tw_from_gate = true;    // WG_TW_FROM_GATE=0: a pass that keeps its planes keeps tanh as well (tw_from_gate below)
    bool start_fold = true;      // WG_START_FOLD=0: the first layer's dilated conv reads h_0 again instead of xa through the composed weight (start_fold_on below)
    int layer_min_chunks = 16;   // WG_LAYER_MIN_CHUNKS=n: the shortest gate product (in 32-deep chunks) that still takes the one-launch layer of wg_gemm16g.h
};
static std::atomic<const EnvSw *> g_env{nullptr};
static const EnvSw *env_load()
{
    auto is = [](const char *name, char c) { const char *e = getenv(name); return e && e[0] == c; };
    EnvSw *n = new EnvSw;        // (a reload leaks the table it replaces: 16 bytes per call of a test-only entry point, never freed under a reader)
    n->g192 = !is("WG_G192", '0');
    n->g192_splitk = !is("WG_G192_SPLITK", '0');
    { const char *e = getenv("WG_G192_OWN"); n->g192_own = e && atoi(e) != 0; }
    n->layer_fusion = is("WG_LAYER_FUSION", '1');
    n->layer_fusion_big = is("WG_LAYER_FUSION_BIG", '1');
    n->layer_g = !is("WG_LAYER_G", '0');
    n->inv_seam = is("WG_INV_SEAM", '1');
    n->lowrank = !is("WG_LOWRANK", '0');
    n->lowrank_all = is("WG_LOWRANK", '2');
    n->tw_from_gate = !is("WG_TW_FROM_GATE", '0');
    n->start_fold = !is("WG_START_FOLD", '0');
    { const char *e = getenv("WG_LAYER_MIN_CHUNKS"); if (e && atoi(e) >= 2) n->layer_min_chunks = atoi(e); }
    return n;
}
static const EnvSw &env_sw()
{
    const EnvSw *e = g_env.load(std::memory_order_acquire);
    if (!e) {
        static std::mutex mu;
        std::lock_guard<std::mutex> lk(mu);
        e = g_env.load(std::memory_order_acquire);
        if (!e) { e = env_load(); g_env.store(e, std::memory_order_release); }
    }
    return *e;
}

// ------------------------------------------------------------------------------------------------
// launch bookkeeping
// ------------------------------------------------------------------------------------------------
struct Ctx {
    hipStream_t st;
    int err;
    int prec;   // 0: exact fp32 MFMA; 1: bf16x3, operands split on the fly (wg_gemm16.h); 2: bf16x3 from pre-split S-planes (wg_gemm16s.h)
    int row_sel1;   // Geo::rows > 0 only: 0 = every plane row; r + 1 = the conv launches cover height row r of every item (WaveFlow's inverse)
    struct FinQueue *fq = nullptr;   // set inside wn_backward: weight-gradient slabs come from its arena, finalisations are batched
    float *gslab = nullptr;          // set inside wn_forward when the workspace has a weight-gradient slab (idle during a forward pass):
    size_t gslab_floats = 0;         // scratch for the K parts of a split gate conv (convgemm16g_kernel<WGG_EPI_PART>)
    struct StageRec *rec = nullptr;  // set while a launch sequence is RECORDED for the stage interpreter (wg_stage.h): nothing is launched
    struct BigCap *cap = nullptr;    // set while run_convgemm only DESCRIBES a chip-filling launch (run_convlayer_big): nothing is launched
    // the gate conv's share of WN's `out` from its own epilogue (ConvGemm16sArgs::part; wn_forward sets these around a layer's gate conv):
    const float *gate_eff = nullptr; // Weff of the layer as A fragments (WnPack::effA)
    float *gate_part = nullptr;      // that layer's partial rows
    int gate_prow = 8;               // floats per row (gate_part_prow)
    int part_written = 0;            // gate convs of this call that were launched with them
    int *probe = nullptr;            // set while run_convgemm only REPORTS whether a gate conv would take the kernel that writes them (1) or not (0)
};
// a launch of the 256 x 128-tile conv kernel as its argument block (ntx x nty x ntz tiles of 256 rows), instead of the launch
struct BigCap {
    ConvGemm16sArgs as;
    bool ok;
};
// the launches of one row step of WaveFlow's inverse, as argument blocks (see wg_stage.h); ok = false: a launch that the interpreter
// cannot run turned up (the caller then launches everything the ordinary way)
struct StageRec {
    std::vector<WgStage> st;
    bool ok = true;
    int widest = 0;
    WgStage &add(int kind, int nblocks)
    {
        st.emplace_back();
        WgStage &x = st.back();
        memset(static_cast<void *>(&x), 0, sizeof(x));
        x.kind = kind; x.nblocks = nblocks;
        widest = std::max(widest, nblocks);
        return x;
    }
};
// (g_last_launch: the kernel expression of the calling thread's most recent launch as written at the launch site -- an attached timer
// keeps it per recorded launch, wg_timer_read_name, so that a benchmark names the instantiation that RAN)
thread_local const char *g_last_launch = "";
#define WG_LAUNCH(ctx, kern, grid, block, shmem, ...)                         \
    do {                                                                      \
        if ((ctx).rec) (ctx).rec->ok = false;   /* not a recordable launch */ \
        else if ((ctx).probe) { }               /* a route is being asked for, nothing runs */ \
        else if ((ctx).err == 0) {                                            \
            g_last_launch = #kern;                                            \
            hipLaunchKernelGGL(kern, grid, block, shmem, (ctx).st, __VA_ARGS__); \
            if (hipGetLastError() != hipSuccess) (ctx).err = WG_ELAUNCH;      \
        }                                                                     \
    } while (0)

// Diagnostics only: an attached timer brackets every launch of ONE kernel class with HIP events on the launch
// stream (bench.py's roofline leg).  Detached (the default) it costs one relaxed pointer load per launch.
struct KernelTimer {
    int kernel_id, capacity, count;      // kernel_id < 0: every timed kernel class
    hipEvent_t *start, *stop;
    long long *info;                     // per recorded launch: class, M, K, columns, algorithmic HBM bytes (wg_timer_read_info)
    const char **name;                   // per recorded launch: the kernel expression of its launch site (wg_timer_read_name)
};
std::atomic<KernelTimer *> g_timer{nullptr};
std::atomic<long long> g_wgrad16t_launches{0};               // diagnostics: launches of wgrad16t_kernel by this process (wg_stat_wgrad16t_launches)
struct TimerScope {
    KernelTimer *t;
    hipStream_t st;
    int slot;
    // M x K product over `cols` columns, `bytes` = what the launch has to move at least (operand planes once + outputs + weights)
    TimerScope(int id, hipStream_t s, long long M = 0, long long K = 0, long long cols = 0, long long bytes = 0)
        : t(g_timer.load(std::memory_order_relaxed)), st(s), slot(-1)
    {
        if (t && id > -1000 && (t->kernel_id == id || t->kernel_id < 0) && t->count < t->capacity) {
            slot = t->count++;
            long long *q = t->info + 5 * (size_t)slot;
            q[0] = id; q[1] = M; q[2] = K; q[3] = cols; q[4] = bytes;
            (void)hipEventRecord(t->start[slot], st);
        }
    }
    ~TimerScope() { if (slot >= 0) { (void)hipEventRecord(t->stop[slot], st); t->name[slot] = g_last_launch; } }
};

// ------------------------------------------------------------------------------------------------
// geometry
// ------------------------------------------------------------------------------------------------
struct WnD {
    int ic, aux, C, Cd, Cs, depth, radix;
    int prec = 0;
    // WN(bias=True) (model/waveglow.py:58): every conv of the WN has a bias.  Built WITHOUT touching a kernel: a bias is one more K
    // segment -- 32 channels of a plane of ones -- whose weight rows hold the bias (row 0; two rows where two biases meet, the dilated
    // conv's and V's; one row per layer in the all-layers skip product), and in the weight-gradient products one more block of 32
    // B columns whose column 0 is the bias gradient.  The parameter table continues behind `end` with V.bias, start.bias,
    // depth x (W.bias, W_o.bias), end.bias.
    int bias = 0;
    int kb() const { return bias ? 32 : 0; }
    int pb(int j) const { return 4 + 4 * depth + 1 + j; }    // table index of bias j: V, start, (W_i, W_o_i) x depth, end
    // WaveFlow's WN2D (model/waveflow.py:70-135) is this same network with a 3x3 kernel over (height, time): radix 9, tap
    // kt = 3 kh + kw reads plane row r + (kh - 2) * hd[layer] (causal along the height axis) at time shift (kw - 1) * 2^layer, and
    // the conditioning has one plane row per item.  The planes then hold one height row per plane row (Geo::rows).
    int mode2d = 0;
    int hd[16] = {0};
    void tap(int layer, int kt, int &tshift, int &row_off) const
    {
        if (!mode2d) { tshift = (kt - (radix - 1) / 2) * (1 << layer); row_off = 0; }
        else { tshift = (kt % 3 - 1) * (1 << layer); row_off = (kt / 3 - 2) * hd[layer]; }
    }
    int auxp() const { return rup(aux, WG_BK); }
    int wo_rows(int i) const { return i == depth - 1 ? Cs : C + Cs; }
    int nparams() const { return 4 + 4 * depth + 1 + (bias ? 2 + 2 * depth + 1 : 0); }
    int maxdil() const { return 1 << (depth - 1); }
};

// skip = sum_i Wskip_i gate_i as one product at the end of the WN (all gates kept) instead of a read-modify-write of the skip plane
// per layer: the residual convs are bound by their HBM bytes, this takes a third of them away
#if !defined(WG_FUSED_SKIP_MIN_COLS)
#define WG_FUSED_SKIP_MIN_COLS 4096
#endif
// The residual stream h_i and its gradient dh_i exist as S-planes ONLY (hi + lo bf16, ~16 significant bits) in the S-plane mode: the
// fp32 copies were written and read back by every residual / data-gradient conv just to carry the running sum, a quarter of those
// launches' HBM bytes.  The contractions read h and dh from the S-planes either way; what changes is that the rounding to hi + lo
// (2^-17 relative) now accumulates along the 8 layers of a WN instead of being refreshed from an exact fp32 chain.
inline bool s_only_chain(const Ctx &cx, const WnD &d)
{
#if !defined(WG_OPT_NO_S_ONLY) && !defined(WG_OPT_MFMA32) && !defined(WG_OPT_NO_WSPEC) && !defined(WG_OPT_DMA)
    // (only convgemm16q / convgemm16h read the accumulate-into value from an S-plane -- ConvGemm16sArgs::saux --: the superseded
    // kernels of the A/B builds ignore it and would lose the residual term)
#if defined(WG_OPT_S_ONLY_1D)                               // A/B build: WN2D keeps its fp32 residual planes (before round 6's last commits)
    return cx.prec == 2 && !d.mode2d;
#else
    // (WN2D since the end of round 6 -- its 64-row products read the accumulate-into S-plane like every other form of convgemm16q: WaveFlow's
    // step 39.4 -> 36.6 ms on one box, residual conv 43 -> 31 us, data-gradient conv 131 -> 119 us; not inside the row-by-row inverse, whose
    // recorded stages keep the planes they were built and tested with)
    return cx.prec == 2 && (!d.mode2d || (!cx.rec && !cx.row_sel1));      // measured (1-D): step 78.8 -> 75.8 ms; errors against the oracle unchanged
#endif
#else                                                       // (tools/experiments/err_report.py: z 3.8e-6, worst gradient 8.9e-6 of its max)
    (void)cx; (void)d; return false;
#endif
}
// Weight gradients of all layers of a WN in ONE grouped launch after its layer loop (WgradGrp, wg_gemm16s.h) instead of two launches
// per layer.  Needs every layer's gate gradient (the 1-D WN keeps them anyway for the one-product conditioning gradient, fused_dy)
// and every layer's dh, each as its own S-plane: + (depth - 1) planes of 2 Cd and of C channels.
inline bool grouped_wgrad(int prec, const WnD &d)
{
#if defined(WG_OPT_NO_WGRAD_GROUP)
    (void)prec; (void)d; return false;
#else
    return prec == 2 && d.depth >= 2 && d.depth <= WG_GRP_MAX && d.radix + 1 + (d.bias ? 1 : 0) <= WG_MAX_SEG;
#endif
}
inline bool fused_skip(const WnD &d)
{
#if defined(WG_OPT_NO_FUSED_SKIP)
    (void)d; return false;
#else
    return d.depth + (d.bias ? 1 : 0) <= WG_MAX_SEG;
#endif
}

// dy = sum_i V_i^T dxy_i as one product after the layer loop (every layer's dxy kept: +(depth - 1) x 2 Cd planes of workspace) instead of
// an HBM-bound launch and a read-modify-write of dy per layer; 1-D WN only (WaveFlow sums dxy over the height axis first)
inline bool fused_dy(const WnD &d)
{
#if defined(WG_OPT_NO_FUSED_DY)
    (void)d; return false;
#else
    return d.depth <= WG_MAX_SEG && !d.mode2d;
#endif
}


// S-plane mode: a pass that keeps its planes for the gate backward (waveglow.py:13-15) keeps sigmoid only.  The gate itself is kept anyway (its
// S-plane, hi + lo: W_o's operand), so tanh = gate / sigmoid costs the gate backward no byte more (8 bytes per element either way) and every
// gate conv of a recompute pass 4 bytes per element less to write: 49 of the ~200 MB a layer launch stores at the training shape.
inline bool tw_from_gate(const Ctx &cx)
{
#if WG_TS_INTERLEAVED && !defined(WG_OPT_KEEP_TANH)
    return cx.prec == 2 && env_sw().tw_from_gate;
#else
    (void)cx; return false;
#endif
}
// The skip sum S = sum_l Wskip_l gate_l and its gradient dS = W_end^T G never formed (Weff_l = W_end Wskip_l, wg_small.h weff_kernel):
// S-plane mode, 1-D WN without biases (a skip bias would enter `out` through W_end as well), every layer's gate kept (fused_skip).
inline bool lowrank_shape(const WnD &d)
{
#if defined(WG_OPT_NO_LOWRANK)
    (void)d; return false;
#else
    return d.prec == 2 && !d.bias && fused_skip(d) && d.Cd % 64 == 0 && 2 * d.ic * d.Cs <= 8192 && (!d.mode2d || d.ic == 1);                           // (W_end in weff_kernel's LDS)
#endif
}

// ... and the gate convs leave their share of `out` themselves (ConvGemm16sArgs::part): 8 floats per column and (row tile, wave row)
inline bool gate_parts_shape(const WnD &d)
{
#if defined(WG_OPT_NO_GATE_PARTS)
    (void)d; return false;
#else
    return lowrank_shape(d) && 2 * d.ic <= 8 && (2 * d.Cd) % 64 == 0 && (size_t)(d.Cd / 32) * 1024 <= 8192;
#endif
}
inline int gate_part_slots(const WnD &d) { return 2 * d.Cd / 64; }      // one per 32 gate channels = per wave row of a gate conv
inline int gate_part_prow(const WnD &d) { return 2 * d.ic <= 2 ? 2 : 8; }   // floats per partial row (2: WaveFlow's WN2D, the 16x16x32 kernels' epilogue only)
// WN.start folded into the first layer's dilated conv: h_0 = W_start xa has rank ic, so W_0 * h_0 = (W_0[kt] W_start) * xa is a conv over the
// flow's ic channels (one 32-deep chunk per tap) instead of C = 256 (eight): the first layer's gate conv is 6 chunks instead of 27 at the
// shipped shape.  S-plane mode, 1-D WN without biases (a start bias would have to ride through W_0 as well); h_0 itself is still made
// (the residual stream starts from it, the backward's weight gradient reads it).  wg_small.h start_fold_kernel forms the weight.
inline bool wn_pack_fused_shape(const WnD &d);
inline bool start_fold_shape(const WnD &d)
{
#if defined(WG_OPT_NO_START_FOLD)
    (void)d; return false;
#else
    return d.prec == 2 && !d.bias && d.ic <= 16 && d.depth >= 2 && wn_pack_fused_shape(d);
#endif
}

Geo make_geo(int B, int T, int halo_need)
{
    Geo g;
    g.B = B;
    g.T = T;
    g.Tt = rup(T, WG_TILE);
    g.H = std::max(16, rup(halo_need, 16));
    g.P = g.H + g.Tt + g.H;
    g.rows = 0;
    return g;
}

int wn_check(const WnD &d)
{
    if (d.ic < 1 || d.aux < 1 || d.depth < 1 || d.depth > 16) return WG_EINVAL;
    if (d.mode2d ? d.radix != 9 : (d.radix < 1 || !(d.radix & 1) || d.radix > 9)) return WG_EUNSUPPORTED;   // 1-D: odd kernels up to 9 taps; 2-D: 3x3
    if (d.C % 32 || d.Cd % 32 || d.Cs % 32) return WG_EUNSUPPORTED;       // MFMA tile granularity
    if (d.bias && d.radix + 2 > WG_MAX_SEG) return WG_EUNSUPPORTED;      // the ones segment needs a K-segment slot (WN2D: 9 taps + conditioning + ones = 11)
    if (2 * d.ic > WG_MAXC) return WG_EUNSUPPORTED;                       // end-conv rows handled by one MFMA tile
    if (d.C * d.radix > WG_FIN_MAXCOLS || d.aux > WG_FIN_MAXCOLS || d.Cd > WG_FIN_MAXCOLS) return WG_EUNSUPPORTED;
    return 0;
}

// ------------------------------------------------------------------------------------------------
// packed weights of one WN (offsets in floats from the WN's base)
// ------------------------------------------------------------------------------------------------
struct WnPack {
    size_t scale_V, scale_start, scale_W[16], scale_Wo[16];
    size_t startT, startN, endT, endN, bias_end;
    size_t Acat[16], WoT[16], WoN[16], WT[16], VN[16], WskT, VNall;
    size_t effA = 0;                                          // gate_parts_shape: Weff as A fragments for the gate conv's epilogue, Cd / 32 KB per layer
    size_t effT = 0, effN = 0, WoG[16] = {0};                // lowrank_shape: Weff^T [depth Cd][32]; Weff per layer [depth][32][Cd]; [Wres_l^T | Weff_l^T] k-major
    size_t w0x = 0, Acat0x = 0;                              // start_fold_shape: the composed weight [2 Cd][ic][radix]; layer 0's gate matrix over xa: [radix x kp_start | auxp] rows
    int kcat0 = 0;
    int ld_startT, ld_startN, ld_endN, ld_Acat, ld_WoT[16], ld_WoN, ld_WT, ld_VN, ld_WskT;
    int kp_start, kp_end, kcat;
    size_t total;
};

// floats taken by an fp32 k-major matrix plus its split image ([chunk][ld][32] bf16 hi, then lo)
inline size_t mat_floats(int K, int ld) { return rupz((size_t)K * ld, 64) + (size_t)(K / WG16_BK + WG_MAX_SEG + 1) * ld * WG16_BK; }
inline unsigned short *mat_img(const float *A32, int K, int ld) { return (unsigned short *)(A32 + rupz((size_t)K * ld, 64)); }

WnPack wn_pack_layout(const WnD &d)
{
    WnPack L;
    size_t off = 0;
    auto take = [&](size_t n) { size_t o = off; off += rupz(n, 64); return o; };
    L.scale_V = take((size_t)2 * d.Cd * d.depth);
    L.scale_start = take(d.C);
    for (int i = 0; i < d.depth; ++i) { L.scale_W[i] = take(2 * d.Cd); L.scale_Wo[i] = take(d.wo_rows(i)); }
    L.kp_start = rup(d.ic, WG_BK);
    L.kp_end = rup(2 * d.ic, WG_BK);
    L.kcat = d.radix * d.C + d.auxp();
    L.ld_startT = rup(d.C, WG_TILE);
    L.ld_startN = rup(d.ic, WG_TILE);
    L.ld_endN = rup(d.Cs, WG_TILE);
    L.ld_Acat = rup(2 * d.Cd, WG_TILE);
    L.ld_WoN = rup(d.Cd, WG_TILE);
    L.ld_WT = rup(d.C, WG_TILE);
    L.ld_VN = rup(d.aux, WG_TILE);
    // a matrix = fp32 k-major [K][ld] followed by its split bf16 image (mat_img)
    auto take_mat = [&](int K, int ld) { return take(mat_floats(K, ld)); };
    const int kb = d.kb();                                   // rows of the ones segment behind a forward matrix's K rows (WnD::bias)
    L.startT = take_mat(L.kp_start + kb, L.ld_startT);
    L.startN = take_mat(d.C, L.ld_startN);
    L.endT = take((size_t)d.Cs * 32);
    L.endN = take_mat(L.kp_end, L.ld_endN);
    for (int i = 0; i < d.depth; ++i) {
        L.ld_WoT[i] = rup(d.wo_rows(i), WG_TILE);
        L.Acat[i] = take_mat(L.kcat + kb, L.ld_Acat);
        L.WoT[i] = take_mat(d.Cd + kb, L.ld_WoT[i]);
        L.WoN[i] = take_mat(d.wo_rows(i), L.ld_WoN);
        L.WT[i] = take_mat(d.radix * 2 * d.Cd, L.ld_WT);
        L.VN[i] = take_mat(2 * d.Cd, L.ld_VN);
    }
    // the skip rows of every layer's W_o, stacked along K: skip = sum_i Wskip_i gate_i as ONE product over all the gates
    L.ld_WskT = rup(d.Cs, WG_TILE);
    L.WskT = take_mat(d.depth * d.Cd + kb, L.ld_WskT);
    L.bias_end = take(32);
    // V^T of every layer stacked along K: dy = sum_i V_i^T dxy_i as ONE product over all the layers' dxy (fused_dy)
    L.VNall = take_mat(d.depth * 2 * d.Cd, L.ld_VN);
    if (lowrank_shape(d)) {
        L.effT = take((size_t)d.depth * d.Cd * 32);
        L.effN = take((size_t)d.depth * 32 * d.Cd);
        if (gate_parts_shape(d)) L.effA = take((size_t)d.depth * (d.Cd / 32) * 256);
        // the gate backward's weights with the skip rows replaced by Weff_l: K = [C residual rows (none on the last layer) | 32 rows of G]
        // (kp_end rows, as the G plane has them: run_convgemm finds a matrix's image behind K = the segments' channels)
        for (int i = 0; i < d.depth; ++i) L.WoG[i] = take_mat((i == d.depth - 1 ? 0 : d.C) + L.kp_end, L.ld_WoN);
    }
    if (start_fold_shape(d)) {
        L.w0x = take((size_t)2 * d.Cd * d.ic * d.radix);
        L.kcat0 = d.radix * L.kp_start + d.auxp();
        // (chunks: one per tap -- kp_start <= 32 channels each -- and the conditioning's: mat_floats leaves room for K / 32 + WG_MAX_SEG + 1)
        L.Acat0x = take_mat(L.kcat0, L.ld_Acat);
    }
    L.total = off;
    return L;
}

// weight images written by the pack jobs themselves (packimg_kernel) instead of a second pass over fp32 matrices
inline bool wn_pack_fused(const WnD &d);
inline bool wn_pack_fused_shape(const WnD &d) { return wn_pack_fused(d); }
inline bool wn_pack_fused(const WnD &d)
{
#if defined(WG_OPT_NO_PACK_FUSE)
    (void)d; return false;
#else
    return !d.bias;
#endif
}
struct JobBatch {
    Ctx *ctx;
    NormArgs na;
    PackArgs pa;
    JobBatch(Ctx *c) : ctx(c) { na.n = 0; pa.n = 0; }
    void norm(const float *g, const float *v, float *scale, int rows, int cols)
    {
        if (na.n == WG_JOBS) flush_norm();
        NormJob &j = na.job[na.n++];
        j.g = g; j.v = v; j.scale = scale; j.rows = rows; j.cols = cols;
    }
    void pack(float *dst, int ldd, int Kp, int Mp, int mode, int no, int ni, int half, const float *src,
              const float *scale, int so, int si, int off)
    {
        if (pa.n == WG_JOBS) flush_pack();
        PackJob &j = pa.job[pa.n++];
        j.dst = dst; j.src = src; j.scale = scale; j.ldd = ldd; j.Kp = Kp; j.Mp = Mp; j.mode = mode;
        j.no = no; j.ni = ni; j.half = half; j.so = so; j.si = si; j.off = off;
        j.img = nullptr; j.chunk0 = 0; j.nchunks = 0;
    }
    // rows [row, row + Kp) of the k-major matrix A32 ([Ktot][ld] fp32 + its split image, mat_floats): with `fuse` the job writes its
    // chunks of the image itself (packimg_kernel) and the fp32 rows only if `want32`; without, it is a plain pack job (img_kernel follows)
    // (chunk0 / nchunks given: a matrix whose K segments are shorter than a chunk -- every segment starts a chunk of the image, so the
    // image's chunk index is no longer row / 32)
    void pack_m(bool fuse, float *A32, int Ktot, int ld, int row, bool want32, int Kp, int mode, int no, int ni, int half,
                const float *src, const float *scale, int so, int si, int off, int chunk0 = -1, int nchunks = -1)
    {
        pack(A32 + (size_t)row * ld, ld, Kp, ld, mode, no, ni, half, src, scale, so, si, off);
        if (!fuse) return;
        PackJob &j = pa.job[pa.n - 1];
        j.img = mat_img(A32, Ktot, ld); j.chunk0 = chunk0 >= 0 ? chunk0 : row / 32; j.nchunks = nchunks >= 0 ? nchunks : (Ktot + 31) / 32;
        if (!want32) j.dst = nullptr;
        if (chunk0 < 0 && row % 32 && !ctx->err) ctx->err = WG_EINVAL;
    }
    void flush_norm()
    {
        if (!na.n) return;
        int maxrows = 0;
        for (int i = 0; i < na.n; ++i) maxrows = std::max(maxrows, na.job[i].rows);
        WG_LAUNCH(*ctx, rownorm_kernel, dim3((maxrows + 3) / 4, na.n), dim3(256), 0, na);
        na.n = 0;
    }
    void flush_pack()
    {
        if (!pa.n) return;
        WG_LAUNCH(*ctx, packimg_kernel, dim3(512, pa.n), dim3(256), 0, pa);  // grid-stride over each job: 64 blocks left 230 M-parameter models dispatch-starved
        pa.n = 0;
    }
};

struct EffBatch {
    Ctx *ctx;
    EffArgs ea;
    int maxcd = 0;
    size_t lds = 0;
    EffBatch(Ctx *c) : ctx(c) { ea.n = 0; }
    void flush()
    {
        if (!ea.n) return;
        WG_LAUNCH(*ctx, weff_kernel, dim3((maxcd + 63) / 64, ea.n), dim3(256), lds, ea);
        ea.n = 0; maxcd = 0; lds = 0;
    }
    void add(const EffJob &j)
    {
        if (ea.n == WG_EFF_JOBS) flush();
        ea.job[ea.n++] = j;
        maxcd = std::max(maxcd, j.Cd);
        lds = std::max(lds, ((size_t)j.ic2 * j.Cs + 3 * 32 * 64) * sizeof(float));      // (lowrank_shape: 2 ic Cs <= 8192 floats -> at most 56 KB)
    }
};
// Weff of every layer (lowrank_shape): behind the row norms (it reads W_o's scales), in front of the pack jobs (they read effN)
void wn_pack_eff(EffBatch &eb, const WnD &d, const WnPack &L, const float *const *p, float *pk)
{
    if (!lowrank_shape(d)) return;
    for (int i = 0; i < d.depth; ++i) {
        const int r0 = d.wo_rows(i) - d.Cs;
        EffJob j;
        j.wE = p[4 + 4 * d.depth];
        j.v = p[7 + 4 * i] + (size_t)r0 * d.Cd;
        j.scale = pk + L.scale_Wo[i] + r0;
        j.effT = pk + L.effT + (size_t)i * d.Cd * 32;
        j.effN = pk + L.effN + (size_t)i * 32 * d.Cd;
        j.effA = L.effA ? (unsigned short *)(pk + L.effA + (size_t)i * (d.Cd / 32) * 256) : nullptr;
        j.ic2 = 2 * d.ic; j.Cs = d.Cs; j.Cd = d.Cd;
        eb.add(j);
    }
}
struct FoldBatch {
    Ctx *ctx;
    FoldArgs fa;
    int maxe = 0;
    FoldBatch(Ctx *c) : ctx(c) { fa.n = 0; }
    void flush()
    {
        if (!fa.n) return;
        WG_LAUNCH(*ctx, start_fold_kernel, dim3((maxe + 255) / 256, fa.n), dim3(256), 0, fa);
        fa.n = 0; maxe = 0;
    }
    void add(const FoldJob &j)
    {
        if (fa.n == WG_FOLD_JOBS) flush();
        fa.job[fa.n++] = j;
        maxe = std::max(maxe, j.M * j.ic * j.radix);
    }
};
// the composed weight of layer 0 (start_fold_shape): behind the row norms (it reads both factors' scales), in front of the pack jobs
void wn_pack_fold(FoldBatch &fb, const WnD &d, const WnPack &L, const float *const *p, float *pk)
{
    if (!start_fold_shape(d) || !L.Acat0x) return;
    FoldJob j;
    j.vW = p[5]; j.sW = pk + L.scale_W[0];
    j.vS = p[3]; j.sS = pk + L.scale_start;
    j.out = pk + L.w0x;
    j.M = 2 * d.Cd; j.C = d.C; j.ic = d.ic; j.radix = d.radix;
    fb.add(j);
}
// params: WN table (nparams entries).  Two passes over the stream: all row norms, then all packs.
void wn_pack_norms(JobBatch &jb, const WnD &d, const WnPack &L, const float *const *p, float *pk)
{
    jb.norm(p[0], p[1], pk + L.scale_V, 2 * d.Cd * d.depth, d.aux);
    jb.norm(p[2], p[3], pk + L.scale_start, d.C, d.ic);
    for (int i = 0; i < d.depth; ++i) {
        jb.norm(p[4 + 4 * i], p[5 + 4 * i], pk + L.scale_W[i], 2 * d.Cd, d.C * d.radix);
        jb.norm(p[6 + 4 * i], p[7 + 4 * i], pk + L.scale_Wo[i], d.wo_rows(i), d.Cd);
    }
}
void wn_pack_mats(JobBatch &jb, const WnD &d, const WnPack &L, const float *const *p, float *pk, float *ones)
{
    const float *vV = p[1], *vS = p[3], *wE = p[4 + 4 * d.depth];
    // Every matrix the MFMA kernels read is written as its split image by the pack job itself (JobBatch::pack_m); the fp32 k-major
    // copy is kept where something reads it: everywhere outside the S-plane mode, and for the small start / end matrices (wg_thin.h,
    // end_affine_kernel).  With biases (WnD::bias) the plain two-step form stays: the bias rows do not start on chunk boundaries.
    const bool fuse = wn_pack_fused(d), f32 = d.prec != 2;
    const int kb = d.kb();
    // start: src [C][ic]
    jb.pack_m(fuse, pk + L.startT, L.kp_start + kb, L.ld_startT, 0, true, L.kp_start, 0, d.C, d.ic, 0, vS, pk + L.scale_start, d.ic, 1, 0);
    jb.pack_m(fuse, pk + L.startN, d.C, L.ld_startN, 0, true, d.C, 1, d.C, d.ic, 0, vS, pk + L.scale_start, d.ic, 1, 0);
    // end: src [2ic][Cs], plain weight (scale = ones)
    jb.pack(pk + L.endT, 32, d.Cs, 32, 0, 2 * d.ic, d.Cs, 0, wE, ones, d.Cs, 1, 0);
    jb.pack_m(fuse, pk + L.endN, L.kp_end, L.ld_endN, 0, true, L.kp_end, 1, 2 * d.ic, d.Cs, 0, wE, ones, d.Cs, 1, 0);
    for (int i = 0; i < d.depth; ++i) {
        const float *vW = p[5 + 4 * i], *vWo = p[7 + 4 * i];
        const int rows = d.wo_rows(i);
        float *acat = pk + L.Acat[i];
        for (int kt = 0; kt < d.radix; ++kt)   // rows kt*C.. : A[kt*C + c][perm m] = W[o][c][kt]
            jb.pack_m(fuse, acat, L.kcat + kb, L.ld_Acat, kt * d.C, f32, d.C, 0, 2 * d.Cd, d.C, d.Cd, vW,
                      pk + L.scale_W[i], d.C * d.radix, d.radix, kt);
        // conditioning rows: A[radix*C + j][perm m] = V[i*2Cd + o][j]
        jb.pack_m(fuse, acat, L.kcat + kb, L.ld_Acat, d.radix * d.C, f32, d.auxp(), 0, 2 * d.Cd, d.aux, d.Cd,
                  vV + (size_t)i * 2 * d.Cd * d.aux, pk + L.scale_V + (size_t)i * 2 * d.Cd, d.aux, 1, 0);
        if (i == 0 && start_fold_shape(d) && L.Acat0x) {
            // layer 0 over xa: rows kt * kp_start + j = the composed weight's [o][j][kt] (a conv weight with ic input channels: the tap jobs' own
            // form), then the conditioning rows as above; one image chunk per tap, the conditioning's chunks behind them
            float *a0 = pk + L.Acat0x;
            const int nck = d.radix + (d.auxp() + 31) / 32;
            for (int kt = 0; kt < d.radix; ++kt)
                jb.pack_m(fuse, a0, L.kcat0, L.ld_Acat, kt * L.kp_start, false, L.kp_start, 0, 2 * d.Cd, d.ic, d.Cd, pk + L.w0x, ones,
                          d.ic * d.radix, d.radix, kt, kt, nck);
            jb.pack_m(fuse, a0, L.kcat0, L.ld_Acat, d.radix * L.kp_start, false, d.auxp(), 0, 2 * d.Cd, d.aux, d.Cd, vV,
                      pk + L.scale_V, d.aux, 1, 0, d.radix, nck);
        }
        jb.pack_m(fuse, pk + L.WoT[i], d.Cd + kb, L.ld_WoT[i], 0, f32, d.Cd, 0, rows, d.Cd, 0, vWo, pk + L.scale_Wo[i], d.Cd, 1, 0);
        {   // rows i*Cd .. of WskT: A[i*Cd + j][m] = Wo_i[skip row m][j]  (the skip rows follow the C residual rows, except on the last layer)
            const int r0 = rows - d.Cs;
            jb.pack_m(fuse, pk + L.WskT, d.depth * d.Cd + kb, L.ld_WskT, i * d.Cd, f32, d.Cd, 0, d.Cs, d.Cd, 0, vWo + (size_t)r0 * d.Cd,
                      pk + L.scale_Wo[i] + r0, d.Cd, 1, 0);
        }
        jb.pack_m(fuse, pk + L.WoN[i], rows, L.ld_WoN, 0, f32, rows, 1, rows, d.Cd, 0, vWo, pk + L.scale_Wo[i], d.Cd, 1, 0);
        if (lowrank_shape(d)) {                                  // [Wres_i^T | Weff_i^T]: the residual rows as in WoN, then Weff_i's 2 ic rows (of kp_end)
            const int rr = rows - d.Cs;
            if (rr) jb.pack_m(fuse, pk + L.WoG[i], rr + L.kp_end, L.ld_WoN, 0, false, rr, 1, rr, d.Cd, 0, vWo, pk + L.scale_Wo[i], d.Cd, 1, 0);
            jb.pack_m(fuse, pk + L.WoG[i], rr + L.kp_end, L.ld_WoN, rr, false, L.kp_end, 1, 2 * d.ic, d.Cd, 0, pk + L.effN + (size_t)i * 32 * d.Cd, ones,
                      d.Cd, 1, 0);
        }
        for (int kt = 0; kt < d.radix; ++kt)   // A[kt*2Cd + o][c] = W[o][c][kt]
            jb.pack_m(fuse, pk + L.WT[i], d.radix * 2 * d.Cd, L.ld_WT, kt * 2 * d.Cd, f32, 2 * d.Cd, 1, 2 * d.Cd, d.C, 0, vW,
                      pk + L.scale_W[i], d.C * d.radix, d.radix, kt);
        jb.pack_m(fuse, pk + L.VN[i], 2 * d.Cd, L.ld_VN, 0, f32, 2 * d.Cd, 1, 2 * d.Cd, d.aux, 0, vV + (size_t)i * 2 * d.Cd * d.aux,
                  pk + L.scale_V + (size_t)i * 2 * d.Cd, d.aux, 1, 0);
        if (fused_dy(d) || (d.mode2d && d.depth <= WG_MAX_SEG))      // (WN2D: the one product over the layers' height-axis sums, wn_backward)
            jb.pack_m(fuse, pk + L.VNall, d.depth * 2 * d.Cd, L.ld_VN, i * 2 * d.Cd, f32, 2 * d.Cd, 1, 2 * d.Cd, d.aux, 0,
                      vV + (size_t)i * 2 * d.Cd * d.aux, pk + L.scale_V + (size_t)i * 2 * d.Cd, d.aux, 1, 0);
    }
    if (!d.bias) return;
    // WnD::bias: the 32 rows of the ones segment behind each forward matrix.  A bias row is a mode-0 job with ONE k (ni = 1, so = 1):
    // dst[0][m] = bias[perm(m)]; `Kp` rows of the job's region beyond the first are zero filled.  Jobs of one launch run concurrently,
    // so no two of them cover the same row.
    auto bias_rows = [&](float *mat, int ld, int row, int nrows, int nvalid, int half, const float *b) {
        jb.pack(mat + (size_t)row * ld, ld, nrows, ld, 0, nvalid, 1, half, b, ones, 1, 1, 0);
    };
    const float *bV = p[d.pb(0)], *bS = p[d.pb(1)], *bE = p[d.pb(2 + 2 * d.depth)];
    bias_rows(pk + L.startT, L.ld_startT, L.kp_start, 32, d.C, 0, bS);
    for (int i = 0; i < d.depth; ++i) {
        const float *bW = p[d.pb(2 + 2 * i)], *bWo = p[d.pb(3 + 2 * i)];
        const int rows = d.wo_rows(i);
        bias_rows(pk + L.Acat[i], L.ld_Acat, L.kcat, 1, 2 * d.Cd, d.Cd, bW);                              // row 0: the dilated conv's bias
        bias_rows(pk + L.Acat[i], L.ld_Acat, L.kcat + 1, 31, 2 * d.Cd, d.Cd, bV + (size_t)i * 2 * d.Cd);  // row 1: V's slice; rows 2.. zero
        bias_rows(pk + L.WoT[i], L.ld_WoT[i], d.Cd, 32, rows, 0, bWo);
        bias_rows(pk + L.WskT, L.ld_WskT, d.depth * d.Cd + i, 1, d.Cs, 0, bWo + (rows - d.Cs));           // row i: layer i's skip rows
    }
    bias_rows(pk + L.WskT, L.ld_WskT, d.depth * d.Cd + d.depth, 32 - d.depth, 0, 0, ones);                // (zero rows)
    jb.pack(pk + L.bias_end, 32, 1, 32, 1, 1, 2 * d.ic, 0, bE, ones, 2 * d.ic, 1, 0);                     // end.bias, read by end_affine_kernel
}

struct ImgBatch {
    Ctx *ctx;
    ImgArgs ia;
    ImgBatch(Ctx *c) : ctx(c) { ia.n = 0; }
    void add(float *A32, int ld, const int *nch, int nseg)
    {
        if (ia.n == WG_IMG_JOBS) flush();
        ImgJob &j = ia.job[ia.n++];
        int K = 0, nc = 0;
        for (int s = 0; s < nseg; ++s) { j.nch[s] = nch[s]; K += nch[s]; nc += (nch[s] + WG16_BK - 1) / WG16_BK; }
        j.A32 = A32; j.img = mat_img(A32, K, ld); j.lda = ld; j.nseg = nseg; j.nchunks = nc;
    }
    void flush()
    {
        if (!ia.n) return;
        int total = 0;
        for (int i = 0; i < ia.n; ++i) {
            ia.start[i] = total;
            total += ia.job[i].nchunks * ((ia.job[i].lda + WG_IMG_COLS - 1) / WG_IMG_COLS);
        }
        ia.start[ia.n] = total;
        WG_LAUNCH(*ctx, img_kernel, dim3(total), dim3(256), 0, ia);
        ia.n = 0;
    }
};
// split images of every matrix convgemm reads (after the fp32 matrices have been packed)
void wn_pack_images(ImgBatch &ib, const WnD &d, const WnPack &L, float *pk)
{
    if (wn_pack_fused(d)) return;                            // the pack jobs wrote the images themselves (wn_pack_mats)
    int one[WG_MAX_SEG];
    const int nb = d.bias ? 1 : 0;                           // WnD::bias: the ones segment closes the segment list of every forward matrix
    one[1] = 32;
    one[0] = L.kp_start; ib.add(pk + L.startT, L.ld_startT, one, 1 + nb);
    one[0] = d.C;        ib.add(pk + L.startN, L.ld_startN, one, 1);
    one[0] = L.kp_end;   ib.add(pk + L.endN, L.ld_endN, one, 1);
    for (int i = 0; i < d.depth; ++i) {
        int sg[WG_MAX_SEG];
        int ns = 0;
        for (int kt = 0; kt < d.radix; ++kt) sg[ns++] = d.C;
        sg[ns++] = d.auxp();
        if (nb) sg[ns++] = 32;
        ib.add(pk + L.Acat[i], L.ld_Acat, sg, ns);
        one[0] = d.Cd; ib.add(pk + L.WoT[i], L.ld_WoT[i], one, 1 + nb);
        ns = 0;
        if (i < d.depth - 1) sg[ns++] = d.C;
        sg[ns++] = d.Cs;
        ib.add(pk + L.WoN[i], L.ld_WoN, sg, ns);
        ns = 0;
        for (int kt = 0; kt < d.radix; ++kt) sg[ns++] = 2 * d.Cd;
        ib.add(pk + L.WT[i], L.ld_WT, sg, ns);
        one[0] = 2 * d.Cd; ib.add(pk + L.VN[i], L.ld_VN, one, 1);
    }
    if (d.depth + nb <= WG_MAX_SEG) {                        // WskT: one K segment of Cd rows per layer (deeper WNs keep the per-layer skip)
        int sgs[WG_MAX_SEG];
        for (int i = 0; i < d.depth; ++i) sgs[i] = d.Cd;
        if (nb) sgs[d.depth] = 32;
        ib.add(pk + L.WskT, L.ld_WskT, sgs, d.depth + nb);
        for (int i = 0; i < d.depth; ++i) sgs[i] = 2 * d.Cd;
        if (fused_dy(d) || d.mode2d) ib.add(pk + L.VNall, L.ld_VN, sgs, d.depth);
    }
}

// ------------------------------------------------------------------------------------------------
// model-level packed layout
// ------------------------------------------------------------------------------------------------
#define WG_LU_STRIDE (3 * WG_MAXC * WG_MAXC + 64)
#define WG_ONES 8192

int flow_channels(const wg_config *cf, int k)
{
    int c = cf->n_group;
    for (int j = 1; j <= k; ++j)
        if (j % cf->n_early_every == 0) c -= cf->n_early_size;   // waveglow.py:140-142
    return c;
}
WnD flow_wn(const wg_config *cf, int k)
{
    WnD d;
    d.ic = flow_channels(cf, k) / 2;
    d.aux = cf->n_mels; d.C = cf->res_ch; d.Cd = cf->dil_ch; d.Cs = cf->skip_ch; d.depth = cf->depth; d.radix = cf->radix;
    d.prec = cf->precision;
    d.bias = cf->bias ? 1 : 0;
    return d;
}
int wn_table_off(const wg_config *cf, int k) { return 3 + cf->n_flows + k * flow_wn(cf, 0).nparams(); }

struct ModelPack {
    size_t ones, lu, up_scale, up_w, up_bias, wn[WG_MAX_FLOWS];
    size_t total;
};
ModelPack model_pack_layout(const wg_config *cf)
{
    ModelPack L;
    size_t off = 0;
    auto take = [&](size_t n) { size_t o = off; off += rupz(n, 64); return o; };
    L.ones = take(WG_ONES);
    L.lu = take((size_t)cf->n_flows * WG_LU_STRIDE);
    L.up_scale = take(cf->n_mels);
    L.up_w = take((size_t)cf->n_mels * cf->up_kernel);
    L.up_bias = take(cf->n_mels);
    for (int k = 0; k < cf->n_flows; ++k) L.wn[k] = take(wn_pack_layout(flow_wn(cf, k)).total);
    L.total = off;
    return L;
}

int cfg_check(const wg_config *cf)
{
    if (!cf || cf->n_flows < 1 || cf->n_flows > WG_MAX_FLOWS || cf->n_group < 2 || cf->n_group > WG_MAXC) return WG_EINVAL;
    if (cf->n_early_every < 1 || cf->n_early_size < 0 || cf->n_mels < 1) return WG_EINVAL;
    for (int k = 0; k < cf->n_flows; ++k) {
        const int c = flow_channels(cf, k);
        if (c < 2 || (c & 1)) return WG_EINVAL;
        int rc = wn_check(flow_wn(cf, k));
        if (rc) return rc;
    }
    return 0;
}

// ------------------------------------------------------------------------------------------------
// workspace
// ------------------------------------------------------------------------------------------------
struct WgradPlan {
    int nts, t_per_split, b_per_split, nsplit;
};
WgradPlan plan_wgrad(const Geo &g, int tiles)
{
    WgradPlan p;
    const int target = 1024;
    p.nts = 1;
    p.b_per_split = 1;
    const int blocks = tiles * g.B;
    if (blocks < target) {
        p.nts = std::min(std::max(1, g.Tt / 256), (target + blocks - 1) / blocks);
    } else {
        while (tiles * ((g.B + p.b_per_split - 1) / p.b_per_split) > 2 * target && p.b_per_split < g.B) p.b_per_split *= 2;
    }
    p.t_per_split = rup((g.Tt + p.nts - 1) / p.nts, WG_WBK);
    p.nts = (g.Tt + p.t_per_split - 1) / p.t_per_split;
    p.nsplit = ((g.B + p.b_per_split - 1) / p.b_per_split) * p.nts;
    return p;
}
// wgrad16s splits the flattened (batch, 32-step chunk) reduction range evenly over nsplit blocks per tile: ONE round of the
// 512 workgroup slots (2 per CU) whenever the tile count allows it -- 28 tiles x 18 splits = 504 blocks at the headline shape
// instead of 1344 blocks in 2.6 rounds -- and 2.7x fewer slab bytes for the finalize pass to read back.
int plan_wgrad_flat(const Geo &g, int tiles)
{
    const int total = g.B * (g.Tt / WG16_BK);
    return std::max(1, std::min(512 / std::max(1, tiles), total / 4));
}
size_t slab_floats(const Geo &g, int Mp, int Np)
{
    const int tiles = (Mp / WG_TILE) * (Np / WG_TILE);
    const WgradPlan p = plan_wgrad(g, tiles);
    return (size_t)std::max(p.nsplit, plan_wgrad_flat(g, tiles)) * Mp * Np;
}

// ---- wgrad16t_kernel (wg_wgrad16t.h): both grouped products of a WN on exactly one workgroup per CU ----
// The plan: which slots of every XCD work on which product over which part of K, in at most WGT_PH_MAX phases (see the header).
//   phase A: the larger product P0 (T0 tiles of 256 x 128 per layer) as n0 = 32 / T0 sets per XCD = s0 = 8 n0 / ng parts of K per layer;
//            in the slots that leaves, n1 sets of P1 (T1 tiles) covering the first s1 parts of P0's length;
//   phase B: the rest of P1's K range over all slots of every XCD.
// A plan exists when the set counts divide evenly over the 8 XCDs and every part holds at least one chunk; it is used when its length
// is within 20 % of the ideal (total work / 256 CUs).  nslab[i] = slabs per layer of product i.
struct WgtPlan {
    bool ok = false;
    int nph = 0;
    WgtPhase ph[WGT_PH_MAX];
    int nslab[2] = {0, 0};
    int length = 0;                                          // chunks on the longest slot
};
static WgtPlan plan_wgt(int tm0, int tn0, int tm1, int tn1, int ng, int K)
{
    WgtPlan pl;
#if defined(WG_OPT_NO_WGRAD16T)
    return pl;
#endif
    const int T0 = tm0 * tn0, T1 = tm1 * tn1;
    if (T0 < 1 || T1 < 1 || T0 > 32 || T1 > 32 || ng < 1 || K < 1) return pl;
    auto sets_per_xcd = [&](int room, int T) {               // most sets of T tiles in `room` slots whose total over 8 XCDs is a multiple of ng
        int n = room / T;
        while (n > 0 && (8 * n) % ng) --n;
        return n;
    };
    const int n0 = sets_per_xcd(32, T0);
    if (n0 == 0) return pl;
    const int s0 = 8 * n0 / ng;
    if (s0 > K) return pl;
    const int dur = (K + s0 - 1) / s0;
    pl.ph[pl.nph++] = WgtPhase{0, 0, n0 * T0, s0, 0, K, 0, 0, 0, 0, 0, 0, 0};
    pl.nslab[0] = s0;
    const int n1 = sets_per_xcd(32 - n0 * T0, T1);
    int kA1 = 0;
    if (n1 > 0) {
        const int s1 = 8 * n1 / ng;
        kA1 = std::min(K, s1 * dur);
        if (s1 > kA1) return pl;
        pl.ph[pl.nph++] = WgtPhase{1, n0 * T0, n1 * T1, s1, 0, kA1, 0, 0, 0, 0, 0, 0, 0};
        pl.nslab[1] = s1;
    }
    int durB = 0;
    if (kA1 < K) {
        int nB = sets_per_xcd(32, T1);                        // (fewer sets when the rest of K is shorter than that many parts)
        while (nB > 0 && 8 * nB / ng > K - kA1) nB = sets_per_xcd(nB * T1 - 1, T1);
        if (nB == 0) return pl;
        const int sB = 8 * nB / ng;
        durB = (K - kA1 + sB - 1) / sB;
        pl.ph[pl.nph++] = WgtPhase{1, 0, nB * T1, sB, kA1, K, pl.nslab[1], 0, 0, 0, 0, 0, 0};
        pl.nslab[1] += sB;
    }
    pl.length = dur + durB;
    const long work = (long)(T0 + T1) * ng * K;
    pl.ok = (long)pl.length * 256 * 5 <= work * 6 + 8L * 256 * 5;
    return pl;
}
// The general form: ROUNDS.  Product 0's row of tiles is cut into column sub-sets of `tns` tile columns (tm0 * tns <= 32 slots), n0 of
// them per XCD and round; every K range is cut into the same s0 parts, so a round is as long as one part and (layer, sub-set, part)
// items are dealt round after round (WgtPhase::set0).  Product 1's sets (4 tiles at the usual widths) sit in the slots product 0 leaves
// free, in parts no longer than product 0's, or -- where there are none -- in rounds of their own behind it.  Covers what the two-phase
// plan above cannot: rows of tiles wider than an XCD (WSRGlow's conditioning: 35 column tiles), layer counts that do not divide 8.
static WgtPlan plan_wgt_rounds(int tm0, int tn0, int tm1, int tn1, int ng, int K)
{
    WgtPlan best;
#if defined(WG_OPT_NO_WGRAD16T) || defined(WG_OPT_NO_WGT_ROUNDS)
    return best;
#endif
    const int T1 = tm1 * tn1;
    if (tm0 < 1 || tn0 < 1 || T1 < 1 || T1 > 32 || tm0 > 32 || ng < 1 || K < 1) return best;
    const long work = ((long)tm0 * tn0 + T1) * ng * K;
    for (int tns = std::min(tn0, 32 / tm0); tns >= 1; --tns) {
        const int T0 = tm0 * tns, n0 = 32 / T0, nsub = (tn0 + tns - 1) / tns, n1 = (32 - n0 * T0) / T1;
        for (int s0 = 1; s0 <= 8 && s0 <= K; ++s0) {
            const int rounds0 = (ng * nsub * s0 + 8 * n0 - 1) / (8 * n0), len0 = (K + s0 - 1) / s0;
            WgtPlan pl;
            bool fits = rounds0 <= WGT_PH_MAX;
            for (int r = 0; r < rounds0 && fits; ++r)
                pl.ph[pl.nph++] = WgtPhase{0, 0, n0 * T0, s0, 0, K, 0, 0, T0, 0, nsub, tns, r * 8 * n0};
            pl.nslab[0] = s0;
            pl.length = rounds0 * len0;
            int s1 = 0, rounds1 = 0;
            if (fits && n1 >= 1) {                            // beside product 0: parts no longer than its parts
                s1 = (K + len0 - 1) / len0;
                rounds1 = (ng * s1 + 8 * n1 - 1) / (8 * n1);
                if (s1 > 8 || rounds1 > rounds0 || pl.nph + rounds1 > WGT_PH_MAX) s1 = 0;
                for (int r = 0; r < rounds1 && s1; ++r)
                    pl.ph[pl.nph++] = WgtPhase{1, n0 * T0, n1 * T1, s1, 0, K, 0, 0, T1, 0, 1, tn1, r * 8 * n1};
            }
            if (fits && !s1) {                                // rounds of its own, all 32 slots of an XCD
                const int nf = 32 / T1;
                int bs = 0, bl = 0, br = 0;
                for (int q = 1; q <= 8 && q <= K; ++q) {
                    const int rr = (ng * q + 8 * nf - 1) / (8 * nf), ll = rr * ((K + q - 1) / q);
                    if (!bs || ll < bl) { bs = q; bl = ll; br = rr; }
                }
                s1 = bs; rounds1 = br;
                if (pl.nph + rounds1 > WGT_PH_MAX) fits = false;
                for (int r = 0; r < rounds1 && fits; ++r)
                    pl.ph[pl.nph++] = WgtPhase{1, 0, nf * T1, s1, 0, K, 0, 0, T1, 0, 1, tn1, r * 8 * nf};
                pl.length += bl;
            }
            if (!fits) continue;
            pl.nslab[1] = s1;
            pl.ok = (long)pl.length * 256 * 5 <= work * 6 + 8L * 256 * 5;
            if (pl.ok && (!best.ok || pl.length < best.length || (pl.length == best.length && pl.nph < best.nph))) best = pl;
        }
    }
    return best;
}
int device_cus();
static WgtPlan plan_wgt_any(int tm0, int tn0, int tm1, int tn1, int ng, int K)
{
    // the plans (and wgrad16t_kernel's slot -> XCD arithmetic) are written for 8 XCDs of 32 CUs: any other part takes the grouped kernel
    if (device_cus() != 256) return WgtPlan{};
    const WgtPlan a = plan_wgt(tm0, tn0, tm1, tn1, ng, K);
    return a.ok ? a : plan_wgt_rounds(tm0, tn0, tm1, tn1, ng, K);
}
struct WnWs {               // plane bases (float offsets) of one WN's activations
    size_t H[16], tw[16], sf[16], gate[16], skip, G, dS, dH, dxy, slab;
    size_t HS[16], gateS[16], XaS, GS, dSS, dHS, dxyS;   // S-planes (precision 2), sized like the fp32 plane of the same tensor
    size_t dxy_step = 0, dxyS_step = 0;                  // fused_dy: layer i's dxy at dxy + i * step (0: one buffer for all layers)
    size_t dHS_step = 0;                                 // grouped_wgrad: dh_i at dHS + i * step (0: accumulated in place in one plane)
    size_t ones = 0, onesS = 0;                          // WnD::bias: 32 channels of ones on [0, T) (fp32 plane, S-plane), filled by every WN pass
    size_t gpart = 0, gpart_step = 0;                    // lowrank_shape, 2 ic <= 8: the gate convs' partial rows of `out`, [depth][slots][B][Tt][8] fp32
    size_t lsync = 0;                                    // precision 2: the one-launch layer's hand-off counters (wg_layer16h.h), WGL_SYNC_WORDS words, zero between launches
    int nH;                 // 2 (ping-pong) or depth
    size_t slab_floats;
};
#define WGL_SYNC_WORDS 16384     // hand-off counters of the one-launch layer (wg_layer16h.h): one 128-byte line per column tile
struct Bump {
    size_t off = 0;
    size_t take(size_t n) { size_t o = off; off += rupz(n, 64); return o; }
};
int device_cus();
std::atomic<long long> g_gate_split_launches{0};             // diagnostics (wg_stat_gate_split_launches)
std::atomic<long long> g_gate_part_launches{0};              // diagnostics (wg_stat_gate_part_launches)
// A gate conv cut along K (run_convgemm's split path: convgemm16g_kernel<WGG_EPI_PART> + gate_finish16g_kernel): nt = output tiles of
// 256 x 192, S = parts per tile (0: the shape does not qualify) -- the tiles fill 1 / S of the CUs, the column tiles are a multiple of
// 8 (XCD placement), every part holds at least 8 and at most WGG_MAXCHUNKS chunks.  ONE function for the launch site and for the
// workspace layout, so that a workspace sized for the cut always has the room the launch asks for.
static void gate_split_plan(int cols_padded, int M, int nc, int cus, int &nt, int &S)
{
    const int nct = (cols_padded + WGG_BN - 1) / WGG_BN, nrb = M / WGG_BM;
    nt = nct * nrb; S = 0;
    if (M % WGG_BM || cus % 8 || nt <= 0 || nct % 8) return;
    const int s = cus / nt;
    if (s >= 2 && s <= 8 && nt * s == cus && nc / s >= 8 && (nc + s - 1) / s <= WGG_MAXCHUNKS) S = s;
}
// floats of scratch a WN's gate conv needs when it is cut (the parts' raw accumulators: 8 waves x 24 blocks x 256 floats per part)
static size_t gate_split_floats(const WnD &d, const Geo &g)
{
    if (g.rows != 0 || g.H < 64) return 0;
    const int nc = d.radix * ((d.C + WG16_BK - 1) / WG16_BK) + (d.aux + WG16_BK - 1) / WG16_BK + (d.bias ? 1 : 0);
    int nt, S;
    gate_split_plan(g.B * g.Tt, 2 * d.Cd, nc, device_cus(), nt, S);
    return S ? (size_t)S * nt * 8 * 24 * 256 : 0;
}
void wn_ws_layout(Bump &bp, const WnD &d, int ic_max, const Geo &g, int mode, int prec, WnWs &w)
{
    const size_t pC = (size_t)g.B * d.C * g.P, pD = (size_t)g.B * d.Cd * g.P, pS = (size_t)g.B * d.Cs * g.P;
    if (prec == 2) {
        const int nHS = mode ? d.depth : 2;
        for (int i = 0; i < nHS; ++i) w.HS[i] = bp.take(pC);
        for (int i = 0; i < d.depth; ++i) w.gateS[i] = (mode || i == 0 || fused_skip(d)) ? bp.take(pD) : w.gateS[0];
        w.XaS = bp.take((size_t)g.B * rup(ic_max, WG_BK) * g.P);
        w.lsync = bp.take(WGL_SYNC_WORDS);
        if (gate_parts_shape(d) && (g.rows == 0 || d.mode2d)) {
            w.gpart_step = rupz((size_t)gate_part_slots(d) * g.B * g.Tt * gate_part_prow(d), 64);
            w.gpart = bp.take(w.gpart_step * d.depth);
        }
        if (mode) {
            w.GS = bp.take((size_t)g.B * rup(2 * ic_max, WG_BK) * g.P);
            w.dSS = bp.take(pS);
            w.dHS = bp.take(pC);
            w.dHS_step = grouped_wgrad(prec, d) ? rupz(pC, 64) : 0;
            for (int i = 1; i < d.depth && w.dHS_step; ++i) (void)bp.take(pC);
            w.dxyS_step = (fused_dy(d) || grouped_wgrad(prec, d)) ? rupz(2 * pD, 64) : 0;
            w.dxyS = bp.take(2 * pD);
            for (int i = 1; i < d.depth && w.dxyS_step; ++i) (void)bp.take(2 * pD);
        }
    }
    if (d.bias) {
        w.ones = bp.take((size_t)g.B * 32 * g.P);
        w.onesS = bp.take((size_t)g.B * 32 * g.P);
    }
    w.nH = mode ? d.depth : 2;
    for (int i = 0; i < w.nH; ++i) w.H[i] = bp.take(pC);
    for (int i = 0; i < d.depth; ++i) {
        w.gate[i] = (mode || i == 0 || (fused_skip(d) && prec != 2)) ? bp.take(pD) : w.gate[0];
        w.tw[i] = mode ? bp.take(pD) : 0;
        w.sf[i] = mode ? bp.take(pD) : 0;
    }
    w.skip = bp.take(pS);
    w.slab_floats = 0;
    if (mode) {
        w.G = bp.take((size_t)g.B * rup(2 * ic_max, WG_BK) * g.P);
        w.dS = bp.take(pS);
        w.dH = bp.take(pC);
        w.dxy_step = (fused_dy(d) && prec != 2) ? rupz(2 * pD, 64) : 0;
        w.dxy = bp.take(2 * pD);
        for (int i = 1; i < d.depth && w.dxy_step; ++i) (void)bp.take(2 * pD);
        size_t s = 0;
        const int kb = d.kb();                                   // (the ones segment: 32 more B columns in every weight-gradient product)
        const int nW = rup(d.radix * rup(d.C, 32) + rup(d.aux, 32) + kb, WG_TILE), nO = rup(rup(d.Cd, 32) + kb, WG_TILE);
        s = std::max(s, slab_floats(g, rup(2 * d.Cd, WG_TILE), nW));
        s = std::max(s, slab_floats(g, rup(d.C + d.Cs, WG_TILE), nO));
        s = std::max(s, slab_floats(g, WG_TILE, rup(rup(d.Cs, 32) + kb, WG_TILE)));
        s = std::max(s, slab_floats(g, rup(d.C, WG_TILE), WG_TILE));
        s = std::max(s, slab_floats(g, WG_TILE, WG_TILE));
        s = std::max(s, wgth_part_floats(2 * device_cus(), std::max(16 * d.C, 32 * d.Cs)) + 64);      // the thin products' partials (wg_thin.h)
        // room for several products' slabs (FinQueue batches the finalisations of a WN): up to 8 of the largest, at most 384 MB
        s = std::max(s, std::min((size_t)8 * s, (size_t)96 << 20));
        if (grouped_wgrad(prec, d)) {                            // both grouped products of a WN at once (run_wgrad_group_pair)
            const size_t oneT = (size_t)rup(2 * d.Cd, WG_TILE) * nW, oneO = (size_t)rup(d.C + d.Cs, WG_TILE) * nO;
            const int nsT = plan_wgrad_flat(g, (int)(oneT / (WG_TILE * WG_TILE)) * d.depth);
            const int nsO = plan_wgrad_flat(g, (int)(oneO / (WG_TILE * WG_TILE)) * d.depth);
            s = std::max(s, rupz((size_t)nsT * d.depth * oneT, 64) + rupz((size_t)nsO * d.depth * oneO, 64) + 4096
                            + rupz((size_t)(nsT + nsO) * d.depth * WG_SYNC_STRIDE, 64));                     // (+ the lock-step counters)
            if (g.rows == 0 && oneT / nW % 256 == 0 && rup(d.C + d.Cs, WG_TILE) % 256 == 0) {              // wgrad16t_kernel's own split
                const WgtPlan pl = plan_wgt_any((int)(oneT / nW) / 256, nW / WG_TILE, rup(d.C + d.Cs, WG_TILE) / 256, nO / WG_TILE,
                                            d.depth, g.B * (g.Tt / WG16_BK));
                // (+ the thin products' partials of the same backward: the start product reserves after the grouped slabs, and a wrap of the
                // arena there means a second finalisation launch per WN for one small job)
                if (pl.ok) s = std::max(s, rupz((size_t)pl.nslab[0] * d.depth * oneT, 64) + rupz((size_t)pl.nslab[1] * d.depth * oneO, 64) + 4096
                                            + 2 * (wgth_part_floats(2 * device_cus(), std::max(16 * d.C, 32 * d.Cs)) + 64));
            }
        }
        if (prec == 2) s = std::max(s, gate_split_floats(d, g));     // (the split gate conv's parts use the idle slab during a forward pass)
        w.slab_floats = s;
        w.slab = bp.take(s);
    } else if (prec == 2) {
        // forward / inverse workspaces get the split gate conv's scratch too: the forward of a training step (mode 1) and a plain
        // forward then run the same arithmetic, K parts included
        const size_t s = gate_split_floats(d, g);
        if (s) { w.slab_floats = s; w.slab = bp.take(s); }
    }
}

// Stored-activation mode (wg_config.keep_activations; the reference's memory_efficient=False, efficient_modules.py:33-35,71-75):
// the planes of one WN that its backward reads -- everything else (fp32 ping-pong at precision 2, gradients, slab) stays shared.
void wn_ws_layout_kept(Bump &bp, const WnD &d, int ic_max, const Geo &g, int prec, WnWs &w)
{
    const size_t pC = (size_t)g.B * d.C * g.P, pD = (size_t)g.B * d.Cd * g.P, pS = (size_t)g.B * d.Cs * g.P;
    if (prec == 2) {
        for (int i = 0; i < d.depth; ++i) { w.HS[i] = bp.take(pC); w.gateS[i] = bp.take(pD); }
        w.XaS = bp.take((size_t)g.B * rup(ic_max, WG_BK) * g.P);
        if (gate_parts_shape(d) && g.rows == 0) {                // (every flow keeps its own: the backward's end conv reads them again)
            w.gpart_step = rupz((size_t)gate_part_slots(d) * g.B * g.Tt * gate_part_prow(d), 64);
            w.gpart = bp.take(w.gpart_step * d.depth);
        }
    } else {
        for (int i = 0; i < d.depth; ++i) { w.H[i] = bp.take(pC); w.gate[i] = bp.take(pD); }
    }
    for (int i = 0; i < d.depth; ++i) { w.tw[i] = bp.take(pD); w.sf[i] = bp.take(pD); }
    w.skip = bp.take(pS);
}

struct ModelWs {
    Geo g;
    int Gp, auxp, ntile;
    size_t X, dX, Y, dY, YS, partial, total;
    WnWs wn;
    std::vector<WnWs> wnk;     // keep_activations: per-flow view (shared scratch + that flow's kept planes); empty otherwise
    const WnWs &flow(int k) const { return wnk.empty() ? wn : wnk[k]; }
};
ModelWs model_ws_layout(const wg_config *cf, int B, int T, int mode)
{
    ModelWs w;
    const WnD d0 = flow_wn(cf, 0);
    w.g = make_geo(B, T, d0.maxdil() * (d0.radix - 1) / 2);
    w.Gp = rup(cf->n_group, WG_BK) + WG_BK;
    w.auxp = d0.auxp();
    w.ntile = w.g.Tt / WG_AFF_T;          // partial log_s sums: one per end_affine workgroup
    Bump bp;
    w.X = bp.take((size_t)B * w.Gp * w.g.P);
    w.Y = bp.take((size_t)B * w.auxp * w.g.P);
    w.partial = bp.take((size_t)cf->n_flows * B * w.ntile);
    w.dX = w.dY = 0;
    if (mode) {
        w.dX = bp.take((size_t)B * w.Gp * w.g.P);
        w.dY = bp.take((size_t)B * w.auxp * w.g.P);
    }
    w.YS = cf->precision == 2 ? bp.take((size_t)B * w.auxp * w.g.P) : 0;
    wn_ws_layout(bp, d0, cf->n_group / 2, w.g, mode, cf->precision, w.wn);
    if (mode && cf->keep_activations) {
        w.wnk.assign(cf->n_flows, w.wn);                       // flow 0 owns the planes laid out above
        for (int k = 1; k < cf->n_flows; ++k) wn_ws_layout_kept(bp, d0, cf->n_group / 2, w.g, cf->precision, w.wnk[k]);
    }
    w.total = bp.off + 4096;   // slack
    return w;
}

// ------------------------------------------------------------------------------------------------
// kernel drivers
// ------------------------------------------------------------------------------------------------
PRef pref(float *p, int Cp, int ch0 = 0) { PRef r; r.p = p; r.Cp = Cp; r.ch0 = ch0; return r; }
PRef pnull() { PRef r; r.p = nullptr; r.Cp = 0; r.ch0 = 0; return r; }

struct SegSpec {
    const float *src;
    int Cp, ch0, nch, shift;
    const float *s;      // S-plane base of the same tensor (precision 2), its rows per item and first row
    int sCp, sch0;
    int row_off = 0, per_item = 0;   // Geo::rows > 0 only: see SSeg
};
SRef sref(const Geo &g, const float *base, int Cp, int ch0 = 0)
{
    SRef r;
    r.hi = (unsigned short *)base; r.lo_off = (size_t)g.B * Cp * g.P; r.Cp = Cp; r.ch0 = ch0;
    return r;
}
SRef snull() { SRef r; r.hi = nullptr; r.lo_off = 0; r.Cp = 8; r.ch0 = 0; return r; }
void run_to_splane(Ctx &cx, const Geo &g, PRef src, int nvalid, const float *dst, int Cp_dst)
{
    if (cx.rec) {                                            // recorded for the row walk: only the current height row of every item is converted
        if (!cx.row_sel1 || g.rows <= 0) { cx.rec->ok = false; return; }
        const int items = g.B / g.rows, tb = (g.T + WGS_THREADS - 1) / WGS_THREADS;
        WgStage &x = cx.rec->add(WGS_TOSPLANE, tb * (Cp_dst / 8) * items);
        x.u.tsp.src = src; x.u.tsp.dst = sref(g, dst, Cp_dst); x.u.tsp.g = g;
        x.u.tsp.nvalid = nvalid; x.u.tsp.cgs = Cp_dst / 8; x.u.tsp.items = items;
        return;
    }
    WG_LAUNCH(cx, to_splane_kernel, dim3((g.T + 255) / 256, Cp_dst / 8, g.B), dim3(256), 0, src, nvalid, sref(g, dst, Cp_dst), g);
}

// CUs of the current device (256 on MI355X), asked once per device
int device_cus()
{
    static std::atomic<int> cache[16];
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) return 256;
    int n = cache[dev].load(std::memory_order_relaxed);
    if (n > 0) return n;
    if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 256;
    cache[dev].store(n, std::memory_order_relaxed);
    return n;
}

// Dynamic LDS beyond the 48 KB a kernel gets by default: the opt-in is set ONCE per (device, kernel) -- `slot` names the kernel -- and a
// device that refuses it (this library is written for gfx950's 160 KB; other CDNA parts have 64 KB) makes the call fail with
// WG_EUNSUPPORTED instead of a launch error further down.
int ensure_dynamic_lds(const void *kernel, int slot, size_t bytes)
{
    static std::atomic<int> granted[16][8];
    if (bytes <= 48 * 1024) return 0;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16 || slot < 0 || slot >= 8) return WG_ELAUNCH;
    if (granted[dev][slot].load(std::memory_order_relaxed) >= (int)bytes) return 0;
    if (hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes) != hipSuccess) {
        (void)hipGetLastError();
        return WG_EUNSUPPORTED;
    }
    granted[dev][slot].store((int)bytes, std::memory_order_relaxed);
    return 0;
}

// the 32-bit byte offsets of convgemm16g_kernel: every operand's hi + lo arrays and the weight image's hi + lo halves span less than 4 GB
static bool g192_fits(const ConvGemm16sArgs &as, size_t img_stride, int nseg)
{
    if (img_stride * 4 >= (1ull << 32)) return false;
    for (int s = 0; s < nseg; ++s)
        if (as.sseg[s].lo_off * 4 >= (1ull << 32)) return false;
    return true;
}

// env WG_G192=0: the conv products never take the 256 x 192-tile kernel of wg_gemm16g.h (A/B runs in one build; EnvSw)
static bool g192_on() { return env_sw().g192; }

void run_convgemm(Ctx &cx, const Geo &g, const float *A, int lda, int M, const SegSpec *segs, int nseg, int epi,
                  PRef out0, PRef out1, PRef out2, PRef aux0, PRef aux1, int nsplit, int accumulate, SRef s0 = snull(), SRef saux = snull())
{
    ConvGemmArgs a;
    memset(&a, 0, sizeof(a));
    a.A = A; a.lda = lda; a.M = M; a.nseg = nseg;
    for (int s = 0; s < nseg; ++s) {
        a.seg[s].src = segs[s].src; a.seg[s].Cp = segs[s].Cp; a.seg[s].ch0 = segs[s].ch0;
        a.seg[s].nch = segs[s].nch; a.seg[s].shift = segs[s].shift;
        a.seg[s].row_off = segs[s].row_off; a.seg[s].per_item = segs[s].per_item;
    }
    a.g = g; a.epi = epi; a.nsplit = nsplit; a.accumulate = accumulate;
    a.out0 = out0; a.out1 = out1; a.out2 = out2; a.aux0 = aux0; a.aux1 = aux1;
    const int mrows = epi == EPI_GATE ? M : M;
    dim3 grid(g.Tt / WG_TILE, rup(mrows, WG_TILE) / WG_TILE, g.B), block(256);
    if (cx.row_sel1) {
        if (g.rows <= 0) { if (!cx.err) cx.err = WG_EINVAL; return; }
        a.row_sel1 = cx.row_sel1;
        grid.z = g.B / g.rows;
    }
    // what this launch has to move at least: every distinct source plane once (hi + lo bf16, or fp32), the weights, the outputs and
    // auxiliary planes of its epilogue (fp32 planes 4 B, S-planes 2 + 2 B per element)
    long long Ksum = 0, in_ch = 0;
    for (int s = 0; s < nseg; ++s) {
        Ksum += segs[s].nch;
        bool seen = false;
        for (int u = 0; u < s; ++u) seen = seen || (segs[u].src == segs[s].src && segs[u].s == segs[s].s && segs[u].ch0 == segs[s].ch0);
        if (!seen) in_ch += segs[s].nch;
    }
    const long long cols = (long long)(cx.row_sel1 && g.rows > 0 ? g.B / g.rows : g.B) * g.T;
    long long out_ch = 0;
    if (epi == EPI_GATE) out_ch = (long long)(M / 2) * ((out0.p ? 1 : 0) + (out1.p ? 2 : 0) + (s0.hi ? 1 : 0));
    else if (epi == EPI_DGATE) out_ch = (long long)M * 2 /* tanh, sigmoid in */ + 2LL * M * ((s0.hi ? 1 : 0) + (out0.p ? 1 : 0));
    else if (epi == EPI_RESSKIP) out_ch = (long long)nsplit * (1 + (out0.p ? 1 : 0) + (s0.hi ? 1 : 0)) + (long long)(M - nsplit) * (1 + accumulate);
    else out_ch = (long long)M * ((out0.p ? 1 : 0) + (s0.hi ? 1 : 0) + (aux0.p ? 1 : 0) + (saux.hi ? 1 : 0));
    const long long alg_bytes = 4 * cols * (in_ch + out_ch) + 4LL * M * Ksum;
    if (cx.cap && cx.prec != 2) { cx.cap->ok = false; return; }
    if (cx.probe) *cx.probe = 0;
    TimerScope ts((cx.rec || cx.cap || cx.probe) ? -1000 : WG_K_CONV_STORE + epi, cx.st, M, Ksum, cols, alg_bytes);      // (nothing is launched while recording)
    if (cx.prec) {
        ConvGemm16Args a16;
        int K = 0, nc = 0;
        for (int s = 0; s < nseg; ++s) { K += segs[s].nch; nc += (segs[s].nch + WG16_BK - 1) / WG16_BK; }
        a16.img = mat_img(A, K, lda);
        a16.img_stride = (size_t)nc * lda * WG16_BK;
        a16.c = a;
        if (cx.prec == 2) {
            ConvGemm16sArgs as;
            as.ntx = as.nty = as.ntz = 0; as.xcd_items = 0; as.tap_il = 0; as.tap_chunks = 0;
#if !defined(WG_OPT_NO_TAP_IL)
            {   // leading segments that are taps of one plane: walked interleaved (ConvGemm16sArgs::tap_il)
                int nt = 1;
                while (nt < nseg && segs[nt].s == segs[0].s && segs[nt].nch == segs[0].nch && segs[nt].sCp == segs[0].sCp &&
                       segs[nt].sch0 == segs[0].sch0 && segs[nt].per_item == segs[0].per_item) ++nt;
                if (nt >= 2 && segs[0].s && segs[0].nch % WG16_BK == 0) { as.tap_il = nt; as.tap_chunks = segs[0].nch / WG16_BK; }
            }
#endif
            as.img = a16.img; as.img_stride = a16.img_stride; as.c = a; as.s0 = s0; as.saux = saux;
            as.eff = nullptr; as.part = nullptr; as.prow = cx.gate_prow;
            if (epi == EPI_GATE && cx.gate_part && (size_t)(M / 64) * 1024 <= WGG_EFF_BYTES) { as.eff = cx.gate_eff; as.part = cx.gate_part; }
            for (int s = 0; s < nseg; ++s) {
                as.sseg[s].hi = (const unsigned short *)segs[s].s;
                as.sseg[s].lo_off = (size_t)(segs[s].per_item ? g.B / g.rows : g.B) * segs[s].sCp * g.P;
                as.sseg[s].Cp = segs[s].sCp; as.sseg[s].ch0 = segs[s].sch0;
                as.sseg[s].row_off = segs[s].row_off; as.sseg[s].per_item = segs[s].per_item;
                if (!segs[s].s && !cx.err) cx.err = WG_EINVAL;
                if ((segs[s].row_off || segs[s].per_item) && g.rows <= 0 && !cx.err) cx.err = WG_EINVAL;
            }
#if defined(WG_OPT_DMA)                           // LDS-DMA loader ring (6 waves per workgroup)
            switch (epi) {
            case EPI_STORE: WG_LAUNCH(cx, convgemm16d_kernel<EPI_STORE>, grid, dim3(384), 0, as); break;
            case EPI_GATE: WG_LAUNCH(cx, convgemm16d_kernel<EPI_GATE>, grid, dim3(384), 0, as); break;
            case EPI_RESSKIP: WG_LAUNCH(cx, convgemm16d_kernel<EPI_RESSKIP>, grid, dim3(384), 0, as); break;
            case EPI_DGATE: WG_LAUNCH(cx, convgemm16d_kernel<EPI_DGATE>, grid, dim3(384), 0, as); break;
            }
            return;
#elif !defined(WG_OPT_NO_WSPEC)                   // default: loader waves + compute waves (8 waves per workgroup)
            // persistent launch: one workgroup per resident slot (two per CU), each walking its share of the tile grid -- see the
            // kernel; the gate backward stays at one workgroup per tile.  -DWG_OPT_NO_PERSIST: one workgroup per tile everywhere.
            const int cus = device_cus();
            bool small = grid.x * grid.y * grid.z < 384;         // fewer 128x128 tiles than 3/4 of the workgroup slots: 128x64 tiles
#if defined(WG_OPT_NI1_MASK)                      // experiment: 128 x 64 tiles (twice the tiles: a fuller last round) for a class of launches
            if ((WG_OPT_NI1_MASK & 1) && epi == EPI_STORE && Ksum <= 256) small = true;
            if ((WG_OPT_NI1_MASK & 2) && epi == EPI_DGATE) small = true;
            if ((WG_OPT_NI1_MASK & 4) && epi == EPI_STORE && Ksum >= 1024 && Ksum < 2048) small = true;
            if ((WG_OPT_NI1_MASK & 8) && epi == EPI_STORE && Ksum >= 2048) small = true;
#endif
            as.ntx = small ? (int)grid.x * 2 : (int)grid.x; as.nty = (int)grid.y; as.ntz = (int)grid.z;
            const int ntiles = as.ntx * as.nty * as.ntz;
            int slots = epi == EPI_DGATE ? ntiles : 2 * cus;
#if defined(WG_OPT_NO_PERSIST)
            slots = ntiles;
#endif
            const dim3 gp = epi == EPI_DGATE ? dim3(as.ntx, as.nty, as.ntz) : dim3(std::min(ntiles, slots));
            // plane rows dealt to XCDs (ConvGemm16sArgs::xcd_items): full persistent grids whose plane rows divide by the 8 XCDs
            const bool xcd_rows = epi != EPI_DGATE && as.ntz % 8 == 0 && g.rows == 0
#if defined(WG_OPT_NO_XCD_ROWS)
                                  && false
#endif
                ;
            if (xcd_rows && ntiles >= slots && slots % 8 == 0) as.xcd_items = as.ntz / 8;
#if !defined(WG_OPT_NO_XCD_COLS)
            // XCD columns (ConvGemm16sArgs::xcd_items < 0): full persistent grids whose plane rows do NOT divide by the 8 XCDs
            // (measured, WSRGlow at batch 12: the conditioning gradient -- 29 row tiles -- 516.9 -> 478.6 us; launches with few row tiles do not
            // gain -- the gate conv, 4 row tiles: 107.6 -> 109.3 us -- so the order is used from 8 row tiles on)
            else if (epi != EPI_DGATE && g.rows == 0 && !cx.row_sel1 && ntiles >= slots && slots % 8 == 0 && as.ntx * as.ntz >= 8 && as.nty >= 8) as.xcd_items = -1;
#endif
            if (cx.cap) {                                     // describe, do not launch: the 256 x 128-tile form with S-plane-only epilogues
                const bool sg = epi == EPI_GATE && as.s0.hi && !a.out0.p && WG_TS_INTERLEAVED;
                const bool se = epi == EPI_STORE && as.s0.hi && !a.out0.p && !a.aux0.p;
                cx.cap->ok = !small && rup(mrows, WG_TILE) % 256 == 0 && (sg || se) && !cx.row_sel1 && g.rows == 0 && !cx.rec;
                if (cx.cap->ok) {
                    as.nty = (int)grid.y / 2;
                    as.xcd_items = as.ntz % 8 == 0 ? as.ntz / 8 : 0;
                    cx.cap->as = as;
                }
                return;
            }
#if !defined(WG_OPT_MFMA32)                       // default: the 16x16x32 form of the same kernel (wg_gemm16q.h)
#if !defined(WG_OPT_NO_HTILE)
            // at most half as many 128 x 64 tiles as CUs (one utterance being synthesised, WaveFlow's row-by-row inverse): such a launch
            // is as long as its slowest CU needs to take in its operands -- 64 x 64 tiles, twice the workgroups (wg_gemm16h.h)
            // (measured and not adopted: also where the 128 x 64 tiles are 1 - 1.5 per CU -- WSRGlow's gate conv, 384 such tiles --
            // three 64 x 64 tiles on every CU instead: 52.9 against 50.5 ms per WSRGlow step)
            if (small && 2 * ntiles <= cus && epi != EPI_DGATE) {
                as.nty = 2 * (int)grid.y;
                const dim3 gh(as.ntx * as.nty * as.ntz);
                if (cx.rec) {                                 // (the stage interpreter runs convgemm16h_body on this argument block)
                    cx.rec->add(epi == EPI_STORE ? WGS_CONV_STORE : epi == EPI_GATE ? WGS_CONV_GATE : WGS_CONV_RESSKIP, (int)gh.x).u.conv = as;
                    return;
                }
                // (measured slower in round 5: the same tiles with the operands by LDS-DMA into a ring of eight chunk buffers, seven chunks in
                // flight, four waves that multiply and issue -- git show 4c099e9:tools/experiments/wg_gemm16m.h: 2.70 against 2.46-2.52 ms per 0.7 s utterance,
                // WaveFlow's row-by-row synthesis 96.6 against 86.7 ms: a CU's intake rate, not the depth of its prefetch, bounds these launches)
                // (2 or 4 k-steps per chunk and barrier measured slower: git show 4c099e9:tools/experiments/wg_gemm16hk.h)
                switch (epi) {
                case EPI_STORE: WG_LAUNCH(cx, convgemm16h_kernel<EPI_STORE>, gh, dim3(512), 0, as); break;
                case EPI_GATE: WG_LAUNCH(cx, convgemm16h_kernel<EPI_GATE>, gh, dim3(512), 0, as); break;
                case EPI_RESSKIP: WG_LAUNCH(cx, convgemm16h_kernel<EPI_RESSKIP>, gh, dim3(512), 0, as); break;
                }
                return;
            }
#endif
            // S-plane-only stores (no fp32 output, no fp32 accumulate-into plane): the instantiation with the hand-issued epilogue (EPI_STORE_SO)
            const bool so_epi = epi == EPI_STORE && as.s0.hi && !a.out0.p && !a.aux0.p
#if defined(WG_OPT_NO_EPI_BATCH)
                                && false
#endif
                ;
            const bool so_gate = epi == EPI_GATE && as.s0.hi && !a.out0.p && WG_TS_INTERLEAVED
#if defined(WG_OPT_NO_EPI_BATCH)
                                 && false
#endif
                ;
            const bool so_dgate = epi == EPI_DGATE && as.s0.hi && !a.out0.p && WG_TS_INTERLEAVED
#if defined(WG_OPT_NO_EPI_BATCH)
                                  && false
#endif
                ;
            const bool fo_epi = epi == EPI_STORE && !as.s0.hi && a.out0.p && !as.saux.hi
#if defined(WG_OPT_NO_EPI_BATCH)
                                && false
#endif
                ;
#if !defined(WG_OPT_NO_G192)
            // 256 x 192 tiles over flattened columns, eight multiplying waves fed by LDS-DMA (wg_gemm16g.h): the S-plane-only gate conv and
            // store / data-gradient / skip products whose tiles deal out evenly over the CUs (env WG_G192=0 restores the 256 x 128 / 128 x 128 forms)
            // (products of fewer than 16 chunks -- the residual conv, K = 256: 36.6 against 35.3 us -- stay on the older kernel: a tile that
            // short is mostly this kernel's longer prologue; S-plane arrays and weight images beyond 4 GB: its 32-bit offsets)
            // a gate conv whose tiles cannot fill the chip, cut along K (WSRGlow: 64 tiles, 139 chunks -> 4 parts of 35 on 256 workgroups) --
            // the parts go to the workspace's slab (wn_ws_layout sizes it for this in every mode, so a plain forward and a training step's
            // forward sum K in the same order; env WG_G192_SPLITK=0: off)
            if (g192_on() && so_gate && g.rows == 0 && !cx.row_sel1 && !cx.rec && M % WGG_BM == 0 && g.H >= 64 && cus % 8 == 0 && cx.gslab &&
                g192_fits(as, a16.img_stride, nseg)) {
                const int nct = (g.B * g.Tt + WGG_BN - 1) / WGG_BN, nrb = M / WGG_BM;
                int nt, S;
                gate_split_plan(g.B * g.Tt, M, nc, cus, nt, S);
                const size_t need = (size_t)S * nt * 8 * 24 * 256;
                if (env_sw().g192_splitk && S && need <= cx.gslab_floats) {
                    if (cx.probe) { *cx.probe = 0; return; }  // (the cut gate conv does not write the partial rows)
                    ConvGemm16sArgs ap = as;
                    ap.eff = nullptr; ap.part = nullptr;
                    ap.ntx = nct; ap.nty = nrb; ap.ntz = S; ap.xcd_items = 2;
                    ap.c.out0.p = cx.gslab;
                    WG_LAUNCH(cx, convgemm16g_kernel<WGG_EPI_PART>, dim3(cus), dim3(512), 0, ap);
                    ap.c.out0 = a.out0;
                    WG_LAUNCH(cx, gate_finish16g_kernel, dim3(nt * 6), dim3(512), 0, ap, (const float *)cx.gslab, S);
                    g_gate_split_launches.fetch_add(1, std::memory_order_relaxed);
                    g_last_launch = "convgemm16g_kernel<WGG_EPI_PART> + gate_finish16g_kernel";      // (one timed class entry covers both)
                    return;
                }
            }
            const bool fo_g = fo_epi && !a.aux0.p;               // (the skip sum: fp32 plane out, nothing to accumulate into)
            if (g192_on() && !small && (so_gate || so_epi || fo_g) && g.rows == 0 && !cx.row_sel1 && !cx.rec && M % WGG_BM == 0 && g.H >= 64 && cus % 8 == 0 &&
                nc >= 16 && nc <= WGG_MAXCHUNKS && g192_fits(as, a16.img_stride, nseg)) {
                const int nct = (g.B * g.Tt + WGG_BN - 1) / WGG_BN, nrb = M / WGG_BM, nt = nct * nrb;
                const int rounds = (nt + cus - 1) / cus;
                if (nt >= cus && (double)(rounds * cus - nt) <= 0.1 * rounds * cus) {
                    as.ntx = nct; as.nty = nrb; as.ntz = 1; as.xcd_items = 0;
                    if (env_sw().g192_own) as.xcd_items = 1;                                   // experiment: column ownership
                    if (as.prow != 8) as.part = nullptr;      // (this kernel's epilogue writes 8-float rows)
                    if (cx.probe) { *cx.probe = (so_gate && as.part) ? 1 : 0; return; }
                    if (so_gate && as.part) { ++cx.part_written; g_gate_part_launches.fetch_add(1, std::memory_order_relaxed); }
                    if (so_gate) WG_LAUNCH(cx, convgemm16g_kernel<EPI_GATE_SO>, dim3(cus), dim3(512), 0, as);
                    else if (fo_g) WG_LAUNCH(cx, convgemm16g_kernel<EPI_STORE_FO>, dim3(cus), dim3(512), 0, as);
                    else WG_LAUNCH(cx, convgemm16g_kernel<EPI_STORE_SO>, dim3(cus), dim3(512), 0, as);
                    return;
                }
            }
#endif
#if !defined(WG_OPT_NO_M64)
            // products with at most 64 rows on 64 x 128 tiles (convgemm16q_kernel<.., M64>): WaveFlow's 64-channel WN2D -- on 128-row tiles
            // half of every MFMA multiplied padding (the data-gradient conv: 181 TF against 314 for the full-height gate conv)
            if (!small && M <= 64 && epi != EPI_GATE && epi != EPI_RESSKIP) {
                if (fo_epi) { WG_LAUNCH(cx, (convgemm16q_kernel<EPI_STORE_FO, 2, 1, true>), gp, dim3(512), 0, as); return; }
                if (so_dgate) { WG_LAUNCH(cx, (convgemm16q_kernel<EPI_DGATE_SO, 2, 1, true>), gp, dim3(512), 0, as); return; }
                if (so_epi) { WG_LAUNCH(cx, (convgemm16q_kernel<EPI_STORE_SO, 2, 1, true>), gp, dim3(512), 0, as); return; }
                if (epi == EPI_STORE) { WG_LAUNCH(cx, (convgemm16q_kernel<EPI_STORE, 2, 1, true>), gp, dim3(512), 0, as); return; }
                if (epi == EPI_DGATE) { WG_LAUNCH(cx, (convgemm16q_kernel<EPI_DGATE, 2, 1, true>), gp, dim3(512), 0, as); return; }
            }
#endif
            if (small) {
                if (fo_epi) { WG_LAUNCH(cx, (convgemm16q_kernel<EPI_STORE_FO, 1>), gp, dim3(512), 0, as); return; }
                if (so_dgate) { WG_LAUNCH(cx, (convgemm16q_kernel<EPI_DGATE_SO, 1>), gp, dim3(512), 0, as); return; }
                if (so_gate) { if (cx.probe) { *cx.probe = as.part ? 1 : 0; return; } if (as.part) { ++cx.part_written; g_gate_part_launches.fetch_add(1, std::memory_order_relaxed); } WG_LAUNCH(cx, (convgemm16q_kernel<EPI_GATE_SO, 1>), gp, dim3(512), 0, as); return; }
                if (so_epi) { WG_LAUNCH(cx, (convgemm16q_kernel<EPI_STORE_SO, 1>), gp, dim3(512), 0, as); return; }
                switch (epi) {
                case EPI_STORE: WG_LAUNCH(cx, (convgemm16q_kernel<EPI_STORE, 1>), gp, dim3(512), 0, as); break;
                case EPI_GATE: WG_LAUNCH(cx, (convgemm16q_kernel<EPI_GATE, 1>), gp, dim3(512), 0, as); break;
                case EPI_RESSKIP: WG_LAUNCH(cx, (convgemm16q_kernel<EPI_RESSKIP, 1>), gp, dim3(512), 0, as); break;
                case EPI_DGATE: WG_LAUNCH(cx, (convgemm16q_kernel<EPI_DGATE, 1>), gp, dim3(512), 0, as); break;
                }
                return;
            }
#if !defined(WG_OPT_NO_CG2)
            // products with ONE 128-row tile (WaveFlow's gate conv, M = 2 Cd = 128) on 128 x 256 tiles: one 16-wave workgroup per CU whose two
            // compute groups share the A image of every chunk (convgemm16q_kernel<.., CG2>) -- 25 % less L2 -> LDS traffic for a launch
            // that sits at the per-CU intake limit
            if (epi == EPI_GATE && (int)grid.y == 1 && g.Tt % 256 == 0 && cus % 8 == 0) {
                as.ntx = (int)grid.x / 2;
                const int nt2 = as.ntx * as.nty * as.ntz;
                if (nt2 >= cus) {
                    as.xcd_items = 0;
                    const dim3 gc(std::min(nt2, cus));
                    if (so_gate) { if (cx.probe) { *cx.probe = as.part ? 1 : 0; return; } if (as.part) { ++cx.part_written; g_gate_part_launches.fetch_add(1, std::memory_order_relaxed); } WG_LAUNCH(cx, (convgemm16q_kernel<EPI_GATE_SO, 2, 2, false, true>), gc, dim3(1024), 0, as); return; }
                    WG_LAUNCH(cx, (convgemm16q_kernel<EPI_GATE, 2, 2, false, true>), gc, dim3(1024), 0, as);
                    return;
                }
                as.ntx = (int)grid.x;
            }
#endif
#if !defined(WG_OPT_NO_MG2)
            // 256 x 128 tiles, one 16-wave workgroup per CU (the compute groups share every chunk's B image: 25 % less L2 -> LDS
            // traffic, no slower co-resident workgroup left to finish alone): gate conv 125.7 -> 118.6 us.  Only where the tiles deal out
            // evenly over the CUs: at 1.5 tiles per CU (the 256-row products of the training shape: 384 such tiles) the half-empty second
            // round costs more than the sharing saves (measured: step 81.8 -> 83.0 ms with every eligible launch on this path).
            // (1.69 such tiles per CU -- the gate conv of a 10 s utterance -- still gain 4 %: 20.2 -> 21.0 MHz; 1.5 per CU lose)
            const int rounds = (ntiles / 2 + cus - 1) / cus;
            const bool mg2_ok = ntiles / 2 >= cus && (double)(rounds * cus - ntiles / 2) <= 0.17 * rounds * cus;
            if (epi != EPI_DGATE && rup(mrows, WG_TILE) % 256 == 0 && mg2_ok) {
                as.nty = (int)grid.y / 2;
                const dim3 g2(std::min(ntiles / 2, cus));
                if (cus % 8) as.xcd_items = 0;                // (mg2_ok: at least one tile per CU)
                if (so_epi) { WG_LAUNCH(cx, (convgemm16q_kernel<EPI_STORE_SO, 2, 2>), g2, dim3(1024), 0, as); return; }
                if (so_gate) { if (cx.probe) { *cx.probe = as.part ? 1 : 0; return; } if (as.part) { ++cx.part_written; g_gate_part_launches.fetch_add(1, std::memory_order_relaxed); } WG_LAUNCH(cx, (convgemm16q_kernel<EPI_GATE_SO, 2, 2>), g2, dim3(1024), 0, as); return; }
                if (fo_epi) { WG_LAUNCH(cx, (convgemm16q_kernel<EPI_STORE_FO, 2, 2>), g2, dim3(1024), 0, as); return; }
                switch (epi) {
                case EPI_STORE: WG_LAUNCH(cx, (convgemm16q_kernel<EPI_STORE, 2, 2>), g2, dim3(1024), 0, as); break;
                case EPI_GATE: WG_LAUNCH(cx, (convgemm16q_kernel<EPI_GATE, 2, 2>), g2, dim3(1024), 0, as); break;
                case EPI_RESSKIP: WG_LAUNCH(cx, (convgemm16q_kernel<EPI_RESSKIP, 2, 2>), g2, dim3(1024), 0, as); break;
                }
                return;
            }
#endif
            if (so_epi) { WG_LAUNCH(cx, (convgemm16q_kernel<EPI_STORE_SO, 2>), gp, dim3(512), 0, as); return; }
            if (so_gate) { if (cx.probe) { *cx.probe = as.part ? 1 : 0; return; } if (as.part) { ++cx.part_written; g_gate_part_launches.fetch_add(1, std::memory_order_relaxed); } WG_LAUNCH(cx, (convgemm16q_kernel<EPI_GATE_SO, 2>), gp, dim3(512), 0, as); return; }
            if (so_dgate) { WG_LAUNCH(cx, (convgemm16q_kernel<EPI_DGATE_SO, 2>), gp, dim3(512), 0, as); return; }
            if (fo_epi) { WG_LAUNCH(cx, (convgemm16q_kernel<EPI_STORE_FO, 2>), gp, dim3(512), 0, as); return; }
            switch (epi) {
            case EPI_STORE: WG_LAUNCH(cx, (convgemm16q_kernel<EPI_STORE, 2>), gp, dim3(512), 0, as); break;
            case EPI_GATE: WG_LAUNCH(cx, (convgemm16q_kernel<EPI_GATE, 2>), gp, dim3(512), 0, as); break;
            case EPI_RESSKIP: WG_LAUNCH(cx, (convgemm16q_kernel<EPI_RESSKIP, 2>), gp, dim3(512), 0, as); break;
            case EPI_DGATE: WG_LAUNCH(cx, (convgemm16q_kernel<EPI_DGATE, 2>), gp, dim3(512), 0, as); break;
            }
            return;
#else                                             // A/B build -DWG_OPT_MFMA32: the 32x32x16 kernel (git show 4c099e9:tools/experiments/wg_gemm16_superseded.h)
            if (small) {
                switch (epi) {
                case EPI_STORE: WG_LAUNCH(cx, (convgemm16w_kernel<EPI_STORE, 1>), gp, dim3(512), 0, as); break;
                case EPI_GATE: WG_LAUNCH(cx, (convgemm16w_kernel<EPI_GATE, 1>), gp, dim3(512), 0, as); break;
                case EPI_RESSKIP: WG_LAUNCH(cx, (convgemm16w_kernel<EPI_RESSKIP, 1>), gp, dim3(512), 0, as); break;
                case EPI_DGATE: WG_LAUNCH(cx, (convgemm16w_kernel<EPI_DGATE, 1>), gp, dim3(512), 0, as); break;
                }
                return;
            }
            switch (epi) {
            case EPI_STORE: WG_LAUNCH(cx, (convgemm16w_kernel<EPI_STORE, 2>), gp, dim3(512), 0, as); break;
            case EPI_GATE: WG_LAUNCH(cx, (convgemm16w_kernel<EPI_GATE, 2>), gp, dim3(512), 0, as); break;
            case EPI_RESSKIP: WG_LAUNCH(cx, (convgemm16w_kernel<EPI_RESSKIP, 2>), gp, dim3(512), 0, as); break;
            case EPI_DGATE: WG_LAUNCH(cx, (convgemm16w_kernel<EPI_DGATE, 2>), gp, dim3(512), 0, as); break;
            }
            return;
#endif
#else                                             // A/B build: the symmetric software-pipelined kernel
            switch (epi) {
            case EPI_STORE: WG_LAUNCH(cx, convgemm16p_kernel<EPI_STORE>, grid, block, 0, as); break;
            case EPI_GATE: WG_LAUNCH(cx, convgemm16p_kernel<EPI_GATE>, grid, block, 0, as); break;
            case EPI_RESSKIP: WG_LAUNCH(cx, convgemm16p_kernel<EPI_RESSKIP>, grid, block, 0, as); break;
            case EPI_DGATE: WG_LAUNCH(cx, convgemm16p_kernel<EPI_DGATE>, grid, block, 0, as); break;
            }
            return;
#endif
        }
        const bool big = (rup(mrows, WG_TILE) % 256) == 0;      // 256-row tiles when M allows it
        if (big) {
            dim3 grid4(g.Tt / WG_TILE, rup(mrows, WG_TILE) / 256, g.B), block4(512);
            switch (epi) {
            case EPI_STORE: WG_LAUNCH(cx, (convgemm16_kernel<EPI_STORE, 4>), grid4, block4, 0, a16); break;
            case EPI_GATE: WG_LAUNCH(cx, (convgemm16_kernel<EPI_GATE, 4>), grid4, block4, 0, a16); break;
            case EPI_RESSKIP: WG_LAUNCH(cx, (convgemm16_kernel<EPI_RESSKIP, 4>), grid4, block4, 0, a16); break;
            case EPI_DGATE: WG_LAUNCH(cx, (convgemm16_kernel<EPI_DGATE, 4>), grid4, block4, 0, a16); break;
            }
        } else {
            switch (epi) {
            case EPI_STORE: WG_LAUNCH(cx, (convgemm16_kernel<EPI_STORE, 2>), grid, block, 0, a16); break;
            case EPI_GATE: WG_LAUNCH(cx, (convgemm16_kernel<EPI_GATE, 2>), grid, block, 0, a16); break;
            case EPI_RESSKIP: WG_LAUNCH(cx, (convgemm16_kernel<EPI_RESSKIP, 2>), grid, block, 0, a16); break;
            case EPI_DGATE: WG_LAUNCH(cx, (convgemm16_kernel<EPI_DGATE, 2>), grid, block, 0, a16); break;
            }
        }
        return;
    }
    switch (epi) {
    case EPI_STORE: WG_LAUNCH(cx, convgemm_kernel<EPI_STORE>, grid, block, 0, a); break;
    case EPI_GATE: WG_LAUNCH(cx, convgemm_kernel<EPI_GATE>, grid, block, 0, a); break;
    case EPI_RESSKIP: WG_LAUNCH(cx, convgemm_kernel<EPI_RESSKIP>, grid, block, 0, a); break;
    case EPI_DGATE: WG_LAUNCH(cx, convgemm_kernel<EPI_DGATE>, grid, block, 0, a); break;
    }
}

struct WSegSpec {
    const float *src;
    int Cp, ch0, nch, shift;
    const float *s;      // S-plane of the same tensor (precision 2) or nullptr
    int sCp, sch0;
    int row_off = 0, per_item = 0;   // Geo::rows > 0 only (B operand)
};
// returns the plan used (finalize needs nsplit / strides)
struct WgradOut {
    int nsplit, Mp, Np;
    float *slab = nullptr;      // where run_wgrad put the slabs (nullptr: the pointer the caller passes to run_finalize)
};
// Weight gradients of one WN backward, finalised together: every run_wgrad takes its split-K slab from the arena, every run_finalize
// only queues its job; the queue is launched as ONE finalize_batch_kernel when the arena or the job table is full and when the WN
// is done.  (Three 4-13 us finalize launches per layer, none of them filling the GPU, were 3 % of the step.)
struct FinQueue {
    Ctx &cx;
    float *arena;
    size_t cap, off = 0;
    FinBatch b;
    FinQueue(Ctx &c, float *a, size_t n) : cx(c), arena(a), cap(n) { b.n = 0; b.start[0] = 0; cx.fq = this; }
    ~FinQueue() { flush(); cx.fq = nullptr; }
    float *reserve(size_t n)
    {
        n = rupz(n, 64);
        if (off + n > cap) {                                  // arena full: finalise what is queued, then start over -- in stream order the
            flush();                                          // batch reads those slabs before any later product rewrites them
            off = 0;
        }
        if (n > cap) { if (!cx.err) cx.err = WG_EWORKSPACE; return arena; }
        float *p = arena + off;
        off += n;
        return p;
    }
    void ensure(size_t n)                                     // the next reserves, n floats in all, will not wrap the arena between them
    {
        if (off + n > cap) { flush(); off = 0; }
    }
    void add(const FinJob &j)
    {
        if (b.n == WG_FIN_JOBS) flush();                      // (the arena keeps growing: the job being added still owns its slab)
        b.job[b.n] = j;
        b.start[b.n + 1] = b.start[b.n] + j.rows;
        ++b.n;
    }
    void flush()
    {
        if (b.n) {
            int maxcols = 0;
            for (int i = 0; i < b.n; ++i) maxcols = std::max(maxcols, b.job[i].I * b.job[i].R);
#if !defined(WG_OPT_FIN_BLOCK_ROWS)
            if (maxcols <= 1024)        // one wave per row, four rows per block (long rows -- WSRGlow's 3659 conditioning columns -- keep a block each)
                WG_LAUNCH(cx, finalize_batch_wave_kernel, dim3((b.start[b.n] + 3) / 4), dim3(256), (size_t)4 * maxcols * sizeof(float), b, maxcols);
            else
#endif
                WG_LAUNCH(cx, finalize_batch_kernel, dim3(b.start[b.n]), dim3(256), 0, b);
        }
        b.n = 0;
    }
};
size_t slab_floats(const Geo &g, int Mp, int Np);
WgradOut run_wgrad(Ctx &cx, const Geo &g, const WSegSpec *sa, int nsa, const WSegSpec *sb, int nsb, float *slab, size_t slab_cap)
{
    WgradArgs a;
    memset(&a, 0, sizeof(a));
    a.nseg_a = nsa; a.nseg_b = nsb;
    int blk = 0;
    for (int s = 0; s < nsa; ++s) {
        a.sa[s].src = sa[s].src; a.sa[s].Cp = sa[s].Cp; a.sa[s].ch0 = sa[s].ch0; a.sa[s].nch = sa[s].nch;
        a.sa[s].shift = 0; a.sa[s].blk0 = blk;
        blk += rup(sa[s].nch, 32) / 32;
    }
    a.Mp = rup(blk * 32, WG_TILE);
    blk = 0;
    for (int s = 0; s < nsb; ++s) {
        a.sb[s].src = sb[s].src; a.sb[s].Cp = sb[s].Cp; a.sb[s].ch0 = sb[s].ch0; a.sb[s].nch = sb[s].nch;
        a.sb[s].shift = sb[s].shift; a.sb[s].blk0 = blk;
        a.sb[s].row_off = sb[s].row_off; a.sb[s].per_item = sb[s].per_item;
        blk += rup(sb[s].nch, 32) / 32;
    }
    a.Np = rup(blk * 32, WG_TILE);
    a.g = g;
    if (cx.fq) {                                             // batched finalisation: this product's slab comes from the queue's arena
        slab_cap = slab_floats(g, a.Mp, a.Np);
        slab = cx.fq->reserve(slab_cap);
    }
    const WgradPlan p = plan_wgrad(g, (a.Mp / WG_TILE) * (a.Np / WG_TILE));
    a.t_per_split = p.t_per_split; a.nts = p.nts; a.b_per_split = p.b_per_split;
    a.slab = slab;
    WgradOut o;
    o.nsplit = p.nsplit; o.Mp = a.Mp; o.Np = a.Np; o.slab = slab;
    if ((size_t)p.nsplit * a.Mp * a.Np > slab_cap) { if (!cx.err) cx.err = WG_EWORKSPACE; return o; }
    dim3 grid(a.Np / WG_TILE, a.Mp / WG_TILE, p.nsplit), block(256);
    TimerScope ts(WG_K_WGRAD, cx.st);
    bool all_s = cx.prec == 2;
    for (int s = 0; s < nsa; ++s) all_s = all_s && sa[s].s;
    for (int s = 0; s < nsb; ++s) all_s = all_s && sb[s].s;
    if (all_s) {
        WgradSArgs q;
        memset(&q, 0, sizeof(q));
        q.nseg_a = nsa; q.nseg_b = nsb; q.g = g;
        q.cpb = g.Tt / WG16_BK; q.total_chunks = g.B * q.cpb;
        // 256-row tiles (8 waves, one workgroup per CU): 25 % less operand stream for the same MFMAs, but MEASURED SLOWER (155 us
        // against 137 us per launch at the headline shape, parity identical): one 8-wave workgroup in barrier lockstep hides less
        // latency than two independent 4-wave ones.  Opt-in A/B build only.
        const bool tall = a.Mp % 256 == 0
#if !defined(WG_OPT_WGRAD_TALL)
                          && false
#endif
            ;
        q.nsplit = plan_wgrad_flat(g, (a.Mp / WG_TILE) * (a.Np / WG_TILE));   // tall: half the tiles for half the slots -- the same split
        o.nsplit = q.nsplit;
        if ((size_t)q.nsplit * a.Mp * a.Np > slab_cap) { if (!cx.err) cx.err = WG_EWORKSPACE; return o; }
        grid.z = q.nsplit;
        q.slab = slab; q.Mp = a.Mp; q.Np = a.Np;
        for (int s = 0; s < nsa; ++s) {
            q.sa[s].hi = (const unsigned short *)sa[s].s; q.sa[s].lo_off = (size_t)g.B * sa[s].sCp * g.P;
            q.sa[s].Cp = sa[s].sCp; q.sa[s].ch0 = sa[s].sch0; q.sa[s].nch = sa[s].nch; q.sa[s].shift = 0; q.sa[s].blk0 = a.sa[s].blk0;
        }
        for (int s = 0; s < nsb; ++s) {
            q.sb[s].hi = (const unsigned short *)sb[s].s; q.sb[s].lo_off = (size_t)(sb[s].per_item ? g.B / g.rows : g.B) * sb[s].sCp * g.P;
            q.sb[s].Cp = sb[s].sCp; q.sb[s].ch0 = sb[s].sch0; q.sb[s].nch = sb[s].nch; q.sb[s].shift = sb[s].shift; q.sb[s].blk0 = a.sb[s].blk0;
            q.sb[s].row_off = sb[s].row_off; q.sb[s].per_item = sb[s].per_item;
        }
        if (tall) {
            grid.y = a.Mp / 256;
            WG_LAUNCH(cx, wgrad16s_kernel<2>, grid, dim3(512), 0, q);
        } else {
            WG_LAUNCH(cx, wgrad16s_kernel<1>, grid, block, 0, q);
        }
    } else if (cx.prec) WG_LAUNCH(cx, wgrad16_kernel, grid, block, 0, a);
    else WG_LAUNCH(cx, wgrad_kernel, grid, block, 0, a);
    if (o.nsplit >= 8 && a.Mp <= 256) {          // few rows = few finalize blocks: fold the slabs with the whole GPU first
        const size_t n = (size_t)a.Mp * a.Np;    // a multiple of 128 * 128
        WG_LAUNCH(cx, slab_reduce_kernel, dim3((unsigned)((n / 4 + 63) / 64)), dim3(256), 0, slab, o.nsplit, n);
        o.nsplit = 1;
    }
    return o;
}

// One launch for `ng` products of one shape (precision 2 only; every operand an S-plane).  gs[k]: the k-th product's segments -- all
// groups have the same segment sizes, only the planes (sa[j].s may be nullptr: zero rows) and the tap shifts differ.  outs[k]: where
// its slabs went.
struct WgradGroupSpec {
    WSegSpec sa[2];
    WSegSpec sb[WG_MAX_SEG];
};
// shape and split of a grouped product (no slab yet)
static bool shape_wgrad_group(Ctx &cx, const Geo &g, const WgradGroupSpec *gs, int ng, int nsa, int nsb, WgradSArgs &q)
{
    memset(&q, 0, sizeof(q));
    if (!cx.fq || ng > WG_GRP_MAX || nsb > WG_MAX_SEG || nsa > 2) { if (!cx.err) cx.err = WG_EINVAL; return false; }
    q.nseg_a = nsa; q.nseg_b = nsb; q.g = g;
    q.cpb = g.Tt / WG16_BK; q.total_chunks = g.B * q.cpb;
    int blk = 0;
    for (int s = 0; s < nsa; ++s) {
        const WSegSpec &x = gs[0].sa[s];
        q.sa[s].hi = nullptr; q.sa[s].lo_off = (size_t)g.B * x.sCp * g.P;
        q.sa[s].Cp = x.sCp; q.sa[s].ch0 = x.sch0; q.sa[s].nch = x.nch; q.sa[s].shift = 0; q.sa[s].blk0 = blk;
        blk += rup(x.nch, 32) / 32;
    }
    q.Mp = rup(blk * 32, WG_TILE);
    blk = 0;
    for (int s = 0; s < nsb; ++s) {
        const WSegSpec &x = gs[0].sb[s];
        q.sb[s].hi = nullptr; q.sb[s].lo_off = (size_t)(x.per_item ? g.B / g.rows : g.B) * x.sCp * g.P;
        q.sb[s].Cp = x.sCp; q.sb[s].ch0 = x.sch0; q.sb[s].nch = x.nch; q.sb[s].shift = 0; q.sb[s].blk0 = blk;
        q.sb[s].row_off = 0; q.sb[s].per_item = x.per_item;
        // the distinct planes of group 0 (at most three) name the planes of every group
        int pl = -1, used = 0;
        for (int u = 0; u < s; ++u) {
            used = std::max(used, q.b_plane_of[u] + 1);
            if (gs[0].sb[u].s == x.s) { pl = q.b_plane_of[u]; break; }
        }
        if (pl < 0) pl = used;
        if (pl > 2) { if (!cx.err) cx.err = WG_EINVAL; return false; }
        q.b_plane_of[s] = (unsigned char)pl;
        blk += rup(x.nch, 32) / 32;
    }
    q.Np = rup(blk * 32, WG_TILE);
    const int tiles = (q.Mp / WG_TILE) * (q.Np / WG_TILE);
    q.nsplit = plan_wgrad_flat(g, tiles * ng);
#if defined(WG_OPT_GRP_NSPLIT_MUL)                             // experiment: more, shorter workgroups (and more slab bytes)
    q.nsplit = std::min(q.nsplit * WG_OPT_GRP_NSPLIT_MUL, std::max(1, q.total_chunks / 4));
#endif
    while (q.nsplit > 1 && rupz((size_t)q.nsplit * ng * q.Mp * q.Np, 64) > cx.fq->cap) --q.nsplit;
    q.ngroups = ng;
    if (rupz((size_t)q.nsplit * ng * q.Mp * q.Np, 64) > cx.fq->cap) { if (!cx.err) cx.err = WG_EWORKSPACE; return false; }
    return true;
}
static size_t group_slab_floats(const WgradSArgs &q) { return rupz((size_t)q.nsplit * q.ngroups * q.Mp * q.Np, 64); }
// slabs (from the finalisation queue's arena) and the groups' planes
static bool bind_wgrad_group(Ctx &cx, const WgradGroupSpec *gs, const float *zero_plane, WgradOut *outs, WgradSArgs &q)
{
    const size_t one = (size_t)q.Mp * q.Np;
    float *slab = cx.fq->reserve(group_slab_floats(q));
    if (cx.err) return false;
    q.zsrc = (const unsigned short *)zero_plane; q.slab = slab;
    for (int k = 0; k < q.ngroups; ++k) {
        for (int s = 0; s < q.nseg_a; ++s) q.grp[k].a_hi[s] = (const unsigned short *)gs[k].sa[s].s;
        q.grp[k].b_plane[0] = q.grp[k].b_plane[1] = q.grp[k].b_plane[2] = nullptr;
        for (int s = 0; s < q.nseg_b; ++s) {
            q.grp[k].b_plane[q.b_plane_of[s]] = (const unsigned short *)gs[k].sb[s].s;
            q.grp[k].b_shift[s] = (short)gs[k].sb[s].shift; q.grp[k].b_row[s] = (short)gs[k].sb[s].row_off;
        }
        q.grp[k].slab = slab + (size_t)k * q.nsplit * one;
        outs[k].nsplit = q.nsplit; outs[k].Mp = q.Mp; outs[k].Np = q.Np; outs[k].slab = q.grp[k].slab;
    }
    return true;
}
void run_wgrad_group(Ctx &cx, const Geo &g, const WgradGroupSpec *gs, int ng, int nsa, int nsb, const float *zero_plane, WgradOut *outs)
{
    WgradSArgs q;
    if (!shape_wgrad_group(cx, g, gs, ng, nsa, nsb, q) || !bind_wgrad_group(cx, gs, zero_plane, outs, q)) return;
    TimerScope ts(WG_K_WGRAD, cx.st);
    WG_LAUNCH(cx, wgrad16s_kernel<1>, dim3(q.Np / WG_TILE, q.Mp / WG_TILE, q.nsplit * ng), dim3(256), 0, q);
}
static int wgt_valid_cols(const WgradSArgs &q)
{
    const WgSSeg &l = q.sb[q.nseg_b - 1];
    return std::min(q.Np, rup(l.blk0 * 32 + l.nch, 16));
}

// two grouped products in one launch (wgrad16s_pair_kernel): the one with the longer workgroups first
void run_wgrad_group_pair(Ctx &cx, const Geo &g, const WgradGroupSpec *gs0, int nsa0, int nsb0, WgradOut *outs0,
                          const WgradGroupSpec *gs1, int nsa1, int nsb1, WgradOut *outs1, int ng, const float *zero_plane)
{
    WgradPairArgs pp;
    if (!shape_wgrad_group(cx, g, gs0, ng, nsa0, nsb0, pp.p[0]) || !shape_wgrad_group(cx, g, gs1, ng, nsa1, nsb1, pp.p[1])) return;
    // one workgroup per CU on 256 x 128 tiles (wg_wgrad16t.h) when the shape has a plan: 1-D planes, 256-row products
    WgtPlan plan;
    if (g.rows == 0 && pp.p[0].Mp % 256 == 0 && pp.p[1].Mp % 256 == 0)
        plan = plan_wgt_any(pp.p[0].Mp / 256, pp.p[0].Np / WG_TILE, pp.p[1].Mp / 256, pp.p[1].Np / WG_TILE, ng, pp.p[0].total_chunks);
    if (plan.ok) { pp.p[0].nsplit = plan.nslab[0]; pp.p[1].nsplit = plan.nslab[1]; }
    // progress counters of the (group, split) sets of both products (soft lock-step, wg_gemm16s.h): one 128-byte line each
    const size_t nctr0 = (size_t)pp.p[0].nsplit * ng, nctr1 = (size_t)pp.p[1].nsplit * ng;
    size_t sync_floats = rupz((nctr0 + nctr1) * WG_SYNC_STRIDE, 64);
#if !defined(WG_OPT_WGRAD_SYNC)                              // measured: holds the HBM traffic at the operand bytes and costs more than that saves
    sync_floats = 0;
#endif
    if (plan.ok) sync_floats = 0;
    if (group_slab_floats(pp.p[0]) + group_slab_floats(pp.p[1]) + sync_floats > cx.fq->cap) { if (!cx.err) cx.err = WG_EWORKSPACE; return; }   // (wn_ws_layout sizes for it)
    cx.fq->ensure(group_slab_floats(pp.p[0]) + group_slab_floats(pp.p[1]) + sync_floats);    // all live until the launch: no wrap between them
    if (!bind_wgrad_group(cx, gs0, zero_plane, outs0, pp.p[0]) || !bind_wgrad_group(cx, gs1, zero_plane, outs1, pp.p[1])) return;
    if (plan.ok) {
        WgtArgs wa;
        memset(&wa, 0, sizeof(wa));
        wa.p[0] = pp.p[0]; wa.p[1] = pp.p[1];
        wa.nph = plan.nph;
        for (int i = 0; i < plan.nph; ++i) {
            wa.ph[i] = plan.ph[i];
            const WgradSArgs &q = wa.p[plan.ph[i].prod];
            wa.ph[i].tn = q.Np / WG_TILE; wa.ph[i].ngroups = ng;
            if (!wa.ph[i].tiles) { wa.ph[i].tiles = (q.Mp / 256) * (q.Np / WG_TILE); wa.ph[i].nsub = 1; wa.ph[i].tns = wa.ph[i].tn; wa.ph[i].set0 = 0; }
        }
        for (int w = 0; w < 2; ++w) wa.nvalid[w] = wgt_valid_cols(wa.p[w]);
        // the launch as ONE entry of the kernel timer: 2 * M * K * cols = its algorithmic FLOPs with M = sum over layers and products of
        // rows x columns of the gradient, K = 1, cols = time steps; bytes = every operand plane once (hi + lo) + the slabs
        long long mm = 0, ch = 0, slabf = 0;
        for (int w = 0; w < 2; ++w) {
            const WgradSArgs &q = wa.p[w];
            long long rows = 0, colsB = 0, distinct = 0;
            for (int u = 0; u < q.nseg_a; ++u) rows += q.sa[u].nch;
            for (int u = 0; u < q.nseg_b; ++u) {
                colsB += q.sb[u].nch;
                bool seen = false;
                for (int v = 0; v < u; ++v) seen = seen || q.b_plane_of[v] == q.b_plane_of[u];
                if (!seen) distinct += q.sb[u].nch;
            }
            mm += (long long)ng * rows * colsB;
            ch += (long long)ng * (rows + distinct);
            slabf += (long long)group_slab_floats(q);
        }
        const long long tsteps = (long long)g.B * g.T;
        TimerScope ts(WG_K_WGRAD, cx.st, mm, 1, tsteps, 4 * ch * tsteps + 4 * slabf);
        WG_LAUNCH(cx, wgrad16t_kernel, dim3(256), dim3(768), 0, wa);
        g_wgrad16t_launches.fetch_add(1, std::memory_order_relaxed);
        return;
    }
    if (sync_floats) {
        unsigned *ctr = reinterpret_cast<unsigned *>(cx.fq->reserve(sync_floats));
        if (cx.err) return;
        if (hipMemsetAsync(ctr, 0, sync_floats * sizeof(float), cx.st) != hipSuccess) { cx.err = WG_ELAUNCH; return; }
        pp.p[0].sync = ctr; pp.p[1].sync = ctr + nctr0 * WG_SYNC_STRIDE;
    }
    for (int w = 0; w < 2; ++w) {
        pp.gx[w] = pp.p[w].Np / WG_TILE; pp.gy[w] = pp.p[w].Mp / WG_TILE;
        pp.n[w] = pp.gx[w] * pp.gy[w] * pp.p[w].nsplit * ng;
        pp.p[w].sync_n = pp.gx[w] * pp.gy[w];
    }
    pp.n0 = pp.n[0];
    // (the launch as ONE timer entry, like the planned kernel above: M = sum of the gradients' sizes, K = 1, columns = time steps)
    long long mm = 0, ch = 0, slabf = 0;
    for (int w = 0; w < 2; ++w) {
        const WgradSArgs &q = pp.p[w];
        long long rows = 0, colsB = 0, distinct = 0;
        for (int u = 0; u < q.nseg_a; ++u) rows += q.sa[u].nch;
        for (int u = 0; u < q.nseg_b; ++u) {
            colsB += q.sb[u].nch;
            bool seen = false;
            for (int v = 0; v < u; ++v) seen = seen || q.b_plane_of[v] == q.b_plane_of[u];
            if (!seen) distinct += q.sb[u].nch;
        }
        mm += (long long)ng * rows * colsB;
        ch += (long long)ng * (rows + distinct);
        slabf += (long long)group_slab_floats(q);
    }
    const long long tsteps = (long long)g.B * g.T;
    TimerScope ts(WG_K_WGRAD, cx.st, mm, 1, tsteps, 4 * ch * tsteps + 4 * slabf);
    WG_LAUNCH(cx, wgrad16s_pair_kernel, dim3(pp.n[0] + pp.n[1]), dim3(256), 0, pp);
}

void run_finalize(Ctx &cx, const float *slab, const WgradOut &wo, int row0, int rows, int I, int R, int col0, int ci, int cr,
                  const float *gp, const float *vp, float *dg, float *dv,
                  const float *extra = nullptr, const float *esrc = nullptr, int n_extra = 0, float emul = 0.f)
{
    if (!dv && !dg) return;
    FinJob j;
    j.slab = wo.slab ? wo.slab : slab; j.nsplit = wo.nsplit; j.sstride = (size_t)wo.Mp * wo.Np; j.ldn = wo.Np; j.row0 = row0;
    j.rows = rows; j.I = I; j.R = R; j.col0 = col0; j.ci = ci; j.cr = cr;
    j.g = gp; j.v = vp; j.dg = dg; j.dv = dv;
    j.extra = extra; j.extra_scale_src = esrc; j.n_extra = n_extra; j.extra_mul = emul;
    if (cx.fq) { cx.fq->add(j); return; }
    WG_LAUNCH(cx, finalize_kernel, dim3(rows), dim3(256), 0, j);
}

void run_mix(Ctx &cx, const Geo &g, PRef X, int c, const float *Mx, int transpose)
{
    dim3 grid((g.T + 255) / 256, g.B), block(256);
    switch (c) {
#define WG_MIX_CASE(CC) case CC: WG_LAUNCH(cx, mix_kernel<CC>, grid, block, 0, X, Mx, transpose, g); break;
        WG_MIX_CASE(2) WG_MIX_CASE(4) WG_MIX_CASE(6) WG_MIX_CASE(8) WG_MIX_CASE(10) WG_MIX_CASE(12)
        WG_MIX_CASE(14) WG_MIX_CASE(16) WG_MIX_CASE(18) WG_MIX_CASE(20) WG_MIX_CASE(22) WG_MIX_CASE(24) WG_MIX_CASE(26) WG_MIX_CASE(28)
        WG_MIX_CASE(30) WG_MIX_CASE(32)
#undef WG_MIX_CASE
    default: if (!cx.err) cx.err = WG_EUNSUPPORTED;
    }
}

// ------------------------------------------------------------------------------------------------
// WN on planes
// ------------------------------------------------------------------------------------------------
struct WnRun {
    WnD d;
    WnPack L;
    const float *pk;     // packed weights of this WN
    Geo g;
    float *ws;           // workspace base
    WnWs w;
    PRef X;              // flow state at ch0 = first channel of this flow
    const float *Y;      // aux plane base (auxp rows per item)
    const float *YS;     // its S-plane (precision 2)
    int save;            // keep all layers (backward) or ping-pong
    Geo gi;              // mode2d: the per-item geometry (conditioning, its gradient)
    float *rs;           // mode2d: S-plane [items][2 Cd][P] for the height-axis sum of dxy
    size_t rs_step = 0;  // floats between the layers' copies of it (0: one plane reused by every layer)
    int start_done = 0;  // h_0 (and its S-plane) are in place: the previous flow's seam launch ran WN.start (run_inv_seam)
};

// WnD::bias: the plane of ones the bias rows multiply (every WN pass refills it: a workspace may have served another shape in between)
__global__ void ones_fill_kernel(float *f32, unsigned short *hi, size_t lo_off, Geo g)
{
    const int p = blockIdx.x * blockDim.x + threadIdx.x, b = blockIdx.y;
    if (p >= g.P) return;
    const bool in = p >= g.H && p < g.H + g.T;
    for (int c = 0; c < 32; ++c) f32[((size_t)b * 32 + c) * g.P + p] = in ? 1.f : 0.f;
    if (hi) {
        const unsigned w = in ? 0x3F803F80u : 0u;             // bf16(1.0) twice; the lo array is zero
        const u32x4 h = {w, w, w, w}, z = {0u, 0u, 0u, 0u};
        for (int cg = 0; cg < 4; ++cg) {
            const size_t i = (((size_t)b * 4 + cg) * g.P + p) * 8;
            *reinterpret_cast<u32x4 *>(hi + i) = h;
            *reinterpret_cast<u32x4 *>(hi + lo_off + i) = z;
        }
    }
}
SegSpec ones_seg(Ctx &cx, const WnRun &r, bool fill)
{
    float *f = r.ws + r.w.ones, *sp = cx.prec == 2 ? r.ws + r.w.onesS : nullptr;
    if (fill) WG_LAUNCH(cx, ones_fill_kernel, dim3((r.g.P + 255) / 256, r.g.B), dim3(256), 0, f, (unsigned short *)sp, (size_t)r.g.B * 32 * r.g.P, r.g);
    SegSpec s = {f, 32, 0, 32, 0, sp, 32, 0};
    return s;
}

// The thin products of WN's backward (wg_thin.h): precision 2, inside a FinQueue (the partials come from its arena).
// Dynamic LDS of the two kernels (floats -> bytes): the [channels][65] tile, the thin operand's tile, the per-wave shares and the weights.
static int thin_icp(const WnD &d) { return d.ic <= 4 ? 4 : d.ic <= 8 ? 8 : 16; }
static int thin_k2p(const WnD &d) { return 2 * d.ic <= 8 ? 8 : 2 * d.ic <= 16 ? 16 : 32; }
static size_t thin_start_lds(const WnD &d) { const int icp = thin_icp(d); return ((size_t)d.C * WGTH_LDT + WGTH_TB * icp + 4 * icp * 64 + (size_t)d.C * icp) * sizeof(float); }
static size_t thin_end_lds(const WnD &d) { const int k2p = thin_k2p(d); return ((size_t)d.Cs * WGTH_LDT + WGTH_TB * k2p + (size_t)k2p * d.Cs) * sizeof(float); }
#define WG_LDS_BYTES (160 * 1024)      // gfx950: LDS per CU = the most one workgroup can ask for
// the shape part of thin_ok (shapes whose tiles do not fit the LDS -- 512 channels with more than 8 thin rows -- keep the split-K MFMA
// products + convs)
static bool thin_shape_ok(const WnD &d)
{
#if defined(WG_OPT_NO_THIN)
    return false;
#else
    return !d.bias && d.ic <= 16 && d.C % 8 == 0 && d.Cs % 8 == 0 && d.C <= WGTH_MAXROWS * WGTH_THREADS && d.Cs <= WGTH_MAXROWS * WGTH_THREADS &&
           thin_start_lds(d) <= WG_LDS_BYTES && thin_end_lds(d) <= WG_LDS_BYTES;
#endif
}

// ------------------------------------------------------------------------------------------------
// One launch per WN layer (wg_layer16h.h) where both of the layer's products would take the 64 x 64-tile kernel: the two launches are
// RECORDED (the recorder of the stage interpreter: run_convgemm fills its argument block instead of launching), checked to be the
// gate conv and the residual / skip conv over the same column tiles, and issued as ONE convlayer16h_kernel.  Returns false when the
// shape does not qualify (nothing was launched: the caller then issues the two launches the ordinary way).
// ------------------------------------------------------------------------------------------------
std::atomic<long long> g_layer_launches{0};                   // diagnostics: launches of convlayer16h_kernel by this process (wg_stat_layer_launches)
template <class FA, class FB>
bool run_convlayer(Ctx &cx, float *ws, size_t lsync, FA &&gate_call, FB &&wo_call)
{
#if defined(WG_OPT_NO_LAYER)
    return false;
#else
    if (cx.prec != 2 || cx.rec || cx.err) return false;
    {   // OPT-IN (WG_LAYER_FUSION=1 in the environment).  Measured on MI355X (gpurun_out/r04e_stress.txt): parity
        // identical, and no faster than the two launches it replaces -- 2.72 against 2.65-2.76 ms per 0.7 s utterance, 95.7 against 95.8 ms
        // for WaveFlow's row-by-row synthesis: the in-launch hand-off (write-through drain, arrival, poll, first sc1 loads) costs what
        // the kernel boundary did.  Kept for the test that pins it and as the starting point should the hand-off get cheaper.
        if (!env_sw().layer_fusion) return false;
    }
    StageRec rec;
    cx.rec = &rec;
    gate_call();
    wo_call();
    cx.rec = nullptr;
    if (!rec.ok || rec.st.size() != 2 || rec.st[0].kind != WGS_CONV_GATE || rec.st[1].kind != WGS_CONV_RESSKIP) return false;
    ConvLayer16hArgs la;
    la.gate = rec.st[0].u.conv;
    la.wo = rec.st[1].u.conv;
    const ConvGemm16sArgs &A = la.gate, &B = la.wo;
    if (A.ntx != B.ntx || A.ntz != B.ntz || A.c.row_sel1 != B.c.row_sel1 || A.c.out0.p || A.c.out1.p || !A.s0.hi) return false;
    if (B.c.nseg != 1 || B.sseg[0].hi != A.s0.hi || B.sseg[0].row_off || B.sseg[0].per_item) return false;      // W_o's operand IS the gate this launch writes
    la.ncol = A.ntx * A.ntz; la.ntx = A.ntx; la.nty = std::max(A.nty, B.nty);
    if (la.ncol * WGL_SYNC_STRIDE > WGL_SYNC_WORDS || la.nty < 1) return false;
    const int grid = 8 * ((la.ncol + 7) / 8) * la.nty;
    if (grid > 2 * device_cus()) return false;               // every workgroup resident: a set never waits for an undispatched member
    la.sync = reinterpret_cast<unsigned *>(ws + lsync);
    la.wo_epi = EPI_RESSKIP;
    long long KA = 0;
    for (int q = 0; q < A.c.nseg; ++q) KA += A.c.seg[q].nch;
    const long long cols = (long long)A.ntz * A.c.g.T;
    TimerScope ts(WG_K_CONV_STORE + EPI_GATE, cx.st, A.c.M, KA, cols, 4 * cols * (KA + A.c.M / 2 + 2 * B.c.M) + 4LL * A.c.M * KA);
    WG_LAUNCH(cx, convlayer16h_kernel<EPI_RESSKIP>, dim3(grid), dim3(512), 0, la);
    g_layer_launches.fetch_add(1, std::memory_order_relaxed);
    return true;
#endif
}
// A layer's gate conv and residual product as ONE persistent launch (wg_layer16q.h) where the gate conv fills the chip with 256 x 128 tiles
// dealt by XCD rows in at least two whole rounds (the training shapes).  Both launches are only DESCRIBED (Ctx::cap), checked, and issued as
// convlayer16q_kernel; false = nothing was launched (the caller issues the two launches the ordinary way).
std::atomic<long long> g_layerq_launches{0};                  // diagnostics: launches of convlayer16q_kernel (wg_stat_layerq_launches)
template <class FA, class FB>
bool run_convlayer_big(Ctx &cx, float *ws, size_t lsync, FA &&gate_call, FB &&res_call)
{
#if defined(WG_OPT_NO_LAYERQ)
    return false;
#else
    if (cx.prec != 2 || cx.rec || cx.cap || cx.err) return false;
    {
        // OPT-IN (WG_LAYER_FUSION_BIG=1).  Measured at the headline shape (gpurun_out/r04g_bigfuse.txt, r04h_bisect.txt):
        // parity identical, 166-169 us per layer against 117.7 + 34.8 = 152.5 us for the two launches (step 65.9 against 64.3 ms).  The R tiles
        // cost 42 us of the launch -- twice their share of chunks: an 8-chunk tile is half overhead (the accumulate-into tile's round trip,
        // the epilogue, the poll), exactly as in the stand-alone residual launch -- and the G tiles run 8 % slower in the two-shape loop.
        if (!env_sw().layer_fusion_big) return false;
    }
    BigCap cg, cr;
    cg.ok = cr.ok = false;
    cx.cap = &cg;
    gate_call();
    cx.cap = &cr;
    res_call();
    cx.cap = nullptr;
    if (!cg.ok || !cr.ok || cx.err) return false;
    const ConvGemm16sArgs &A = cg.as, &R = cr.as;
    const int cus = device_cus();
    if (cus % 16 || A.nty != 2 || R.nty != 1 || A.ntx != R.ntx || A.ntz != R.ntz || A.xcd_items <= 0) return false;
    if (R.c.nseg != 1 || R.sseg[0].hi != A.s0.hi || R.c.seg[0].shift || !R.saux.hi || R.c.M > 256) return false;     // the residual's operand IS the gate
    const int xper = A.ntx * A.nty, xl = A.xcd_items * xper, xslots = cus / 8;
    if (xl % xslots || xl / xslots < 2) return false;        // whole rounds, at least two (an R tile needs an item between it and its own G tile)
    const int ncol = A.ntx * A.ntz;
    if ((size_t)ncol * WGL_SYNC_STRIDE > WGL_SYNC_WORDS) return false;
    ConvLayer16qArgs la;
    la.p[0] = A; la.p[1] = R;
    la.sync = reinterpret_cast<unsigned *>(ws + lsync);
    long long KA = 0, in_ch = 0;
    for (int q = 0; q < A.c.nseg; ++q) {
        KA += A.c.seg[q].nch;
        bool seen = false;
        for (int u = 0; u < q; ++u) seen = seen || A.sseg[u].hi == A.sseg[q].hi;
        if (!seen) in_ch += A.c.seg[q].nch;
    }
    const long long KR = R.c.seg[0].nch, cols = (long long)A.ntz * A.c.g.T;
    const long long Keff = KA + KR * R.c.M / std::max(1, A.c.M);
    // every operand plane once (h, y, the gate written and read back, tanh / sigmoid where saved, h in and out of the residual) + the weights
    const long long bytes = 4 * cols * (in_ch + (long long)A.c.M / 2 * (A.c.out1.p ? 4 : 2) + 2LL * R.c.M) + 4LL * A.c.M * KA + 4LL * R.c.M * KR;
    TimerScope ts(WG_K_LAYER, cx.st, A.c.M, Keff, cols, bytes);
    WG_LAUNCH(cx, convlayer16q_kernel, dim3(cus), dim3(1024), 0, la);
    g_layerq_launches.fetch_add(1, std::memory_order_relaxed);
    return true;
#endif
}

// The layer as ONE launch of convlayer16g_kernel (wg_gemm16g.h): a workgroup owns whole 192-column tiles, computes both 256-row gate tiles
// for them and then, from its own stores, the residual product -- nothing crosses workgroups.  The two launches are described through
// the same capture as above; returns false (nothing launched) when the shapes do not qualify.  env WG_LAYER_G=0 switches it off.
std::atomic<long long> g_layerg_launches{0};                  // diagnostics (wg_stat_layer_launches)
template <class FA, class FB>
bool run_convlayer_g(Ctx &cx, FA &&gate_call, FB &&res_call)
{
#if defined(WG_OPT_NO_G192) || defined(WG_OPT_NO_LAYERG)
    return false;
#else
    // (WG_LAYER_FUSION_BIG=1 asks for the older one-launch layer instead)
    if (!env_sw().layer_g || env_sw().layer_fusion_big) return false;
    if (!g192_on() || cx.prec != 2 || cx.rec || cx.cap || cx.err) return false;
    BigCap cg, cr;
    cg.ok = cr.ok = false;
    cx.cap = &cg;
    gate_call();
    cx.cap = &cr;
    res_call();
    cx.cap = nullptr;
    if (!cg.ok || !cr.ok || cx.err) return false;
    ConvLayer16gArgs la;
    la.p[0] = cg.as; la.p[1] = cr.as;
    ConvGemm16sArgs &A = la.p[0], &R = la.p[1];
    const Geo &g = A.c.g;
    const int cus = device_cus();
    // the residual's operand IS the gate (one K segment, no shift), it accumulates into an S-plane, both products over the same planes
    if (R.c.nseg != 1 || R.sseg[0].hi != A.s0.hi || R.c.seg[0].shift || !R.saux.hi || R.c.M != WGG_BM || A.c.M % WGG_BM || A.c.M / WGG_BM < 1) return false;
    if (g.rows != 0 || g.H < 64 || cus % 8 || R.c.g.B != g.B || R.c.g.Tt != g.Tt || R.c.g.P != g.P) return false;
    int ncA = 0, ncR = 0;
    for (int q = 0; q < A.c.nseg; ++q) ncA += (A.c.seg[q].nch + WG16_BK - 1) / WG16_BK;
    for (int q = 0; q < R.c.nseg; ++q) ncR += (R.c.seg[q].nch + WG16_BK - 1) / WG16_BK;
    // (a gate product of a few chunks -- layer 0 over xa, start_fold_on -- measured the same as two launches: 89.5 against 44.8 + 39.5 us,
    // gpurun_out/r06w; env WG_LAYER_MIN_CHUNKS lowers the bar for A/B runs)
    if (ncA < env_sw().layer_min_chunks || ncA > WGG_MAXCHUNKS || ncR < 2 || ncR > WGG_MAXCHUNKS) return false;
    if (!g192_fits(A, A.img_stride, A.c.nseg) || !g192_fits(R, R.img_stride, R.c.nseg)) return false;
    const int nct = (g.B * g.Tt + WGG_BN - 1) / WGG_BN;
    const int rounds = (nct + cus - 1) / cus;
    if (nct < cus || (double)(rounds * cus - nct) > 0.1 * rounds * cus) return false;       // column tiles deal out evenly over the CUs
    A.ntx = R.ntx = nct; A.nty = A.c.M / WGG_BM; R.nty = 1; A.ntz = R.ntz = 1; A.xcd_items = R.xcd_items = 1;
    long long KA = 0, in_ch = 0;
    for (int q = 0; q < A.c.nseg; ++q) {
        KA += A.c.seg[q].nch;
        bool seen = false;
        for (int u = 0; u < q; ++u) seen = seen || A.sseg[u].hi == A.sseg[q].hi;
        if (!seen) in_ch += A.c.seg[q].nch;
    }
    const long long KR = R.c.seg[0].nch, cols = (long long)g.B * g.T;
    const long long Keff = KA + KR * R.c.M / std::max(1, A.c.M);
    // every operand plane once (h, y, the gate written and read back, tanh / sigmoid where saved, h in and out of the residual) + the weights
    const long long bytes = 4 * cols * (in_ch + (long long)A.c.M / 2 * (A.c.out1.p ? 4 : 2) + 2LL * R.c.M) + 4LL * A.c.M * KA + 4LL * R.c.M * KR;
    TimerScope ts(WG_K_LAYER, cx.st, A.c.M, Keff, cols, bytes);
    if (A.part) { ++cx.part_written; g_gate_part_launches.fetch_add(1, std::memory_order_relaxed); }
    WG_LAUNCH(cx, convlayer16g_kernel, dim3(cus), dim3(512), 0, la);
    g_layerg_launches.fetch_add(1, std::memory_order_relaxed);
    return true;
#endif
}

// (the counters are left at zero by every launch; a call that was cut short -- an error half way -- is the reason they are cleared
// once at the start of every entry point that may use them)
// (only where one of the one-launch layer kernels can run at all: they are opt-in through WG_LAYER_FUSION / WG_LAYER_FUSION_BIG; without them
// every forward / inverse call paid a 64 KB memset for counters nobody reads)
void layer_sync_clear(Ctx &cx, float *ws, size_t lsync)
{
    // the hand-off counters of the opt-in one-launch layers: cleared whenever a call may use them.  (A call that ended in an error left them
    // wherever it stopped; the next call clears them here before anything reads them -- and a process that switched the opt-in off in
    // between does not read them at all.)
    if (!(env_sw().layer_fusion || env_sw().layer_fusion_big)) return;
    if (cx.prec == 2 && !cx.err && hipMemsetAsync(ws + lsync, 0, WGL_SYNC_WORDS * sizeof(unsigned), cx.st) != hipSuccess) cx.err = WG_ELAUNCH;
}

struct WnRun;
// layer i's gate conv (model/waveglow.py:42-44): dilated taps of h_i (plane / S-plane `hin`) + the conditioning -> gate (tanh, sigmoid
// kept where the pass saves them); keepg: every layer's gate has its own plane
static void wn_gate_conv(Ctx &cx, const WnRun &r, int i, int hin, bool keepg);
// The WN runs in the rank-2ic form of its skip path (lowrank_shape; wg_small.h weff_kernel): no skip sum, no dS.  One predicate for the
// forward, the recompute pass and the backward of a shape, so that a kept flow and a recomputed one produce the same bits: the shapes
// whose forward keeps every layer's gate anyway (the one-product skip sum's, fs below).
static bool lowrank_base(const Ctx &cx, const WnRun &r)
{
    const Geo &g = r.g;
    // (2 ic <= 8: where the gate convs can leave their share of `out` themselves.  WSRGlow -- 2 ic = 16, 6 144 columns per launch -- runs
    // the form that reads the gate planes, on 96 workgroups: measured 41.9 against 40.7 ms per step, gpurun_out/r06j_wsr_ab.txt)
    return env_sw().lowrank && cx.prec == 2 && lowrank_shape(r.d) && (2 * r.d.ic <= 8 || env_sw().lowrank_all) && r.L.effT && !cx.rec &&
           !cx.row_sel1 && (g.rows == 0 || r.d.mode2d) && g.B * g.Tt >= WG_FUSED_SKIP_MIN_COLS;
}
// Do this WN's gate convs leave their share of `out` (ConvGemm16sArgs::part)?  Asked of run_convgemm itself (Ctx::probe: the launch is
// described, routed, and not run), so that the answer cannot drift from the routing: 1 = the kernel that writes the partial rows.
static bool gate_parts_on(Ctx &cx, const WnRun &r);
// (WaveFlow's WN2D, mode2d: only together with the partial rows -- its coupling kernel has no form that reads the gate planes)
static bool lowrank_on(Ctx &cx, const WnRun &r) { return lowrank_base(cx, r) && (!r.d.mode2d || gate_parts_on(cx, r)); }
// where end_affine_kernel takes `out` from: 0 = W_end . S (the skip plane), 1 = sum_l Weff_l gate_l straight from the gate planes,
// 2 = the partial rows the gate convs left
static int affine_source(Ctx &cx, const WnRun &r, AffineArgs &a)
{
    a.bias = r.d.bias ? r.pk + r.L.bias_end : nullptr;
    a.Cs = r.d.Cs; a.ic = r.d.ic;
    if (!lowrank_on(cx, r)) {
        a.endT = r.pk + r.L.endT;
        a.S = pref(r.ws + r.w.skip, r.d.Cs);
        return 0;
    }
    a.endT = r.pk + r.L.effT;
    a.S = pnull();
    if (gate_parts_on(cx, r)) {
        a.part = r.ws + r.w.gpart;
        a.nsrc = r.d.depth * gate_part_slots(r.d);
        if (r.w.gpart_step != (size_t)gate_part_slots(r.d) * r.g.B * r.g.Tt * gate_part_prow(r.d) && !cx.err) cx.err = WG_EINVAL;      // (the sources are one array)
        return 2;
    }
    for (int i = 0; i < r.d.depth; ++i) a.gS[i] = (const unsigned short *)(r.ws + r.w.gateS[i]);
    a.g_lo_off = (size_t)r.g.B * r.d.Cd * r.g.P;
    a.Cd = r.d.Cd; a.nl = r.d.depth;
    return 1;
}
static void launch_end_affine(Ctx &cx, const AffineArgs &a, int src)
{
    const dim3 grid(a.g.Tt / WG_AFF_T, a.g.B);
    // (timed as its own class where it replaces the skip sum: what it adds per column, and the bytes of that + the flow's channels in and out)
    const long long cols = (long long)a.g.B * a.g.T, kk = src == 2 ? 8LL * a.nsrc : src == 1 ? (long long)a.Cd * a.nl : a.Cs;
    TimerScope ts(src ? WG_K_THIN : -1000, cx.st, 2 * a.ic, kk, cols, 4 * cols * (kk + 4 * a.ic));
    if (src == 2) {
        WG_LAUNCH(cx, (end_affine_kernel<8, false, 2>), grid, dim3(256), 0, a);
    } else if (src == 1) {
        if (2 * a.ic <= 8) WG_LAUNCH(cx, (end_affine_kernel<8, false, 1>), grid, dim3(256), 0, a);
        else WG_LAUNCH(cx, (end_affine_kernel<32, false, 1>), grid, dim3(256), 0, a);
    } else {
        if (2 * a.ic <= 8) WG_LAUNCH(cx, end_affine_kernel<8>, grid, dim3(256), 0, a);
        else WG_LAUNCH(cx, end_affine_kernel<32>, grid, dim3(256), 0, a);
    }
}

// Layer 0's dilated conv over xa through the composed weight (start_fold_shape): one predicate for every pass of a shape -- forward,
// recompute, inverse -- so that a kept flow and a recomputed one produce the same bits.  (Not inside the recorded row walk; env
// WG_START_FOLD=0 restores the conv over h_0.)
static bool start_fold_on(const Ctx &cx, const WnRun &r)
{
    return env_sw().start_fold && cx.prec == 2 && start_fold_shape(r.d) && r.L.Acat0x && !cx.rec && !cx.row_sel1 && (r.g.rows == 0 || r.d.mode2d);
}
static void wn_gate_conv(Ctx &cx, const WnRun &r, int i, int hin, bool keepg)
{
    const WnD &d = r.d;
    const Geo &g = r.g;
    float *ws = r.ws;
    const bool sp = cx.prec == 2;
    const int nb = d.bias ? 1 : 0;
    float *Hin = ws + r.w.H[hin];
    float *gate = ws + r.w.gate[keepg ? i : 0];
    const float *gateS = ws + r.w.gateS[keepg ? i : 0];
    SegSpec sg[WG_MAX_SEG];
    int ns = 0;
    const bool fold = i == 0 && start_fold_on(cx, r);
    for (int kt = 0; kt < d.radix; ++kt) {
        int ts, ro;
        d.tap(i, kt, ts, ro);
        if (fold) sg[ns++] = {r.X.p, r.X.Cp, r.X.ch0, r.L.kp_start, ts, ws + r.w.XaS, r.L.kp_start, 0, ro, 0};    // (xa's S-plane: wn_forward)
        else sg[ns++] = {Hin, d.C, 0, d.C, ts, ws + r.w.HS[hin], d.C, 0, ro, 0};
    }
#if defined(WG_DBG_NOCOND)      // timing experiment only (results are garbage): the gate conv without its conditioning segment
    if (false)
#endif
    sg[ns++] = {r.Y, d.auxp(), 0, d.auxp(), 0, r.YS, d.auxp(), 0, 0, d.mode2d};
    if (nb) sg[ns++] = ones_seg(cx, r, false);               // (wn_forward filled the plane of ones at the start of the pass)
    run_convgemm(cx, g, r.pk + (fold ? r.L.Acat0x : r.L.Acat[i]), r.L.ld_Acat, 2 * d.Cd, sg, ns, EPI_GATE, sp ? pnull() : pref(gate, d.Cd),
                 (r.save && !tw_from_gate(cx)) ? pref(ws + r.w.tw[i], d.Cd) : pnull(), r.save ? pref(ws + r.w.sf[i], d.Cd) : pnull(),
                 pnull(), pnull(), 0, 0, sp ? sref(g, gateS, d.Cd) : snull());             // waveglow.py:42-44
}
// the weight-gradient slab of a training workspace is idle while a WN runs forward (wn_backward's queue is flushed when it returns): scratch
// for a split gate conv's parts.  `asked_outside`: the routing probe asks what a FORWARD pass of this WN does from wherever the answer is
// needed -- behind wn_forward (the end conv's source), inside wn_backward (its queue owns the slab then) -- and must see what that pass saw:
// without it a WN whose gate convs are cut along K (no partial rows) was told, after the pass, that they had been written.
struct GslabScope {
    Ctx &c;
    float *p0;
    size_t n0;
    GslabScope(Ctx &cx_, const WnRun &r, bool asked_outside) : c(cx_), p0(cx_.gslab), n0(cx_.gslab_floats)
    {
        const size_t n = (r.w.slab_floats && (asked_outside || !c.fq)) ? r.w.slab_floats : 0;
        c.gslab = n ? r.ws + r.w.slab : nullptr; c.gslab_floats = n;
    }
    ~GslabScope() { c.gslab = p0; c.gslab_floats = n0; }
};
static bool gate_parts_on(Ctx &cx, const WnRun &r)
{
    if (!lowrank_base(cx, r) || !gate_parts_shape(r.d) || !r.w.gpart_step || !r.L.effA || cx.probe) return false;
    if (env_sw().layer_fusion_big) return false;              // (the opt-in one-launch layer on 256 x 128 tiles has its own gate epilogue)
    GslabScope gslab_scope(cx, r, true);
    int route = 0;
    cx.probe = &route;
    cx.gate_eff = r.pk + r.L.effA; cx.gate_part = r.ws + r.w.gpart; cx.gate_prow = gate_part_prow(r.d);
    wn_gate_conv(cx, r, 0, 0, true);
    cx.probe = nullptr; cx.gate_eff = nullptr; cx.gate_part = nullptr; cx.gate_prow = 8;
    return route == 1;
}

void wn_forward(Ctx &cx, const WnRun &r)
{
    const WnD &d = r.d;
    const Geo &g = r.g;
    float *ws = r.ws;
    const bool sp = cx.prec == 2;
    GslabScope gslab_scope(cx, r, false);                     // (scratch for a split gate conv's parts)
    const int nb = d.bias ? 1 : 0;
    const SegSpec sone = d.bias ? ones_seg(cx, r, true) : SegSpec{};
    const int cols0 = (cx.row_sel1 && g.rows > 0 ? g.B / g.rows : g.B) * g.Tt;
    const bool so = s_only_chain(cx, d) && fused_skip(d) && cols0 >= WG_FUSED_SKIP_MIN_COLS;      // residual stream as S-planes only
    // WN.start on the vector ALU in ONE launch (start_fwd_kernel, wg_thin.h) instead of an S-plane conversion + an MFMA conv on a K of 2-4
    // channels; a pass that keeps its activations takes it only where the backward's thin start product reads xa itself (thin_shape_ok)
    // -- the MFMA weight gradient would want xa's S-plane
    const bool vstart = sp && !nb && !cx.rec && d.ic <= 16 && d.C % 8 == 0 && (!r.save || thin_shape_ok(d))
#if defined(WG_OPT_NO_VSTART)
                        && false
#endif
        ;
    if (r.start_done) {
        // (the seam launch of the flow visited before wrote h_0: same arithmetic as start_fwd_kernel below)
        if (start_fold_on(cx, r)) run_to_splane(cx, g, r.X, d.ic, ws + r.w.XaS, r.L.kp_start);      // (layer 0 reads xa itself: wn_gate_conv)
    } else if (vstart) {
        StartFwdArgs a;
        memset(&a, 0, sizeof(a));
        a.X = r.X; a.W = r.pk + r.L.startN; a.ldw = r.L.ld_startN; a.C = d.C; a.ic = d.ic;
        a.H = so ? pnull() : pref(ws + r.w.H[0], d.C);
        a.HS = sref(g, ws + r.w.HS[0], d.C);
        if (start_fold_on(cx, r)) a.XS = sref(g, ws + r.w.XaS, r.L.kp_start);      // (layer 0 reads xa itself: wn_gate_conv)
        a.g = g; a.row_sel1 = cx.row_sel1;
        WG_LAUNCH(cx, start_fwd_kernel, dim3((g.T + 255) / 256, d.C / 8, cx.row_sel1 && g.rows > 0 ? g.B / g.rows : g.B), dim3(256), 0, a);
    } else {
        if (sp) run_to_splane(cx, g, r.X, d.ic, ws + r.w.XaS, r.L.kp_start);      // xa -> S-plane (re-based to channel 0)
        SegSpec s0[2] = {{r.X.p, r.X.Cp, r.X.ch0, r.L.kp_start, 0, ws + r.w.XaS, r.L.kp_start, 0}, sone};
        run_convgemm(cx, g, r.pk + r.L.startT, r.L.ld_startT, d.C, s0, 1 + nb, EPI_STORE, so ? pnull() : pref(ws + r.w.H[0], d.C), pnull(), pnull(),
                     pnull(), pnull(), 0, 0, sp ? sref(g, ws + r.w.HS[0], d.C) : snull());             // waveglow.py:99
    }
    // one long product (depth x Cd / 32 chunks in a row) only pays where launches are bound by bytes, not by their chunk latency chain:
    // single-utterance synthesis (2 048 columns) lost 9 % with it, the training shapes gain 2.5 % per step
    const int cols = (cx.row_sel1 && g.rows > 0 ? g.B / g.rows : g.B) * g.Tt;
    const bool fs = fused_skip(d) && cols >= WG_FUSED_SKIP_MIN_COLS;
    // (lowrank) the gate convs leave their share of `out` themselves: asked once per pass; every layer must then have written its rows
    const bool gparts = fs && gate_parts_on(cx, r);
    const int written0 = cx.part_written;
    for (int i = 0; i < d.depth; ++i) {
        const int hin = r.save ? i : (i & 1), hout = r.save ? std::min(i + 1, d.depth - 1) : ((i + 1) & 1);
        float *Hin = ws + r.w.H[hin], *Hout = ws + r.w.H[hout];
        float *gate = ws + r.w.gate[(r.save || fs) ? i : 0];
        const float *gateS = ws + r.w.gateS[(r.save || fs) ? i : 0];
        // fp32 gate plane: only the on-the-fly weight-gradient kernel still reads it (backward); the S-plane feeds W_o
        auto gate_call = [&]() {
            if (gparts) {
                cx.gate_eff = r.pk + r.L.effA + (size_t)i * (d.Cd / 32) * 256; cx.gate_part = ws + r.w.gpart + (size_t)i * r.w.gpart_step;
                cx.gate_prow = gate_part_prow(d);
            }
            wn_gate_conv(cx, r, i, hin, r.save || fs);
            cx.gate_eff = nullptr; cx.gate_part = nullptr; cx.gate_prow = 8;
        };
        SegSpec sgt[2] = {{gate, d.Cd, 0, d.Cd, 0, gateS, d.Cd, 0}, sone};
        const int last = i == d.depth - 1;
        auto wo_call = [&]() {
            run_convgemm(cx, g, r.pk + r.L.WoT[i], r.L.ld_WoT[i], d.wo_rows(i), sgt, 1 + nb, EPI_RESSKIP, pref(Hout, d.C),
                         pref(ws + r.w.skip, d.Cs), pnull(), pref(Hin, d.C), pnull(), last ? 0 : d.C, i > 0,
                         (sp && !last) ? sref(g, ws + r.w.HS[hout], d.C) : snull());               // :45-46,104
        };
        // the whole layer as ONE launch where both products are small-grid launches (single-utterance synthesis, WaveFlow's row steps)
        if (!fs && !r.save && !nb && run_convlayer(cx, ws, r.w.lsync, gate_call, wo_call)) continue;
        if (fs) {
            // residual rows only: h_{i+1} = h_i + Wres_i gate_i (the first C rows of W_o); the skip rows of all layers follow in one product
            auto res_call = [&]() {
                run_convgemm(cx, g, r.pk + r.L.WoT[i], r.L.ld_WoT[i], d.C, sgt, 1 + nb, EPI_STORE, so ? pnull() : pref(Hout, d.C), pnull(), pnull(),
                             so ? pnull() : pref(Hin, d.C), pnull(), 0, 0, sp ? sref(g, ws + r.w.HS[hout], d.C) : snull(),
                             so ? sref(g, ws + r.w.HS[hin], d.C) : snull());                                   // :45-46
            };
            // gate conv + residual product as ONE persistent launch where the gate conv fills the chip in whole rounds (wg_layer16q.h)
            if (!last && so && !nb && run_convlayer_g(cx, gate_call, res_call)) continue;
            if (!last && so && !nb && run_convlayer_big(cx, ws, r.w.lsync, gate_call, res_call)) continue;
            gate_call();
            if (!last) res_call();
            continue;
        }
        gate_call();
        wo_call();
    }
    if (gparts && cx.part_written - written0 != d.depth && !cx.err) cx.err = WG_ELAUNCH;      // (a gate conv took another kernel than the probe said)
    if (fs && lowrank_on(cx, r)) return;                      // (the end conv reads the gate planes or the partial rows: affine_source)
    if (fs) {                                                 // cum_skip = sum_i skip_i (waveglow.py:104) = [Wskip_0 .. Wskip_{d-1}] [gate_0; ..; gate_{d-1}]
        SegSpec sk[WG_MAX_SEG];
        for (int i = 0; i < d.depth; ++i) sk[i] = {ws + r.w.gate[i], d.Cd, 0, d.Cd, 0, ws + r.w.gateS[i], d.Cd, 0};
        if (nb) sk[d.depth] = sone;
        run_convgemm(cx, g, r.pk + r.L.WskT, r.L.ld_WskT, d.Cs, sk, d.depth + nb, EPI_STORE, pref(ws + r.w.skip, d.Cs), pnull(), pnull(),
                     pnull(), pnull(), 0, 0);
    }
}

// fused WN.end + coupling (mode AFF_*)
void run_end_affine(Ctx &cx, const WnRun &r, int mode, PRef dX, float *log_s_out, const float *dls_plain, const float *dld,
                    float *partial)
{
    AffineArgs a;
    memset(&a, 0, sizeof(a));
    const int fromg = affine_source(cx, r, a);
    a.X = r.X; a.dX = dX;
    a.Gp = pref(r.ws + r.w.G, r.L.kp_end);
    a.log_s_out = log_s_out; a.dls_plain = dls_plain; a.dld = dld; a.partial = partial;
    a.g = r.g; a.mode = mode;
    launch_end_affine(cx, a, fromg);
}

// Synthesis, between two WNs of the inverse direction: AFF_REV of flow k, its inverse 1x1 conv and WN.start of the flow visited next as
// ONE launch (end_affine_kernel<8, true>) instead of three dependent ~5 us ones -- what VERDICT r04 #6 asked to be cut.  Built, x bit
// for bit the three launches' (tests/test_gpu_parity.py), and MEASURED: 16.0 us against 8.8 + 4.9 + 5.1 per flow, a 0.7 s synthesis call
// 2.455-2.480 against 2.443-2.555 ms (gpurun_out/r05s.txt): within the noise of the boxes, because a 32-workgroup launch is a chain of
// round trips either way and the tail adds two of them.  Opt-in (env WG_INV_SEAM=1); the call's time is in its 192 conv launches.
// `nxt` is the next WN's run (its X already re-based); returns false where the shapes are outside the seam kernel's.
bool run_inv_seam(Ctx &cx, const WnRun &r, const float *Winv, float *partial, const WnRun &nxt)
{
    if (!env_sw().inv_seam || lowrank_on(cx, r)) return false;
    const WnD &d = r.d, &n = nxt.d;
    const Geo &g = r.g;
    const int rel = nxt.X.ch0 - r.X.ch0;
    if (cx.prec != 2 || cx.rec || cx.row_sel1 || r.save || nxt.save || d.bias || n.bias || g.rows != 0 || 2 * d.ic > 8 || n.ic > 8 || n.C % 8 ||
        n.C * n.ic > WG_SEAM_MAXW || nxt.X.p != r.X.p || nxt.X.Cp != r.X.Cp || r.X.ch0 + rel < 0)
        return false;
    const int cols0 = g.B * g.Tt;
    const bool so = s_only_chain(cx, n) && fused_skip(n) && cols0 >= WG_FUSED_SKIP_MIN_COLS;      // (as wn_forward decides it for the next WN)
    AffineArgs a;
    memset(&a, 0, sizeof(a));
    a.endT = r.pk + r.L.endT;
    a.S = pref(r.ws + r.w.skip, d.Cs);
    a.Cs = d.Cs; a.ic = d.ic;
    a.X = r.X;
    a.partial = partial;
    a.g = g; a.mode = AFF_REV;
    a.mixM = Winv;
    a.nW = nxt.pk + nxt.L.startN; a.n_ldw = nxt.L.ld_startN; a.n_C = n.C; a.n_ic = n.ic; a.n_rel = rel;
    a.nH = so ? pnull() : pref(nxt.ws + nxt.w.H[0], n.C);
    a.nHS = sref(g, nxt.ws + nxt.w.HS[0], n.C);
    WG_LAUNCH(cx, (end_affine_kernel<8, true>), dim3(g.Tt / WG_AFF_T, g.B), dim3(256), 0, a);
    return true;
}

bool thin_ok(const Ctx &cx, const WnD &d)
{
#if defined(WG_OPT_NO_THIN)
    return false;
#else
    return cx.prec == 2 && cx.fq && !cx.rec && thin_shape_ok(d);
#endif
}
static int thin_grid(int tiles)
{
    const int slots = 2 * device_cus(), rounds = (tiles + slots - 1) / slots;
    return (tiles + rounds - 1) / rounds;
}
// dW_start = sum dh_0 (x) xa (finalised into dg / dv) ; dxa += W_start^T dh_0
void run_thin_start(Ctx &cx, const WnRun &r, const float *dhS, PRef dX, const float *gp, const float *vp, float *dg, float *dv)
{
    const WnD &d = r.d;
    const Geo &g = r.g;
    const int icp = thin_icp(d);
    ThinStartArgs a;
    memset(&a, 0, sizeof(a));
    a.dh = sref(g, const_cast<float *>(dhS), d.C);
    a.X = r.X; a.dX = dX;
    a.W = r.pk + r.L.startN; a.ldw = r.L.ld_startN; a.C = d.C; a.ic = d.ic;
    a.g = g; a.tiles = g.B * (g.Tt / WGTH_TB);
    const int grid = thin_grid(a.tiles), n = d.C * icp;
    a.part = cx.fq->reserve(wgth_part_floats(grid, n));
    float *out = a.part + (size_t)grid * n;
    const size_t lds = thin_start_lds(d);
    if (cx.err) return;
    switch (icp) {
#define WG_THIN_CASE(P, SLOT)                                                                                         \
    case P:                                                                                                           \
        cx.err = ensure_dynamic_lds((const void *)thin_start_kernel<P>, SLOT, lds);                                   \
        if (!cx.err) WG_LAUNCH(cx, thin_start_kernel<P>, dim3(grid), dim3(WGTH_THREADS), lds, a);                      \
        break;
        WG_THIN_CASE(4, 2) WG_THIN_CASE(8, 3) WG_THIN_CASE(16, 4)
#undef WG_THIN_CASE
    }
    WG_LAUNCH(cx, thin_fold_kernel, dim3(n / 32), dim3(256), 0, (const float *)a.part, grid, n, out);
    WgradOut wo;
    wo.nsplit = 1; wo.Mp = d.C; wo.Np = icp; wo.slab = out;
    run_finalize(cx, out, wo, 0, d.C, d.ic, 1, 0, 1, 0, gp, vp, dg, dv);
}
// dW_end = sum G (x) skip ; dS = W_end^T G as an S-plane
void run_thin_end(Ctx &cx, const WnRun &r, float *G, int Gc, float *skip, float *dSS, float *dW)
{
    const WnD &d = r.d;
    const Geo &g = r.g;
    const int k2 = 2 * d.ic, k2p = thin_k2p(d);
    ThinEndArgs a;
    memset(&a, 0, sizeof(a));
    a.G = pref(G, Gc); a.skip = pref(skip, d.Cs); a.dS = sref(g, dSS, d.Cs);
    a.W = r.pk + r.L.endN; a.ldw = r.L.ld_endN; a.Cs = d.Cs; a.K2 = k2;
    a.g = g; a.tiles = g.B * (g.Tt / WGTH_TB);
    const int grid = thin_grid(a.tiles), n = k2p * d.Cs;
    a.part = cx.fq->reserve(wgth_part_floats(grid, n));
    float *out = a.part + (size_t)grid * n;
    const size_t lds = thin_end_lds(d);
    if (cx.err) return;
    switch (k2p) {
#define WG_THIN_CASE(P, SLOT)                                                                                         \
    case P:                                                                                                           \
        cx.err = ensure_dynamic_lds((const void *)thin_end_kernel<P>, SLOT, lds);                                     \
        if (!cx.err) WG_LAUNCH(cx, thin_end_kernel<P>, dim3(grid), dim3(WGTH_THREADS), lds, a);                        \
        break;
        WG_THIN_CASE(8, 5) WG_THIN_CASE(16, 6) WG_THIN_CASE(32, 7)
#undef WG_THIN_CASE
    }
    WG_LAUNCH(cx, thin_fold_kernel, dim3(n / 32), dim3(256), 0, (const float *)a.part, grid, n, out);
    WgradOut wo;
    wo.nsplit = 1; wo.Mp = k2p; wo.Np = d.Cs; wo.slab = out;
    run_finalize(cx, out, wo, 0, k2, d.Cs, 1, 0, 1, 0, nullptr, nullptr, nullptr, dW);
}

// The skip path's gradients in their rank-2ic form (wg_thin.h pgate_kernel): P_l = sum G (x) gate_l for every layer in one pass over the
// gate planes, then dWskip_l = W_end^T P_l (returned: [depth][Cs][Cd] in the finalisation queue's arena, for the caller's run_finalize of
// W_o's skip rows) and dW_end = sum_l P_l Wskip_l^T (finalised here).
static const float *run_lowrank_end(Ctx &cx, const WnRun &r, const float *const *p, float *dWend)
{
    const WnD &d = r.d;
    const Geo &g = r.g;
    const int ic2 = 2 * d.ic, mrows = rup(ic2, 8), nblk = g.B * (g.Tt / 64);
    // column ranges: 8 where the unit rows alone give 64 workgroups per range (the 256-channel WN: 512 workgroups), more where they do not
    // (WaveFlow's 64 channels: 16 workgroups per range -- with 8 ranges half the chip ran this pass: 199 us for the bytes the headline's takes 105 for)
#if defined(WG_OPT_PGATE_NCR8)                           // A/B build: eight ranges whatever the channel count (before round 6's last commit)
    const int want = 8;
#else
    const int wgr = (d.depth * (d.Cd / 8) + 3) / 4, want = std::min(32, std::max(8, (512 + wgr - 1) / wgr));
#endif
    const int ncr = std::max(1, std::min(want, nblk / 4)), per = (nblk + ncr - 1) / ncr;
    const size_t n = (size_t)d.depth * mrows * d.Cd;
    PGateArgs a;
    memset(&a, 0, sizeof(a));
    for (int i = 0; i < d.depth; ++i) a.gS[i] = (const unsigned short *)(r.ws + r.w.gateS[i]);
    a.g_lo_off = (size_t)g.B * d.Cd * g.P;
    a.G = pref(r.ws + r.w.G, r.L.kp_end);
    a.Cd = d.Cd; a.nl = d.depth; a.ic2 = ic2; a.mrows = mrows;
    a.g = g; a.nblk = nblk; a.per = per;
    // partials, P, the dWskip matrices and dW_end: ONE reservation (nothing of it may be recycled before the queued finalisations ran)
    const size_t fl = wgth_part_floats(ncr, (int)n) + rupz((size_t)d.depth * d.Cs * d.Cd, 64) + (size_t)32 * d.Cs;
    a.part = cx.fq->reserve(fl);
    if (cx.err) return nullptr;
    float *P = a.part + (size_t)ncr * n, *dWsk = P + rupz(n, 64), *dWe = dWsk + rupz((size_t)d.depth * d.Cs * d.Cd, 64);
    {
        const long long cols = (long long)g.B * g.T, kk = (long long)d.depth * d.Cd;
        TimerScope ts(WG_K_THIN, cx.st, ic2, kk, cols, 4 * cols * (kk + ic2));
        WG_LAUNCH(cx, pgate_kernel, dim3((d.depth * (d.Cd / 8) + 3) / 4, ncr, mrows / 8), dim3(256), 0, a);
    }
    WG_LAUNCH(cx, thin_fold_kernel, dim3((unsigned)(n / 32)), dim3(256), 0, (const float *)a.part, ncr, (int)n, P);
    LrFinArgs f;
    memset(&f, 0, sizeof(f));
    f.P = P; f.wE = p[4 + 4 * d.depth];
    for (int i = 0; i < d.depth; ++i) {
        const int r0 = d.wo_rows(i) - d.Cs;
        f.v[i] = p[7 + 4 * i] + (size_t)r0 * d.Cd;
        f.scale[i] = r.pk + r.L.scale_Wo[i] + r0;
    }
    f.dWsk = dWsk; f.dWend = dWe;
    f.nl = d.depth; f.Cs = d.Cs; f.Cd = d.Cd; f.ic2 = ic2; f.mrows = mrows;
    WG_LAUNCH(cx, lr_dwsk_kernel, dim3((d.Cs * d.Cd + 255) / 256, d.depth), dim3(256), 0, f);
    WG_LAUNCH(cx, lr_dwend_kernel, dim3(d.Cs), dim3(256), 0, f);
    WgradOut wo;
    wo.nsplit = 1; wo.Mp = 32; wo.Np = d.Cs; wo.slab = dWe;
    run_finalize(cx, dWe, wo, 0, ic2, d.Cs, 1, 0, 1, 0, nullptr, nullptr, nullptr, dWend);
    return dWsk;
}

// backward through WN given the G plane (what autograd.grad at efficient_modules.py:143 evaluates).
// p/grads: this WN's parameter / gradient tables.  dX: gradient plane at the same ch0 as r.X (dxa accumulates into it).
void wn_backward(Ctx &cx, const WnRun &r, const float *const *p, float *const *grads, PRef dX, float *dY)
{
    const WnD &d = r.d;
    const Geo &g = r.g;
    float *ws = r.ws;
    const int nd = d.depth;
    float *slab = ws + r.w.slab;
    const size_t cap = r.w.slab_floats;
    float *G = ws + r.w.G, *dS = ws + r.w.dS, *dH = ws + r.w.dH, *skip = ws + r.w.skip;
    const bool fdy = dY && fused_dy(d);                       // every layer keeps its dxy; dy is one product after the loop
    const int Gc = r.L.kp_end;
    const bool sp = cx.prec == 2;
    // the layers' weight gradients as ONE grouped launch behind the loop (every layer's dxy and dh is kept then: dxyS_step, dHS_step)
    const bool gw = grouped_wgrad(cx.prec, d) && r.w.dHS_step && r.w.dxyS_step && nd >= 2;
    auto dHSp = [&](int j) { return ws + r.w.dHS + (size_t)j * r.w.dHS_step; };      // S-plane of dh_j (one plane for all j unless gw)
    WgradGroupSpec gsT[WG_GRP_MAX], gsO[WG_GRP_MAX], gsV[WG_GRP_MAX];
    // WN2D: the conditioning is broadcast over the height axis, so dV_i = (sum over an item's rows of dxy_i) (x) y -- the row sums exist
    // anyway (the conditioning's own gradient is taken from them).  With a plane of them per layer the conditioning leaves the big
    // weight-gradient product (672 -> 576 columns at the shipped width: five column tiles instead of six) and dV becomes one small grouped
    // product over items x T columns instead of plane rows x T.
    const bool hv = d.mode2d && dY && sp && gw && r.rs_step > 0 && !d.bias;
    auto rsp = [&](int j) { return r.rs + (size_t)j * (hv ? r.rs_step : 0); };
    // WnD::bias: the plane of ones closes the B side of every weight-gradient product; column 0 of its 32-column block is the bias gradient
    const int nb = d.bias ? 1 : 0;
    WSegSpec wone = {nullptr, 32, 0, 32, 0, nullptr, 32, 0};
    if (nb) {
        const SegSpec so1 = ones_seg(cx, r, true);
        wone.src = so1.src; wone.s = so1.s;
    }
    auto fin_bias = [&](const float *slabp, const WgradOut &wo, int row0, int rows, int col0, float *db) {
        if (nb && db) run_finalize(cx, slabp, wo, row0, rows, 1, 1, col0, 1, 0, nullptr, nullptr, nullptr, db);
    };
    auto gb = [&](int j) -> float * { return nb ? grads[d.pb(j)] : nullptr; };      // gradient of bias j (WnD::pb), nullable
#if !defined(WG_OPT_NO_FIN_BATCH)
    FinQueue fq(cx, slab, cap);                               // flushed when it goes out of scope: before the caller's next launch
#endif
    // the rank-2ic form of the skip path (lowrank_on): neither S nor dS = W_end^T G exists; G itself (as an S-plane of kp_end channels)
    // is the K segment that stands for dS in every gate backward, and the skip rows' weight gradients come from P_l = G gate_l^T
    const bool lr = lowrank_on(cx, r);
    const float *dWsk = nullptr;
    // end: dW_end = sum G (x) S ; dS = W_end^T G
    if (lr) {
        run_to_splane(cx, g, pref(G, Gc), Gc, ws + r.w.GS, Gc);
        dWsk = run_lowrank_end(cx, r, p, grads[4 + 4 * nd]);
        // W_o's skip rows: dWskip_i = W_end^T P_i, [Cs][Cd] at dWsk + i Cs Cd; parameter rows [wo_rows - Cs, wo_rows).  Queued at once:
        // the matrices live in the finalisation queue's arena, and a later reservation that wraps it flushes what is queued first
        for (int i = 0; i < nd && dWsk; ++i) {
            const int r0 = d.wo_rows(i) - d.Cs;
            WgradOut wk;
            wk.nsplit = 1; wk.Mp = d.Cs; wk.Np = d.Cd; wk.slab = const_cast<float *>(dWsk) + (size_t)i * d.Cs * d.Cd;
            run_finalize(cx, wk.slab, wk, 0, d.Cs, d.Cd, 1, 0, 1, 0, p[6 + 4 * i] ? p[6 + 4 * i] + r0 : nullptr, p[7 + 4 * i] + (size_t)r0 * d.Cd,
                         grads[6 + 4 * i] ? grads[6 + 4 * i] + r0 : nullptr, grads[7 + 4 * i] ? grads[7 + 4 * i] + (size_t)r0 * d.Cd : nullptr);
        }
    } else {
        // skip (fp32 only: it feeds the fp32 end conv) has no S-plane -> this small product runs on the on-the-fly kernel
        if (thin_ok(cx, d)) run_thin_end(cx, r, G, Gc, skip, ws + r.w.dSS, grads[4 + 4 * nd]);      // both in one pass over skip (wg_thin.h)
        else {
            WSegSpec sa = {G, Gc, 0, 2 * d.ic, 0, nullptr, 0, 0}, sb[2] = {{skip, d.Cs, 0, d.Cs, 0, nullptr, 0, 0}, wone};
            sb[1].s = nullptr;                                 // (skip has no S-plane: the on-the-fly kernel reads the fp32 ones)
            WgradOut wo = run_wgrad(cx, g, &sa, 1, sb, 1 + nb, slab, cap);
            run_finalize(cx, slab, wo, 0, 2 * d.ic, d.Cs, 1, 0, 1, 0, nullptr, nullptr, nullptr, grads[4 + 4 * nd]);
            fin_bias(slab, wo, 0, 2 * d.ic, rup(d.Cs, 32), gb(2 + 2 * nd));
            if (sp) run_to_splane(cx, g, pref(G, Gc), Gc, ws + r.w.GS, Gc);
            SegSpec s = {G, Gc, 0, Gc, 0, ws + r.w.GS, Gc, 0};
            run_convgemm(cx, g, r.pk + r.L.endN, r.L.ld_endN, d.Cs, &s, 1, EPI_STORE, sp ? pnull() : pref(dS, d.Cs), pnull(), pnull(), pnull(), pnull(), 0, 0,
                         sp ? sref(g, ws + r.w.dSS, d.Cs) : snull());
        }
    }
    for (int i = nd - 1; i >= 0; --i) {
        const int rows = d.wo_rows(i), last = i == nd - 1;
        float *Hi = ws + r.w.H[i], *gate = ws + r.w.gate[i];
        float *dxy = ws + r.w.dxy + (size_t)i * r.w.dxy_step;                    // (step 0: one buffer for all layers)
        float *dxyS = ws + r.w.dxyS + (size_t)i * r.w.dxyS_step;
        // dW_o = sum do (x) gate,  do = last ? dS : cat(dh_{i+1}, dS)
        if (gw) {                                                              // (the last layer's group has no dh rows: zero, skipped by row0)
            gsO[i].sa[0] = {nullptr, d.C, 0, d.C, 0, last ? nullptr : dHSp(i + 1), d.C, 0};
            gsO[i].sa[1] = {nullptr, d.Cs, 0, d.Cs, 0, ws + r.w.dSS, d.Cs, 0};          // (lowrank: the group has its dh rows only, nsa = 1 below)
            gsO[i].sb[0] = {nullptr, d.Cd, 0, d.Cd, 0, ws + r.w.gateS[i], d.Cd, 0};
            if (nb) gsO[i].sb[1] = wone;
        } else if (lr) {
            if (!last) {                                                         // the residual rows' product; the skip rows come from P
                WSegSpec sa = {dH, d.C, 0, d.C, 0, dHSp(i + 1), d.C, 0};
                WSegSpec sb = {gate, d.Cd, 0, d.Cd, 0, ws + r.w.gateS[i], d.Cd, 0};
                WgradOut wo = run_wgrad(cx, g, &sa, 1, &sb, 1, slab, cap);
                run_finalize(cx, slab, wo, 0, d.C, d.Cd, 1, 0, 1, 0, p[6 + 4 * i], p[7 + 4 * i], grads[6 + 4 * i], grads[7 + 4 * i]);
            }
        } else {
            WSegSpec sa[2];
            int nsa = 0;
            if (!last) sa[nsa++] = {dH, d.C, 0, d.C, 0, sp ? dHSp(i + 1) : nullptr, d.C, 0};
            sa[nsa++] = {dS, d.Cs, 0, d.Cs, 0, sp ? ws + r.w.dSS : nullptr, d.Cs, 0};
            WSegSpec sb[2] = {{gate, d.Cd, 0, d.Cd, 0, sp ? ws + r.w.gateS[i] : nullptr, d.Cd, 0}, wone};
            WgradOut wo = run_wgrad(cx, g, sa, nsa, sb, 1 + nb, slab, cap);
            run_finalize(cx, slab, wo, 0, rows, d.Cd, 1, 0, 1, 0, p[6 + 4 * i], p[7 + 4 * i], grads[6 + 4 * i], grads[7 + 4 * i]);
            fin_bias(slab, wo, 0, rows, rup(d.Cd, 32), gb(3 + 2 * i));
        }
        // dgate = W_o^T do  ->  dxy (gate backward, waveglow.py:13-15)
        {
            SegSpec s[2];
            int ns = 0;
            if (!last) s[ns++] = {dH, d.C, 0, d.C, 0, dHSp(i + 1), d.C, 0};
            if (lr) s[ns++] = {G, Gc, 0, Gc, 0, ws + r.w.GS, Gc, 0};          // Wskip_i^T dS = Weff_i^T G: 2 ic channels (of kp_end) instead of Cs
            else s[ns++] = {dS, d.Cs, 0, d.Cs, 0, ws + r.w.dSS, d.Cs, 0};
            const bool twg = tw_from_gate(cx);              // (tanh from the gate's S-plane: the recompute pass did not keep it)
            run_convgemm(cx, g, r.pk + (lr ? r.L.WoG[i] : r.L.WoN[i]), r.L.ld_WoN, d.Cd, s, ns, EPI_DGATE, sp ? pnull() : pref(dxy, 2 * d.Cd), pnull(), pnull(),
                         twg ? pnull() : pref(ws + r.w.tw[i], d.Cd), pref(ws + r.w.sf[i], d.Cd), d.Cd, 0, sp ? sref(g, dxyS, 2 * d.Cd) : snull(),
                         twg ? sref(g, ws + r.w.gateS[i], d.Cd) : snull());
        }
        // dW (taps) and dV (conditioning) in one wgrad
        if (gw) {
            gsT[i].sa[0] = {nullptr, 2 * d.Cd, 0, 2 * d.Cd, 0, dxyS, 2 * d.Cd, 0};
            for (int kt = 0; kt < d.radix; ++kt) {
                int ts, ro;
                d.tap(i, kt, ts, ro);
                gsT[i].sb[kt] = {nullptr, d.C, 0, d.C, ts, ws + r.w.HS[i], d.C, 0, ro, 0};
            }
            if (!hv) gsT[i].sb[d.radix] = {nullptr, d.auxp(), 0, d.aux, 0, r.YS, d.auxp(), 0, 0, d.mode2d};
            if (nb) gsT[i].sb[d.radix + 1] = wone;
            if (hv) {
                gsV[i].sa[0] = {nullptr, 2 * d.Cd, 0, 2 * d.Cd, 0, rsp(i), 2 * d.Cd, 0};
                gsV[i].sb[0] = {nullptr, d.auxp(), 0, d.aux, 0, r.YS, d.auxp(), 0};
            }
        } else {
            WSegSpec sa = {dxy, 2 * d.Cd, 0, 2 * d.Cd, 0, sp ? dxyS : nullptr, 2 * d.Cd, 0};
            WSegSpec sb[WG_MAX_SEG];
            int nsb = 0;
            for (int kt = 0; kt < d.radix; ++kt) {
                int ts, ro;
                d.tap(i, kt, ts, ro);
                sb[nsb++] = {Hi, d.C, 0, d.C, ts, sp ? ws + r.w.HS[i] : nullptr, d.C, 0, ro, 0};
            }
            sb[nsb++] = {r.Y, d.auxp(), 0, d.aux, 0, sp ? r.YS : nullptr, d.auxp(), 0, 0, d.mode2d};
            if (nb) sb[nsb++] = wone;
            WgradOut wo = run_wgrad(cx, g, &sa, 1, sb, nsb, slab, cap);
            const int C32 = rup(d.C, 32);
            fin_bias(slab, wo, 0, 2 * d.Cd, d.radix * C32 + rup(d.aux, 32), gb(2 + 2 * i));                          // W_i.bias and V.bias share
            fin_bias(slab, wo, 0, 2 * d.Cd, d.radix * C32 + rup(d.aux, 32), gb(0) ? gb(0) + (size_t)i * 2 * d.Cd : nullptr);   // the pre-activation
            run_finalize(cx, slab, wo, 0, 2 * d.Cd, d.C, d.radix, 0, 1, C32, p[4 + 4 * i], p[5 + 4 * i], grads[4 + 4 * i], grads[5 + 4 * i]);
            const size_t ro = (size_t)i * 2 * d.Cd;
            run_finalize(cx, slab, wo, 0, 2 * d.Cd, d.aux, 1, d.radix * C32, 1, 0, p[0] ? p[0] + ro : nullptr, p[1] + ro * d.aux,
                         grads[0] ? grads[0] + ro : nullptr, grads[1] ? grads[1] + ro * d.aux : nullptr);
        }
        // dy += V_i^T dxy
        if (dY && d.mode2d) {
            // the conditioning is broadcast over the height axis: sum dxy over the rows of an item first (64x fewer columns for
            // the product, no per-row gradient plane), then dy[item] += V_i^T rowsum
            if (sp)
                WG_LAUNCH(cx, wf_rowsum_s_kernel, dim3((g.T + 63) / 64, 2 * d.Cd / 8, r.gi.B), dim3(256), 0, sref(g, dxyS, 2 * d.Cd), g,
                          sref(r.gi, rsp(i), 2 * d.Cd), r.gi);
            else                                              // exact-fp32 mode: the same sum on the fp32 plane (r.rs holds 2 Cd fp32 channels then)
                WG_LAUNCH(cx, wf_rowsum_kernel, dim3((g.T + 255) / 256, 2 * d.Cd, r.gi.B), dim3(256), 0, pref(dxy, 2 * d.Cd), g,
                          pref(r.rs, 2 * d.Cd), r.gi, 2 * d.Cd);
            // (every layer's sums kept -- hv: the weight gradient of V reads them as well --: ONE product over all of them behind the loop)
            if (!(hv && nd <= WG_MAX_SEG)) {
                SegSpec s = {sp ? nullptr : r.rs, 2 * d.Cd, 0, 2 * d.Cd, 0, sp ? rsp(i) : nullptr, 2 * d.Cd, 0};
                run_convgemm(cx, r.gi, r.pk + r.L.VN[i], r.L.ld_VN, d.aux, &s, 1, EPI_STORE, pref(dY, d.auxp()), pnull(), pnull(),
                             pref(dY, d.auxp()), pnull(), 0, 0);
            }
        } else if (dY && !fdy) {
            SegSpec s = {dxy, 2 * d.Cd, 0, 2 * d.Cd, 0, dxyS, 2 * d.Cd, 0};
            run_convgemm(cx, g, r.pk + r.L.VN[i], r.L.ld_VN, d.aux, &s, 1, EPI_STORE, pref(dY, d.auxp()), pnull(), pnull(),
                         pref(dY, d.auxp()), pnull(), 0, 0);
        }
        // dh_i = (last ? 0 : dh_{i+1}) + sum_k W[:,:,k]^T dxy[t - (k-mid) d]
        {
            SegSpec s[WG_MAX_SEG];
            int ns = 0;
            for (int kt = 0; kt < d.radix; ++kt) {
                int ts, ro;
                d.tap(i, kt, ts, ro);
                s[ns++] = {dxy, 2 * d.Cd, 0, 2 * d.Cd, -ts, dxyS, 2 * d.Cd, 0, -ro, 0};
            }
            const bool sod = s_only_chain(cx, d);            // dh as an S-plane only (accumulated from dh_{i+1}'s hi + lo: in place, or
            run_convgemm(cx, g, r.pk + r.L.WT[i], r.L.ld_WT, d.C, s, ns, EPI_STORE, sod ? pnull() : pref(dH, d.C), pnull(), pnull(),      // plane to plane with gw)
                         (last || sod) ? pnull() : pref(dH, d.C), pnull(), 0, 0, sp ? sref(g, dHSp(i), d.C) : snull(),
                         (sod && !last) ? sref(g, dHSp(i + 1), d.C) : snull());
        }
    }
    if (gw) {
        WgradOut wo[WG_GRP_MAX], woO[WG_GRP_MAX];
        const int C32 = rup(d.C, 32);
        bool pair = true;
#if defined(WG_OPT_NO_WGRAD_PAIR)
        pair = false;
#endif
        // (one after the other, each finalisation queued behind its own launch: the second product's slabs may then reuse the arena)
        const int nsbT = d.radix + (hv ? 0 : 1) + nb;
        if (pair) run_wgrad_group_pair(cx, g, gsT, 1, nsbT, wo, gsO, lr ? 1 : 2, 1 + nb, woO, nd, ws + r.w.dSS);
        else run_wgrad_group(cx, g, gsT, nd, 1, nsbT, ws + r.w.dSS, wo);
        for (int i = 0; i < nd && !cx.err; ++i) {
            run_finalize(cx, slab, wo[i], 0, 2 * d.Cd, d.C, d.radix, 0, 1, C32, p[4 + 4 * i], p[5 + 4 * i], grads[4 + 4 * i], grads[5 + 4 * i]);
            const size_t ro = (size_t)i * 2 * d.Cd;
            if (!hv)
            run_finalize(cx, slab, wo[i], 0, 2 * d.Cd, d.aux, 1, d.radix * C32, 1, 0, p[0] ? p[0] + ro : nullptr, p[1] + ro * d.aux,
                         grads[0] ? grads[0] + ro : nullptr, grads[1] ? grads[1] + ro * d.aux : nullptr);
            fin_bias(slab, wo[i], 0, 2 * d.Cd, d.radix * C32 + rup(d.aux, 32), gb(2 + 2 * i));
            fin_bias(slab, wo[i], 0, 2 * d.Cd, d.radix * C32 + rup(d.aux, 32), gb(0) ? gb(0) + ro : nullptr);
        }
        if (!pair) run_wgrad_group(cx, g, gsO, nd, lr ? 1 : 2, 1 + nb, ws + r.w.dSS, woO);
        if (hv) {                                             // dV of every layer: items x T columns against the conditioning itself
            WgradOut woV[WG_GRP_MAX];
            run_wgrad_group(cx, r.gi, gsV, nd, 1, 1, ws + r.w.dSS, woV);
            for (int i = 0; i < nd && !cx.err; ++i) {
                const size_t ro = (size_t)i * 2 * d.Cd;
                run_finalize(cx, slab, woV[i], 0, 2 * d.Cd, d.aux, 1, 0, 1, 0, p[0] ? p[0] + ro : nullptr, p[1] + ro * d.aux,
                             grads[0] ? grads[0] + ro : nullptr, grads[1] ? grads[1] + ro * d.aux : nullptr);
            }
        }
        for (int i = 0; i < nd && !cx.err; ++i) {
            const int last = i == nd - 1;
            if (lr) {                                         // the product holds the C residual rows (none on the last layer); the skip rows from P
                if (!last) run_finalize(cx, slab, woO[i], 0, d.C, d.Cd, 1, 0, 1, 0, p[6 + 4 * i], p[7 + 4 * i], grads[6 + 4 * i], grads[7 + 4 * i]);
                continue;
            }
            run_finalize(cx, slab, woO[i], last ? d.C : 0, d.wo_rows(i), d.Cd, 1, 0, 1, 0, p[6 + 4 * i], p[7 + 4 * i], grads[6 + 4 * i], grads[7 + 4 * i]);
            fin_bias(slab, woO[i], last ? d.C : 0, d.wo_rows(i), rup(d.Cd, 32), gb(3 + 2 * i));
        }
    }
    if (hv && nd <= WG_MAX_SEG) {                             // WN2D: dy[item] += [V_0^T .. V_{d-1}^T] [rowsum_0; ..; rowsum_{d-1}] (eight launches of K = 2 Cd before)
        SegSpec sv[WG_MAX_SEG];
        for (int i = 0; i < nd; ++i) sv[i] = {nullptr, 2 * d.Cd, 0, 2 * d.Cd, 0, rsp(i), 2 * d.Cd, 0};
        run_convgemm(cx, r.gi, r.pk + r.L.VNall, r.L.ld_VN, d.aux, sv, nd, EPI_STORE, pref(dY, d.auxp()), pnull(), pnull(),
                     pref(dY, d.auxp()), pnull(), 0, 0);
    }
    if (fdy) {                                                // dy += [V_0^T .. V_{d-1}^T] [dxy_0; ..; dxy_{d-1}]
        SegSpec sv[WG_MAX_SEG];
        for (int i = 0; i < nd; ++i)
            sv[i] = {ws + r.w.dxy + (size_t)i * r.w.dxy_step, 2 * d.Cd, 0, 2 * d.Cd, 0, ws + r.w.dxyS + (size_t)i * r.w.dxyS_step, 2 * d.Cd, 0};
        run_convgemm(cx, g, r.pk + r.L.VNall, r.L.ld_VN, d.aux, sv, nd, EPI_STORE, pref(dY, d.auxp()), pnull(), pnull(),
                     pref(dY, d.auxp()), pnull(), 0, 0);
    }
    // start: dW_start = sum dh_0 (x) xa ; dxa += W_start^T dh_0
    if (thin_ok(cx, d)) run_thin_start(cx, r, dHSp(0), dX, p[2], p[3], grads[2], grads[3]);      // both in one pass over dh_0 (wg_thin.h)
    else {
        WSegSpec sa = {dH, d.C, 0, d.C, 0, sp ? dHSp(0) : nullptr, d.C, 0};
        WSegSpec sb[2] = {{r.X.p, r.X.Cp, r.X.ch0, d.ic, 0, sp ? ws + r.w.XaS : nullptr, r.L.kp_start, 0}, wone};
        WgradOut wo = run_wgrad(cx, g, &sa, 1, sb, 1 + nb, slab, cap);
        run_finalize(cx, slab, wo, 0, d.C, d.ic, 1, 0, 1, 0, p[2], p[3], grads[2], grads[3]);
        fin_bias(slab, wo, 0, d.C, rup(d.ic, 32), gb(1));
        SegSpec s = {dH, d.C, 0, d.C, 0, dHSp(0), d.C, 0};
        run_convgemm(cx, g, r.pk + r.L.startN, r.L.ld_startN, d.ic, &s, 1, EPI_STORE, dX, pnull(), pnull(), dX, pnull(), 0, 0);
    }
}

void run_upsample(Ctx &cx, const wg_config *cf, const float *w, const float *bias, const float *h, int F, const Geo &g,
                  PRef Y, float *yplain)
{
    WG_LAUNCH(cx, upsample_fwd_kernel, dim3((g.T + 255) / 256, cf->n_mels, g.B), dim3(256), 0, h, w, bias, Y, yplain, g,
              cf->n_mels, F, cf->up_kernel, cf->up_stride, cf->up_pad);
}

int shape_check(const wg_config *cf, int B, int N, int F, int *T)
{
    if (B < 1 || N < 1 || F < 1) return WG_EINVAL;
    if (N % cf->n_group) return WG_ESHAPE;
    *T = N / cf->n_group;
    const long L = (long)(F - 1) * cf->up_stride - 2 * cf->up_pad + cf->up_kernel;
    if (*T > L) return WG_ESHAPE;                      // assert x.size(2) <= y.size(2)   waveglow.py:156,187
    return 0;
}


// ================================================================================================
// WaveFlow (model/waveflow.py)
// ================================================================================================
WnD wf_wn(const wg_wf_config *cf)
{
    WnD d;
    d.ic = 1; d.aux = cf->n_mels; d.C = cf->res_ch; d.Cd = cf->dil_ch; d.Cs = cf->skip_ch; d.depth = 8; d.radix = 9;
    d.prec = cf->precision; d.mode2d = 1; d.bias = cf->bias ? 1 : 0;
    static const int d8[8] = {1, 1, 1, 1, 1, 1, 1, 1}, d32[8] = {1, 2, 4, 1, 2, 4, 1, 2}, d64[8] = {1, 2, 4, 8, 16, 1, 2, 4},
                     d128[8] = {1, 2, 4, 8, 16, 32, 64, 1};                 // waveflow.py:81-87
    const int *hd = cf->n_group == 32 ? d32 : cf->n_group == 64 ? d64 : cf->n_group == 128 ? d128 : d8;
    for (int i = 0; i < 8; ++i) d.hd[i] = hd[i];
    return d;
}
int wf_check(const wg_wf_config *cf)
{
    if (!cf || cf->flows < 1 || cf->flows > WG_MAX_FLOWS || cf->n_mels < 1) return WG_EINVAL;
    const int H = cf->n_group;
    if (H != 8 && H != 16 && H != 32 && H != 64 && H != 128) return WG_EUNSUPPORTED;     // the keys of dilation_dict (waveflow.py:81-87)
    if (cf->precision < WG_PREC_F32 || cf->precision > WG_PREC_BF16X3_PLANES) return WG_EUNSUPPORTED;
    if (cf->n_mels * (2 * (256 / H) + 1) > WG_FIN_MAXCOLS) return WG_EUNSUPPORTED;           // upsampler weight row [n_mels x (2s+1)] in finalize's LDS
    // use_conv1x1: the H x H weight is factored / its gradient gathered in ONE workgroup's LDS (lu_big_kernel: H (H + 2) floats,
    // wf_hgram_kernel: 130 H floats = 66 560 B at H = 128): must fit gfx950's 160 KB; the opt-in itself is asked for once per device
    // (ensure_dynamic_lds), and a device that refuses it fails that call with WG_EUNSUPPORTED
    if (cf->use_conv1x1 && std::max((size_t)H * (H + 2), (size_t)130 * H) * sizeof(float) > 160 * 1024) return WG_EUNSUPPORTED;
    return wn_check(wf_wn(cf));
}
struct WfPack {
    size_t ones, up_scale, wn[WG_MAX_FLOWS], mix, total;
    int mix_stride;             // use_conv1x1: per flow [W | W^-1 | logdet W] (floats from `mix`)
};
// parameter-table entries of one flow's WN2D: 37 weights (V.g V.v start.g start.v {W.g W.v W_o.g W_o.v} x 8 end.weight) and, with
// bias=True, the 19 biases behind them (WnD::pb)
inline int wf_pf(const wg_wf_config *cf) { return 37 + (cf->bias ? 19 : 0); }
inline int wf_nparams(const wg_wf_config *cf) { return 3 + cf->flows * wf_pf(cf) + (cf->use_conv1x1 ? cf->flows : 0); }
inline const float *wf_end_bias(const wg_wf_config *cf, const float *const *p, int k) { return cf->bias ? p[3 + wf_pf(cf) * k + 37 + 18] : nullptr; }
WfPack wf_pack_layout(const wg_wf_config *cf)
{
    WfPack L;
    size_t off = 0;
    auto take = [&](size_t n) { size_t o = off; off += rupz(n, 64); return o; };
    L.ones = take(WG_ONES);
    L.up_scale = take(cf->n_mels);
    const size_t wn = wn_pack_layout(wf_wn(cf)).total;
    for (int k = 0; k < cf->flows; ++k) L.wn[k] = take(wn);
    L.mix_stride = (int)rupz((size_t)2 * cf->n_group * cf->n_group + 1, 64);
    L.mix = take(cf->use_conv1x1 ? (size_t)cf->flows * L.mix_stride : 0);
    L.total = off;
    return L;
}
#define WF_PROG_STAGES 24       // a row step of WN2D: S-plane conversion, start, depth x (gate conv, residual / skip conv), end + coupling
struct WfWs {
    Geo g, gi;              // one plane row per (item, height row) / one per item
    int auxp;
    size_t Y, YS, X[2], rowsum, dX[2], dYrow, rs, rs_step, gp, dwup, Xt, dXt, gram, total;    // Xt / dXt / gram: use_conv1x1 only
    size_t prog;            // the inverse's recorded row-step program (wg_stage.h): WF_PROG_STAGES argument blocks + the barrier words
    int gram_blocks;
    WnWs wn;
};
WfWs wf_ws_layout(const wg_wf_config *cf, int B, int Wd, int mode)
{
    WfWs w;
    const WnD d = wf_wn(cf);
    const int H = cf->n_group, s = 256 / H;
    w.gi = make_geo(B, Wd, d.maxdil());
    w.g = w.gi;
    w.g.B = B * H;
    w.g.rows = H;
    w.auxp = d.auxp();
    Bump bp;
    const size_t xplane = (size_t)w.g.B * w.g.P;
    w.Y = bp.take((size_t)B * w.auxp * w.gi.P);
    w.YS = bp.take((size_t)B * w.auxp * w.gi.P);
    w.X[0] = bp.take(xplane); w.X[1] = bp.take(xplane);
    w.rowsum = bp.take((size_t)cf->flows * w.g.B);
    w.Xt = w.dXt = w.gram = 0;
    w.gram_blocks = B * ((Wd + 63) / 64);
    if (cf->use_conv1x1) {
        w.Xt = bp.take(xplane);                                          // cat(x[0], xout): the 1x1's input
        if (mode) { w.dXt = bp.take(xplane); w.gram = bp.take((size_t)w.gram_blocks * H * H); }
    }
    w.dX[0] = w.dX[1] = w.dYrow = w.rs = w.rs_step = w.gp = w.dwup = 0;
    if (mode) {
        w.dX[0] = bp.take(xplane); w.dX[1] = bp.take(xplane);
        w.dYrow = bp.take((size_t)B * w.auxp * w.gi.P);                 // conditioning gradient, per item
        // (S-plane mode, no biases: one plane per layer -- the conditioning's weight gradient is taken from these sums after the layer loop,
        // wn_backward)
        w.rs_step = (cf->precision == WG_PREC_BF16X3_PLANES && !d.bias && grouped_wgrad(cf->precision, d)) ? rupz((size_t)B * 2 * d.Cd * w.gi.P, 64) : 0;
        w.rs = bp.take(w.rs_step ? w.rs_step * d.depth : (size_t)B * 2 * d.Cd * w.gi.P);
        w.gp = bp.take((size_t)B * cf->n_mels * Wd);
        w.dwup = bp.take((size_t)cf->n_mels * cf->n_mels * (2 * s + 1));
    }
    wn_ws_layout(bp, d, 1, w.g, mode, cf->precision, w.wn);
    w.prog = bp.take((WF_PROG_STAGES * sizeof(WgStage) + 256) / sizeof(float));
    w.total = bp.off + 4096;
    return w;
}
int wf_shape(const wg_wf_config *cf, int B, int N, int F, int *Wd)
{
    if (B < 1 || N < 1 || F < 1) return WG_EINVAL;
    if (N % cf->n_group) return WG_ESHAPE;
    *Wd = N / cf->n_group;
    const int s = 256 / cf->n_group;
    if (*Wd > F * s - 2 * (s / 2) + 2 * s + 1) return WG_ESHAPE;          // y[..., :W] needs W upsampled frames (waveflow.py:187)
    return 0;
}
void wf_upsample(Ctx &cx, const wg_wf_config *cf, const float *const *p, const float *pk, const WfPack &L, const float *mel, int F,
                 const WfWs &W, float *ws)
{
    const int s = 256 / cf->n_group;
    WfUpArgs a;
    a.mel = mel; a.v = p[2]; a.scale = pk + L.up_scale; a.bias = p[0];
    a.M = cf->n_mels; a.F = F; a.K = 2 * s + 1; a.s = s; a.pad = s / 2;
    a.Y = pref(ws + W.Y, W.auxp); a.gi = W.gi;
    WG_LAUNCH(cx, wf_upsample_fwd_kernel, dim3((W.gi.T + 255) / 256, cf->n_mels, W.gi.B), dim3(256), 0, a);
    if (cx.prec == 2) run_to_splane(cx, W.gi, pref(ws + W.Y, W.auxp), cf->n_mels, ws + W.YS, W.auxp);
}
void run_hmix(Ctx &cx, const Geo &g, PRef src, PRef dst, const float *Mx, int transpose)
{
    WG_LAUNCH(cx, wf_hmix_kernel, dim3((g.T + 255) / 256, (g.rows + WF_MIX_ROWS - 1) / WF_MIX_ROWS, g.B / g.rows), dim3(256), 0, src, dst, g, Mx,
              transpose);
}
void wf_couple(Ctx &cx, const WnRun &r, const float *endw, const float *endb, int mode, PRef X, PRef Xn, PRef dXn, PRef dX, const float *dld,
               float *rowsum, int row_sel, int noflip = 0)
{
    WfCoupleArgs a;
    memset(&a, 0, sizeof(a));
    a.endw = endw; a.endb = endb;
    a.S = pref(r.ws + r.w.skip, r.d.Cs); a.Cs = r.d.Cs;
    if (lowrank_on(cx, r)) {                                  // (mode2d: with the partial rows, or not at all)
        a.part = r.ws + r.w.gpart; a.nsrc = r.d.depth * gate_part_slots(r.d);
        if (r.w.gpart_step != (size_t)gate_part_slots(r.d) * r.g.B * r.g.Tt * 2 && !cx.err) cx.err = WG_EINVAL;
    }
    a.X = X; a.Xn = Xn; a.dXn = dXn; a.dX = dX;
    a.G = pref(r.ws + r.w.G, r.L.kp_end);
    a.dld = dld; a.rowsum = rowsum; a.row_sel = row_sel; a.g = r.g; a.mode = mode; a.noflip = noflip;
    if (cx.rec && mode == 2) { cx.rec->add(WGS_WFCOUPLE, r.g.B / r.g.rows).u.cpl = a; return; }
    if (mode == 2 && a.Cs % 4 == 0) { WG_LAUNCH(cx, wf_couple_row_kernel, dim3(r.g.B / r.g.rows), dim3(1024), 0, a); return; }
    WG_LAUNCH(cx, wf_couple_kernel, dim3(mode == 2 ? r.g.B / r.g.rows : r.g.B), dim3(256), 0, a);
}
}  // namespace

// ================================================================================================
// C ABI
// ================================================================================================
extern "C" {

const char *wg_strerror(int code)
{
    switch (code) {
    case WG_OK: return "ok";
    case WG_EINVAL: return "invalid argument";
    case WG_ESHAPE: return "shape mismatch (audio length % n_group != 0, or mel shorter than audio)";
    case WG_EUNSUPPORTED: return "configuration not supported by the HIP kernels";
    case WG_ELAUNCH: return "HIP kernel launch failed";
    case WG_EWORKSPACE: return "workspace too small";
    }
    return "unknown error";
}
int wg_abi_version(void) { return WG_ABI_VERSION; }
void wg_reload_env(void) { g_env.store(env_load(), std::memory_order_release); }

#if defined(WG_DBG_TRACE)
int wg_dbg_trace_read(unsigned long long *out, int n)
{
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(wg_dbg_trace), (size_t)n * sizeof(unsigned long long)) == hipSuccess ? 0 : -1;
}
int wg_dbg_trace_read_cycles(unsigned long long *out, int n)
{
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(wg_dbg_trace_cyc), (size_t)n * sizeof(unsigned long long)) == hipSuccess ? 0 : -1;
}
#endif
long long wg_stat_wgrad16t_launches(void) { return g_wgrad16t_launches.load(std::memory_order_relaxed); }
long long wg_stat_layer_launches(void) { return g_layer_launches.load(std::memory_order_relaxed) + g_layerq_launches.load(std::memory_order_relaxed); }
long long wg_stat_layerg_launches(void) { return g_layerg_launches.load(std::memory_order_relaxed); }
long long wg_stat_gate_split_launches(void) { return g_gate_split_launches.load(std::memory_order_relaxed); }
long long wg_stat_gate_part_launches(void) { return g_gate_part_launches.load(std::memory_order_relaxed); }
void *wg_timer_create(int kernel_id, int capacity)
{
    if (capacity < 1) return nullptr;
    KernelTimer *t = new KernelTimer;
    t->kernel_id = kernel_id; t->capacity = capacity; t->count = 0;
    t->start = new hipEvent_t[capacity];
    t->stop = new hipEvent_t[capacity];
    t->info = new long long[5 * (size_t)capacity]();
    t->name = new const char *[capacity]();
    for (int i = 0; i < capacity; ++i) { (void)hipEventCreate(&t->start[i]); (void)hipEventCreate(&t->stop[i]); }
    return t;
}
void wg_timer_attach(void *timer) { g_timer.store((KernelTimer *)timer); }
int wg_timer_count(void *timer) { return timer ? ((KernelTimer *)timer)->count : 0; }
/* call after the stream has been synchronised; writes one duration (ms) per recorded launch */
int wg_timer_read(void *timer, float *ms, int n)
{
    KernelTimer *t = (KernelTimer *)timer;
    if (!t || !ms) return WG_EINVAL;
    const int m = std::min(n, t->count);
    for (int i = 0; i < m; ++i)
        if (hipEventElapsedTime(&ms[i], t->start[i], t->stop[i]) != hipSuccess) return WG_ELAUNCH;
    return m;
}
/* per recorded launch five values: kernel class (WG_K_*), M, K, columns of the product, algorithmic HBM bytes; returns launches written */
int wg_timer_read_info(void *timer, long long *info, int n)
{
    KernelTimer *t = (KernelTimer *)timer;
    if (!t || !info) return WG_EINVAL;
    const int m = std::min(n, t->count);
    memcpy(info, t->info, (size_t)m * 5 * sizeof(long long));
    return m;
}
void wg_timer_destroy(void *timer)
{
    KernelTimer *t = (KernelTimer *)timer;
    if (!t) return;
    if (g_timer.load() == t) g_timer.store(nullptr);
    for (int i = 0; i < t->capacity; ++i) { (void)hipEventDestroy(t->start[i]); (void)hipEventDestroy(t->stop[i]); }
    delete[] t->start; delete[] t->stop; delete[] t->info; delete[] t->name; delete t;
}
/* the kernel expression of recorded launch `index` as written at its launch site, e.g. "(convgemm16q_kernel<EPI_GATE_SO, 2, 2>)"
 * (template arguments by name; defaulted ones absent); returns its length, or a negative error */
int wg_timer_read_name(void *timer, int index, char *buf, int n)
{
    KernelTimer *t = (KernelTimer *)timer;
    if (!t || !buf || n < 1 || index < 0 || index >= t->count) return WG_EINVAL;
    const char *q = t->name[index] ? t->name[index] : "";
    const int len = (int)strlen(q);
    snprintf(buf, (size_t)n, "%s", q);
    return len;
}

/* Box calibration: a FIXED matrix-pipe + LDS loop (every CU: eight waves, 64 x 64 tile per wave, hi / lo fragments re-read from LDS, three
 * v_mfma_f32_16x16x32_bf16 per fragment pair -- the conv kernels' mix with no global traffic; tools/experiments/shape_probe.hip) on random
 * data, launched back to back for about `ms` milliseconds.  out[0] = issued TFLOP/s of the last launches, out[1] = the clock held inside
 * the kernel (GHz: shader cycles / 100 MHz wall clock, median over the workgroups), out[2] = ms per launch.  The boxes of a pool differ
 * by a few per cent on exactly this (MI355X_MICROARCH.md, DVFS give-back item 5): a benchmark line that carries these numbers lets a
 * reader tell a slower box from a slower build.  scratch: at least wg_box_probe_bytes() bytes of device memory. */
size_t wg_box_probe_bytes(void) { return (size_t)65536 * 16 + (size_t)256 * 512 * 4 + 512 * 8; }
int wg_box_probe(void *scratch, int ms, double *out, void *stream)
{
    if (!scratch || !out || ms < 1) return WG_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    char *base = (char *)scratch;
    u32x4 *rnd = (u32x4 *)base;
    float *res = (float *)(base + (size_t)65536 * 16);
    unsigned long long *stamps = (unsigned long long *)(base + (size_t)65536 * 16 + (size_t)256 * 512 * 4);
    struct Events {                                           // destroyed on every way out
        hipEvent_t e0 = nullptr, e1 = nullptr;
        bool ok() { return hipEventCreate(&e0) == hipSuccess && hipEventCreate(&e1) == hipSuccess; }
        ~Events() { if (e0) (void)hipEventDestroy(e0); if (e1) (void)hipEventDestroy(e1); }
    } ev;
    if (!ev.ok()) return WG_ELAUNCH;
    // (the clock stamps start from zero: after a launch that failed they must not read as a clock)
    if (hipMemsetAsync(stamps, 0, 512 * sizeof(unsigned long long), st) != hipSuccess) return WG_ELAUNCH;
    hipLaunchKernelGGL(box_fill_kernel, dim3(256), dim3(256), 0, st, rnd);
    if (hipGetLastError() != hipSuccess) return WG_ELAUNCH;
    float one = 0.f, last = 0.f;
    (void)hipEventRecord(ev.e0, st);
    hipLaunchKernelGGL(box_probe_kernel, dim3(256), dim3(512), 0, st, rnd, res, stamps);
    if (hipGetLastError() != hipSuccess) return WG_ELAUNCH;
    (void)hipEventRecord(ev.e1, st);
    if (hipStreamSynchronize(st) != hipSuccess || hipEventElapsedTime(&one, ev.e0, ev.e1) != hipSuccess) return WG_ELAUNCH;
    const int batch = 8, rounds = std::max(1, (int)((double)ms / (std::max(one, 0.05f) * batch)));
    for (int r = 0; r < rounds; ++r) {
        (void)hipEventRecord(ev.e0, st);
        for (int i = 0; i < batch; ++i) {
            hipLaunchKernelGGL(box_probe_kernel, dim3(256), dim3(512), 0, st, rnd, res, stamps);
            if (hipGetLastError() != hipSuccess) return WG_ELAUNCH;
        }
        (void)hipEventRecord(ev.e1, st);
        if (hipStreamSynchronize(st) != hipSuccess || hipEventElapsedTime(&last, ev.e0, ev.e1) != hipSuccess) return WG_ELAUNCH;
    }
    std::vector<unsigned long long> h(512);
    if (hipMemcpyAsync(h.data(), stamps, 512 * sizeof(unsigned long long), hipMemcpyDeviceToHost, st) != hipSuccess ||
        hipStreamSynchronize(st) != hipSuccess)
        return WG_ELAUNCH;
    std::vector<double> ghz;
    for (int b = 0; b < 256; ++b) ghz.push_back(h[2 * b + 1] ? (double)h[2 * b] / (double)h[2 * b + 1] / 10.0 : 0.0);
    std::sort(ghz.begin(), ghz.end());
    const double per = last / batch, flops = 256.0 * 8 * (double)WG_BOX_CHUNKS * 2.0 * 64 * 64 * 32 * 3;
    out[0] = flops / (per * 1e-3) / 1e12; out[1] = ghz[128]; out[2] = per;
    return 0;
}

int wg_param_count(const wg_config *cf) { return cf ? 3 + cf->n_flows + cf->n_flows * flow_wn(cf, 0).nparams() : WG_EINVAL; }
size_t wg_packed_bytes(const wg_config *cf) { return cfg_check(cf) ? 0 : model_pack_layout(cf).total * sizeof(float); }
size_t wg_workspace_bytes(const wg_config *cf, int B, int N, int mode)
{
    if (cfg_check(cf) || B < 1 || N < 1 || N % cf->n_group) return 0;
    return model_ws_layout(cf, B, N / cf->n_group, mode).total * sizeof(float);
}
int wg_workspace_init(void *ws, size_t bytes, void *stream)
{
    return hipMemsetAsync(ws, 0, bytes, (hipStream_t)stream) == hipSuccess ? WG_OK : WG_ELAUNCH;
}

static WnD wnd_from(const wg_wn_dims *d)
{
    WnD w;
    w.ic = d->in_ch; w.aux = d->aux_ch; w.C = d->res_ch; w.Cd = d->dil_ch; w.Cs = d->skip_ch; w.depth = d->depth; w.radix = d->radix;
    w.prec = d->precision;
    w.bias = d->bias ? 1 : 0;
    return w;
}
int wg_wn_param_count(const wg_wn_dims *d) { return d ? wnd_from(d).nparams() : WG_EINVAL; }
size_t wg_wn_packed_bytes(const wg_wn_dims *d)
{
    if (!d || wn_check(wnd_from(d))) return 0;
    return (rupz(WG_ONES, 64) + wn_pack_layout(wnd_from(d)).total) * sizeof(float);
}

int wg_wn_pack_weights(const wg_wn_dims *dd, const void *const *params, void *packed, void *stream)
{
    if (!dd || !params || !packed) return WG_EINVAL;
    const WnD d = wnd_from(dd);
    int rc = wn_check(d);
    if (rc) return rc;
    Ctx cx = {(hipStream_t)stream, 0, 0};
    float *pk = (float *)packed;
    float *ones = pk;
    float *wn = pk + rupz(WG_ONES, 64);
    const WnPack L = wn_pack_layout(d);
    WG_LAUNCH(cx, fill_rows_kernel, dim3(WG_ONES / 256, 1, 1), dim3(256), 0, pref(ones, 1), Geo{1, WG_ONES, WG_ONES, 0, WG_ONES}, (const float *)nullptr, 1.0f);
    JobBatch jb(&cx);
    wn_pack_norms(jb, d, L, (const float *const *)params, wn);
    jb.flush_norm();
    EffBatch eb(&cx);
    wn_pack_eff(eb, d, L, (const float *const *)params, wn);
    eb.flush();
    FoldBatch fb(&cx);
    wn_pack_fold(fb, d, L, (const float *const *)params, wn);
    fb.flush();
    wn_pack_mats(jb, d, L, (const float *const *)params, wn, ones);
    jb.flush_pack();
    ImgBatch ib(&cx);
    wn_pack_images(ib, d, L, wn);
    ib.flush();
    return cx.err;
}

int wg_pack_weights(const wg_config *cf, const void *const *params, void *packed, void *stream)
{
    int rc = cfg_check(cf);
    if (rc) return rc;
    if (!params || !packed) return WG_EINVAL;
    Ctx cx = {(hipStream_t)stream, 0, 0};
    const float *const *p = (const float *const *)params;
    float *pk = (float *)packed;
    const ModelPack M = model_pack_layout(cf);
    float *ones = pk + M.ones;
    WG_LAUNCH(cx, fill_rows_kernel, dim3(WG_ONES / 256, 1, 1), dim3(256), 0, pref(ones, 1), Geo{1, WG_ONES, WG_ONES, 0, WG_ONES}, (const float *)nullptr, 1.0f);
    // 1x1 weights: LU -> logdet, inverse   (efficient_modules.py:221,235)
    // (one workgroup per matrix, the matrix in LDS: the one-thread-per-matrix lu_kernel, whose arrays live in scratch memory, took
    // 0.11 ms for WaveGlow's 8 x 8 and 0.62 ms for WSRGlow's 16 x 16 weights at the head of every step)
    LuBigArgs lu;
    memset(&lu, 0, sizeof(lu));
    lu.n = cf->n_flows; lu.c = cf->n_group; lu.out = pk + M.lu; lu.ostride = WG_LU_STRIDE;
    lu.inv_off = WG_MAXC * WG_MAXC; lu.det_off = 2 * WG_MAXC * WG_MAXC;
    for (int k = 0; k < cf->n_flows; ++k) { lu.W[k] = p[3 + k]; lu.cs[k] = (unsigned char)flow_channels(cf, k); }
    WG_LAUNCH(cx, lu_big_kernel, dim3(cf->n_flows), dim3(256), ((size_t)cf->n_group * (cf->n_group + 1) + cf->n_group) * sizeof(float), lu);
    JobBatch jb(&cx);
    jb.norm(p[1], p[2], pk + M.up_scale, cf->n_mels, cf->up_kernel);
    for (int k = 0; k < cf->n_flows; ++k) {
        const WnD d = flow_wn(cf, k);
        wn_pack_norms(jb, d, wn_pack_layout(d), p + wn_table_off(cf, k), pk + M.wn[k]);
    }
    jb.flush_norm();
    EffBatch eb(&cx);
    for (int k = 0; k < cf->n_flows; ++k) {
        const WnD d = flow_wn(cf, k);
        wn_pack_eff(eb, d, wn_pack_layout(d), p + wn_table_off(cf, k), pk + M.wn[k]);
    }
    eb.flush();
    FoldBatch fb(&cx);
    for (int k = 0; k < cf->n_flows; ++k) {
        const WnD d = flow_wn(cf, k);
        wn_pack_fold(fb, d, wn_pack_layout(d), p + wn_table_off(cf, k), pk + M.wn[k]);
    }
    fb.flush();
    jb.pack(pk + M.up_w, cf->up_kernel, cf->n_mels, cf->up_kernel, 1, cf->n_mels, cf->up_kernel, 0, p[2], pk + M.up_scale, cf->up_kernel, 1, 0);
    jb.pack(pk + M.up_bias, cf->n_mels, 1, cf->n_mels, 1, 1, cf->n_mels, 0, p[0], ones, cf->n_mels, 1, 0);
    for (int k = 0; k < cf->n_flows; ++k) {
        const WnD d = flow_wn(cf, k);
        wn_pack_mats(jb, d, wn_pack_layout(d), p + wn_table_off(cf, k), pk + M.wn[k], ones);
    }
    jb.flush_pack();
    ImgBatch ib(&cx);
    for (int k = 0; k < cf->n_flows; ++k) {
        const WnD d = flow_wn(cf, k);
        wn_pack_images(ib, d, wn_pack_layout(d), pk + M.wn[k]);
    }
    ib.flush();
    return cx.err;
}

// ---- model level -------------------------------------------------------------------------------

// ws_mode 1 + save_last: the training step's forward.  It runs in the BACKWARD workspace and keeps every layer of the flow it
// processes last, so that wg_train_step can start the backward without recomputing that flow (its activations are the only ones
// that fit the O(1)-in-depth budget; every other flow is recomputed from its rebuilt input as usual).
static int model_run_fwd_or_inv(const wg_config *cf, const void *packed, const float *in, const float *h,
                                int B, int N, int F, int inverse, float *out, float *logdet, void *wsv, size_t ws_bytes, void *stream,
                                int ws_mode = 0, int save_last = 0)
{
    int rc = cfg_check(cf);
    if (rc) return rc;
    int T;
    rc = shape_check(cf, B, N, F, &T);
    if (rc) return rc;
    if (!packed || !in || !h || !out || !logdet || !wsv) return WG_EINVAL;
    const bool keep = cf->keep_activations && !inverse;   // stored-activation forward: mode-1 workspace, every flow keeps its layers
    if (keep) ws_mode = 1;
    const ModelWs W = model_ws_layout(cf, B, T, ws_mode);
    if (W.total * sizeof(float) > ws_bytes) return WG_EWORKSPACE;
    Ctx cx = {(hipStream_t)stream, 0, cf->precision};
    const float *pk = (const float *)packed;
    const ModelPack M = model_pack_layout(cf);
    float *ws = (float *)wsv;
    const Geo g = W.g;
    const int G = cf->n_group;
    const int last_k = (cf->reverse_mode != 0) == (inverse != 0) ? cf->n_flows - 1 : 0;     // the flow this direction processes last
    PRef X = pref(ws + W.X, W.Gp);
    layer_sync_clear(cx, ws, W.wn.lsync);
    WG_LAUNCH(cx, squeeze_kernel, dim3((T + 255) / 256, B), dim3(256), 0, in, X, g, G, N);           // waveglow.py:153 / :184
    run_upsample(cx, cf, pk + M.up_w, pk + M.up_bias, h, F, g, pref(ws + W.Y, W.auxp), nullptr);                // :151,157
    if (cx.prec == 2) run_to_splane(cx, g, pref(ws + W.Y, W.auxp), cf->n_mels, ws + W.YS, W.auxp);
    float *partial = ws + W.partial;
    WnRun r;
    r.g = g; r.ws = ws; r.w = W.wn; r.Y = ws + W.Y; r.YS = ws + W.YS; r.save = 0;
    // one coupling (WN + affine, forward or inverse formulas) and one 1x1 mix on the channels [base, base + c_k)
    auto coupling = [&](int k, int base, int aff_mode) {
        r.d = flow_wn(cf, k); r.L = wn_pack_layout(r.d); r.pk = pk + M.wn[k]; r.X = pref(ws + W.X, W.Gp, base);
        r.save = keep || (save_last && k == last_k);
        if (keep) r.w = W.flow(k);
        wn_forward(cx, r);
        run_end_affine(cx, r, aff_mode, pnull(), nullptr, nullptr, nullptr, partial + (size_t)k * B * W.ntile);
    };
    auto mix = [&](int k, int base, bool inv) {
        const float *lu = pk + M.lu + (size_t)k * WG_LU_STRIDE;
        run_mix(cx, g, pref(ws + W.X, W.Gp, base), flow_channels(cf, k), lu + (inv ? WG_MAXC * WG_MAXC : 0), 0);
    };
    const int c_last = flow_channels(cf, cf->n_flows - 1);
    if (!cf->reverse_mode) {
        if (!inverse) {                                   // waveglow.py:150-179
            int base = 0;
            for (int k = 0; k < cf->n_flows; ++k) {
                if (k % cf->n_early_every == 0 && k) base += cf->n_early_size;                        // :164-170
                mix(k, base, false);                                                                  // :172
                coupling(k, base, AFF_FWD);                                                           // :173
            }
        } else {                                          // waveglow.py:181-208
            int base = G - c_last;
            int start_done = 0;
            for (int k = cf->n_flows - 1; k >= 0; --k) {
                const int nbase = (k % cf->n_early_every == 0 && k) ? base - cf->n_early_size : base;
                // the WN, then affine + 1x1 + the next flow's WN.start in one launch where the seam kernel serves the shapes
                r.d = flow_wn(cf, k); r.L = wn_pack_layout(r.d); r.pk = pk + M.wn[k]; r.X = pref(ws + W.X, W.Gp, base);
                r.save = keep || (save_last && k == last_k);
                if (keep) r.w = W.flow(k);
                r.start_done = start_done;
                wn_forward(cx, r);                                                                    // :199
                r.start_done = start_done = 0;
                bool seam = false;
                if (k > 0 && !keep) {
                    WnRun nx = r;
                    nx.d = flow_wn(cf, k - 1); nx.L = wn_pack_layout(nx.d); nx.pk = pk + M.wn[k - 1]; nx.X = pref(ws + W.X, W.Gp, nbase);
                    nx.save = save_last && k - 1 == last_k;
                    const float *lu = pk + M.lu + (size_t)k * WG_LU_STRIDE;
                    if (flow_channels(cf, k) == 2 * r.d.ic)
                        seam = run_inv_seam(cx, r, lu + WG_MAXC * WG_MAXC, partial + (size_t)k * B * W.ntile, nx);
                }
                if (seam) start_done = 1;
                else {
                    run_end_affine(cx, r, AFF_REV, pnull(), nullptr, nullptr, nullptr, partial + (size_t)k * B * W.ntile);
                    mix(k, base, true);                                                               // :200
                }
                base = nbase;                                                                         // :204-205
            }
        }
    } else {
        // reverse_mode=True (base.py:20-28): model.forward runs the loop of :181-208 with the blocks' FORWARD formulas,
        // model.reverse the loop of :150-179 with their inverse formulas.
        if (!inverse) {
            int base = G - c_last;
            for (int k = cf->n_flows - 1; k >= 0; --k) {
                coupling(k, base, AFF_FWD);
                mix(k, base, false);
                if (k % cf->n_early_every == 0 && k) base -= cf->n_early_size;
            }
        } else {
            int base = 0;
            for (int k = 0; k < cf->n_flows; ++k) {
                if (k % cf->n_early_every == 0 && k) base += cf->n_early_size;
                mix(k, base, true);
                coupling(k, base, AFF_REV);
            }
        }
    }
    WG_LAUNCH(cx, unsqueeze_kernel, dim3((T + 255) / 256, B), dim3(256), 0, X, out, g, G, N);          // :179 / :207
    WG_LAUNCH(cx, logdet_finalize_kernel, dim3(B), dim3(64), 0, pk + M.lu, WG_LU_STRIDE, cf->n_flows,
              inverse ? -(float)T : (float)T, partial, W.ntile, B, logdet);                            // :175 / :202
    return cx.err;
}

int wg_forward(const wg_config *cf, const void *packed, const float *audio, const float *h, int B, int N, int F,
               float *z, float *logdet, void *ws, size_t ws_bytes, void *stream)
{
    return model_run_fwd_or_inv(cf, packed, audio, h, B, N, F, 0, z, logdet, ws, ws_bytes, stream);
}
int wg_inverse(const wg_config *cf, const void *packed, const float *z, const float *h, int B, int N, int F,
               float *x, float *logdet, void *ws, size_t ws_bytes, void *stream)
{
    return model_run_fwd_or_inv(cf, packed, z, h, B, N, F, 1, x, logdet, ws, ws_bytes, stream);
}


// A gradient bucket's event gates its all-reduce on the caller's communication stream: a failed record would leave that stream
// waiting on a stale record and reducing half-written gradients, so it is an error of the call.
static void record_event(Ctx &cx, void *const *flow_events, int k)
{
    if (!flow_events || !flow_events[k] || cx.err) return;
    if (hipEventRecord((hipEvent_t)flow_events[k], cx.st) != hipSuccess) cx.err = WG_ELAUNCH;
}

// resume: called by wg_train_step right after the training forward ran in this workspace -- X, Y (and YS) are in place and the
// flow the backward visits first still has all its layers, so neither the squeeze / upsample nor that flow's recompute is repeated.
static int model_backward(const wg_config *cf, const void *const *params, const void *packed, const float *z, const float *h,
                          const float *dz, const float *dlogdet, int B, int N, int F, void *const *grads, float *dh, float *dx,
                          float *x_rebuilt, void *wsv, size_t ws_bytes, void *stream, void *const *flow_events, int resume)
{
    int rc = cfg_check(cf);
    if (rc) return rc;
    int T;
    rc = shape_check(cf, B, N, F, &T);
    if (rc) return rc;
    if (!params || !packed || (!z && !resume) || !h || !dz || !dlogdet || !grads || !wsv) return WG_EINVAL;
    const ModelWs W = model_ws_layout(cf, B, T, 1);
    if (W.total * sizeof(float) > ws_bytes) return WG_EWORKSPACE;
    Ctx cx = {(hipStream_t)stream, 0, cf->precision};
    const float *pk = (const float *)packed;
    const float *const *p = (const float *const *)params;
    float *const *gr = (float *const *)grads;
    const ModelPack M = model_pack_layout(cf);
    float *ws = (float *)wsv;
    const Geo g = W.g;
    const int G = cf->n_group;
    PRef X = pref(ws + W.X, W.Gp), dX = pref(ws + W.dX, W.Gp);
    layer_sync_clear(cx, ws, W.wn.lsync);                   // (the recompute pass may take the one-launch layer: an aborted call must not leave its counters set)
    if (!resume) WG_LAUNCH(cx, squeeze_kernel, dim3((T + 255) / 256, B), dim3(256), 0, z, X, g, G, N);
    WG_LAUNCH(cx, squeeze_kernel, dim3((T + 255) / 256, B), dim3(256), 0, dz, dX, g, G, N);
    if (!resume) {
        run_upsample(cx, cf, pk + M.up_w, pk + M.up_bias, h, F, g, pref(ws + W.Y, W.auxp), nullptr);
        if (cx.prec == 2) run_to_splane(cx, g, pref(ws + W.Y, W.auxp), cf->n_mels, ws + W.YS, W.auxp);
    }
    bool have_first = resume != 0;                        // the first flow visited still holds its activations
    const bool keep = cf->keep_activations != 0;          // every flow does: the forward (wg_forward / wg_train_step) ran in this workspace
    if (cx.err == 0 && hipMemsetAsync(ws + W.dY, 0, (size_t)B * W.auxp * g.P * sizeof(float), cx.st) != hipSuccess) cx.err = WG_ELAUNCH;
    WnRun r;
    r.g = g; r.ws = ws; r.w = W.wn; r.Y = ws + W.Y; r.YS = ws + W.YS; r.save = 1;
    // AffineCouplingFunc.backward (efficient_modules.py:118-154) on channels [base, base + c_k)
    auto coupling_bwd = [&](int k, int base) {
        PRef Xk = pref(ws + W.X, W.Gp, base), dXk = pref(ws + W.dX, W.Gp, base);
        r.d = flow_wn(cf, k); r.L = wn_pack_layout(r.d); r.pk = pk + M.wn[k]; r.X = Xk;
        if (keep) r.w = W.flow(k);                                                           // stored by the forward (memory_efficient=False)
        else if (!have_first) wn_forward(cx, r);                                             // recompute :127-130
        have_first = false;
        run_end_affine(cx, r, AFF_BWD, dXk, nullptr, nullptr, dlogdet, nullptr);              // :132-148 (log_s.sum feeds logdet[b], waveglow.py:175)
        wn_backward(cx, r, p + wn_table_off(cf, k), gr + wn_table_off(cf, k), dXk, ws + W.dY);
    };
    // Conv1x1Func.backward (efficient_modules.py:230-244)
    auto invconv_bwd = [&](int k, int base) {
        const int c = flow_channels(cf, k);
        const float *lu = pk + M.lu + (size_t)k * WG_LU_STRIDE;
        PRef Xk = pref(ws + W.X, W.Gp, base), dXk = pref(ws + W.dX, W.Gp, base);
#if !defined(WG_OPT_NO_FUSED_INVCONV_BWD)
        const dim3 fgrid((T + 256 * WG_ICB_T - 1) / (256 * WG_ICB_T), B);
        if (c <= 8 && (c & 1) == 0 && (size_t)fgrid.x * fgrid.y * c * c <= W.wn.slab_floats) {      // one pass: x, dx and the shares of dW
            float *part = ws + W.wn.slab;
            const float *Wm = lu, *Wi = lu + WG_MAXC * WG_MAXC;
            switch (c) {
            case 2: WG_LAUNCH(cx, invconv_bwd_kernel<2>, fgrid, dim3(256), 0, Xk, dXk, Wm, Wi, g, part); break;
            case 4: WG_LAUNCH(cx, invconv_bwd_kernel<4>, fgrid, dim3(256), 0, Xk, dXk, Wm, Wi, g, part); break;
            case 6: WG_LAUNCH(cx, invconv_bwd_kernel<6>, fgrid, dim3(256), 0, Xk, dXk, Wm, Wi, g, part); break;
            default: WG_LAUNCH(cx, invconv_bwd_kernel<8>, fgrid, dim3(256), 0, Xk, dXk, Wm, Wi, g, part); break;
            }
            WgradOut wo;
            wo.nsplit = (int)(fgrid.x * fgrid.y); wo.Mp = c; wo.Np = c;
            run_finalize(cx, part, wo, 0, c, c, 1, 0, 1, 0, nullptr, nullptr, nullptr, gr[3 + k],
                         lu + WG_MAXC * WG_MAXC, dlogdet, B, (float)T);                       // + W^-T dlogdet T :242
            return;
        }
#endif
        run_mix(cx, g, Xk, c, lu + WG_MAXC * WG_MAXC, 0);                                     // x = W^-1 z   :235-237
        WSegSpec sa = {dXk.p, dXk.Cp, dXk.ch0, c, 0, nullptr, 0, 0}, sb = {Xk.p, Xk.Cp, Xk.ch0, c, 0, nullptr, 0, 0};
        const int prec_keep = cx.prec;
        cx.prec = 0;                                                                          // tiny c x c product: keep it exact
        WgradOut wo = run_wgrad(cx, g, &sa, 1, &sb, 1, ws + W.wn.slab, W.wn.slab_floats);     // dW = dz x^T     :240
        cx.prec = prec_keep;
        run_finalize(cx, ws + W.wn.slab, wo, 0, c, c, 1, 0, 1, 0, nullptr, nullptr, nullptr, gr[3 + k],
                     lu + WG_MAXC * WG_MAXC, dlogdet, B, (float)T);                           // + W^-T dlogdet T :242
        run_mix(cx, g, dXk, c, lu, 1);                                                        // dx = W^T dz  :239
    };
    if (!cf->reverse_mode) {
        int base = G - flow_channels(cf, cf->n_flows - 1);
        for (int k = cf->n_flows - 1; k >= 0; --k) {
            coupling_bwd(k, base);
            invconv_bwd(k, base);
            record_event(cx, flow_events, k);                                                        // flow k's grads final
            if (k % cf->n_early_every == 0 && k) base -= cf->n_early_size;
        }
    } else {                                              // reverse_mode architecture: per flow the 1x1 came last, flows ran n-1..0
        int base = 0;
        for (int k = 0; k < cf->n_flows; ++k) {
            if (k % cf->n_early_every == 0 && k) base += cf->n_early_size;
            invconv_bwd(k, base);
            coupling_bwd(k, base);
            record_event(cx, flow_events, k);
        }
    }
    if (x_rebuilt) WG_LAUNCH(cx, unsqueeze_kernel, dim3((T + 255) / 256, B), dim3(256), 0, X, x_rebuilt, g, G, N);
    if (dx) WG_LAUNCH(cx, unsqueeze_kernel, dim3((T + 255) / 256, B), dim3(256), 0, dX, dx, g, G, N);
    // upsampler backward (+ its weight norm)
    WG_LAUNCH(cx, upsample_bwd_kernel, dim3(cf->n_mels), dim3(256), (size_t)(cf->up_kernel + 512) * sizeof(float), h, pk + M.up_w,
              pref(ws + W.dY, W.auxp), g, cf->n_mels, F, cf->up_kernel, cf->up_stride, cf->up_pad, p[1], p[2], gr[0], gr[1], gr[2], dh);
    record_event(cx, flow_events, cf->n_flows);
    return cx.err;
}

int wg_backward(const wg_config *cf, const void *const *params, const void *packed, const float *z, const float *h,
                const float *dz, const float *dlogdet, int B, int N, int F, void *const *grads, float *dh, float *dx,
                float *x_rebuilt, void *wsv, size_t ws_bytes, void *stream, void *const *flow_events)
{
    return model_backward(cf, params, packed, z, h, dz, dlogdet, B, N, F, grads, dh, dx, x_rebuilt, wsv, ws_bytes, stream, flow_events, 0);
}

// The whole training step of model/lightning.py:52-56 in one call: z, logdet = model(x, h); loss = WaveGlowLoss(sigma)(z, logdet);
// loss.backward().  Same kernels as wg_forward + wg_nll_loss + wg_nll_loss_backward + wg_backward, but the forward runs in the
// backward's workspace and keeps the last flow's layers, so the backward starts without recomputing that flow and without a second
// squeeze / upsample.  scratch: wg_train_scratch_floats(B, N) floats (d loss / d z, d loss / d logdet, the loss kernels' partials).
static void run_nll(Ctx &cx, const float *z, const float *logdet, int B, int N, float inv_s2, int elementwise_mean, float *loss,
                    float *metrics, float *part)
{
    WG_LAUNCH(cx, nll_partial_kernel, dim3(WG_NLL_BLK, B), dim3(256), 0, z, N, part);
    WG_LAUNCH(cx, nll_finish_kernel, dim3(1), dim3(256), 0, (const float *)part, logdet, B, N, inv_s2, elementwise_mean, loss, metrics);
}
size_t wg_nll_scratch_floats(int B) { return B < 1 ? 0 : (size_t)2 * WG_NLL_BLK * B; }
size_t wg_train_scratch_floats(int B, int N) { return (B < 1 || N < 1) ? 0 : (size_t)B * N + B + wg_nll_scratch_floats(B); }

int wg_train_step(const wg_config *cf, const void *const *params, const void *packed, const float *audio, const float *h,
                  int B, int N, int F, float sigma, int elementwise_mean, float *z, float *logdet, float *loss, float *metrics,
                  void *const *grads, float *dh, float *scratch, void *ws, size_t ws_bytes, void *stream, void *const *flow_events)
{
    if (!z || !logdet || !loss || !scratch || !(sigma > 0.f)) return WG_EINVAL;
    int rc = model_run_fwd_or_inv(cf, packed, audio, h, B, N, F, 0, z, logdet, ws, ws_bytes, stream, 1, 1);
    if (rc) return rc;
    Ctx cx = {(hipStream_t)stream, 0, 0};
    const float inv_s2 = 1.0f / (sigma * sigma);
    float *dz = scratch, *dld = scratch + (size_t)B * N, *part = dld + B;
    run_nll(cx, z, logdet, B, N, inv_s2, elementwise_mean, loss, metrics, part);
    const size_t n = (size_t)B * N;
    WG_LAUNCH(cx, nll_loss_bwd_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, z, B, N, inv_s2, elementwise_mean,
              (const float *)nullptr, dz, dld);
    if (cx.err) return cx.err;
    return model_backward(cf, params, packed, nullptr, h, dz, dld, B, N, F, grads, dh, nullptr, nullptr, ws, ws_bytes, stream, flow_events, 1);
}

int wg_nll_loss(const float *z, const float *logdet, int B, int N, float sigma, int elementwise_mean, float *loss, float *metrics,
                float *scratch, void *stream)
{
    if (!z || !logdet || !loss || !scratch || B < 1 || N < 1 || !(sigma > 0.f)) return WG_EINVAL;
    Ctx cx = {(hipStream_t)stream, 0, 0};
    run_nll(cx, z, logdet, B, N, 1.0f / (sigma * sigma), elementwise_mean, loss, metrics, scratch);
    return cx.err;
}
int wg_nll_loss_backward(const float *z, int B, int N, float sigma, int elementwise_mean, const float *dloss, float *dz,
                         float *dlogdet, void *stream)
{
    if (!z || !dz || !dlogdet || B < 1 || N < 1 || !(sigma > 0.f)) return WG_EINVAL;
    Ctx cx = {(hipStream_t)stream, 0, 0};
    const size_t n = (size_t)B * N;
    WG_LAUNCH(cx, nll_loss_bwd_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, z, B, N, 1.0f / (sigma * sigma), elementwise_mean,
              dloss, dz, dlogdet);
    return cx.err;
}

int wg_adam_step(float *param, const float *grad, float *exp_avg, float *exp_avg_sq, size_t n, double lr, double beta1, double beta2,
                 double eps, double weight_decay, int step, void *stream)
{
    if (!param || !grad || !exp_avg || !exp_avg_sq || step < 1 || !(lr >= 0.0) || !(beta1 >= 0.0 && beta1 < 1.0) ||
        !(beta2 >= 0.0 && beta2 < 1.0))
        return WG_EINVAL;
    if (n == 0) return WG_OK;
    if (((uintptr_t)param | (uintptr_t)grad | (uintptr_t)exp_avg | (uintptr_t)exp_avg_sq) & 15) return WG_EINVAL;   // float4 access
    Ctx cx = {(hipStream_t)stream, 0, 0};
    AdamArgs a;
    a.p = param; a.g = grad; a.m = exp_avg; a.v = exp_avg_sq; a.n = n;
    // hyper-parameters arrive as doubles and every derived scalar is rounded to fp32 once, as torch does with its Python floats
    const double bc1 = 1.0 - std::pow(beta1, (double)step), bc2 = 1.0 - std::pow(beta2, (double)step);
    a.one_minus_b1 = (float)(1.0 - beta1); a.b2 = (float)beta2; a.one_minus_b2 = (float)(1.0 - beta2); a.eps = (float)eps;
    a.wd = (float)weight_decay;
    a.step_size = (float)(lr / bc1);
    a.inv_bc2_sqrt = (float)(1.0 / std::sqrt(bc2));
    const size_t n4 = (n + 3) / 4;
    const unsigned blocks = (unsigned)std::min<size_t>((n4 + 255) / 256, 256 * 16);
    WG_LAUNCH(cx, adam_kernel, dim3(blocks), dim3(256), 0, a);
    return cx.err;
}

int wg_upsample(const wg_config *cf, const void *packed, const float *h, int B, int F, int T, float *y, void *stream)
{
    int rc = cfg_check(cf);
    if (rc) return rc;
    if (!packed || !h || !y || B < 1 || F < 1 || T < 1) return WG_EINVAL;
    const long L = (long)(F - 1) * cf->up_stride - 2 * cf->up_pad + cf->up_kernel;
    if (T > L) return WG_ESHAPE;
    Ctx cx = {(hipStream_t)stream, 0, 0};
    const float *pk = (const float *)packed;
    const ModelPack M = model_pack_layout(cf);
    const Geo g = make_geo(B, T, 0);
    run_upsample(cx, cf, pk + M.up_w, pk + M.up_bias, h, F, g, pnull(), y);
    return cx.err;
}

// ---- WaveFlow -------------------------------------------------------------------------------------
int wg_wf_param_count(const wg_wf_config *cf) { return cf ? wf_nparams(cf) : WG_EINVAL; }
size_t wg_wf_packed_bytes(const wg_wf_config *cf) { return wf_check(cf) ? 0 : wf_pack_layout(cf).total * sizeof(float); }
size_t wg_wf_workspace_bytes(const wg_wf_config *cf, int B, int N, int mode)
{
    if (wf_check(cf) || B < 1 || N < 1 || N % cf->n_group) return 0;
    return wf_ws_layout(cf, B, N / cf->n_group, mode ? 1 : 0).total * sizeof(float);
}
size_t wg_wf_tape_bytes(const wg_wf_config *cf, int B, int N)
{
    if (wf_check(cf) || B < 1 || N < 1 || N % cf->n_group) return 0;
    const WfWs W = wf_ws_layout(cf, B, N / cf->n_group, 0);
    return (size_t)(cf->flows + 1) * W.g.B * W.g.P * sizeof(float);
}

int wg_wf_pack_weights(const wg_wf_config *cf, const void *const *params, void *packed, void *stream)
{
    int rc = wf_check(cf);
    if (rc) return rc;
    if (!params || !packed) return WG_EINVAL;
    Ctx cx = {(hipStream_t)stream, 0, 0};
    const float *const *p = (const float *const *)params;
    float *pk = (float *)packed;
    const WfPack L = wf_pack_layout(cf);
    const WnD d = wf_wn(cf);
    const WnPack WL = wn_pack_layout(d);
    const int s = 256 / cf->n_group;
    float *ones = pk + L.ones;
    WG_LAUNCH(cx, fill_rows_kernel, dim3(WG_ONES / 256, 1, 1), dim3(256), 0, pref(ones, 1), Geo{1, WG_ONES, WG_ONES, 0, WG_ONES}, (const float *)nullptr, 1.0f);
    JobBatch jb(&cx);
    jb.norm(p[1], p[2], pk + L.up_scale, cf->n_mels, cf->n_mels * (2 * s + 1));      // ConvTranspose1d: dim 0 is the input channel
    for (int k = 0; k < cf->flows; ++k) wn_pack_norms(jb, d, WL, p + 3 + wf_pf(cf) * k, pk + L.wn[k]);
    jb.flush_norm();
    EffBatch eb(&cx);
    for (int k = 0; k < cf->flows; ++k) wn_pack_eff(eb, d, WL, p + 3 + wf_pf(cf) * k, pk + L.wn[k]);
    eb.flush();
    FoldBatch fb(&cx);
    for (int k = 0; k < cf->flows; ++k) wn_pack_fold(fb, d, WL, p + 3 + wf_pf(cf) * k, pk + L.wn[k]);
    fb.flush();
    for (int k = 0; k < cf->flows; ++k) wn_pack_mats(jb, d, WL, p + 3 + wf_pf(cf) * k, pk + L.wn[k], ones);
    jb.flush_pack();
    ImgBatch ib(&cx);
    for (int k = 0; k < cf->flows; ++k) wn_pack_images(ib, d, WL, pk + L.wn[k]);
    ib.flush();
    if (cf->use_conv1x1) {                                   // W, W^-1, logdet W of every flow's 1x1 (efficient_modules.py:37-41,49-54)
        LuBigArgs lu;
        memset(&lu, 0, sizeof(lu));
        lu.n = cf->flows; lu.c = cf->n_group; lu.ostride = L.mix_stride; lu.out = pk + L.mix;
        for (int k = 0; k < cf->flows; ++k) lu.W[k] = p[3 + wf_pf(cf) * cf->flows + k];
        const size_t lds = ((size_t)lu.c * (lu.c + 1) + lu.c) * sizeof(float);
        if (!cx.err) cx.err = ensure_dynamic_lds((const void *)lu_big_kernel, 0, lds);
        WG_LAUNCH(cx, lu_big_kernel, dim3(cf->flows), dim3(256), lds, lu);
    }
    return cx.err;
}

// WaveFlow._upsample_h (waveflow.py:255-257) alone: mel[B,n_mels,F] -> y[B,n_mels,T], T <= F s - 2 (s // 2) + 2 s + 1 (plain layout).
int wg_wf_upsample(const wg_wf_config *cf, const void *const *params, const void *packed, const float *mel, int B, int F, int T,
                   float *y, void *stream)
{
    int rc = wf_check(cf);
    if (rc) return rc;
    if (!params || !packed || !mel || !y || B < 1 || F < 1 || T < 1) return WG_EINVAL;
    const int s = 256 / cf->n_group;
    if (T > F * s - 2 * (s / 2) + 2 * s + 1) return WG_ESHAPE;
    Ctx cx = {(hipStream_t)stream, 0, 0};
    const float *const *p = (const float *const *)params;
    const WfPack L = wf_pack_layout(cf);
    WfUpArgs a;
    a.mel = mel; a.v = p[2]; a.scale = (const float *)packed + L.up_scale; a.bias = p[0];
    a.M = cf->n_mels; a.F = F; a.K = 2 * s + 1; a.s = s; a.pad = s / 2;
    a.Y = pref(y, cf->n_mels); a.gi = Geo{B, T, T, 0, T, 0};          // a "plane" with no halo and pitch T = the plain [B][M][T] layout
    WG_LAUNCH(cx, wf_upsample_fwd_kernel, dim3((T + 255) / 256, cf->n_mels, B), dim3(256), 0, a);
    return cx.err;
}

// WaveFlow.forward_computation (waveflow.py:182-208).  tape (optional, wg_wf_tape_bytes, zero-initialised once by the caller) receives
// the input of every flow and the final state: what wg_wf_backward needs, since this flow has no cheap inverse to rebuild them from.
int wg_wf_forward(const wg_wf_config *cf, const void *const *params, const void *packed, const float *audio, const float *mel,
                  int B, int N, int F, float *z, float *logdet, void *tape, void *wsv, size_t ws_bytes, void *stream)
{
    int rc = wf_check(cf);
    if (rc) return rc;
    int Wd;
    rc = wf_shape(cf, B, N, F, &Wd);
    if (rc) return rc;
    if (!params || !packed || !audio || !mel || !z || !logdet || !wsv) return WG_EINVAL;
    const WfWs W = wf_ws_layout(cf, B, Wd, 0);
    if (W.total * sizeof(float) > ws_bytes) return WG_EWORKSPACE;
    Ctx cx = {(hipStream_t)stream, 0, cf->precision};
    const float *const *p = (const float *const *)params;
    const float *pk = (const float *)packed;
    const WfPack L = wf_pack_layout(cf);
    float *ws = (float *)wsv;
    const Geo g = W.g;
    const size_t xplane = (size_t)g.B * g.P;
    auto xk = [&](int k) { return tape ? (float *)tape + (size_t)k * xplane : ws + W.X[k & 1]; };
    layer_sync_clear(cx, ws, W.wn.lsync);
    WG_LAUNCH(cx, wf_squeeze_kernel, dim3((g.T + 255) / 256, g.B), dim3(256), 0, audio, pref(xk(0), 1), g, N);        // waveflow.py:186
    wf_upsample(cx, cf, p, pk, L, mel, F, W, ws);                                                                    // :183,187
    WnRun r;
    r.d = wf_wn(cf); r.L = wn_pack_layout(r.d); r.g = g; r.ws = ws; r.w = W.wn; r.Y = ws + W.Y; r.YS = ws + W.YS; r.save = 0;
    const int conv = cf->use_conv1x1;
    for (int k = 0; k < cf->flows; ++k) {
        r.pk = pk + L.wn[k]; r.X = pref(xk(k), 1);
        wn_forward(cx, r);                                                                                           // :197
        // x_next = cat(flip(xout), x0), or with the 1x1 conv W cat(x0, xout)                                        // :198-206
        wf_couple(cx, r, p[3 + wf_pf(cf) * k + 36], wf_end_bias(cf, p, k), 0, pref(xk(k), 1), pref(conv ? ws + W.Xt : xk(k + 1), 1), pnull(), pnull(), nullptr,
                  ws + W.rowsum + (size_t)k * g.B, 0, conv);
        if (conv) run_hmix(cx, g, pref(ws + W.Xt, 1), pref(xk(k + 1), 1), pk + L.mix + (size_t)k * L.mix_stride, 0);
    }
    WG_LAUNCH(cx, wf_logdet_kernel, dim3((B + 63) / 64), dim3(64), 0, ws + W.rowsum, cf->flows, B, g.rows, logdet,
              conv ? pk + L.mix : (const float *)nullptr, L.mix_stride, (float)g.T);
    WG_LAUNCH(cx, wf_unsqueeze_kernel, dim3((g.T + 255) / 256, g.B), dim3(256), 0, pref(xk(cf->flows), 1), g, N, z);  // :208
    return cx.err;
}

// WN2D.forward on its own (waveflow.py:128-135)
int wg_wf_wn_apply(const wg_wf_config *cf, const void *const *params, const void *packed, const float *x, const float *y, int B, int rows, int Wd,
                   float *log_s, float *t, void *wsv, size_t ws_bytes, void *stream)
{
    int rc = wf_check(cf);
    if (rc) return rc;
    if (!params || !params[3 + 36] || !packed || !x || !y || !log_s || !t || !wsv || B < 1 || Wd < 1 || rows < 1 || rows > cf->n_group) return WG_EINVAL;
    const WfWs W = wf_ws_layout(cf, B, Wd, 0);
    if (W.total * sizeof(float) > ws_bytes) return WG_EWORKSPACE;
    Ctx cx = {(hipStream_t)stream, 0, cf->precision};
    const float *pk = (const float *)packed;
    const WfPack L = wf_pack_layout(cf);
    float *ws = (float *)wsv;
    const Geo g = W.g;
    layer_sync_clear(cx, ws, W.wn.lsync);
    WG_LAUNCH(cx, wf_rows_in_kernel, dim3((g.T + 255) / 256, g.B), dim3(256), 0, x, pref(ws + W.X[0], 1), g, rows);
    WG_LAUNCH(cx, import_kernel, dim3((Wd + 255) / 256, cf->n_mels, B), dim3(256), 0, y, pref(ws + W.Y, W.auxp), W.gi, cf->n_mels);
    if (cx.prec == 2) run_to_splane(cx, W.gi, pref(ws + W.Y, W.auxp), cf->n_mels, ws + W.YS, W.auxp);
    WnRun r;
    r.d = wf_wn(cf); r.L = wn_pack_layout(r.d); r.g = g; r.ws = ws; r.w = W.wn; r.Y = ws + W.Y; r.YS = ws + W.YS; r.save = 0;
    r.pk = pk + L.wn[0]; r.X = pref(ws + W.X[0], 1);
    wn_forward(cx, r);
    WfCoupleArgs a;
    memset(&a, 0, sizeof(a));
    a.endw = (const float *)params[3 + 36];                    // end.weight [2][Cs][1][1], read as it is (like wf_couple)
    a.endb = wf_end_bias(cf, (const float *const *)params, 0);
    a.S = pref(ws + W.wn.skip, r.d.Cs); a.Cs = r.d.Cs;
    a.g = g; a.mode = 3; a.raw_ls = log_s; a.raw_t = t; a.raw_rows = rows;
    WG_LAUNCH(cx, wf_couple_kernel, dim3(g.B), dim3(256), 0, a);
    return cx.err;
}

// What autograd computes upstream for `log_s, t = wn2d(x, y)` (waveflow.py:128-135, an ordinary differentiable module): the recompute of
// wg_wf_wn_apply with the layers kept, the gradients of (log_s, t) as the seeds of the WN backward, then wn_backward as inside wg_wf_backward.
int wg_wf_wn_backward(const wg_wf_config *cf, const void *const *params, const void *packed, const float *x, const float *y,
                      const float *dlog_s, const float *dt, int B, int rows, int Wd, float *dx, float *dy, void *const *grads,
                      void *wsv, size_t ws_bytes, void *stream)
{
    int rc = wf_check(cf);
    if (rc) return rc;
    if (!params || !params[3 + 36] || !packed || !x || !y || !dlog_s || !dt || !grads || !wsv || B < 1 || Wd < 1 || rows < 1 || rows > cf->n_group)
        return WG_EINVAL;
    const WfWs W = wf_ws_layout(cf, B, Wd, 1);
    if (W.total * sizeof(float) > ws_bytes) return WG_EWORKSPACE;
    Ctx cx = {(hipStream_t)stream, 0, cf->precision};
    const float *const *p = (const float *const *)params;
    float *const *gr = (float *const *)grads;
    const float *pk = (const float *)packed;
    const WfPack L = wf_pack_layout(cf);
    float *ws = (float *)wsv;
    const Geo g = W.g;
    const dim3 rgrid((g.T + 255) / 256, g.B);
    layer_sync_clear(cx, ws, W.wn.lsync);
    WG_LAUNCH(cx, wf_rows_in_kernel, rgrid, dim3(256), 0, x, pref(ws + W.X[0], 1), g, rows);
    WG_LAUNCH(cx, import_kernel, dim3((Wd + 255) / 256, cf->n_mels, B), dim3(256), 0, y, pref(ws + W.Y, W.auxp), W.gi, cf->n_mels);
    if (cx.prec == 2) run_to_splane(cx, W.gi, pref(ws + W.Y, W.auxp), cf->n_mels, ws + W.YS, W.auxp);
    if (cx.err == 0 && hipMemsetAsync(ws + W.dYrow, 0, (size_t)B * W.auxp * W.gi.P * sizeof(float), cx.st) != hipSuccess) cx.err = WG_ELAUNCH;
    WnRun r;
    r.d = wf_wn(cf); r.L = wn_pack_layout(r.d); r.g = g; r.ws = ws; r.w = W.wn; r.Y = ws + W.Y; r.YS = ws + W.YS; r.save = 1;
    r.gi = W.gi; r.rs = ws + W.rs; r.rs_step = W.rs_step;
    r.pk = pk + L.wn[0]; r.X = pref(ws + W.X[0], 1);
    wn_forward(cx, r);
    WG_LAUNCH(cx, wf_seed_kernel, rgrid, dim3(256), 0, dlog_s, dt, pref(ws + W.wn.G, r.L.kp_end), pref(ws + W.dX[0], 1), g, rows);
    wn_backward(cx, r, p + 3, gr + 3, pref(ws + W.dX[0], 1), ws + W.dYrow);
    if (dx) WG_LAUNCH(cx, wf_rows_out_kernel, rgrid, dim3(256), 0, pref(ws + W.dX[0], 1), g, rows, dx);
    if (dy) WG_LAUNCH(cx, export_kernel, dim3((Wd + 255) / 256, cf->n_mels, B), dim3(256), 0, pref(ws + W.dYrow, W.auxp), dy, W.gi, cf->n_mels, 1.0f);
    return cx.err;
}

// WaveFlow.reverse_computation (waveflow.py:210-253): per flow (last first) flip, then one height row at a time -- WN2D on row r
// from rows <= r (what reverse_mode_forward's ring buffers hold), x[r+1] = (z[r+1] - t[r]) / exp(log_s[r]).
int wg_wf_inverse(const wg_wf_config *cf, const void *const *params, const void *packed, const float *z, const float *mel,
                  int B, int N, int F, float *x, float *logdet, void *wsv, size_t ws_bytes, void *stream)
{
    int rc = wf_check(cf);
    if (rc) return rc;
    int Wd;
    rc = wf_shape(cf, B, N, F, &Wd);
    if (rc) return rc;
    if (!params || !packed || !z || !mel || !x || !logdet || !wsv) return WG_EINVAL;
    const WfWs W = wf_ws_layout(cf, B, Wd, 1);
    if (W.total * sizeof(float) > ws_bytes) return WG_EWORKSPACE;
    Ctx cx = {(hipStream_t)stream, 0, cf->precision};
    const float *const *p = (const float *const *)params;
    const float *pk = (const float *)packed;
    const WfPack L = wf_pack_layout(cf);
    float *ws = (float *)wsv;
    const Geo g = W.g;
    const int H = g.rows;
    const dim3 rgrid((g.T + 255) / 256, g.B), igrid((g.T + 255) / 256, B);
    float *Z = ws + W.X[0], *Zf = ws + W.X[1], *Xb = ws + W.dX[0];
    layer_sync_clear(cx, ws, W.wn.lsync);
    WG_LAUNCH(cx, wf_squeeze_kernel, rgrid, dim3(256), 0, z, pref(Z, 1), g, N);
    wf_upsample(cx, cf, p, pk, L, mel, F, W, ws);
    WnRun r;
    r.d = wf_wn(cf); r.L = wn_pack_layout(r.d); r.g = g; r.ws = ws; r.w = W.wn; r.Y = ws + W.Y; r.YS = ws + W.YS; r.save = 1;
    for (int k = cf->flows - 1; k >= 0; --k) {
        if (cf->use_conv1x1) run_hmix(cx, g, pref(Z, 1), pref(Zf, 1), pk + L.mix + (size_t)k * L.mix_stride + (size_t)H * H, 0);   // z = W^-1 z  :224-226
        else WG_LAUNCH(cx, wf_flip_kernel, rgrid, dim3(256), 0, pref(Z, 1), pref(Zf, 1), g);                         // :222
        WG_LAUNCH(cx, wf_copy_row_kernel, igrid, dim3(256), 0, pref(Zf, 1), pref(Xb, 1), g, 0, 0);                   // :228
        r.pk = pk + L.wn[k]; r.X = pref(Xb, 1);
        // One row step's launches, recorded: they are the same for every row but for the row index, so ONE persistent kernel walks
        // rows x stages on the device with a grid barrier where the launches had kernel boundaries (wg_stage.h).
        // MEASURED and therefore OFF by default (-DWG_OPT_ROWWALK turns it on; bit-identical results): 105.8 against 99.3 ms for a 0.7 s
        // utterance, 227 against 214 ms for 10 s.  A stage is not its launch overhead: the 64 x 64-tile conv of a row step is a chain of
        // 21 chunk latencies on 8 of 256 CUs (~8 us), and a grid barrier with the release / acquire fences that make planes visible
        // across XCDs (L2 write-back + invalidate) costs about what a kernel boundary does.  What the row step needs is more workgroups
        // per stage (the K range of a tile split over several CUs), see DESIGN.md section 9.
        bool walked = false;
#if defined(WG_OPT_ROWWALK)
        {
            StageRec rec;
            cx.rec = &rec;
            cx.row_sel1 = 1;
            wn_forward(cx, r);
            wf_couple(cx, r, p[3 + wf_pf(cf) * k + 36], wf_end_bias(cf, p, k), 2, pref(Zf, 1), pref(Xb, 1), pnull(), pnull(), nullptr, ws + W.rowsum + (size_t)k * g.B, 0);
            cx.rec = nullptr;
            cx.row_sel1 = 0;
#if defined(WG_DBG_ROWWALK_HOSTREPLAY)   // debugging aid: the recorded program replayed launch by launch from the host
            if (rec.ok && !cx.err) {
                for (int row = 0; row < H - 1; ++row)
                    for (size_t si = 0; si < rec.st.size(); ++si) {
                        WgStage stg = rec.st[si];
                        if (stg.kind <= WGS_CONV_RESSKIP) {
                            stg.u.conv.c.row_sel1 = row + 1;
                            const dim3 gh(stg.nblocks);
                            if (stg.kind == WGS_CONV_STORE) WG_LAUNCH(cx, convgemm16h_kernel<EPI_STORE>, gh, dim3(512), 0, stg.u.conv);
                            else if (stg.kind == WGS_CONV_GATE) WG_LAUNCH(cx, convgemm16h_kernel<EPI_GATE>, gh, dim3(512), 0, stg.u.conv);
                            else WG_LAUNCH(cx, convgemm16h_kernel<EPI_RESSKIP>, gh, dim3(512), 0, stg.u.conv);
                        } else if (stg.kind == WGS_TOSPLANE) {
                            const WgsToSplane &t = stg.u.tsp;
                            WG_LAUNCH(cx, to_splane_kernel, dim3((t.g.T + 255) / 256, t.cgs, t.g.B), dim3(256), 0, t.src, t.nvalid, t.dst, t.g);
                        } else {
                            stg.u.cpl.row_sel = row;
                            WG_LAUNCH(cx, wf_couple_kernel, dim3(stg.nblocks), dim3(256), 0, stg.u.cpl);
                        }
                    }
                walked = true;
            } else
#endif
            if (rec.ok && !cx.err && !rec.st.empty() && rec.st.size() <= WF_PROG_STAGES) {
                WgStage *prog = reinterpret_cast<WgStage *>(ws + W.prog);
                unsigned *bar = reinterpret_cast<unsigned *>(reinterpret_cast<char *>(prog) + WF_PROG_STAGES * sizeof(WgStage));
                const int n = (int)rec.st.size();
#if defined(WG_DBG_ROWWALK_MEMCPY)       // debugging aid: the program through a host copy (the buffer is leaked on purpose)
                {
                    WgStage *keep = new WgStage[n];
                    memcpy(static_cast<void *>(keep), static_cast<const void *>(rec.st.data()), (size_t)n * sizeof(WgStage));
                    if (hipMemcpyAsync(prog, keep, (size_t)n * sizeof(WgStage), hipMemcpyHostToDevice, cx.st) != hipSuccess) cx.err = WG_ELAUNCH;
                }
                for (int i = n; i < n; i += WGS_PER_STORE) {
#else
                for (int i = 0; i < n; i += WGS_PER_STORE) {
#endif
                    WgsStoreArgs sa;
                    const int m = std::min(WGS_PER_STORE, n - i);
                    memcpy(static_cast<void *>(sa.st), static_cast<const void *>(&rec.st[i]), (size_t)m * sizeof(WgStage));
                    WG_LAUNCH(cx, wgs_store_kernel, dim3(1), dim3(256), 0, sa, m, prog + i);
                }
                if (!cx.err && hipMemsetAsync(bar, 0, 64, cx.st) != hipSuccess) cx.err = WG_ELAUNCH;
                const int grid = std::max(1, std::min(rec.widest, device_cus()));     // every workgroup resident: the barrier needs them all
                WG_LAUNCH(cx, wf_rowsteps_kernel, dim3(grid), dim3(WGS_THREADS), 0, (const WgStage *)prog, n, H - 1, bar, (int *)(bar + 8));
                walked = true;
            }
        }
#endif
        for (int row = 0; row < H - 1 && !walked; ++row) {
            cx.row_sel1 = row + 1;
            wn_forward(cx, r);
            wf_couple(cx, r, p[3 + wf_pf(cf) * k + 36], wf_end_bias(cf, p, k), 2, pref(Zf, 1), pref(Xb, 1), pnull(), pnull(), nullptr,
                      ws + W.rowsum + (size_t)k * g.B, row);                                                         // :236-243
        }
        cx.row_sel1 = 0;
        std::swap(Z, Xb);
    }
    // rowsum rows H-1 are never written by mode 2: they were zeroed with the workspace
    // (the row walk's failure word is only ever written by the -DWG_OPT_ROWWALK build: the default build must not read a workspace
    // region nothing initialises -- a workspace that served another layout could turn every logdet into NaN)
#if defined(WG_OPT_ROWWALK)
    const int *walk_fail = reinterpret_cast<const int *>(reinterpret_cast<const char *>(ws + W.prog) + WF_PROG_STAGES * sizeof(WgStage)) + 8;
#else
    const int *walk_fail = nullptr;
#endif
    WG_LAUNCH(cx, wf_logdet_kernel, dim3((B + 63) / 64), dim3(64), 0, ws + W.rowsum, cf->flows, B, H, logdet,
              cf->use_conv1x1 ? pk + L.mix : (const float *)nullptr, L.mix_stride, -(float)g.T, walk_fail);          // :227-229
    WG_LAUNCH(cx, wf_unsqueeze_kernel, rgrid, dim3(256), 0, pref(Z, 1), g, N, x);
    return cx.err;
}

// Backward of wg_wf_forward + NLL seed: dz [B][N], dlogdet [B] -> every parameter gradient (table order), optional dmel, optional dx.
// Per flow (last first): recompute WN2D from the taped flow input keeping every layer, seed (d log_s, d t) from the coupling, walk
// WN2D backwards.  The conditioning gradient is accumulated per plane row and summed over the height axis at the end.
int wg_wf_backward(const wg_wf_config *cf, const void *const *params, const void *packed, const void *tape, const float *mel,
                   const float *dz, const float *dlogdet, int B, int N, int F, void *const *grads, float *dmel, float *dx,
                   void *wsv, size_t ws_bytes, void *stream)
{
    int rc = wf_check(cf);
    if (rc) return rc;
    int Wd;
    rc = wf_shape(cf, B, N, F, &Wd);
    if (rc) return rc;
    if (!params || !packed || !tape || !mel || !dz || !dlogdet || !grads || !wsv) return WG_EINVAL;
    const WfWs W = wf_ws_layout(cf, B, Wd, 1);
    if (W.total * sizeof(float) > ws_bytes) return WG_EWORKSPACE;
    Ctx cx = {(hipStream_t)stream, 0, cf->precision};
    const float *const *p = (const float *const *)params;
    float *const *gr = (float *const *)grads;
    const float *pk = (const float *)packed;
    const WfPack L = wf_pack_layout(cf);
    float *ws = (float *)wsv;
    const Geo g = W.g;
    const int M = cf->n_mels, s = 256 / cf->n_group, K = 2 * s + 1;
    const size_t xplane = (size_t)g.B * g.P;
    const dim3 rgrid((g.T + 255) / 256, g.B);
    float *dXn = ws + W.dX[0], *dXc = ws + W.dX[1];
    WG_LAUNCH(cx, wf_squeeze_kernel, rgrid, dim3(256), 0, dz, pref(dXn, 1), g, N);
    wf_upsample(cx, cf, p, pk, L, mel, F, W, ws);
    if (cx.err == 0 && hipMemsetAsync(ws + W.dYrow, 0, (size_t)B * W.auxp * W.gi.P * sizeof(float), cx.st) != hipSuccess) cx.err = WG_ELAUNCH;
    WnRun r;
    r.d = wf_wn(cf); r.L = wn_pack_layout(r.d); r.g = g; r.ws = ws; r.w = W.wn; r.Y = ws + W.Y; r.YS = ws + W.YS; r.save = 1;
    r.gi = W.gi; r.rs = ws + W.rs; r.rs_step = W.rs_step;
    const int conv = cf->use_conv1x1, H = g.rows;
    if (conv) {
        const size_t lds = (size_t)2 * H * 65 * sizeof(float);
        if (!cx.err) cx.err = ensure_dynamic_lds((const void *)wf_hgram_kernel, 1, lds);
    }
    for (int k = cf->flows - 1; k >= 0; --k) {
        const float *Xk = (const float *)tape + (size_t)k * xplane;
        if (conv) {
            // Conv1x1 backward over the height axis (efficient_modules.py:235-242): u = W^-1 z (the 1x1's input, rebuilt from the taped
            // flow output), dW = sum dz u^T + W^-T (sum_b dlogdet_b) W_time, du = W^T dz
            const float *mx = pk + L.mix + (size_t)k * L.mix_stride;
            const float *Xn = (const float *)tape + (size_t)(k + 1) * xplane;
            run_hmix(cx, g, pref((float *)Xn, 1), pref(ws + W.Xt, 1), mx + (size_t)H * H, 0);
            WG_LAUNCH(cx, wf_hgram_kernel, dim3((g.T + 63) / 64, B), dim3(256), (size_t)2 * H * 65 * sizeof(float), pref(dXn, 1), pref(ws + W.Xt, 1), g,
                      ws + W.gram);
            float *dWk = gr[3 + wf_pf(cf) * cf->flows + k];
            if (dWk)
                WG_LAUNCH(cx, wf_hgram_reduce_kernel, dim3((H * H + 255) / 256), dim3(256), 0, (const float *)(ws + W.gram), W.gram_blocks, H,
                          mx + (size_t)H * H, dlogdet, B, (float)g.T, dWk);
            run_hmix(cx, g, pref(dXn, 1), pref(ws + W.dXt, 1), mx, 1);
        }
        r.pk = pk + L.wn[k]; r.X = pref((float *)Xk, 1);
        wn_forward(cx, r);
        wf_couple(cx, r, p[3 + wf_pf(cf) * k + 36], wf_end_bias(cf, p, k), 1, pref((float *)Xk, 1), pnull(), pref(conv ? ws + W.dXt : dXn, 1), pref(dXc, 1), dlogdet, nullptr, 0, conv);
        wn_backward(cx, r, p + 3 + wf_pf(cf) * k, gr + 3 + wf_pf(cf) * k, pref(dXc, 1), ws + W.dYrow);
        std::swap(dXn, dXc);
    }
    if (dx) WG_LAUNCH(cx, wf_unsqueeze_kernel, rgrid, dim3(256), 0, pref(dXn, 1), g, N, dx);
    // upsampler backward: sum the per-row conditioning gradient over the height axis, LeakyReLU', transposed conv, weight norm
    Geo g1 = W.gi;
    g1.rows = 1;                                            // the conditioning gradient is already summed over the height axis
    WG_LAUNCH(cx, wf_rowsum_leaky_kernel, dim3((g.T + 255) / 256, M, B), dim3(256), 0, pref(ws + W.dYrow, W.auxp), g1, pref(ws + W.Y, W.auxp),
              W.gi, M, ws + W.gp);
    WfUpBwdArgs a;
    a.mel = mel; a.v = p[2]; a.scale = pk + L.up_scale; a.gp = ws + W.gp;
    a.B = B; a.M = M; a.F = F; a.K = K; a.s = s; a.pad = s / 2; a.W = Wd;
    a.dw = ws + W.dwup; a.dbias = gr[0]; a.dmel = dmel;
    WG_LAUNCH(cx, wf_upsample_bwd_kernel, dim3(M, 8), dim3(256), 0, a);
    WgradOut wo;
    wo.nsplit = 1; wo.Mp = M; wo.Np = M * K;
    run_finalize(cx, ws + W.dwup, wo, 0, M, M * K, 1, 0, 1, 0, p[1], p[2], gr[1], gr[2]);
    return cx.err;
}

// ---- log-mel conditioner --------------------------------------------------------------------------
int wg_melspec_frames(int N, int n_fft, int hop) { return (N < 1 || hop < 1) ? WG_EINVAL : N / hop + 1; }
int wg_melspec(const float *audio, int B, int N, int sr, int n_fft, int hop, double f_min, double f_max, int n_mels, float *mel,
               float *power, void *stream)
{
    if (!audio || !mel || B < 1 || N < 2 || sr < 2 || hop < 1 || n_mels < 1) return WG_EINVAL;
    if (n_fft < 2 || n_fft > WG_MEL_MAXFFT || (n_fft & (n_fft - 1)) || n_mels > 256 || hop > n_fft) return WG_EUNSUPPORTED;
    if (n_fft / 2 + hop / 2 >= N) return WG_ESHAPE;                      // reflection padding needs pad < length
    if (f_max <= 0.0) f_max = (double)(sr / 2);
    if (!(f_min >= 0.0) || !(f_max > f_min)) return WG_EINVAL;
    Ctx cx = {(hipStream_t)stream, 0, 0};
    MelArgs a;
    a.audio = audio; a.mel = mel; a.power = power; a.N = N; a.n_fft = n_fft; a.hop = hop; a.n_mels = n_mels;
    a.frames = N / hop + 1;                                               // (N + n_fft - n_fft) / hop + 1 with center=False
    a.pad_left = n_fft / 2 - hop / 2;
    a.sr_half = sr / 2;
    a.m_min = (float)(2595.0 * std::log10(1.0 + f_min / 700.0));
    a.m_max = (float)(2595.0 * std::log10(1.0 + f_max / 700.0));
    WG_LAUNCH(cx, melspec_kernel, dim3(a.frames, B), dim3(256), 0, a);
    return cx.err;
}

// ---- LowPass / STFTDecimate -----------------------------------------------------------------------
size_t wg_lowpass_workspace_bytes(int B, int T, int n_fft, int hop)
{
    if (B < 1 || T < 1 || n_fft < 2 || hop < 1) return 0;
    return (size_t)B * ((size_t)(T + n_fft) / hop + 1) * n_fft * sizeof(float);
}
int wg_lowpass(const float *x, int B, int T, int n_fft, int hop, int cut_bins, int step, float *out, void *ws, size_t ws_bytes, void *stream)
{
    if (!x || !out || !ws || B < 1 || T < 1 || step < 1 || cut_bins < 1) return WG_EINVAL;
    if (n_fft < 2 || n_fft > WG_MEL_MAXFFT || (n_fft & (n_fft - 1)) || hop < 1 || hop > n_fft || cut_bins > n_fft / 2 + 1) return WG_EUNSUPPORTED;
    if (n_fft / 2 >= T + n_fft) return WG_ESHAPE;
    if (ws_bytes < wg_lowpass_workspace_bytes(B, T, n_fft, hop)) return WG_EWORKSPACE;
    Ctx cx = {(hipStream_t)stream, 0, 0};
    LowPassArgs a;
    a.x = x; a.frames = (float *)ws; a.out = out; a.T = T; a.n_fft = n_fft; a.hop = hop;
    a.nframes = (T + n_fft) / hop + 1;                                  // stft of T + n_fft samples with center=True
    a.cut = cut_bins; a.step = step; a.nout = (T + step - 1) / step;
    WG_LAUNCH(cx, lowpass_frame_kernel, dim3(a.nframes, B), dim3(256), 0, a);
    WG_LAUNCH(cx, lowpass_ola_kernel, dim3((a.nout + 255) / 256, B), dim3(256), 0, a);
    return cx.err;
}

// ---- WSRGlow conditioning front-end ---------------------------------------------------------------
int wg_wsr_cond(const float *c, int B, int L, const float *mu_table, const float *ang_table, float *cond, void *stream)
{
    if (!c || !mu_table || !ang_table || !cond || B < 1 || L < 8) return WG_EINVAL;
    if (L % 8) return WG_ESHAPE;
    Ctx cx = {(hipStream_t)stream, 0, 0};
    const int F = L / 8, nslice = 32, per = (WSR_COND + nslice - 1) / nslice;
    WG_LAUNCH(cx, wsr_cond_kernel, dim3((F + WSR_FT - 1) / WSR_FT, nslice, B), dim3(256), 0, c, L, mu_table, ang_table, cond, per);
    return cx.err;
}

int wg_wsr_cond_pre(const float *c, int B, int L, float *mu_pre, float *ang_pre, void *stream)
{
    if (!c || !mu_pre || !ang_pre || B < 1 || L < 8) return WG_EINVAL;
    if (L % 8) return WG_ESHAPE;
    Ctx cx = {(hipStream_t)stream, 0, 0};
    WG_LAUNCH(cx, wsr_cond_pre_kernel, dim3((L / 8 + 255) / 256, B), dim3(256), 0, c, L, mu_pre, ang_pre);
    return cx.err;
}

int wg_wsr_cond_backward(const float *c, int B, int L, const float *dcond, float *dmu_table, float *dang_table, void *stream)
{
    if (!c || !dcond || !dmu_table || !dang_table || B < 1 || L < 8) return WG_EINVAL;
    if (L % 8) return WG_ESHAPE;
    Ctx cx = {(hipStream_t)stream, 0, 0};
    if (hipMemsetAsync(dmu_table, 0, sizeof(float) * WSR_MU * WSR_MU_DIM, cx.st) != hipSuccess ||
        hipMemsetAsync(dang_table, 0, sizeof(float) * WSR_ANG * WSR_ANG_DIM, cx.st) != hipSuccess)
        return WG_ELAUNCH;
    WG_LAUNCH(cx, wsr_table_grad_kernel<false>, dim3((WSR_MU_DIM + 15) / 16, 8), dim3(256), 0, c, B, L, dcond, dmu_table);
    WG_LAUNCH(cx, wsr_table_grad_kernel<true>, dim3((WSR_ANG_DIM + 15) / 16, WSR_BINS), dim3(256), 0, c, B, L, dcond, dang_table);
    return cx.err;
}

// ---- block level: invertible 1x1 -----------------------------------------------------------------
struct InvWs {
    Geo g;
    int Cp;
    size_t X, dX, lu, slab, slab_floats, total;
};
static InvWs inv_ws_layout(int c, int B, int T)
{
    InvWs w;
    w.g = make_geo(B, T, 0);
    w.Cp = rup(c, WG_BK);
    Bump bp;
    w.X = bp.take((size_t)B * w.Cp * w.g.P);
    w.dX = bp.take((size_t)B * w.Cp * w.g.P);
    w.lu = bp.take(WG_LU_STRIDE);
    w.slab_floats = slab_floats(w.g, WG_TILE, WG_TILE);
    w.slab = bp.take(w.slab_floats);
    w.total = bp.off + 4096;
    return w;
}
size_t wg_invconv_workspace_bytes(int c, int B, int T)
{
    if (c < 1 || c > WG_MAXC || B < 1 || T < 1) return 0;
    return inv_ws_layout(c, B, T).total * sizeof(float);
}

int wg_invconv_apply(const float *Wm, int c, const float *x, int B, int T, int reverse, float *z, float *logdet,
                     void *wsv, size_t ws_bytes, void *stream)
{
    if (!Wm || !x || !z || !logdet || !wsv || c < 1 || c > WG_MAXC || B < 1 || T < 1) return WG_EINVAL;
    const InvWs W = inv_ws_layout(c, B, T);
    if (W.total * sizeof(float) > ws_bytes) return WG_EWORKSPACE;
    Ctx cx = {(hipStream_t)stream, 0, 0};
    float *ws = (float *)wsv;
    LuArgs lu;
    lu.n = 1; lu.out = ws + W.lu; lu.ostride = WG_LU_STRIDE; lu.job[0].W = Wm; lu.job[0].c = c;
    WG_LAUNCH(cx, lu_kernel, dim3(1), dim3(64), 0, lu);
    PRef X = pref(ws + W.X, W.Cp);
    WG_LAUNCH(cx, import_kernel, dim3((T + 255) / 256, c, B), dim3(256), 0, x, X, W.g, c);
    run_mix(cx, W.g, X, c, ws + W.lu + (reverse ? WG_MAXC * WG_MAXC : 0), 0);                 // efficient_modules.py:40 / :53
    WG_LAUNCH(cx, export_kernel, dim3((T + 255) / 256, c, B), dim3(256), 0, X, z, W.g, c, 1.0f);
    WG_LAUNCH(cx, scalar_logdet_kernel, dim3(1), dim3(1), 0, ws + W.lu, reverse ? -(float)T : (float)T, logdet);   // :39 / :51-52
    return cx.err;
}

int wg_invconv_backward(const float *Wm, int c, const float *z, const float *dz, const float *dlogdet, int B, int T, int reverse,
                        float *x, float *dx, float *dW, void *wsv, size_t ws_bytes, void *stream)
{
    if (!Wm || !z || !dz || !dlogdet || !x || !dx || !dW || !wsv || c < 1 || c > WG_MAXC || B < 1 || T < 1) return WG_EINVAL;
    const InvWs W = inv_ws_layout(c, B, T);
    if (W.total * sizeof(float) > ws_bytes) return WG_EWORKSPACE;
    Ctx cx = {(hipStream_t)stream, 0, 0};
    float *ws = (float *)wsv;
    const Geo g = W.g;
    LuArgs lu;
    lu.n = 1; lu.out = ws + W.lu; lu.ostride = WG_LU_STRIDE; lu.job[0].W = Wm; lu.job[0].c = c;
    WG_LAUNCH(cx, lu_kernel, dim3(1), dim3(64), 0, lu);
    const float *Wd = ws + W.lu, *Wi = ws + W.lu + WG_MAXC * WG_MAXC;
    PRef X = pref(ws + W.X, W.Cp), dX = pref(ws + W.dX, W.Cp);
    WG_LAUNCH(cx, import_kernel, dim3((T + 255) / 256, c, B), dim3(256), 0, z, X, g, c);
    WG_LAUNCH(cx, import_kernel, dim3((T + 255) / 256, c, B), dim3(256), 0, dz, dX, g, c);
    WSegSpec sa = {dX.p, dX.Cp, 0, c, 0, nullptr, 0, 0}, sb = {X.p, X.Cp, 0, c, 0, nullptr, 0, 0};
    if (!reverse) {                      // Conv1x1Func.backward  (efficient_modules.py:230-244)
        run_mix(cx, g, X, c, Wi, 0);                                                          // x = W^-1 z
        WgradOut wo = run_wgrad(cx, g, &sa, 1, &sb, 1, ws + W.slab, W.slab_floats);           // dz x^T
        run_finalize(cx, ws + W.slab, wo, 0, c, c, 1, 0, 1, 0, nullptr, nullptr, nullptr, dW, Wi, dlogdet, 1, (float)T);
        run_mix(cx, g, dX, c, Wd, 1);                                                         // dx = W^T dz
    } else {                             // InvConv1x1Func.backward (efficient_modules.py:262-279)
        run_mix(cx, g, X, c, Wd, 0);                                                          // x = W z       :267
        WgradOut wo = run_wgrad(cx, g, &sa, 1, &sb, 1, ws + W.slab, W.slab_floats);           // dw = dz x^T   :274-275
        InvRevFinArgs fa;
        fa.slab = ws + W.slab; fa.nsplit = wo.nsplit; fa.sstride = (size_t)wo.Mp * wo.Np; fa.ldn = wo.Np;
        fa.c = c; fa.Winv = Wi; fa.dlogdet = dlogdet; fa.T = (float)T; fa.dW = dW;
        WG_LAUNCH(cx, invconv_rev_finalize_kernel, dim3(1), dim3(256), 0, fa);                // :276-277
        run_mix(cx, g, dX, c, Wi, 1);                                                         // dx = W^-T dz  :271-273
    }
    WG_LAUNCH(cx, export_kernel, dim3((T + 255) / 256, c, B), dim3(256), 0, X, x, g, c, 1.0f);
    WG_LAUNCH(cx, export_kernel, dim3((T + 255) / 256, c, B), dim3(256), 0, dX, dx, g, c, 1.0f);
    return cx.err;
}

// ---- block level: affine coupling with F = WN ------------------------------------------------------
struct CplWs {
    Geo g;
    int Xp, auxp;
    size_t X, dX, Y, dY, YS, total;
    WnWs wn;
};
static CplWs cpl_ws_layout(const WnD &d, int B, int T, int mode)
{
    CplWs w;
    w.g = make_geo(B, T, d.maxdil() * (d.radix - 1) / 2);
    w.Xp = rup(2 * d.ic, WG_BK) + WG_BK;
    w.auxp = d.auxp();
    Bump bp;
    w.X = bp.take((size_t)B * w.Xp * w.g.P);
    w.Y = bp.take((size_t)B * w.auxp * w.g.P);
    w.dX = w.dY = 0;
    if (mode) {
        w.dX = bp.take((size_t)B * w.Xp * w.g.P);
        w.dY = bp.take((size_t)B * w.auxp * w.g.P);
    }
    w.YS = d.prec == 2 ? bp.take((size_t)B * w.auxp * w.g.P) : 0;
    wn_ws_layout(bp, d, d.ic, w.g, mode, d.prec, w.wn);
    w.total = bp.off + 4096;
    return w;
}
size_t wg_coupling_workspace_bytes(const wg_wn_dims *dd, int B, int T, int mode)
{
    if (!dd || wn_check(wnd_from(dd)) || B < 1 || T < 1) return 0;
    return cpl_ws_layout(wnd_from(dd), B, T, mode).total * sizeof(float);
}

int wg_coupling_apply(const wg_wn_dims *dd, const void *packed, const float *x, const float *y, int B, int T, int reverse,
                      float *z, float *log_s, void *wsv, size_t ws_bytes, void *stream)
{
    if (!dd || !packed || !x || !y || !z || !log_s || !wsv || B < 1 || T < 1) return WG_EINVAL;
    const WnD d = wnd_from(dd);
    int rc = wn_check(d);
    if (rc) return rc;
    const CplWs W = cpl_ws_layout(d, B, T, 0);
    if (W.total * sizeof(float) > ws_bytes) return WG_EWORKSPACE;
    Ctx cx = {(hipStream_t)stream, 0, dd->precision};
    float *ws = (float *)wsv;
    const Geo g = W.g;
    PRef X = pref(ws + W.X, W.Xp);
    layer_sync_clear(cx, ws, W.wn.lsync);
    WG_LAUNCH(cx, import_kernel, dim3((T + 255) / 256, 2 * d.ic, B), dim3(256), 0, x, X, g, 2 * d.ic);
    WG_LAUNCH(cx, import_kernel, dim3((T + 255) / 256, d.aux, B), dim3(256), 0, y, pref(ws + W.Y, W.auxp), g, d.aux);
    if (cx.prec == 2) run_to_splane(cx, g, pref(ws + W.Y, W.auxp), d.aux, ws + W.YS, W.auxp);
    WnRun r;
    r.d = d; r.L = wn_pack_layout(d); r.pk = (const float *)packed + rupz(WG_ONES, 64); r.g = g; r.ws = ws; r.w = W.wn;
    r.X = X; r.Y = ws + W.Y; r.YS = ws + W.YS; r.save = 0;
    wn_forward(cx, r);
    run_end_affine(cx, r, reverse ? AFF_REV : AFF_FWD, pnull(), log_s, nullptr, nullptr, nullptr);
    WG_LAUNCH(cx, export_kernel, dim3((T + 255) / 256, 2 * d.ic, B), dim3(256), 0, X, z, g, 2 * d.ic, 1.0f);
    return cx.err;
}

int wg_wn_apply(const wg_wn_dims *dd, const void *packed, const float *x, const float *y, int B, int T, float *log_s, float *t,
                void *wsv, size_t ws_bytes, void *stream)
{
    if (!dd || !packed || !x || !y || !log_s || !t || !wsv || B < 1 || T < 1) return WG_EINVAL;
    const WnD d = wnd_from(dd);
    int rc = wn_check(d);
    if (rc) return rc;
    const CplWs W = cpl_ws_layout(d, B, T, 0);
    if (W.total * sizeof(float) > ws_bytes) return WG_EWORKSPACE;
    Ctx cx = {(hipStream_t)stream, 0, dd->precision};
    float *ws = (float *)wsv;
    const Geo g = W.g;
    PRef X = pref(ws + W.X, W.Xp);
    layer_sync_clear(cx, ws, W.wn.lsync);
    WG_LAUNCH(cx, import_kernel, dim3((T + 255) / 256, d.ic, B), dim3(256), 0, x, X, g, d.ic);
    WG_LAUNCH(cx, import_kernel, dim3((T + 255) / 256, d.aux, B), dim3(256), 0, y, pref(ws + W.Y, W.auxp), g, d.aux);
    if (cx.prec == 2) run_to_splane(cx, g, pref(ws + W.Y, W.auxp), d.aux, ws + W.YS, W.auxp);
    WnRun r;
    r.d = d; r.L = wn_pack_layout(d); r.pk = (const float *)packed + rupz(WG_ONES, 64); r.g = g; r.ws = ws; r.w = W.wn;
    r.X = X; r.Y = ws + W.Y; r.YS = ws + W.YS; r.save = 0;
    wn_forward(cx, r);
    AffineArgs a;
    memset(&a, 0, sizeof(a));
    const int fromg = affine_source(cx, r, a);
    a.X = X;
    a.log_s_out = log_s; a.t_out = t; a.g = g; a.mode = AFF_RAW;
    launch_end_affine(cx, a, fromg);
    return cx.err;
}

// ---- stand-alone NonCausalLayer / NonCausalLayer2D (waveglow.py:18-46, waveflow.py:14-51) ----
struct LayerWs {
    Geo g, gi;           // gi: one plane row per item (the 2-D layer's conditioning); == g for the 1-D layer
    int Cp, Yp, Dp, Sp, ldA, ldO, KA, R, taps;
    size_t X, Y, gate, res, skip, Acat, WoT, total;
};
static int layer_check(const wg_layer_dims *d)
{
    if (!d || d->res_ch < 16 || d->dil_ch < 32 || d->skip_ch < 16 || d->res_ch % 16 || d->dil_ch % 32 || d->skip_ch % 16) return WG_EUNSUPPORTED;
    if (d->radix < 1 || !(d->radix & 1) || d->dilation < 1 || d->h_dilation < 0) return WG_EUNSUPPORTED;
    const int taps = d->h_dilation > 0 ? d->radix * d->radix : d->radix;
    if (taps > WG_MAX_SEG - 1) return WG_EUNSUPPORTED;
    if ((d->h_dilation > 0) != (d->rows > 0)) return WG_EINVAL;
    return 0;
}
static LayerWs layer_ws_layout(const wg_layer_dims *d, int B, int T)
{
    LayerWs w;
    Bump bp;
    w.gi = make_geo(B, T, d->dilation * (d->radix - 1) / 2);
    w.g = w.gi;
    if (d->rows > 0) { w.g.B = B * d->rows; w.g.rows = d->rows; }
    w.taps = d->h_dilation > 0 ? d->radix * d->radix : d->radix;
    w.Cp = d->res_ch; w.Yp = 2 * d->dil_ch; w.Dp = d->dil_ch; w.Sp = d->skip_ch;
    w.R = d->last_layer ? d->skip_ch : d->res_ch + d->skip_ch;
    w.KA = w.taps * d->res_ch + 2 * d->dil_ch;
    w.ldA = rup(2 * d->dil_ch, WG_TILE); w.ldO = rup(w.R, WG_TILE);
    w.X = bp.take((size_t)w.g.B * w.Cp * w.g.P);
    w.Y = bp.take((size_t)B * w.Yp * w.gi.P);
    w.gate = bp.take((size_t)w.g.B * w.Dp * w.g.P);
    w.res = bp.take((size_t)w.g.B * w.Cp * w.g.P);
    w.skip = bp.take((size_t)w.g.B * w.Sp * w.g.P);
    w.Acat = bp.take((size_t)w.KA * w.ldA);
    w.WoT = bp.take((size_t)d->dil_ch * w.ldO);
    w.total = bp.off + 1024;
    return w;
}
size_t wg_layer_workspace_bytes(const wg_layer_dims *d, int B, int T)
{
    if (layer_check(d) || B < 1 || T < 1) return 0;
    return layer_ws_layout(d, B, T).total * sizeof(float);
}
int wg_layer_apply(const wg_layer_dims *d, const void *const *params, const float *x, const float *y, int B, int T, float *res, float *skip,
                   void *wsv, size_t ws_bytes, void *stream)
{
    int rc = layer_check(d);
    if (rc) return rc;
    if (!params || !params[1] || !params[3] || !x || !y || !skip || (!res && !d->last_layer) || !wsv || B < 1 || T < 1) return WG_EINVAL;
    const LayerWs W = layer_ws_layout(d, B, T);
    if (W.total * sizeof(float) > ws_bytes) return WG_EWORKSPACE;
    Ctx cx = {(hipStream_t)stream, 0, 0};                      // the exact-fp32 conv kernel: fp32 planes, fp32 k-major weights, no images
    float *ws = (float *)wsv;
    const Geo g = W.g;
    const int C = d->res_ch, Cd = d->dil_ch, Cs = d->skip_ch;
    const bool two_d = d->rows > 0;
    // (the whole workspace is zeroed by every call: plane halos and the padding of the weight matrices; this entry point exists for API
    // parity, not for speed)
    if (hipMemsetAsync(ws, 0, W.total * sizeof(float), cx.st) != hipSuccess) return WG_ELAUNCH;
    LayerPackArgs pa;
    pa.g = (const float *)params[0]; pa.v = (const float *)params[1]; pa.dst = ws + W.Acat; pa.rows = 2 * Cd; pa.fan = C * W.taps; pa.ld = W.ldA;
    pa.kind = 0; pa.C = C; pa.Cd = Cd; pa.radix = W.taps;      // (the 2-D weight [2 Cd][C][kh][kw] is a radix^2-tap weight: tap = kh * radix + kw)
    WG_LAUNCH(cx, layer_pack_kernel, dim3(pa.rows), dim3(256), 0, pa);
    pa.g = (const float *)params[2]; pa.v = (const float *)params[3]; pa.dst = ws + W.WoT; pa.rows = W.R; pa.fan = Cd; pa.ld = W.ldO; pa.kind = 1;
    WG_LAUNCH(cx, layer_pack_kernel, dim3(pa.rows), dim3(256), 0, pa);
    PRef X = pref(ws + W.X, W.Cp), Y = pref(ws + W.Y, W.Yp);
    if (two_d) WG_LAUNCH(cx, import2d_kernel, dim3((T + 255) / 256, C, g.B), dim3(256), 0, x, X, g, C);
    else WG_LAUNCH(cx, import_kernel, dim3((T + 255) / 256, C, B), dim3(256), 0, x, X, g, C);
    WG_LAUNCH(cx, import_kernel, dim3((T + 255) / 256, 2 * Cd, B), dim3(256), 0, y, Y, W.gi, 2 * Cd);
    SegSpec sg[WG_MAX_SEG];
    int ns = 0;
    const int half = (d->radix - 1) / 2;
    if (two_d) {                                               // waveflow.py:42: F.pad(x, [pad, pad, h_pad, 0]): causal along the height axis
        for (int kh = 0; kh < d->radix; ++kh)
            for (int kw = 0; kw < d->radix; ++kw)
                sg[ns++] = {ws + W.X, W.Cp, 0, C, (kw - half) * d->dilation, nullptr, 0, 0, (kh - (d->radix - 1)) * d->h_dilation, 0};
        sg[ns++] = {ws + W.Y, W.Yp, 0, 2 * Cd, 0, nullptr, 0, 0, 0, 1};      // + y, one plane row per item: the identity block of Acat
    } else {
        for (int kt = 0; kt < d->radix; ++kt) sg[ns++] = {ws + W.X, W.Cp, 0, C, (kt - half) * d->dilation, nullptr, 0, 0};
        sg[ns++] = {ws + W.Y, W.Yp, 0, 2 * Cd, 0, nullptr, 0, 0};           // + y: the identity block of Acat
    }
    run_convgemm(cx, g, ws + W.Acat, W.ldA, 2 * Cd, sg, ns, EPI_GATE, pref(ws + W.gate, W.Dp), pnull(), pnull(), pnull(), pnull(), 0, 0);     // waveglow.py:42-44
    SegSpec sgt[1] = {{ws + W.gate, W.Dp, 0, Cd, 0, nullptr, 0, 0}};
    run_convgemm(cx, g, ws + W.WoT, W.ldO, W.R, sgt, 1, EPI_RESSKIP, pref(ws + W.res, W.Cp), pref(ws + W.skip, W.Sp), pnull(), X, pnull(),
                 d->last_layer ? 0 : C, 0);                                                                                                     // :45-46
    if (two_d) {
        if (!d->last_layer) WG_LAUNCH(cx, export2d_kernel, dim3((T + 255) / 256, C, g.B), dim3(256), 0, pref(ws + W.res, W.Cp), res, g, C);
        WG_LAUNCH(cx, export2d_kernel, dim3((T + 255) / 256, Cs, g.B), dim3(256), 0, pref(ws + W.skip, W.Sp), skip, g, Cs);
    } else {
        if (!d->last_layer) WG_LAUNCH(cx, export_kernel, dim3((T + 255) / 256, C, B), dim3(256), 0, pref(ws + W.res, W.Cp), res, g, C, 1.0f);
        WG_LAUNCH(cx, export_kernel, dim3((T + 255) / 256, Cs, B), dim3(256), 0, pref(ws + W.skip, W.Sp), skip, g, Cs, 1.0f);
    }
    return cx.err;
}

// What autograd computes upstream for `res, skip = layer(x, y)` (waveglow.py:41-46 / waveflow.py:41-51; the classes are ordinary
// differentiable modules): the forward again with tanh / sigmoid kept, then the four products of a WN layer's backward as wn_backward
// runs them in the exact-fp32 mode (W_o's weight gradient, the gate backward, W's weight gradient, the data gradient), on matrices
// normalised from (g, v) by this call.
struct LayerBwdWs {
    LayerWs f;
    size_t tw, sf, dO, dxy, dXp, rs, WT, WoN, slab, slab_floats, total;
    int ldT, ldN;
};
static LayerBwdWs layer_bwd_ws_layout(const wg_layer_dims *d, int B, int T)
{
    LayerBwdWs w;
    w.f = layer_ws_layout(d, B, T);
    Bump bp;
    bp.off = w.f.total;
    const Geo &g = w.f.g;
    const size_t cols = (size_t)g.B * g.P;
    w.tw = bp.take(cols * w.f.Dp); w.sf = bp.take(cols * w.f.Dp);
    w.dO = bp.take(cols * w.f.R);
    w.dxy = bp.take(cols * 2 * d->dil_ch);
    w.dXp = bp.take(cols * w.f.Cp);
    w.rs = bp.take((size_t)B * 2 * d->dil_ch * w.f.gi.P);
    w.ldT = rup(d->res_ch, WG_TILE); w.ldN = rup(d->dil_ch, WG_TILE);
    w.WT = bp.take((size_t)w.f.taps * 2 * d->dil_ch * w.ldT);
    w.WoN = bp.take((size_t)w.f.R * w.ldN);
    const int MpW = rup(rup(2 * d->dil_ch, 32), WG_TILE), NpW = rup(w.f.taps * rup(d->res_ch, 32), WG_TILE);
    const int MpO = rup(rup(d->res_ch, 32) + rup(d->skip_ch, 32), WG_TILE), NpO = rup(rup(d->dil_ch, 32), WG_TILE);
    w.slab_floats = std::max(slab_floats(g, MpW, NpW), slab_floats(g, MpO, NpO));
    w.slab = bp.take(w.slab_floats);
    w.total = bp.off + 1024;
    return w;
}
size_t wg_layer_backward_workspace_bytes(const wg_layer_dims *d, int B, int T)
{
    if (layer_check(d) || B < 1 || T < 1) return 0;
    return layer_bwd_ws_layout(d, B, T).total * sizeof(float);
}
int wg_layer_backward(const wg_layer_dims *d, const void *const *params, const float *x, const float *y, const float *dres, const float *dskip,
                      int B, int T, float *dx, float *dy, void *const *grads, void *wsv, size_t ws_bytes, void *stream)
{
    int rc = layer_check(d);
    if (rc) return rc;
    if (!params || !params[1] || !params[3] || !x || !y || !dskip || !grads || !wsv || B < 1 || T < 1) return WG_EINVAL;
    const LayerBwdWs Wb = layer_bwd_ws_layout(d, B, T);
    const LayerWs &W = Wb.f;
    if (Wb.total * sizeof(float) > ws_bytes) return WG_EWORKSPACE;
    Ctx cx = {(hipStream_t)stream, 0, 0};
    float *ws = (float *)wsv;
    const Geo g = W.g;
    const int C = d->res_ch, Cd = d->dil_ch, Cs = d->skip_ch, last = d->last_layer ? 1 : 0;
    const bool two_d = d->rows > 0;
    const float *const *p = (const float *const *)params;
    float *const *gr = (float *const *)grads;
    if (hipMemsetAsync(ws, 0, Wb.total * sizeof(float), cx.st) != hipSuccess) return WG_ELAUNCH;
    // the four matrices: the forward's two, W per tap transposed, W_o as it is
    LayerPackArgs pa;
    pa.g = p[0]; pa.v = p[1]; pa.dst = ws + W.Acat; pa.rows = 2 * Cd; pa.fan = C * W.taps; pa.ld = W.ldA; pa.kind = 0; pa.C = C; pa.Cd = Cd; pa.radix = W.taps;
    WG_LAUNCH(cx, layer_pack_kernel, dim3(pa.rows), dim3(256), 0, pa);
    pa.dst = ws + Wb.WT; pa.ld = Wb.ldT; pa.kind = 2;
    WG_LAUNCH(cx, layer_pack_kernel, dim3(pa.rows), dim3(256), 0, pa);
    pa.g = p[2]; pa.v = p[3]; pa.dst = ws + W.WoT; pa.rows = W.R; pa.fan = Cd; pa.ld = W.ldO; pa.kind = 1;
    WG_LAUNCH(cx, layer_pack_kernel, dim3(pa.rows), dim3(256), 0, pa);
    pa.dst = ws + Wb.WoN; pa.ld = Wb.ldN; pa.kind = 3;
    WG_LAUNCH(cx, layer_pack_kernel, dim3(pa.rows), dim3(256), 0, pa);
    PRef X = pref(ws + W.X, W.Cp), Y = pref(ws + W.Y, W.Yp), dO = pref(ws + Wb.dO, W.R);
    const dim3 gx((T + 255) / 256, C, two_d ? g.B : B), gs((T + 255) / 256, Cs, two_d ? g.B : B);
    PRef dOs = dO;
    dOs.ch0 = last ? 0 : C;                                     // do = last ? dskip : cat(dres, dskip)
    if (two_d) {
        WG_LAUNCH(cx, import2d_kernel, gx, dim3(256), 0, x, X, g, C);
        if (!last && dres) WG_LAUNCH(cx, import2d_kernel, gx, dim3(256), 0, dres, dO, g, C);
        WG_LAUNCH(cx, import2d_kernel, gs, dim3(256), 0, dskip, dOs, g, Cs);
    } else {
        WG_LAUNCH(cx, import_kernel, gx, dim3(256), 0, x, X, g, C);
        if (!last && dres) WG_LAUNCH(cx, import_kernel, gx, dim3(256), 0, dres, dO, g, C);
        WG_LAUNCH(cx, import_kernel, gs, dim3(256), 0, dskip, dOs, g, Cs);
    }
    WG_LAUNCH(cx, import_kernel, dim3((T + 255) / 256, 2 * Cd, B), dim3(256), 0, y, Y, W.gi, 2 * Cd);
    // forward gate conv, tanh and sigmoid kept (waveglow.py:42-44)
    const int half = (d->radix - 1) / 2;
    int ts[WG_MAX_SEG], ro[WG_MAX_SEG];
    if (two_d) {
        for (int kh = 0; kh < d->radix; ++kh)
            for (int kw = 0; kw < d->radix; ++kw) { ts[kh * d->radix + kw] = (kw - half) * d->dilation; ro[kh * d->radix + kw] = (kh - (d->radix - 1)) * d->h_dilation; }
    } else {
        for (int kt = 0; kt < d->radix; ++kt) { ts[kt] = (kt - half) * d->dilation; ro[kt] = 0; }
    }
    SegSpec sg[WG_MAX_SEG];
    int ns = 0;
    for (int k = 0; k < W.taps; ++k) sg[ns++] = {ws + W.X, W.Cp, 0, C, ts[k], nullptr, 0, 0, ro[k], 0};
    sg[ns++] = {ws + W.Y, W.Yp, 0, 2 * Cd, 0, nullptr, 0, 0, 0, two_d ? 1 : 0};
    run_convgemm(cx, g, ws + W.Acat, W.ldA, 2 * Cd, sg, ns, EPI_GATE, pref(ws + W.gate, W.Dp), pref(ws + Wb.tw, W.Dp), pref(ws + Wb.sf, W.Dp), pnull(), pnull(), 0, 0);
    float *slab = ws + Wb.slab;
    // dW_o = sum do (x) gate
    {
        WSegSpec sa = {ws + Wb.dO, W.R, 0, W.R, 0, nullptr, W.R, 0};
        WSegSpec sb = {ws + W.gate, W.Dp, 0, Cd, 0, nullptr, W.Dp, 0};
        WgradOut wo = run_wgrad(cx, g, &sa, 1, &sb, 1, slab, Wb.slab_floats);
        run_finalize(cx, slab, wo, 0, W.R, Cd, 1, 0, 1, 0, p[2], p[2] ? p[3] : nullptr, p[2] ? gr[2] : nullptr, gr[3]);
    }
    // dgate = W_o^T do -> dxy (gate backward, waveglow.py:13-15)
    {
        SegSpec s = {ws + Wb.dO, W.R, 0, W.R, 0, nullptr, 0, 0};
        run_convgemm(cx, g, ws + Wb.WoN, Wb.ldN, Cd, &s, 1, EPI_DGATE, pref(ws + Wb.dxy, 2 * Cd), pnull(), pnull(), pref(ws + Wb.tw, W.Dp),
                     pref(ws + Wb.sf, W.Dp), Cd, 0);
    }
    // dW = sum dxy (x) x[tap]
    {
        WSegSpec sa = {ws + Wb.dxy, 2 * Cd, 0, 2 * Cd, 0, nullptr, 2 * Cd, 0};
        WSegSpec sb[WG_MAX_SEG];
        for (int k = 0; k < W.taps; ++k) sb[k] = {ws + W.X, W.Cp, 0, C, ts[k], nullptr, W.Cp, 0, ro[k], 0};
        WgradOut wo = run_wgrad(cx, g, &sa, 1, sb, W.taps, slab, Wb.slab_floats);
        run_finalize(cx, slab, wo, 0, 2 * Cd, C, W.taps, 0, 1, rup(C, 32), p[0], p[0] ? p[1] : nullptr, p[0] ? gr[0] : nullptr, gr[1]);
    }
    // dx = sum_k W[:, :, k]^T dxy[t - shift_k] (+ dres: res = o[:C] + x)
    if (dx) {
        SegSpec s[WG_MAX_SEG];
        for (int k = 0; k < W.taps; ++k) s[k] = {ws + Wb.dxy, 2 * Cd, 0, 2 * Cd, -ts[k], nullptr, 0, 0, -ro[k], 0};
        run_convgemm(cx, g, ws + Wb.WT, Wb.ldT, C, s, W.taps, EPI_STORE, pref(ws + Wb.dXp, W.Cp), pnull(), pnull(),
                     (!last && dres) ? pref(ws + Wb.dO, W.R) : pnull(), pnull(), 0, 0);
        if (two_d) WG_LAUNCH(cx, export2d_kernel, gx, dim3(256), 0, pref(ws + Wb.dXp, W.Cp), dx, g, C);
        else WG_LAUNCH(cx, export_kernel, gx, dim3(256), 0, pref(ws + Wb.dXp, W.Cp), dx, g, C, 1.0f);
    }
    // dy = dxy (2-D: summed over the height axis the conditioning was broadcast over)
    if (dy) {
        if (two_d) {
            WG_LAUNCH(cx, wf_rowsum_kernel, dim3((T + 255) / 256, 2 * Cd, B), dim3(256), 0, pref(ws + Wb.dxy, 2 * Cd), g, pref(ws + Wb.rs, 2 * Cd), W.gi, 2 * Cd);
            WG_LAUNCH(cx, export_kernel, dim3((T + 255) / 256, 2 * Cd, B), dim3(256), 0, pref(ws + Wb.rs, 2 * Cd), dy, W.gi, 2 * Cd, 1.0f);
        } else
            WG_LAUNCH(cx, export_kernel, dim3((T + 255) / 256, 2 * Cd, B), dim3(256), 0, pref(ws + Wb.dxy, 2 * Cd), dy, g, 2 * Cd, 1.0f);
    }
    return cx.err;
}

int wg_affine_apply(const float *in, const float *log_s, const float *t, size_t n, int reverse, float *out, void *stream)
{
    if (!in || !log_s || !t || !out || n < 1) return WG_EINVAL;
    Ctx cx = {(hipStream_t)stream, 0, 0};
    AffinePlainArgs a;
    memset(&a, 0, sizeof(a));
    a.in = in; a.log_s = log_s; a.t = t; a.out = out; a.n = n; a.reverse = reverse;
    WG_LAUNCH(cx, affine_plain_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, a);
    return cx.err;
}
int wg_affine_backward(const float *out_half, const float *log_s, const float *t, const float *dout, const float *dlog_s, size_t n, int reverse,
                       float *in_rebuilt, float *g_log_s, float *g_t, float *din, void *stream)
{
    if (!out_half || !log_s || !t || !dout || !in_rebuilt || !g_log_s || !g_t || !din || n < 1) return WG_EINVAL;
    Ctx cx = {(hipStream_t)stream, 0, 0};
    AffinePlainArgs a;
    memset(&a, 0, sizeof(a));
    a.in = out_half; a.log_s = log_s; a.t = t; a.dout = dout; a.dls = dlog_s; a.out = in_rebuilt; a.g_ls = g_log_s; a.g_t = g_t; a.din = din;
    a.n = n; a.reverse = reverse; a.backward = 1;
    WG_LAUNCH(cx, affine_plain_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, a);
    return cx.err;
}

int wg_coupling_backward(const wg_wn_dims *dd, const void *const *params, const void *packed, const float *z, const float *y,
                         const float *dz, const float *dlog_s, int B, int T, int reverse, float *x, float *dx, float *dy,
                         void *const *grads, void *wsv, size_t ws_bytes, void *stream)
{
    if (!dd || !params || !packed || !z || !y || !dz || !dlog_s || !x || !dx || !grads || !wsv || B < 1 || T < 1) return WG_EINVAL;
    const WnD d = wnd_from(dd);
    int rc = wn_check(d);
    if (rc) return rc;
    const CplWs W = cpl_ws_layout(d, B, T, 1);
    if (W.total * sizeof(float) > ws_bytes) return WG_EWORKSPACE;
    Ctx cx = {(hipStream_t)stream, 0, dd->precision};
    float *ws = (float *)wsv;
    const Geo g = W.g;
    PRef X = pref(ws + W.X, W.Xp), dX = pref(ws + W.dX, W.Xp);
    layer_sync_clear(cx, ws, W.wn.lsync);
    WG_LAUNCH(cx, import_kernel, dim3((T + 255) / 256, 2 * d.ic, B), dim3(256), 0, z, X, g, 2 * d.ic);
    WG_LAUNCH(cx, import_kernel, dim3((T + 255) / 256, 2 * d.ic, B), dim3(256), 0, dz, dX, g, 2 * d.ic);
    WG_LAUNCH(cx, import_kernel, dim3((T + 255) / 256, d.aux, B), dim3(256), 0, y, pref(ws + W.Y, W.auxp), g, d.aux);
    if (cx.prec == 2) run_to_splane(cx, g, pref(ws + W.Y, W.auxp), d.aux, ws + W.YS, W.auxp);
    if (dy && cx.err == 0 && hipMemsetAsync(ws + W.dY, 0, (size_t)B * W.auxp * g.P * sizeof(float), cx.st) != hipSuccess) cx.err = WG_ELAUNCH;
    WnRun r;
    r.d = d; r.L = wn_pack_layout(d); r.pk = (const float *)packed + rupz(WG_ONES, 64); r.g = g; r.ws = ws; r.w = W.wn;
    r.X = X; r.Y = ws + W.Y; r.YS = ws + W.YS; r.save = 1;
    wn_forward(cx, r);
    run_end_affine(cx, r, reverse ? AFF_BWD_REV : AFF_BWD, dX, nullptr, dlog_s, nullptr, nullptr);
    wn_backward(cx, r, (const float *const *)params, (float *const *)grads, dX, dy ? ws + W.dY : nullptr);
    WG_LAUNCH(cx, export_kernel, dim3((T + 255) / 256, 2 * d.ic, B), dim3(256), 0, X, x, g, 2 * d.ic, 1.0f);
    WG_LAUNCH(cx, export_kernel, dim3((T + 255) / 256, 2 * d.ic, B), dim3(256), 0, dX, dx, g, 2 * d.ic, 1.0f);
    if (dy) WG_LAUNCH(cx, export_kernel, dim3((T + 255) / 256, d.aux, B), dim3(256), 0, pref(ws + W.dY, W.auxp), dy, g, d.aux, 1.0f);
    return cx.err;
}

}  // extern "C"
