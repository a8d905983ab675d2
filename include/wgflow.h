/*
 * wgflow.h -- C ABI of the MI355X-native WaveGlow flow engine (libwgflow.so).
 *
 * This is the drop-in boundary for the hot path of yoyololicon/constant-memory-waveglow:
 * every entry point replaces a specific piece of the reference's Python/ATen path (cited
 * below as path:line under the upstream repo).  Plain C types only: device pointers, sizes,
 * a stream handle.  No allocation, no global state: every call works inside a caller-provided
 * workspace whose size comes from the matching *_bytes() query, is re-entrant per stream, and
 * only ENQUEUES work on `stream` (hipStream_t passed as void*; NULL = the default stream).
 * All tensors are float32, contiguous, device resident unless a parameter says "host".
 *
 * Return value: 0 on success, a negative WG_E* code otherwise (wg_strerror() names it).
 *
 * Parameter table ("params"): an array of device pointers in the order of the reference
 * model's named_parameters():
 *     [0] upsampler.bias [n_mels]   [1] upsampler.weight_g [n_mels,1,1]   [2] upsampler.weight_v [n_mels,1,K]
 *     [3 .. 3+flows)                invconv1x1.{k}.weight [c_k,c_k,1]
 *     then per flow k, 4+4*depth+1 entries:
 *         V.weight_g, V.weight_v, start.weight_g, start.weight_v,
 *         layers.{i}.W.weight_g, .W.weight_v, .W_o.weight_g, .W_o.weight_v   (i = 0..depth-1),
 *         end.weight
 * A NULL weight_g entry means the convolution carries a plain weight in its weight_v slot
 * (what the module looks like after utils.remove_weight_norms, utils.py:9-11).
 * The gradient table ("grads") has the same order and shapes; NULL entries are skipped.
 */
#ifndef WGFLOW_H
#define WGFLOW_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define WG_OK 0
#define WG_EINVAL (-1)      /* bad argument / unsupported shape */
#define WG_ESHAPE (-2)      /* audio length not a multiple of n_group, or mel too short (waveglow.py:156 assert) */
#define WG_EUNSUPPORTED (-3)/* valid for the reference but outside this engine's kernels */
#define WG_ELAUNCH (-4)     /* a HIP launch failed */
#define WG_EWORKSPACE (-5)  /* workspace too small */

/* Arithmetic of the matrix contractions; every tensor in memory is fp32 in both modes.
 *   WG_PREC_F32    v_mfma_f32_32x32x2_f32: bit-exact fp32 fma chains.
 *   WG_PREC_BF16X3 each fp32 operand is split into bf16 hi + lo and a*b is evaluated as a_lo*b_hi + a_hi*b_lo + a_hi*b_hi
 *                  on the bf16 matrix pipe with fp32 accumulation (drops a_lo*b_lo, 2^-16 of a product).
 *   WG_PREC_BF16X3_PLANES  same arithmetic; producers additionally keep every MFMA operand tensor pre-split in bf16 hi/lo
 *                  "S-planes" (wg_gemm16s.h), so the contraction kernels do no conversion work. */
#define WG_PREC_F32 0
#define WG_PREC_BF16X3 1
#define WG_PREC_BF16X3_PLANES 2

/* Constructor arguments of model.WaveGlow (model/waveglow.py:109-118) after the arithmetic of
 * :125-129 (upsampler geometry).  WN arguments are the **kwargs forwarded to WN (waveglow.py:50-59). */
typedef struct wg_config {
    int32_t n_flows, n_group, n_early_every, n_early_size, n_mels;
    int32_t up_stride, up_kernel, up_pad;            /* ConvTranspose1d(n_mels,n_mels,K,stride,pad,groups=n_mels) */
    int32_t res_ch, dil_ch, skip_ch, depth, radix;   /* WN: residual/dilation/skip channels, layers, kernel size (odd, <= 9) */
    int32_t precision;                               /* WG_PREC_* : arithmetic of the MFMA contractions (not upstream) */
    int32_t reverse_mode;                            /* WaveGlow(reverse_mode=...) (waveglow.py:116, base.py:20-28): wg_forward is what
                                                        model.forward computes in that architecture, wg_inverse what model.reverse does */
    int32_t keep_activations;                        /* WaveGlow(memory_efficient=False) (waveglow.py:118, efficient_modules.py:33-35,71-75):
                                                        wg_forward / wg_train_step run in the MODE-1 workspace and leave every flow's WN layers
                                                        there; wg_backward with the same workspace reads them instead of recomputing each WN.
                                                        The caller guarantees nothing else used that workspace in between.  0 = the
                                                        constant-memory scheme (only one flow's layers exist at a time). */
    int32_t bias;                                    /* WN(bias=True) (waveglow.py:58): every conv of every WN carries a bias.  The WN's
                                                        part of the parameter table then continues behind `end.weight` with V.bias,
                                                        start.bias, depth x (layers.i.W.bias, layers.i.W_o.bias), end.bias */
} wg_config;

/* Dimensions of one WN as AffineCouplingBlock builds it (efficient_modules.py:58-65, waveglow.py:50-59). */
typedef struct wg_wn_dims {
    int32_t in_ch, aux_ch, res_ch, dil_ch, skip_ch, depth, radix;
    int32_t precision;                               /* WG_PREC_* */
    int32_t bias;                                    /* WN(bias=True): see wg_config.bias (the block-level table grows the same way) */
} wg_wn_dims;

const char *wg_strerror(int code);
/* ABI revision of this header (2: wg_config gained keep_activations; 3: wg_nll_loss / wg_train_step produce the logged
 * training scalars and take their scratch from the caller, wg_melspec returns the power spectrogram on request, wg_wf_config gained
 * use_conv1x1, wg_wf_upsample; 4: wg_timer_create(-1, ..) times every kernel class, wg_timer_read_info, wg_stat_wgrad16t_launches,
 * wg_wf_* accept every WG_PREC_*; 5: wg_config and wg_wn_dims gained bias; 6: wg_stat_layer_launches, the workspaces carry the one-launch
 * layer's hand-off counters, wg_layer_apply / wg_layer_workspace_bytes, wg_wf_wn_apply; 7: wg_wf_config gained bias; 8: wg_timer_read_name,
 * wg_box_probe / wg_box_probe_bytes, wg_stat_layerg_launches, wg_stat_gate_split_launches,
 * wg_wf_wn_backward, wg_layer_backward / wg_layer_backward_workspace_bytes, wg_affine_apply / wg_affine_backward; 9: wg_reload_env,
 * wg_stat_gate_part_launches, wg_wsr_cond_pre).  A binding built against another revision must not pass its
 * structs: the Python loader compares this with its own ABI_VERSION and refuses the library otherwise. */
#define WG_ABI_VERSION 9
int wg_abi_version(void);
/* Developer switches (WG_G192, WG_G192_SPLITK, WG_LOWRANK, WG_TW_FROM_GATE, WG_START_FOLD, WG_LAYER_G, WG_LAYER_MIN_CHUNKS, WG_LAYER_FUSION,
 * WG_LAYER_FUSION_BIG, WG_INV_SEAM: A/B switches between
 * kernels that compute the same thing, csrc/wgflow.hip EnvSw) are read from the environment ONCE per process, at the first call that needs
 * one; a caller that changes one afterwards -- a test, an A/B run -- calls this to have them read again.  Not to be called while
 * another thread is inside the library.  No counterpart upstream. */
void wg_reload_env(void);

/* Diagnostics (no counterpart upstream; the reference times with wall-clock time(), inference.py:39-53): while a
 * timer is attached, every launch of ONE kernel class is bracketed with HIP events on its launch stream, so a
 * benchmark can read that kernel's per-launch durations over the very steps it times.  Process-wide: attach from
 * one thread, detach (attach NULL) before destroying. */
#define WG_K_CONV_STORE 0    /* convgemm, plain / accumulate epilogue: start, dgrad (W^T), V^T, W_end^T, W_start^T */
#define WG_K_CONV_GATE 1     /* convgemm, dilated conv + conditioning + tanh*sigmoid gate   (the dominant kernel) */
#define WG_K_CONV_RESSKIP 2  /* convgemm, W_o + residual/skip epilogue */
#define WG_K_CONV_DGATE 3    /* convgemm, W_o^T + gate backward */
#define WG_K_WGRAD 4         /* weight-gradient kernel */
#define WG_K_LAYER 5         /* convlayer16q_kernel: a layer's gate conv + residual product in one persistent launch (reported as M = gate rows,
                              * K = gate K + residual K * residual rows / gate rows, so that 2 M K columns = the launch's FLOPs) */
#define WG_K_THIN 6          /* the byte-bound passes of the rank-2ic skip path: WN's end conv from the gate planes or from the gate convs' partial rows
                              * (M = 2 ic, K = the channels or rows it adds), and P_l = G gate_l^T over the gate planes (M = 2 ic, K = depth x dil_ch) */
void *wg_timer_create(int kernel_id, int capacity);   /* kernel_id < 0: every class above */
void  wg_timer_attach(void *timer);
int   wg_timer_count(void *timer);
int   wg_timer_read(void *timer, float *ms, int n);   /* after a stream sync; returns the number written */
/* five values per recorded launch: class (WG_K_*), M, K, columns of the product (2 M K columns = its algorithmic FLOPs; a paired
 * weight-gradient launch reports M = the summed sizes of its gradients, K = 1), algorithmic HBM bytes (every operand plane once, the
 * weights, the outputs and auxiliary planes of the epilogue); launches that are not conv / weight-gradient products report zeros */
int   wg_timer_read_info(void *timer, long long *info, int n);
/* the kernel expression of recorded launch `index` as written at its launch site in csrc/wgflow.hip, e.g.
 * "(convgemm16q_kernel<EPI_GATE_SO, 2, 2>)" (template arguments by name, defaulted ones absent): which instantiation RAN.  Returns the
 * name's length (it is truncated to n - 1 characters), or a negative error */
int   wg_timer_read_name(void *timer, int index, char *buf, int n);
void  wg_timer_destroy(void *timer);
/* Box calibration (no counterpart upstream): a fixed matrix-pipe + LDS loop without global traffic on random data (csrc/wg_probe.h), run
 * back to back for about `ms` milliseconds on `stream`.  out[0] = issued TFLOP/s, out[1] = the clock held inside the kernel (GHz),
 * out[2] = ms per launch.  The boxes of a pool differ by a few per cent on exactly this loop; a benchmark line that carries these
 * numbers tells a slower box from a slower build.  scratch: wg_box_probe_bytes() bytes of device memory.  Synchronises the stream. */
size_t wg_box_probe_bytes(void);
int    wg_box_probe(void *scratch, int ms, double *out, void *stream);
/* diagnostics: how many times this process has launched wgrad16t_kernel (the one-workgroup-per-CU weight-gradient kernel; shapes
 * without a plan take wgrad16s_pair_kernel) -- lets a test assert which kernel a shape ran on */
long long wg_stat_wgrad16t_launches(void);
/* diagnostics: launches of convlayer16h_kernel (ONE launch per WN layer: gate conv -> gate -> W_o -> residual / skip, wg_layer16h.h; small grids
 * only -- single-utterance synthesis, WaveFlow's row steps); larger shapes take two launches per layer */
long long wg_stat_layer_launches(void);
/* diagnostics: launches of convlayer16g_kernel (csrc/wg_gemm16g.h: a layer's gate conv and residual product in one launch on 256 x 192 tiles of
 * flattened columns, a workgroup owning whole column tiles -- the training shapes; default on, env WG_LAYER_G=0 switches it off) */
long long wg_stat_layerg_launches(void);
/* diagnostics: gate convs that ran cut along K (convgemm16g_kernel<8> writes S partial 256 x 192 tiles per output tile into the workspace,
 * gate_finish16g_kernel sums them in a fixed order and applies the gate): shapes whose column tiles fill a fraction 1/S of the CUs and
 * whose K is long -- WSRGlow's 512 x 4432 gate conv at 12 x 512 columns; env WG_G192_SPLITK=0 switches it off */
long long wg_stat_gate_split_launches(void);
/* diagnostics: gate convs launched with the partial rows of WN's `out` (csrc/wg_gemm16g.h, wgg_gate_nb: the rank-2ic form of the skip
 * path, csrc/wgflow.hip lowrank_on / gate_parts_on; env WG_LOWRANK=0 switches the form off) -- lets a test assert that a shape took it */
long long wg_stat_gate_part_launches(void);

/* ---- sizes -------------------------------------------------------------------------------- */
int    wg_param_count(const wg_config *cfg);                 /* entries of the parameter table */
size_t wg_packed_bytes(const wg_config *cfg);                /* materialised-weight buffer of the model */
/* mode 0: forward / inverse (2 activation planes per WN);  mode 1: backward (all layers kept for ONE flow) */
size_t wg_workspace_bytes(const wg_config *cfg, int B, int N, int mode);
int    wg_wn_param_count(const wg_wn_dims *d);
size_t wg_wn_packed_bytes(const wg_wn_dims *d);
size_t wg_coupling_workspace_bytes(const wg_wn_dims *d, int B, int T, int mode);
size_t wg_invconv_workspace_bytes(int c, int B, int T);

/* Workspaces keep zero halos between calls; zero a freshly allocated one ONCE with this. */
int wg_workspace_init(void *ws, size_t bytes, void *stream);

/* ---- weights ------------------------------------------------------------------------------ */
/* Replaces the weight_norm forward pre-hooks (utils.py:14-16; w = g*v/||v||, recomputed on every module
 * call upstream) and the per-call logdet/inverse of the 1x1 weights (efficient_modules.py:221,235):
 * materialises every effective weight ONCE per step in the layouts the kernels read. */
int wg_pack_weights(const wg_config *cfg, const void *const *params, void *packed, void *stream);
int wg_wn_pack_weights(const wg_wn_dims *d, const void *const *params, void *packed, void *stream);

/* ---- model level  (model/waveglow.py:150-208, model/base.py:20-55) ------------------------- */
/* WaveGlow.forward_computation (waveglow.py:150-179): audio[B,N], h[B,n_mels,F] -> z[B,N], logdet[B]. */
int wg_forward(const wg_config *cfg, const void *packed, const float *audio, const float *h,
               int B, int N, int F, float *z, float *logdet, void *ws, size_t ws_bytes, void *stream);

/* WaveGlow.reverse_computation (waveglow.py:181-208): z[B,N], h -> x[B,N], logdet[B].  FlowBase.infer
 * (base.py:42-55) is this call on a latent the caller draws. */
int wg_inverse(const wg_config *cfg, const void *packed, const float *z, const float *h,
               int B, int N, int F, float *x, float *logdet, void *ws, size_t ws_bytes, void *stream);

/* Backward of wg_forward with the reference's constant-memory protocol (efficient_modules.py:118-154,
 * 230-244): starts from the flow OUTPUT z, walks flows last to first, rebuilds each block's input from
 * its output, recomputes that flow's WN activations into the workspace and produces the gradient of every
 * parameter.  dz[B,N], dlogdet[B] are the incoming gradients.  grads: table as params (written, not
 * accumulated).  dh[B,n_mels,F] and dx[B,N] (gradient wrt the audio) may be NULL.  x_rebuilt (nullable)
 * receives the re-materialised input audio [B,N].
 * flow_events (nullable): n_flows+1 caller-created hipEvent_t; [k] is recorded on `stream` as soon as every parameter
 * gradient of flow k (its WN and its 1x1 weight) is final, [n_flows] after the upsampler's -- lets a data-parallel caller
 * all-reduce one flow's bucket over RCCL while the remaining flows are still in backward (what DDP's bucket hooks do
 * for the reference, train.py:77). */
int wg_backward(const wg_config *cfg, const void *const *params, const void *packed,
                const float *z, const float *h, const float *dz, const float *dlogdet,
                int B, int N, int F, void *const *grads, float *dh, float *dx, float *x_rebuilt,
                void *ws, size_t ws_bytes, void *stream, void *const *flow_events);

/* WaveGlowLoss.forward (model/loss.py:10-15): loss = mean_b(0.5*sum z^2/sigma^2 - logdet_b) [/N].
 * loss is a device scalar.  metrics (nullable, 4 device floats) receives the scalars the reference's training step logs
 * next to the loss (model/lightning.py:58-64): [0] logdet.sum() / z.numel(), [1] z.mean(), [2] z.std() (unbiased),
 * [3] the loss -- per process; a data-parallel caller mean-reduces the vector (Lightning's sync_dist=True).
 * scratch: wg_nll_scratch_floats(B) floats.  The backward writes dz[B,N], dlogdet[B] for an upstream gradient dloss
 * (device scalar, NULL = 1). */
size_t wg_nll_scratch_floats(int B);
int wg_nll_loss(const float *z, const float *logdet, int B, int N, float sigma, int elementwise_mean,
                float *loss, float *metrics, float *scratch, void *stream);
int wg_nll_loss_backward(const float *z, int B, int N, float sigma, int elementwise_mean,
                         const float *dloss, float *dz, float *dlogdet, void *stream);

/* ---- block level  (model/efficient_modules.py) ---------------------------------------------- */
/* InvertibleConv1x1.forward_computation / reverse_computation (efficient_modules.py:31-54): W[c,c] device,
 * x[B,c,T] -> z[B,c,T]; logdet = device scalar (+-T*logdet W, NaN when det W < 0 as torch.logdet). */
int wg_invconv_apply(const float *W, int c, const float *x, int B, int T, int reverse,
                     float *z, float *logdet, void *ws, size_t ws_bytes, void *stream);
/* Conv1x1Func.backward / InvConv1x1Func.backward (efficient_modules.py:230-244, 262-279): from the block
 * OUTPUT z and its gradient dz plus dlogdet (device scalar): rebuilt input x, dx, dW[c,c]. */
int wg_invconv_backward(const float *W, int c, const float *z, const float *dz, const float *dlogdet,
                        int B, int T, int reverse, float *x, float *dx, float *dW,
                        void *ws, size_t ws_bytes, void *stream);

/* AffineCouplingBlock.forward_computation / reverse_computation (efficient_modules.py:70-96) with F = WN:
 * x[B,2*in_ch,T], y[B,aux,T] -> z[B,2*in_ch,T], log_s[B,in_ch,T] (negated when reverse). */
int wg_coupling_apply(const wg_wn_dims *d, const void *packed, const float *x, const float *y,
                      int B, int T, int reverse, float *z, float *log_s,
                      void *ws, size_t ws_bytes, void *stream);
/* WN.forward (waveglow.py:98-105): x[B,in_ch,T], y[B,aux,T] -> (log_s, t), each [B,in_ch,T]. */
int wg_wn_apply(const wg_wn_dims *d, const void *packed, const float *x, const float *y, int B, int T,
                float *log_s, float *t, void *ws, size_t ws_bytes, void *stream);
/* NonCausalLayer.forward / NonCausalLayer2D.forward on its own (waveglow.py:18-46, waveflow.py:14-51): xy = W(x) + y; gate = tanh(xy[:Cd]) * sigmoid(xy[Cd:]); o = W_o(gate);
 * returns (o[:C] + x, o[C:]) -- or (none, o) for the last layer.  Any dilation, odd radix <= 9, no bias; exact fp32 MFMA arithmetic.
 * params = {W.weight_g (NULL: plain weight), W.weight_v [2 Cd, C, radix], W_o.weight_g (NULL: plain), W_o.weight_v [C + Cs or Cs, Cd, 1]};
 * x[B,C,T], y[B,2 Cd,T] (the layer's slice of the conditioning projection V(y)) -> res[B,C,T] (NULL when last_layer), skip[B,Cs,T]. */
typedef struct wg_layer_dims {
    int32_t res_ch, dil_ch, skip_ch, radix, dilation, last_layer;
    int32_t h_dilation, rows;   /* NonCausalLayer2D (waveflow.py:14-51): h_dilation > 0 = a radix x radix conv, causal along the height axis
                                 * (dilation h_dilation there); x[B,C,rows,T], y[B,2 Cd,1,T] -> res[B,C,rows,T], skip[B,Cs,rows,T].  0, 0: the 1-D layer */
} wg_layer_dims;
size_t wg_layer_workspace_bytes(const wg_layer_dims *d, int B, int T);
int wg_layer_apply(const wg_layer_dims *d, const void *const *params, const float *x, const float *y, int B, int T,
                   float *res, float *skip, void *ws, size_t ws_bytes, void *stream);
/* What autograd computes upstream for that call (NonCausalLayer / NonCausalLayer2D are ordinary differentiable modules): from d res
 * (NULL: none -- the last layer, or an output nobody used) and d skip, the gradient of x (nullable), of y (nullable; [B,2 Cd,T] -- for the
 * 2-D layer summed over the height axis y was broadcast over) and of the four parameters (`grads`: the layout of `params`; entries of a
 * plain weight's g and entries nobody needs are NULL).  Workspace: wg_layer_backward_workspace_bytes. */
size_t wg_layer_backward_workspace_bytes(const wg_layer_dims *d, int B, int T);
int wg_layer_backward(const wg_layer_dims *d, const void *const *params, const float *x, const float *y, const float *dres, const float *dskip,
                      int B, int T, float *dx, float *dy, void *const *grads, void *ws, size_t ws_bytes, void *stream);
/* AffineCouplingFunc.backward / InvAffineCouplingFunc.backward (efficient_modules.py:118-154, 175-212):
 * from the block OUTPUT z, y, dz, dlog_s: rebuilt input x, dx, dy (nullable), parameter grads. */
int wg_coupling_backward(const wg_wn_dims *d, const void *const *params, const void *packed,
                         const float *z, const float *y, const float *dz, const float *dlog_s,
                         int B, int T, int reverse, float *x, float *dx, float *dy, void *const *grads,
                         void *ws, size_t ws_bytes, void *stream);

/* mel upsampler alone (waveglow.py:126-130,210-212, cropped to T as :157): h[B,n_mels,F] -> y[B,n_mels,T] */
int wg_upsample(const wg_config *cfg, const void *packed, const float *h, int B, int F, int T, float *y, void *stream);

/* ---- WSRGlow conditioning front-end -------------------------------------------------------------
 * Replaces WSRGlow._get_cond (model/wsrglow.py:37-50): c[B,L] (low-rate audio, L a multiple of 8) ->
 * cond[B, WG_WSR_COND_CHANNELS, L/8] = cat(mu-law(256) embedding [8*400], |STFT16| [9], phase embedding [9*50]).
 * mu_table is `mu_enc.1.weight` [256,400] (wsrglow.py:27-30), ang_table is `angle_embed.embed.weight` [120,50]
 * (wsrglow.py:8-18,31).  c is read clipped to [-1,1] and NOT modified (the reference clips it in place, wsrglow.py:38:
 * the Python mirror does that).  The result feeds wg_forward / wg_inverse as `h` of a WaveGlow with n_mels = 3659. */
#define WG_WSR_COND_CHANNELS 3659
int wg_wsr_cond(const float *c, int B, int L, const float *mu_table, const float *ang_table, float *cond, void *stream);
/* Diagnostics: the two quantisers' float32 values BEFORE truncation, from the functions wg_wsr_cond truncates: mu_pre[B][L] (the mu-law
 * value (x_mu + 1) / 2 * 255 + 0.5 of every clipped low-rate sample, wsrglow.py:39) and ang_pre[B][9][L/8] ((angle / pi + 1) * 0.5 * 119 of
 * every STFT bin, wsrglow.py:15-18).  A parity test uses them to forgive ONLY decisions that sit within float noise of a bin edge. */
int wg_wsr_cond_pre(const float *c, int B, int L, float *mu_pre, float *ang_pre, void *stream);
/* Its backward: the two nn.Embedding weight gradients from dcond[B,3659,L/8] (c has no gradient path: wsrglow.py:39,48 go
 * through integer indices, :47 has no parameters).  Outputs are overwritten. */
int wg_wsr_cond_backward(const float *c, int B, int L, const float *dcond, float *dmu_table, float *dang_table, void *stream);

/* The training step of model/lightning.py:52-65 in ONE call: forward, WaveGlowLoss(sigma) (loss.py:10-15), backward to every
 * parameter gradient (+ dh when not NULL), and the four logged scalars (metrics, nullable: see wg_nll_loss).  Equivalent to
 * wg_forward + wg_nll_loss + wg_nll_loss_backward + wg_backward; the forward runs in the backward's workspace
 * (wg_workspace_bytes(.., mode 1)) and keeps the layers of the flow it processes last, which the backward then does not
 * recompute.  scratch: wg_train_scratch_floats(B, N) floats.  flow_events as in wg_backward. */
size_t wg_train_scratch_floats(int B, int N);
int wg_train_step(const wg_config *cfg, const void *const *params, const void *packed, const float *audio, const float *h,
                  int B, int N, int F, float sigma, int elementwise_mean, float *z, float *logdet, float *loss, float *metrics,
                  void *const *grads, float *dh, float *scratch, void *ws, size_t ws_bytes, void *stream, void *const *flow_events);

/* ---- WaveFlow (model/waveflow.py) -------------------------------------------------------------------
 * WaveFlow(flows, n_group, n_mels, use_conv1x1=False, ..., dilation/residual/skip_channels, bias=False): audio [B,N] viewed as
 * [B, n_group (height), N/n_group (time)], 8-layer WN2D with 3x3 dilated convs causal along the height axis, autoregressive
 * affine coupling along the height axis, flip between flows.  Parameter table = named_parameters() order (3 + 37 per flow):
 *   upsampler.1.{bias, weight_g, weight_v}; WNs.k.{V.g, V.v, start.g, start.v, layers.i.{W.g, W.v, W_o.g, W_o.v} x 8, end.weight};
 *   with use_conv1x1 followed by invconv1x1.k.weight for every flow (3 + 38 per flow).
 * Every WG_PREC_* arithmetic mode is built for this model (ABI 4).  The hop length is 256 (waveflow.py:160). */
typedef struct wg_wf_config {
    int32_t flows, n_group, n_mels;
    int32_t res_ch, dil_ch, skip_ch;
    int32_t precision;
    int32_t use_conv1x1;    /* WaveFlow(use_conv1x1=True) (waveflow.py:176-181,203-206,224-229): an InvertibleConv1x1(n_group) over the height
                               axis replaces the flip between flows; the parameter table then ends with invconv1x1.{k}.weight [H,H,1] per flow */
    int32_t bias;           /* WN2D(bias=True) (waveflow.py:77,100-122): every conv of every WN2D has a bias.  A flow's part of the table then
                               continues behind end.weight with V.bias, start.bias, 8 x (layers.i.W.bias, layers.i.W_o.bias), end.bias
                               (3 + 56 per flow; the 1x1 weights, when present, still come last) */
} wg_wf_config;
int wg_wf_param_count(const wg_wf_config *cfg);
size_t wg_wf_packed_bytes(const wg_wf_config *cfg);
size_t wg_wf_workspace_bytes(const wg_wf_config *cfg, int B, int N, int mode);      /* 0: forward; 1: backward / inverse */
size_t wg_wf_tape_bytes(const wg_wf_config *cfg, int B, int N);
int wg_wf_pack_weights(const wg_wf_config *cfg, const void *const *params, void *packed, void *stream);
/* WaveFlow._upsample_h (waveflow.py:163-169,255-257): mel[B,n_mels,F] -> y[B,n_mels,T] for T <= F*s - 2*(s/2) + 2*s + 1, s = 256/n_group. */
int wg_wf_upsample(const wg_wf_config *cfg, const void *const *params, const void *packed, const float *mel, int B, int F, int T,
                   float *y, void *stream);
/* WaveFlow.forward_computation (waveflow.py:182-208): z[B,N], logdet[B].  `tape` (nullable; zero-initialised once) receives every
 * flow's input for wg_wf_backward. */
int wg_wf_forward(const wg_wf_config *cfg, const void *const *params, const void *packed, const float *audio, const float *mel,
                  int B, int N, int F, float *z, float *logdet, void *tape, void *ws, size_t ws_bytes, void *stream);
/* WaveFlow.reverse_computation (waveflow.py:210-253): the row-by-row autoregressive inverse. */
int wg_wf_inverse(const wg_wf_config *cfg, const void *const *params, const void *packed, const float *z, const float *mel,
                  int B, int N, int F, float *x, float *logdet, void *ws, size_t ws_bytes, void *stream);
/* WN2D.forward on its own (waveflow.py:128-135): x[B,1,rows,W] with rows <= n_group, y[B,n_mels,W] (already at the flow's time
 * resolution) -> log_s, t, each [B,1,rows,W].  `params` / `packed`: the table and wg_wf_pack_weights of a configuration with flows = 1 whose
 * WN2D entries are this module's parameters (the three upsampler entries are packed but not read here). */
int wg_wf_wn_apply(const wg_wf_config *cfg, const void *const *params, const void *packed, const float *x, const float *y, int B, int rows, int W,
                   float *log_s, float *t, void *ws, size_t ws_bytes, void *stream);
/* What autograd computes upstream for that call (WN2D is an ordinary differentiable module, waveflow.py:94-135): from the gradients of
 * log_s and t, each [B,1,rows,W], the gradient of x (nullable, [B,1,rows,W]), of y (nullable, [B,n_mels,W]) and of every WN2D parameter
 * (`grads`: the table layout of `params`, null entries skipped; the three upsampler entries are not touched).  Workspace:
 * wg_wf_workspace_bytes(cfg, B, W * n_group, 1). */
int wg_wf_wn_backward(const wg_wf_config *cfg, const void *const *params, const void *packed, const float *x, const float *y,
                      const float *dlog_s, const float *dt, int B, int rows, int W, float *dx, float *dy, void *const *grads,
                      void *ws, size_t ws_bytes, void *stream);
/* What autograd computes upstream for z, logdet = model(x, mel) (the reference trains this model with memory_efficient=False):
 * every parameter gradient (table order), d mel (nullable), d audio (nullable), from the tape wg_wf_forward wrote. */
int wg_wf_backward(const wg_wf_config *cfg, const void *const *params, const void *packed, const void *tape, const float *mel,
                   const float *dz, const float *dlogdet, int B, int N, int F, void *const *grads, float *dmel, float *dx,
                   void *ws, size_t ws_bytes, void *stream);

/* ---- log-mel conditioner ---------------------------------------------------------------------------
 * Replaces MelSpec.forward (model/condition.py:7-19): reflection pad (n_fft/2 - hop/2, n_fft/2 + hop/2), torchaudio
 * MelSpectrogram(sample_rate, n_fft, hop_length, center=False, f_min, f_max, n_mels) with torchaudio's defaults for the rest
 * (periodic Hann, power 2, HTK mel scale, norm None), then log(x + 1e-7).  audio[B,N] -> mel[B,n_mels,N/hop + 1].
 * f_max <= 0 means sr // 2.  n_fft must be a power of two <= 2048.  power (nullable, [B, n_fft/2+1, frames]) additionally receives
 * the power spectrogram before the mel filterbank -- the half of the conditioner that is plain torch upstream
 * (nn.ReflectionPad1d + torch.stft(periodic Hann, center=False).abs()**2) and is pinned to torch's own output
 * (tests/golden/cond_melspec_power.npz); the HTK filterbank on top restates torchaudio's published formula.  torchaudio is not part of the reference tree: its published
 * algorithm is restated (parity unpinned against torchaudio itself). */
int wg_melspec_frames(int N, int n_fft, int hop);
int wg_melspec(const float *audio, int B, int N, int sr, int n_fft, int hop, double f_min, double f_max, int n_mels, float *mel,
               float *power, void *stream);

/* LowPass.forward / STFTDecimate.forward (model/condition.py:22-66), the conditioner of the WSRGlow configs: right zero-pad by
 * n_fft, STFT (center, reflect, periodic Hann), zero the bins >= cut_bins, ISTFT, crop to T, keep every `step`-th sample.
 * STFTDecimate(r) = cut_bins int((n_fft/2+1) / r), step r; LowPass(x, i) = cut_bins int((n_fft/2+1) * ratio[i]), step 1.
 * x[B,T] -> out[B, ceil(T/step)].  ws: wg_lowpass_workspace_bytes. */
size_t wg_lowpass_workspace_bytes(int B, int T, int n_fft, int hop);
int wg_lowpass(const float *x, int B, int T, int n_fft, int hop, int cut_bins, int step, float *out, void *ws, size_t ws_bytes, void *stream);

/* ---- optimizer step (trainer parity) ---------------------------------------------------------------
 * torch.optim.Adam (amsgrad = false, maximize = false) on one contiguous fp32 range: what the reference's
 * configure_optimizers builds from `optimizer` in its configs (model/lightning.py:41-44; configs/waveglow_LJ_speech.json:
 * lr 1e-4; configs/wsrglow_vctk_2x.json: betas (0.9, 0.98)).  `step` is 1-based.  All four ranges must be 16-byte aligned.
 * Hyper-parameters are doubles (Python floats upstream): 1 - beta is formed in double before it is rounded to fp32. */
int wg_adam_step(float *param, const float *grad, float *exp_avg, float *exp_avg_sq, size_t n, double lr, double beta1, double beta2,
                 double eps, double weight_decay, int step, void *stream);

/* The affine coupling by itself, on plain arrays of n = B * ic * T floats: what AffineCouplingBlock computes around a transform that is
 * NOT this library's WN (efficient_modules.py:58-62 accepts any `transform_type`; the transform then runs as the caller's module).
 * apply: out = in * exp(log_s) + t, or (reverse) (in - t) / exp(log_s)  (:81 / :94).
 * backward (AffineCouplingFunc.backward :132-148 / InvAffineCouplingFunc.backward :194-206): from the block output's second half, the
 * recomputed log_s and t, the gradient of that half and of the returned log_s (nullable): the block input's second half rebuilt, the
 * gradients w.r.t. the transform's outputs (g_log_s, g_t: the seeds of its backward) and the gradient of the input half. */
int wg_affine_apply(const float *in, const float *log_s, const float *t, size_t n, int reverse, float *out, void *stream);
int wg_affine_backward(const float *out_half, const float *log_s, const float *t, const float *dout, const float *dlog_s, size_t n, int reverse,
                       float *in_rebuilt, float *g_log_s, float *g_t, float *din, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* WGFLOW_H */
