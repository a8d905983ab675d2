/*
 * wf_oracle.c -- CPU restatement of the WaveFlow hot path.  TEST INFRASTRUCTURE ONLY.
 *
 * Same rules as wg_oracle.c: only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load it; the
 * product never links, imports or calls anything under oracle/.
 *
 * Restates, in plain C loops (citations: path:line under yoyololicon/constant-memory-waveglow):
 *   - WaveFlow.forward_computation / reverse_computation      model/waveflow.py:182-208, 210-253; with use_conv1x1 the flip
 *     between flows is replaced by an InvertibleConv1x1(n_group) over the height axis (waveflow.py:179-181,203-206,219-226;
 *     model/efficient_modules.py:37-54: z = W x, logdet += W_time * logdet W)
 *   - the upsampler: ReplicationPad1d((0,1)) -> ConvTranspose1d(n_mels, n_mels, 2s+1, s, padding s//2) -> LeakyReLU(0.4)
 *                                                              model/waveflow.py:163-169, 255-257
 *   - WN2D: start 1x1, V conditioning, 8 NonCausalLayer2D (3x3 dilated conv, causal along the height axis, gate, W_o,
 *     residual + skip), end 1x1 -> (log_s, t)                  model/waveflow.py:14-51, 70-135
 *   - the row-by-row inverse that reverse_mode_forward implements with ring buffers (model/waveflow.py:53-67, 137-153)
 *   - weight norm (utils.py:14-16) and the NLL loss (model/loss.py:10-15)
 * and their gradients (the reference trains this model with plain autograd, memory_efficient = False).
 *
 * Layout: audio[b, w*H + h] = x[b][h][w] (waveflow.py:186, x.view(B,1,-1,H).transpose(2,3)); H = n_group rows, W = N/H columns.
 * The FlowBase hop length is fixed to 256 (waveflow.py:160), so the upsampling stride is s = 256 / n_group.
 *
 * Parity pinning: tests/golden/make_golden.py runs the imported reference on fill.py inputs -> tests/golden/model_wf*.npz.
 * Build: make -C oracle   (libwforacle.so: float arithmetic, libwforacle64.so: double)
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#ifndef WGO_REAL
#define WGO_REAL float
#endif
typedef WGO_REAL real;
#define WFO_API __attribute__((visibility("default")))
#define WF_DEPTH 8

typedef struct {
    int32_t flows, n_group, n_mels;
    int32_t res_ch, dil_ch, skip_ch;
    int32_t use_conv1x1;        /* WaveFlow(use_conv1x1=True): parameters invconv1x1.{k}.weight [H,H,1] follow the WN2D tables */
    int32_t bias;               /* WN2D(bias=True) (waveflow.py:77,100-122): every conv of every WN2D has a bias */
} wfo_config;

static void *xmalloc(size_t n)
{
    void *p = NULL;
    if (posix_memalign(&p, 64, n ? n : 64)) abort();
    return p;
}
static real *ralloc(size_t n) { return (real *)xmalloc(n * sizeof(real)); }
static real *rzalloc(size_t n)
{
    real *p = ralloc(n);
    memset(p, 0, (n ? n : 1) * sizeof(real));
    return p;
}

/* h_dilations of WN2D (waveflow.py:81-87); width dilations are 2^i (waveflow.py:90-91) */
static int h_dilations(int n_group, int *hd)
{
    static const int d8[8] = {1, 1, 1, 1, 1, 1, 1, 1}, d32[8] = {1, 2, 4, 1, 2, 4, 1, 2}, d64[8] = {1, 2, 4, 8, 16, 1, 2, 4},
                     d128[8] = {1, 2, 4, 8, 16, 32, 64, 1};
    const int *s = NULL;
    if (n_group == 8 || n_group == 16) s = d8;
    else if (n_group == 32) s = d32;
    else if (n_group == 64) s = d64;
    else if (n_group == 128) s = d128;
    if (!s) return -1;
    memcpy(hd, s, sizeof(int) * 8);
    return 0;
}

/* ---- invertible 1x1 over the height axis (efficient_modules.py:17-54) ------------------------------------------------------ */
/* LU with partial pivoting: log|det|, sign and (optionally) the inverse of a c x c matrix */
static void lu_logdet_inverse(const real *W, int c, real *logabs, int *sign, real *Winv)
{
    real *a = ralloc((size_t)c * c), *yv = ralloc(c);
    int *perm = (int *)xmalloc(sizeof(int) * c);
    memcpy(a, W, sizeof(real) * c * c);
    for (int i = 0; i < c; ++i) perm[i] = i;
    int sg = 1;
    real la = 0;
    for (int k = 0; k < c; ++k) {
        int p = k;
        real best = (real)fabs((double)a[k * c + k]);
        for (int r = k + 1; r < c; ++r) {
            const real v = (real)fabs((double)a[r * c + k]);
            if (v > best) { best = v; p = r; }
        }
        if (p != k) {
            for (int j = 0; j < c; ++j) { real tmp = a[k * c + j]; a[k * c + j] = a[p * c + j]; a[p * c + j] = tmp; }
            int ti = perm[k]; perm[k] = perm[p]; perm[p] = ti;
            sg = -sg;
        }
        const real piv = a[k * c + k];
        if (piv < 0) sg = -sg;
        la += (real)log(fabs((double)piv));
        for (int r = k + 1; r < c; ++r) {
            const real f = a[r * c + k] / piv;
            a[r * c + k] = f;
            for (int j = k + 1; j < c; ++j) a[r * c + j] -= f * a[k * c + j];
        }
    }
    *logabs = la;
    *sign = sg;
    if (Winv)
        for (int col = 0; col < c; ++col) {
            for (int r = 0; r < c; ++r) {
                real s = (perm[r] == col) ? (real)1 : (real)0;
                for (int j = 0; j < r; ++j) s -= a[r * c + j] * yv[j];
                yv[r] = s;
            }
            for (int r = c - 1; r >= 0; --r) {
                real s = yv[r];
                for (int j = r + 1; j < c; ++j) s -= a[r * c + j] * Winv[j * c + col];
                Winv[r * c + col] = s / a[r * c + r];
            }
        }
    free(a); free(yv); free(perm);
}
typedef struct { real *W, *Wi, ld; } mix_w;           /* ld = logdet W (NaN if det < 0, as torch.logdet) */
static void mix_w_build(const float *w, int H, mix_w *m)
{
    m->W = ralloc((size_t)H * H); m->Wi = ralloc((size_t)H * H);
    for (int i = 0; i < H * H; ++i) m->W[i] = (real)w[i];
    real la; int sg;
    lu_logdet_inverse(m->W, H, &la, &sg, m->Wi);
    m->ld = sg > 0 ? la : (real)NAN;
}
/* out[o][t] = sum_h M[o][h] x[h][t]   (transpose: M[h][o]) */
static void hmix(const real *M, int H, int Wd, const real *x, real *out, int transpose)
{
    for (int o = 0; o < H; ++o)
        for (int t = 0; t < Wd; ++t) {
            real acc = 0;
            for (int h = 0; h < H; ++h) acc += (transpose ? M[h * H + o] : M[o * H + h]) * x[(long)h * Wd + t];
            out[(long)o * Wd + t] = acc;
        }
}

/* weight norm, dim 0 (utils.py:14-16): w[o,:] = g[o] v[o,:] / ||v[o,:]|| */
static void wn_fwd(const float *g, const float *v, int rows, int cols, real *w)
{
    for (int o = 0; o < rows; ++o) {
        real ss = 0;
        for (int j = 0; j < cols; ++j) ss += (real)v[(long)o * cols + j] * (real)v[(long)o * cols + j];
        const real sc = (real)g[o] / (real)sqrt((double)ss);
        for (int j = 0; j < cols; ++j) w[(long)o * cols + j] = sc * (real)v[(long)o * cols + j];
    }
}
static void wn_bwd(const float *g, const float *v, const real *dw, int rows, int cols, float *dg, float *dv)
{
    for (int o = 0; o < rows; ++o) {
        const float *vo = v + (long)o * cols;
        const real *dwo = dw + (long)o * cols;
        real ss = 0, dot = 0;
        for (int j = 0; j < cols; ++j) { ss += (real)vo[j] * (real)vo[j]; dot += dwo[j] * (real)vo[j]; }
        const real nrm = (real)sqrt((double)ss);
        dg[o] = (float)(dot / nrm);
        const real a = (real)g[o] / nrm, bq = dot / ss;
        for (int j = 0; j < cols; ++j) dv[(long)o * cols + j] = (float)(a * (dwo[j] - (real)vo[j] * bq));
    }
}

/* parameter table = named_parameters() order of WaveFlow(use_conv1x1=False, bias=False):
 *   0 upsampler.1.bias  1 upsampler.1.weight_g  2 upsampler.1.weight_v
 *   per flow (37 entries): V.g V.v start.g start.v {W.g W.v W_o.g W_o.v} x 8  end.weight
 *   bias=True: 19 more per flow behind end.weight -- V.bias start.bias {W.bias W_o.bias} x 8 end.bias (the order of the 1-D WN's table) */
#define WF_PW (4 + 4 * WF_DEPTH + 1)
#define WF_PF (WF_PW + (cf->bias ? 2 + 2 * WF_DEPTH + 1 : 0))
WFO_API int wfo_param_count(const wfo_config *cf) { return 3 + cf->flows * WF_PF + (cf->use_conv1x1 ? cf->flows : 0); }

typedef struct {            /* effective (weight-normed) weights of one flow */
    real *V;                /* [16 Cd][n_mels] */
    real *start;            /* [C] */
    real *W[WF_DEPTH];      /* [2 Cd][C][3][3] */
    real *Wo[WF_DEPTH];     /* [rows_i][Cd] */
    real *end;              /* [2][Cs] */
    real *bV, *bstart, *bW[WF_DEPTH], *bWo[WF_DEPTH], *bend;      /* the biases (zeros without bias=True) */
} flow_w;

static int wo_rows(const wfo_config *cf, int i) { return i == WF_DEPTH - 1 ? cf->skip_ch : cf->res_ch + cf->skip_ch; }

static void flow_w_build(const wfo_config *cf, const float *const *p, flow_w *w)
{
    const int C = cf->res_ch, Cd = cf->dil_ch, Cs = cf->skip_ch;
    w->V = ralloc((size_t)16 * Cd * cf->n_mels);
    wn_fwd(p[0], p[1], 16 * Cd, cf->n_mels, w->V);
    w->start = ralloc(C);
    wn_fwd(p[2], p[3], C, 1, w->start);
    for (int i = 0; i < WF_DEPTH; ++i) {
        w->W[i] = ralloc((size_t)2 * Cd * C * 9);
        wn_fwd(p[4 + 4 * i], p[5 + 4 * i], 2 * Cd, C * 9, w->W[i]);
        w->Wo[i] = ralloc((size_t)wo_rows(cf, i) * Cd);
        wn_fwd(p[6 + 4 * i], p[7 + 4 * i], wo_rows(cf, i), Cd, w->Wo[i]);
    }
    w->end = ralloc((size_t)2 * Cs);
    for (int j = 0; j < 2 * Cs; ++j) w->end[j] = (real)p[4 + 4 * WF_DEPTH][j];
    const float *const *pb = cf->bias ? p + WF_PW : NULL;
    w->bV = rzalloc((size_t)16 * Cd); w->bstart = rzalloc(C); w->bend = rzalloc(2);
    if (pb) {
        for (int j = 0; j < 16 * Cd; ++j) w->bV[j] = (real)pb[0][j];
        for (int j = 0; j < C; ++j) w->bstart[j] = (real)pb[1][j];
        for (int j = 0; j < 2; ++j) w->bend[j] = (real)pb[2 + 2 * WF_DEPTH][j];
    }
    for (int i = 0; i < WF_DEPTH; ++i) {
        w->bW[i] = rzalloc((size_t)2 * Cd); w->bWo[i] = rzalloc((size_t)wo_rows(cf, i));
        if (pb) {
            for (int j = 0; j < 2 * Cd; ++j) w->bW[i][j] = (real)pb[2 + 2 * i][j];
            for (int j = 0; j < wo_rows(cf, i); ++j) w->bWo[i][j] = (real)pb[3 + 2 * i][j];
        }
    }
}
static void flow_w_free(flow_w *w)
{
    free(w->V); free(w->start); free(w->end); free(w->bV); free(w->bstart); free(w->bend);
    for (int i = 0; i < WF_DEPTH; ++i) { free(w->W[i]); free(w->Wo[i]); free(w->bW[i]); free(w->bWo[i]); }
}

/* ---- upsampler ------------------------------------------------------------------------------------------------------- */
/* y[o][j] = leaky(bias[o] + sum_c sum_i xpad[c][i] w[c][o][j + pad - s i]),  xpad = h with its last frame repeated once.
 * pre (optional) receives the pre-activation (for the backward).  Only columns j < Wd are produced (waveflow.py:187). */
static void upsample_fwd(const wfo_config *cf, const real *wup, const float *bias, const float *mel, int F, int Wd, real *y, real *pre)
{
    const int M = cf->n_mels, s = 256 / cf->n_group, K = 2 * s + 1, pad = s / 2, Fp = F + 1;
    for (int o = 0; o < M; ++o)
        for (int j = 0; j < Wd; ++j) {
            real acc = (real)bias[o];
            for (int i = 0; i < Fp; ++i) {
                const int k = j + pad - s * i;
                if (k < 0 || k >= K) continue;
                const int isrc = i < F ? i : F - 1;
                for (int c = 0; c < M; ++c) acc += (real)mel[(long)c * F + isrc] * wup[((long)c * M + o) * K + k];
            }
            if (pre) pre[(long)o * Wd + j] = acc;
            y[(long)o * Wd + j] = acc > 0 ? acc : (real)0.4 * acc;
        }
}

/* ---- WN2D on one batch item ---------------------------------------------------------------------------------------------
 * x: [R][Wd] (R = H-1 input rows), y: [n_mels][Wd].  Saves what the backward needs when sv != NULL. */
typedef struct {
    real *hin[WF_DEPTH];    /* layer inputs [C][R][Wd] */
    real *tw[WF_DEPTH], *sf[WF_DEPTH], *gate[WF_DEPTH];   /* [Cd][R][Wd] */
    real *S;                /* cumulated skip [Cs][R][Wd] */
} wn_saved;

static void wn_saved_alloc(const wfo_config *cf, int R, int Wd, wn_saved *sv)
{
    const size_t nC = (size_t)cf->res_ch * R * Wd, nD = (size_t)cf->dil_ch * R * Wd;
    for (int i = 0; i < WF_DEPTH; ++i) {
        sv->hin[i] = ralloc(nC); sv->tw[i] = ralloc(nD); sv->sf[i] = ralloc(nD); sv->gate[i] = ralloc(nD);
    }
    sv->S = ralloc((size_t)cf->skip_ch * R * Wd);
}
static void wn_saved_free(wn_saved *sv)
{
    for (int i = 0; i < WF_DEPTH; ++i) { free(sv->hin[i]); free(sv->tw[i]); free(sv->sf[i]); free(sv->gate[i]); }
    free(sv->S);
}

/* rows [r0, r1) of one layer: xy = W (*) hin + V_i y ; gate ; o = W_o gate ; res -> hout rows, skip accumulated into S rows */
static void layer_rows(const wfo_config *cf, const flow_w *w, int i, int hd, int R, int Wd, int r0, int r1, const real *hin,
                       const real *vy /* [16Cd][Wd] */, real *tw, real *sf, real *gate, real *hout, real *S)
{
    const int C = cf->res_ch, Cd = cf->dil_ch, Cs = cf->skip_ch, d = 1 << i, rows = wo_rows(cf, i);
    real *xy = ralloc((size_t)2 * Cd), *o = ralloc((size_t)rows);
    for (int r = r0; r < r1; ++r)
        for (int t = 0; t < Wd; ++t) {
            for (int m = 0; m < 2 * Cd; ++m) {
                real acc = vy[((long)i * 2 * Cd + m) * Wd + t] + w->bW[i][m];
                for (int kh = 0; kh < 3; ++kh) {
                    const int rr = r + (kh - 2) * hd;
                    if (rr < 0) continue;
                    for (int kw = 0; kw < 3; ++kw) {
                        const int tt = t + (kw - 1) * d;
                        if (tt < 0 || tt >= Wd) continue;
                        const real *wk = w->W[i] + (long)m * C * 9 + kh * 3 + kw;
                        const real *hp = hin + (long)rr * Wd + tt;
                        for (int c = 0; c < C; ++c) acc += wk[(long)c * 9] * hp[(long)c * R * Wd];
                    }
                }
                xy[m] = acc;
            }
            for (int c = 0; c < Cd; ++c) {
                const real a = (real)tanh((double)xy[c]), b = (real)(1.0 / (1.0 + exp(-(double)xy[Cd + c])));
                const long idx = ((long)c * R + r) * Wd + t;
                if (tw) { tw[idx] = a; sf[idx] = b; }
                gate[idx] = a * b;
            }
            for (int m = 0; m < rows; ++m) {
                real acc = w->bWo[i][m];
                for (int c = 0; c < Cd; ++c) acc += w->Wo[i][(long)m * Cd + c] * gate[((long)c * R + r) * Wd + t];
                o[m] = acc;
            }
            if (i < WF_DEPTH - 1) {
                for (int c = 0; c < C; ++c) hout[((long)c * R + r) * Wd + t] = o[c] + hin[((long)c * R + r) * Wd + t];
                for (int c = 0; c < Cs; ++c) S[((long)c * R + r) * Wd + t] += o[C + c];
            } else {
                for (int c = 0; c < Cs; ++c) S[((long)c * R + r) * Wd + t] += o[c];
            }
        }
    free(xy); free(o);
}

/* V y for all layers: [16 Cd][Wd] */
static void cond_project(const wfo_config *cf, const flow_w *w, const real *y, int Wd, real *vy)
{
    const int M = cf->n_mels, rows = 16 * cf->dil_ch;
    for (int m = 0; m < rows; ++m)
        for (int t = 0; t < Wd; ++t) {
            real acc = w->bV[m];
            for (int c = 0; c < M; ++c) acc += w->V[(long)m * M + c] * y[(long)c * Wd + t];
            vy[(long)m * Wd + t] = acc;
        }
}

/* full WN2D forward on rows [0, R): log_s, t: [R][Wd] */
static void wn_forward(const wfo_config *cf, const flow_w *w, const int *hd, const real *x, const real *y, int R, int Wd, real *ls, real *tt,
                       wn_saved *sv_out)
{
    const int C = cf->res_ch, Cs = cf->skip_ch;
    wn_saved local, *sv = sv_out;
    if (!sv) { wn_saved_alloc(cf, R, Wd, &local); sv = &local; }
    real *vy = ralloc((size_t)16 * cf->dil_ch * Wd);
    cond_project(cf, w, y, Wd, vy);
    for (int c = 0; c < C; ++c)
        for (long e = 0; e < (long)R * Wd; ++e) sv->hin[0][(long)c * R * Wd + e] = w->start[c] * x[e] + w->bstart[c];
    memset(sv->S, 0, sizeof(real) * Cs * R * Wd);
    real *spare = ralloc((size_t)C * R * Wd);
    for (int i = 0; i < WF_DEPTH; ++i)
        layer_rows(cf, w, i, hd[i], R, Wd, 0, R, sv->hin[i], vy, sv->tw[i], sv->sf[i], sv->gate[i], i < WF_DEPTH - 1 ? sv->hin[i + 1] : spare, sv->S);
    for (long e = 0; e < (long)R * Wd; ++e) {
        real a = w->bend[0], b = w->bend[1];
        for (int c = 0; c < Cs; ++c) { a += w->end[c] * sv->S[(long)c * R * Wd + e]; b += w->end[Cs + c] * sv->S[(long)c * R * Wd + e]; }
        ls[e] = a; tt[e] = b;
    }
    free(vy); free(spare);
    if (!sv_out) wn_saved_free(&local);
}

/* ---- model forward ----------------------------------------------------------------------------------------------------- */
static void squeeze_in(const float *audio, int H, int Wd, real *x)
{
    for (int h = 0; h < H; ++h)
        for (int t = 0; t < Wd; ++t) x[(long)h * Wd + t] = (real)audio[(long)t * H + h];
}
static void squeeze_out(const real *x, int H, int Wd, float *audio)
{
    for (int h = 0; h < H; ++h)
        for (int t = 0; t < Wd; ++t) audio[(long)t * H + h] = (float)x[(long)h * Wd + t];
}

WFO_API int wfo_forward(const wfo_config *cf, const float *const *params, const float *audio, const float *mel, int B, int N, int F,
                        float *z, float *logdet)
{
    int hd[8];
    const int H = cf->n_group;
    if (h_dilations(H, hd) || N % H) return -1;
    const int Wd = N / H, R = H - 1, s = 256 / H;
    if (Wd > (F + 1 - 1) * s - 2 * (s / 2) + 2 * s + 1) return -2;
    real *wup = ralloc((size_t)cf->n_mels * cf->n_mels * (2 * s + 1));
    wn_fwd(params[1], params[2], cf->n_mels, cf->n_mels * (2 * s + 1), wup);
    flow_w *fw = (flow_w *)xmalloc(sizeof(flow_w) * cf->flows);
    for (int k = 0; k < cf->flows; ++k) flow_w_build(cf, params + 3 + k * WF_PF, &fw[k]);
    const int conv = cf->use_conv1x1;
    mix_w *mw = (mix_w *)xmalloc(sizeof(mix_w) * cf->flows);
    if (conv) for (int k = 0; k < cf->flows; ++k) mix_w_build(params[3 + cf->flows * WF_PF + k], H, &mw[k]);
#pragma omp parallel for schedule(dynamic)
    for (int b = 0; b < B; ++b) {
        real *x = ralloc((size_t)H * Wd), *xn = ralloc((size_t)H * Wd), *y = ralloc((size_t)cf->n_mels * Wd);
        real *ls = ralloc((size_t)R * Wd), *tt = ralloc((size_t)R * Wd);
        upsample_fwd(cf, wup, params[0], mel + (long)b * cf->n_mels * F, F, Wd, y, NULL);
        squeeze_in(audio + (long)b * N, H, Wd, x);
        real ld = 0;
        for (int k = 0; k < cf->flows; ++k) {
            wn_forward(cf, &fw[k], hd, x, y, R, Wd, ls, tt, NULL);
            /* xout[r] = x[r+1] exp(ls[r]) + t[r] ; x_next = cat(flip(xout), x0), or W cat(x0, xout) with the 1x1   waveflow.py:198-206 */
            for (int r = 0; r < R; ++r)
                for (int t = 0; t < Wd; ++t) {
                    const long e = (long)r * Wd + t;
                    xn[(long)(conv ? r + 1 : R - 1 - r) * Wd + t] = x[(long)(r + 1) * Wd + t] * (real)exp((double)ls[e]) + tt[e];
                    ld += ls[e];
                }
            memcpy(xn + (long)(conv ? 0 : H - 1) * Wd, x, sizeof(real) * Wd);
            if (conv) {
                hmix(mw[k].W, H, Wd, xn, x, 0);                      /* efficient_modules.py:40 */
                ld += (real)Wd * mw[k].ld;                           /* :39, waveflow.py:206 */
            } else {
                real *tmp = x; x = xn; xn = tmp;
            }
        }
        squeeze_out(x, H, Wd, z + (long)b * N);
        logdet[b] = (float)ld;
        free(x); free(xn); free(y); free(ls); free(tt);
    }
    for (int k = 0; k < cf->flows; ++k) { flow_w_free(&fw[k]); if (conv) { free(mw[k].W); free(mw[k].Wi); } }
    free(fw); free(wup); free(mw);
    return 0;
}

/* ---- inverse: row by row (what reverse_mode_forward's buffers compute, waveflow.py:53-67,137-153,230-249) ------------------ */
WFO_API int wfo_inverse(const wfo_config *cf, const float *const *params, const float *z, const float *mel, int B, int N, int F,
                        float *xout, float *logdet)
{
    int hd[8];
    const int H = cf->n_group;
    if (h_dilations(H, hd) || N % H) return -1;
    const int Wd = N / H, R = H - 1, s = 256 / H, C = cf->res_ch, Cs = cf->skip_ch;
    real *wup = ralloc((size_t)cf->n_mels * cf->n_mels * (2 * s + 1));
    wn_fwd(params[1], params[2], cf->n_mels, cf->n_mels * (2 * s + 1), wup);
    flow_w *fw = (flow_w *)xmalloc(sizeof(flow_w) * cf->flows);
    for (int k = 0; k < cf->flows; ++k) flow_w_build(cf, params + 3 + k * WF_PF, &fw[k]);
    const int conv = cf->use_conv1x1;
    mix_w *mw = (mix_w *)xmalloc(sizeof(mix_w) * cf->flows);
    if (conv) for (int k = 0; k < cf->flows; ++k) mix_w_build(params[3 + cf->flows * WF_PF + k], H, &mw[k]);
#pragma omp parallel for schedule(dynamic)
    for (int b = 0; b < B; ++b) {
        real *zc = ralloc((size_t)H * Wd), *x = ralloc((size_t)H * Wd), *y = ralloc((size_t)cf->n_mels * Wd);
        real *vy = ralloc((size_t)16 * cf->dil_ch * Wd), *spare = ralloc((size_t)C * R * Wd);
        wn_saved sv;
        wn_saved_alloc(cf, R, Wd, &sv);
        upsample_fwd(cf, wup, params[0], mel + (long)b * cf->n_mels * F, F, Wd, y, NULL);
        squeeze_in(z + (long)b * N, H, Wd, zc);
        real ld = 0;
        for (int k = cf->flows - 1; k >= 0; --k) {
            const flow_w *w = &fw[k];
            if (conv) {                                             /* z = W^-1 z ; logdet -= W_time logdet W   (waveflow.py:224-229) */
                hmix(mw[k].Wi, H, Wd, zc, x, 0);
                memcpy(zc, x, sizeof(real) * H * Wd);
                ld -= (real)Wd * mw[k].ld;
            } else
            /* z = z.flip(2)  (waveflow.py:222) : rows [xout_flipped.., x0] -> [x0, xout..] */
            for (int h = 0; h < H / 2; ++h)
                for (int t = 0; t < Wd; ++t) {
                    const real a = zc[(long)h * Wd + t];
                    zc[(long)h * Wd + t] = zc[(long)(H - 1 - h) * Wd + t];
                    zc[(long)(H - 1 - h) * Wd + t] = a;
                }
            cond_project(cf, w, y, Wd, vy);
            memset(sv.S, 0, sizeof(real) * Cs * R * Wd);
            memcpy(x, zc, sizeof(real) * Wd);                       /* row 0 */
            for (int r = 0; r < R; ++r) {
                /* WN row r from input rows <= r, then x[r+1] = (z[r+1] - t[r]) / exp(ls[r]) */
                for (int c = 0; c < C; ++c)
                    for (int t = 0; t < Wd; ++t) sv.hin[0][((long)c * R + r) * Wd + t] = w->start[c] * x[(long)r * Wd + t] + w->bstart[c];
                for (int i = 0; i < WF_DEPTH; ++i)
                    layer_rows(cf, w, i, hd[i], R, Wd, r, r + 1, sv.hin[i], vy, NULL, NULL, sv.gate[i], i < WF_DEPTH - 1 ? sv.hin[i + 1] : spare, sv.S);
                for (int t = 0; t < Wd; ++t) {
                    real a = w->bend[0], bb = w->bend[1];
                    for (int c = 0; c < Cs; ++c) {
                        const real sv_ = sv.S[((long)c * R + r) * Wd + t];
                        a += w->end[c] * sv_; bb += w->end[Cs + c] * sv_;
                    }
                    x[(long)(r + 1) * Wd + t] = (zc[(long)(r + 1) * Wd + t] - bb) / (real)exp((double)a);
                    ld -= a;
                }
            }
            memcpy(zc, x, sizeof(real) * H * Wd);
        }
        squeeze_out(zc, H, Wd, xout + (long)b * N);
        logdet[b] = (float)ld;
        wn_saved_free(&sv);
        free(zc); free(x); free(y); free(vy); free(spare);
    }
    for (int k = 0; k < cf->flows; ++k) { flow_w_free(&fw[k]); if (conv) { free(mw[k].W); free(mw[k].Wi); } }
    free(fw); free(wup); free(mw);
    return 0;
}

/* ---- training step: forward, NLL (loss.py:10-15), backward ----------------------------------------------------------------- */
typedef struct {            /* gradient accumulators of one flow's EFFECTIVE weights */
    real *V, *start, *W[WF_DEPTH], *Wo[WF_DEPTH], *end;
    real *bV, *bstart, *bW[WF_DEPTH], *bWo[WF_DEPTH], *bend;
} flow_g;
static void flow_g_alloc(const wfo_config *cf, flow_g *g)
{
    const int C = cf->res_ch, Cd = cf->dil_ch, Cs = cf->skip_ch;
    g->V = rzalloc((size_t)16 * Cd * cf->n_mels);
    g->start = rzalloc(C);
    for (int i = 0; i < WF_DEPTH; ++i) { g->W[i] = rzalloc((size_t)2 * Cd * C * 9); g->Wo[i] = rzalloc((size_t)wo_rows(cf, i) * Cd); }
    g->end = rzalloc((size_t)2 * Cs);
    g->bV = rzalloc((size_t)16 * Cd); g->bstart = rzalloc(C); g->bend = rzalloc(2);
    for (int i = 0; i < WF_DEPTH; ++i) { g->bW[i] = rzalloc((size_t)2 * Cd); g->bWo[i] = rzalloc((size_t)wo_rows(cf, i)); }
}
static void flow_g_free(flow_g *g)
{
    free(g->V); free(g->start); free(g->end); free(g->bV); free(g->bstart); free(g->bend);
    for (int i = 0; i < WF_DEPTH; ++i) { free(g->W[i]); free(g->Wo[i]); free(g->bW[i]); free(g->bWo[i]); }
}
static void flow_g_add(const wfo_config *cf, flow_g *a, const flow_g *b)
{
    const int C = cf->res_ch, Cd = cf->dil_ch, Cs = cf->skip_ch;
    for (long j = 0; j < (long)16 * Cd * cf->n_mels; ++j) a->V[j] += b->V[j];
    for (int j = 0; j < C; ++j) a->start[j] += b->start[j];
    for (int i = 0; i < WF_DEPTH; ++i) {
        for (long j = 0; j < (long)2 * Cd * C * 9; ++j) a->W[i][j] += b->W[i][j];
        for (long j = 0; j < (long)wo_rows(cf, i) * Cd; ++j) a->Wo[i][j] += b->Wo[i][j];
    }
    for (int j = 0; j < 2 * Cs; ++j) a->end[j] += b->end[j];
    for (int j = 0; j < 16 * Cd; ++j) a->bV[j] += b->bV[j];
    for (int j = 0; j < C; ++j) a->bstart[j] += b->bstart[j];
    for (int j = 0; j < 2; ++j) a->bend[j] += b->bend[j];
    for (int i = 0; i < WF_DEPTH; ++i) {
        for (int j = 0; j < 2 * Cd; ++j) a->bW[i][j] += b->bW[i][j];
        for (int j = 0; j < wo_rows(cf, i); ++j) a->bWo[i][j] += b->bWo[i][j];
    }
}

/* WN2D backward on one item: given d log_s, d t [R][Wd] -> dx [R][Wd] (added), dy [n_mels][Wd] (added), weight grads (added) */
static void wn_backward(const wfo_config *cf, const flow_w *w, const int *hd, const real *x, const real *y, int R, int Wd, const wn_saved *sv,
                        const real *dls, const real *dtt, real *dx, real *dy, flow_g *g)
{
    const int C = cf->res_ch, Cd = cf->dil_ch, Cs = cf->skip_ch, M = cf->n_mels;
    const long RW = (long)R * Wd;
    real *dS = ralloc((size_t)Cs * RW), *dh = rzalloc((size_t)C * RW), *dhn = ralloc((size_t)C * RW);
    real *dxy = ralloc((size_t)2 * Cd * RW), *dvy = rzalloc((size_t)16 * Cd * Wd);
    /* end: out = W_end S */
    for (int c = 0; c < Cs; ++c) {
        real ga = 0, gb = 0;
        for (long e = 0; e < RW; ++e) {
            const real sv_ = sv->S[(long)c * RW + e];
            ga += dls[e] * sv_; gb += dtt[e] * sv_;
            dS[(long)c * RW + e] = w->end[c] * dls[e] + w->end[Cs + c] * dtt[e];
        }
        g->end[c] += ga; g->end[Cs + c] += gb;
    }
    for (long e = 0; e < RW; ++e) { g->bend[0] += dls[e]; g->bend[1] += dtt[e]; }
    for (int i = WF_DEPTH - 1; i >= 0; --i) {
        const int d = 1 << i, rows = wo_rows(cf, i), last = i == WF_DEPTH - 1;
        /* do = last ? dS : cat(dh_{i+1}, dS) ; dW_o += do gate^T ; dgate = W_o^T do */
        for (int c = 0; c < Cd; ++c) {
            const real *gp = sv->gate[i] + (long)c * RW, *twp = sv->tw[i] + (long)c * RW, *sfp = sv->sf[i] + (long)c * RW;
            for (long e = 0; e < RW; ++e) {
                real dgt = 0;
                for (int m = 0; m < rows; ++m) {
                    const real dom = last ? dS[(long)m * RW + e] : (m < C ? dh[(long)m * RW + e] : dS[(long)(m - C) * RW + e]);
                    dgt += w->Wo[i][(long)m * Cd + c] * dom;
                }
                dxy[(long)c * RW + e] = dgt * sfp[e] * (1 - twp[e] * twp[e]);
                dxy[(long)(Cd + c) * RW + e] = dgt * twp[e] * sfp[e] * (1 - sfp[e]);
            }
            for (int m = 0; m < rows; ++m) {
                real acc = 0;
                for (long e = 0; e < RW; ++e) {
                    const real dom = last ? dS[(long)m * RW + e] : (m < C ? dh[(long)m * RW + e] : dS[(long)(m - C) * RW + e]);
                    acc += dom * gp[e];
                }
                g->Wo[i][(long)m * Cd + c] += acc;
            }
        }
        for (int m = 0; m < rows; ++m) {
            real acc = 0;
            for (long e = 0; e < RW; ++e) acc += last ? dS[(long)m * RW + e] : (m < C ? dh[(long)m * RW + e] : dS[(long)(m - C) * RW + e]);
            g->bWo[i][m] += acc;
        }
        /* xy = W (*) hin + V_i y : dW, dV_i y part, dhin */
        for (int m = 0; m < 2 * Cd; ++m) {
            const real *dp = dxy + (long)m * RW;
            real tot = 0;
            for (int t = 0; t < Wd; ++t) {
                real acc = 0;
                for (int r = 0; r < R; ++r) acc += dp[(long)r * Wd + t];
                dvy[((long)i * 2 * Cd + m) * Wd + t] = acc;          /* the conditioning is broadcast over rows (waveflow.py:123) */
                tot += acc;
            }
            g->bW[i][m] += tot;
            g->bV[(long)i * 2 * Cd + m] += tot;                      /* V.bias and W.bias meet in the same pre-activation */
        }
        /* residual path: dh_i = (last ? 0 : dh_{i+1}) + W^T (*) dxy */
        if (last) memset(dhn, 0, sizeof(real) * C * RW);
        else memcpy(dhn, dh, sizeof(real) * C * RW);
        for (int m = 0; m < 2 * Cd; ++m)
            for (int c = 0; c < C; ++c)
                for (int kh = 0; kh < 3; ++kh)
                    for (int kw = 0; kw < 3; ++kw) {
                        const real wk = w->W[i][((long)m * C + c) * 9 + kh * 3 + kw];
                        real gacc = 0;
                        for (int r = 0; r < R; ++r) {
                            const int rr = r + (kh - 2) * hd[i];
                            if (rr < 0) continue;
                            for (int t = 0; t < Wd; ++t) {
                                const int tt = t + (kw - 1) * d;
                                if (tt < 0 || tt >= Wd) continue;
                                const real dv = dxy[(long)m * RW + (long)r * Wd + t];
                                gacc += dv * sv->hin[i][(long)c * RW + (long)rr * Wd + tt];
                                dhn[(long)c * RW + (long)rr * Wd + tt] += wk * dv;
                            }
                        }
                        g->W[i][((long)m * C + c) * 9 + kh * 3 + kw] += gacc;
                    }
        real *tmp = dh; dh = dhn; dhn = tmp;
    }
    /* V: vy = V y */
    for (int m = 0; m < 16 * Cd; ++m)
        for (int c = 0; c < M; ++c) {
            real acc = 0;
            for (int t = 0; t < Wd; ++t) acc += dvy[(long)m * Wd + t] * y[(long)c * Wd + t];
            g->V[(long)m * M + c] += acc;
        }
    for (int c = 0; c < M; ++c)
        for (int t = 0; t < Wd; ++t) {
            real acc = 0;
            for (int m = 0; m < 16 * Cd; ++m) acc += w->V[(long)m * M + c] * dvy[(long)m * Wd + t];
            dy[(long)c * Wd + t] += acc;
        }
    /* start: h0 = w_start[c] x */
    for (int c = 0; c < C; ++c) {
        real acc = 0;
        real bacc = 0;
        for (long e = 0; e < RW; ++e) { acc += dh[(long)c * RW + e] * x[e]; dx[e] += w->start[c] * dh[(long)c * RW + e]; bacc += dh[(long)c * RW + e]; }
        g->start[c] += acc;
        g->bstart[c] += bacc;
    }
    free(dS); free(dh); free(dhn); free(dxy); free(dvy);
}

/* grads: one float buffer per parameter (NULL entries are skipped); dmel optional [B][n_mels][F] */
WFO_API int wfo_train_step(const wfo_config *cf, const float *const *params, const float *audio, const float *mel, int B, int N, int F,
                           float sigma, float *z, float *logdet, float *loss, float *const *grads, float *dmel)
{
    int hd[8];
    const int H = cf->n_group, M = cf->n_mels;
    if (h_dilations(H, hd) || N % H) return -1;
    const int Wd = N / H, R = H - 1, s = 256 / H, K = 2 * s + 1, pad = s / 2, nf = cf->flows;
    const long HW = (long)H * Wd, RW = (long)R * Wd;
    real *wup = ralloc((size_t)M * M * K);
    wn_fwd(params[1], params[2], M, M * K, wup);
    flow_w *fw = (flow_w *)xmalloc(sizeof(flow_w) * nf);
    flow_g *fg = (flow_g *)xmalloc(sizeof(flow_g) * nf);
    for (int k = 0; k < nf; ++k) { flow_w_build(cf, params + 3 + k * WF_PF, &fw[k]); flow_g_alloc(cf, &fg[k]); }
    const int conv = cf->use_conv1x1;
    mix_w *mw = (mix_w *)xmalloc(sizeof(mix_w) * nf);
    real **gmix = (real **)xmalloc(sizeof(real *) * nf);          /* d loss / d W of the 1x1 convs, summed over the batch */
    for (int k = 0; k < nf; ++k) { gmix[k] = NULL; if (conv) { mix_w_build(params[3 + nf * WF_PF + k], H, &mw[k]); gmix[k] = rzalloc((size_t)H * H); } }
    real *gwup = rzalloc((size_t)M * M * K), *gbias = rzalloc(M);
    double loss_acc = 0;
    const real inv_s2 = (real)(1.0 / ((double)sigma * sigma)), scale = (real)(1.0 / ((double)B * N));
#pragma omp parallel for schedule(dynamic) reduction(+ : loss_acc)
    for (int b = 0; b < B; ++b) {
        real **xs = (real **)xmalloc(sizeof(real *) * (nf + 1));
        real **lss = (real **)xmalloc(sizeof(real *) * nf);
        for (int k = 0; k <= nf; ++k) xs[k] = ralloc((size_t)HW);
        real *y = ralloc((size_t)M * Wd), *pre = ralloc((size_t)M * Wd), *tt = ralloc((size_t)RW);
        const float *melb = mel + (long)b * M * F;
        upsample_fwd(cf, wup, params[0], melb, F, Wd, y, pre);
        squeeze_in(audio + (long)b * N, H, Wd, xs[0]);
        real ld = 0;
        real **pm = (real **)xmalloc(sizeof(real *) * nf);          /* use_conv1x1: cat(x0, xout), the input of flow k's 1x1 */
        for (int k = 0; k < nf; ++k) {
            lss[k] = ralloc((size_t)RW);
            pm[k] = conv ? ralloc((size_t)HW) : NULL;
            real *dst = conv ? pm[k] : xs[k + 1];
            wn_forward(cf, &fw[k], hd, xs[k], y, R, Wd, lss[k], tt, NULL);
            for (int r = 0; r < R; ++r)
                for (int t = 0; t < Wd; ++t) {
                    const long e = (long)r * Wd + t;
                    dst[(long)(conv ? r + 1 : R - 1 - r) * Wd + t] = xs[k][(long)(r + 1) * Wd + t] * (real)exp((double)lss[k][e]) + tt[e];
                    ld += lss[k][e];
                }
            memcpy(dst + (long)(conv ? 0 : H - 1) * Wd, xs[k], sizeof(real) * Wd);
            if (conv) { hmix(mw[k].W, H, Wd, pm[k], xs[k + 1], 0); ld += (real)Wd * mw[k].ld; }
        }
        squeeze_out(xs[nf], H, Wd, z + (long)b * N);
        logdet[b] = (float)ld;
        double zz = 0;
        for (long e = 0; e < HW; ++e) zz += (double)xs[nf][e] * (double)xs[nf][e];
        loss_acc += 0.5 * zz * (double)inv_s2 - (double)ld;
        /* backward */
        real *dxn = ralloc((size_t)HW), *dx = ralloc((size_t)HW), *dy = rzalloc((size_t)M * Wd);
        real *dls = ralloc((size_t)RW), *dtt = ralloc((size_t)RW);
        for (long e = 0; e < HW; ++e) dxn[e] = xs[nf][e] * inv_s2 * scale;
        flow_g *lg = (flow_g *)xmalloc(sizeof(flow_g) * nf);
        wn_saved sv;
        wn_saved_alloc(cf, R, Wd, &sv);
        for (int k = nf - 1; k >= 0; --k) {
            flow_g_alloc(cf, &lg[k]);
            wn_forward(cf, &fw[k], hd, xs[k], y, R, Wd, lss[k], tt, &sv);       /* activations of this flow (plain autograd keeps them) */
            real *lgm = NULL;
            if (conv) {
                /* z = W u, logdet += W_time logdet W:  dW = sum_t dz u^T + W^-T (d logdet) W_time,  du = W^T dz   (efficient_modules.py:239-242) */
                lgm = rzalloc((size_t)H * H);
                for (int o = 0; o < H; ++o)
                    for (int h2 = 0; h2 < H; ++h2) {
                        real acc = 0;
                        for (int t = 0; t < Wd; ++t) acc += dxn[(long)o * Wd + t] * pm[k][(long)h2 * Wd + t];
                        lgm[o * H + h2] = acc + mw[k].Wi[h2 * H + o] * (-scale) * (real)Wd;
                    }
                hmix(mw[k].W, H, Wd, dxn, dx, 1);
                memcpy(dxn, dx, sizeof(real) * HW);
#pragma omp critical
                for (int j = 0; j < H * H; ++j) gmix[k][j] += lgm[j];
                free(lgm);
            }
            memset(dx, 0, sizeof(real) * HW);
            for (int r = 0; r < R; ++r)
                for (int t = 0; t < Wd; ++t) {
                    const long e = (long)r * Wd + t;
                    const real gout = dxn[(long)(conv ? r + 1 : R - 1 - r) * Wd + t], es = (real)exp((double)lss[k][e]);
                    const real xv = xs[k][(long)(r + 1) * Wd + t];
                    dx[(long)(r + 1) * Wd + t] += gout * es;
                    dls[e] = gout * xv * es - scale;                            /* + d loss / d logdet = -1/(B N) */
                    dtt[e] = gout;
                }
            for (int t = 0; t < Wd; ++t) dx[t] += dxn[(long)(conv ? 0 : H - 1) * Wd + t];
            wn_backward(cf, &fw[k], hd, xs[k], y, R, Wd, &sv, dls, dtt, dx, dy, &lg[k]);
            real *tmp = dxn; dxn = dx; dx = tmp;
        }
        wn_saved_free(&sv);
        /* upsampler backward: y = leaky(pre) */
        real *dmelb = rzalloc((size_t)M * (F + 1)), *lgw = rzalloc((size_t)M * M * K), *lgb = rzalloc(M);
        for (int o = 0; o < M; ++o)
            for (int j = 0; j < Wd; ++j) {
                const real gp = dy[(long)o * Wd + j] * (pre[(long)o * Wd + j] > 0 ? (real)1 : (real)0.4);
                lgb[o] += gp;
                for (int i = 0; i < F + 1; ++i) {
                    const int kk = j + pad - s * i;
                    if (kk < 0 || kk >= K) continue;
                    const int isrc = i < F ? i : F - 1;
                    for (int c = 0; c < M; ++c) {
                        lgw[((long)c * M + o) * K + kk] += (real)melb[(long)c * F + isrc] * gp;
                        dmelb[(long)c * (F + 1) + i] += wup[((long)c * M + o) * K + kk] * gp;
                    }
                }
            }
        if (dmel)
            for (int c = 0; c < M; ++c)
                for (int i = 0; i < F; ++i)
                    dmel[((long)b * M + c) * F + i] = (float)(dmelb[(long)c * (F + 1) + i] + (i == F - 1 ? dmelb[(long)c * (F + 1) + F] : 0));
#pragma omp critical
        {
            for (int k = 0; k < nf; ++k) flow_g_add(cf, &fg[k], &lg[k]);
            for (long j = 0; j < (long)M * M * K; ++j) gwup[j] += lgw[j];
            for (int j = 0; j < M; ++j) gbias[j] += lgb[j];
        }
        for (int k = 0; k < nf; ++k) { flow_g_free(&lg[k]); free(lss[k]); free(pm[k]); }
        free(pm);
        for (int k = 0; k <= nf; ++k) free(xs[k]);
        free(lg); free(xs); free(lss); free(y); free(pre); free(tt); free(dxn); free(dx); free(dy); free(dls); free(dtt);
        free(dmelb); free(lgw); free(lgb);
    }
    *loss = (float)(loss_acc / ((double)B * N));
    /* effective-weight grads -> parameter grads */
    if (grads[0]) for (int j = 0; j < M; ++j) grads[0][j] = (float)gbias[j];
    if (grads[2]) wn_bwd(params[1], params[2], gwup, M, M * K, grads[1], grads[2]);
    for (int k = 0; k < nf; ++k) {
        const float *const *p = params + 3 + k * WF_PF;
        float *const *g = grads + 3 + k * WF_PF;
        wn_bwd(p[0], p[1], fg[k].V, 16 * cf->dil_ch, M, g[0], g[1]);
        wn_bwd(p[2], p[3], fg[k].start, cf->res_ch, 1, g[2], g[3]);
        for (int i = 0; i < WF_DEPTH; ++i) {
            wn_bwd(p[4 + 4 * i], p[5 + 4 * i], fg[k].W[i], 2 * cf->dil_ch, cf->res_ch * 9, g[4 + 4 * i], g[5 + 4 * i]);
            wn_bwd(p[6 + 4 * i], p[7 + 4 * i], fg[k].Wo[i], wo_rows(cf, i), cf->dil_ch, g[6 + 4 * i], g[7 + 4 * i]);
        }
        for (int j = 0; j < 2 * cf->skip_ch; ++j) g[4 + 4 * WF_DEPTH][j] = (float)fg[k].end[j];
        if (cf->bias) {
            float *const *gb = g + WF_PW;
            for (int j = 0; j < 16 * cf->dil_ch; ++j) gb[0][j] = (float)fg[k].bV[j];
            for (int j = 0; j < cf->res_ch; ++j) gb[1][j] = (float)fg[k].bstart[j];
            for (int i = 0; i < WF_DEPTH; ++i) {
                for (int j = 0; j < 2 * cf->dil_ch; ++j) gb[2 + 2 * i][j] = (float)fg[k].bW[i][j];
                for (int j = 0; j < wo_rows(cf, i); ++j) gb[3 + 2 * i][j] = (float)fg[k].bWo[i][j];
            }
            for (int j = 0; j < 2; ++j) gb[2 + 2 * WF_DEPTH][j] = (float)fg[k].bend[j];
        }
        flow_w_free(&fw[k]); flow_g_free(&fg[k]);
        if (conv) {
            float *gw = grads[3 + nf * WF_PF + k];
            if (gw) for (int j = 0; j < H * H; ++j) gw[j] = (float)gmix[k][j];
            free(gmix[k]); free(mw[k].W); free(mw[k].Wi);
        }
    }
    free(fw); free(fg); free(wup); free(gwup); free(gbias); free(mw); free(gmix);
    return 0;
}

WFO_API int wfo_real_bytes(void) { return (int)sizeof(real); }
#ifdef _OPENMP
#include <omp.h>
WFO_API int wfo_set_threads(int n) { if (n > 0) omp_set_num_threads(n); return omp_get_max_threads(); }
#else
WFO_API int wfo_set_threads(int n) { (void)n; return 1; }
#endif
