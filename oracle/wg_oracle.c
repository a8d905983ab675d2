/*
 * wg_oracle.c -- CPU restatement of the WaveGlow flow hot path.  TEST INFRASTRUCTURE ONLY.
 *
 * This file is the parity oracle for the HIP engine in constant-memory-waveglow_amd/csrc.
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load it; the
 * product path never links, imports or calls anything in oracle/.
 *
 * It restates, in plain C loops, what the reference computes (citations are
 * path:line under the upstream repo yoyololicon/constant-memory-waveglow):
 *   - weight-norm parameterisation            utils.py:14-16 (nn.utils.weight_norm, dim 0)
 *   - mel upsampler (depthwise ConvTranspose) model/waveglow.py:126-130,210-212
 *   - squeeze / unsqueeze / early outputs     model/waveglow.py:153,164-170,178-179,190-205
 *   - WN transform net                        model/waveglow.py:13-15,41-46,98-105
 *   - invertible 1x1 convolution              model/efficient_modules.py:37-54,215-279
 *   - affine coupling fwd / inverse / bwd     model/efficient_modules.py:77-96,99-212
 *   - NLL loss                                model/loss.py:10-15
 *   - WSRGlow conditioning front-end          model/wsrglow.py:8-18,27-50 (mu-law: torchaudio, restated)
 * The backward pass follows the reference's constant-memory protocol: nothing but the
 * flow outputs is kept; each block rebuilds its input from its output
 * (efficient_modules.py:127-136, 235-237) and that REBUILT input is what enters the
 * gradient formulas, exactly as upstream.
 *
 * Parity pinning: the reference has no golden vectors of its own; this oracle
 * is pinned against outputs of the reference itself, imported in the build container by
 * tests/golden/make_golden.py, whose outputs are committed under tests/golden/.
 *
 * Build:  make -C oracle      (float: libwgoracle.so, double: libwgoracle64.so)
 * The arithmetic type is WGO_REAL (float or double); all I/O buffers are float.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#ifndef WGO_REAL
#define WGO_REAL float
#endif
typedef WGO_REAL real;

#define WGO_API __attribute__((visibility("default")))

typedef struct {
    int32_t n_flows, n_group, n_early_every, n_early_size, n_mels;
    int32_t up_stride, up_kernel, up_pad;           /* ConvTranspose1d(n_mels, n_mels, k, stride, pad, groups=n_mels) */
    int32_t res_ch, dil_ch, skip_ch, depth, radix;  /* WN */
    int32_t bias;                                   /* WN(bias=True) (waveglow.py:58): every conv of the WN carries a bias */
} wgo_config;

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
static inline int imin(int a, int b) { return a < b ? a : b; }

/* ------------------------------------------------------------------------------------------
 * dense primitives
 * ---------------------------------------------------------------------------------------- */

/* out[o][t] += sum_i w[o*so + i*si] * in[i][t+shift]   (in is [K][T], zero outside [0,T)) */
static void conv_tap_acc(real *out, int M, int T, const real *w, long so, long si,
                         const real *in, int K, int shift)
{
    const int t_lo = shift < 0 ? -shift : 0;
    const int t_hi = shift > 0 ? T - shift : T;
    if (t_hi <= t_lo) return;
#pragma omp parallel for schedule(static)
    for (int ob = 0; ob < M; ob += 4) {
        const int mb = imin(4, M - ob);
        for (int tb = t_lo; tb < t_hi; tb += 1024) {
            const int te = imin(tb + 1024, t_hi);
            for (int i = 0; i < K; ++i) {
                const real *src = in + (long)i * T + shift;
                if (mb == 4) {
                    const real w0 = w[(ob + 0) * so + i * si], w1 = w[(ob + 1) * so + i * si];
                    const real w2 = w[(ob + 2) * so + i * si], w3 = w[(ob + 3) * so + i * si];
                    real *o0 = out + (long)(ob + 0) * T, *o1 = out + (long)(ob + 1) * T;
                    real *o2 = out + (long)(ob + 2) * T, *o3 = out + (long)(ob + 3) * T;
#pragma omp simd
                    for (int t = tb; t < te; ++t) {
                        const real s = src[t];
                        o0[t] += w0 * s;
                        o1[t] += w1 * s;
                        o2[t] += w2 * s;
                        o3[t] += w3 * s;
                    }
                } else {
                    for (int m = 0; m < mb; ++m) {
                        const real wm = w[(ob + m) * so + i * si];
                        real *om = out + (long)(ob + m) * T;
#pragma omp simd
                        for (int t = tb; t < te; ++t) om[t] += wm * src[t];
                    }
                }
            }
        }
    }
}

/* dw[o*so + i*si] += sum_t a[o][t] * in[i][t+shift]    (a is [M][T], in is [K][T]) */
static void wgrad_tap_acc(real *dw, long so, long si, const real *a, int M, const real *in,
                          int K, int T, int shift)
{
    const int t_lo = shift < 0 ? -shift : 0;
    const int t_hi = shift > 0 ? T - shift : T;
    if (t_hi <= t_lo) return;
#pragma omp parallel for schedule(static)
    for (int o = 0; o < M; ++o) {
        const real *ao = a + (long)o * T;
        for (int i = 0; i < K; ++i) {
            const real *src = in + (long)i * T + shift;
            real acc = 0;
#pragma omp simd reduction(+ : acc)
            for (int t = t_lo; t < t_hi; ++t) acc += ao[t] * src[t];
            dw[o * so + i * si] += acc;
        }
    }
}

/* ------------------------------------------------------------------------------------------
 * weight norm  (utils.py:14-16 -> torch.nn.utils.weight_norm, dim=0)
 *   w[o,:] = g[o] * v[o,:] / ||v[o,:]||_2 ;  g == NULL means "plain weight" (after remove_weight_norm)
 * ---------------------------------------------------------------------------------------- */
static void weight_norm_fwd(const float *g, const float *v, int rows, int cols, real *w)
{
    for (int o = 0; o < rows; ++o) {
        const float *vo = v + (long)o * cols;
        real scale = 1;
        if (g) {
            real ss = 0;
            for (int j = 0; j < cols; ++j) ss += (real)vo[j] * (real)vo[j];
            scale = (real)g[o] / (real)sqrt((double)ss);
        }
        for (int j = 0; j < cols; ++j) w[(long)o * cols + j] = scale * (real)vo[j];
    }
}

/* dg[o] = <dw[o],v[o]>/||v[o]|| ; dv[o] = g[o]/||v[o]|| * (dw[o] - v[o] <dw[o],v[o]>/||v[o]||^2) */
static void weight_norm_bwd(const float *g, const float *v, const real *dw, int rows, int cols,
                            float *dg, float *dv)
{
    for (int o = 0; o < rows; ++o) {
        const float *vo = v + (long)o * cols;
        const real *dwo = dw + (long)o * cols;
        if (!g) {
            for (int j = 0; j < cols; ++j) dv[(long)o * cols + j] = (float)dwo[j];
            continue;
        }
        real ss = 0, dot = 0;
        for (int j = 0; j < cols; ++j) {
            ss += (real)vo[j] * (real)vo[j];
            dot += dwo[j] * (real)vo[j];
        }
        const real nrm = (real)sqrt((double)ss);
        if (dg) dg[o] = (float)(dot / nrm);
        const real a = (real)g[o] / nrm, bq = dot / ss;
        for (int j = 0; j < cols; ++j) dv[(long)o * cols + j] = (float)(a * (dwo[j] - (real)vo[j] * bq));
    }
}

/* ------------------------------------------------------------------------------------------
 * small dense linear algebra for the invertible 1x1 conv (c <= 64)
 * ---------------------------------------------------------------------------------------- */
#define WGO_MAXC 64

/* LU with partial pivoting; returns log|det| in *logabs and the sign; optionally the inverse. */
static void lu_logdet_inverse(const real *W, int c, real *logabs, int *sign, real *Winv)
{
    real a[WGO_MAXC * WGO_MAXC];
    int perm[WGO_MAXC];
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
    if (!Winv) return;
    /* solve (P W) X = P I column by column:  L U x = e_perm */
    for (int col = 0; col < c; ++col) {
        real yv[WGO_MAXC];
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
}

/* torch.logdet semantics: NaN for det < 0 (efficient_modules.py:38 comment) */
static real logdet_of(const real *W, int c)
{
    real la; int sg;
    lu_logdet_inverse(W, c, &la, &sg, NULL);
    return sg > 0 ? la : (real)NAN;
}

/* z[b][o][t] = sum_i M[o][i] x[b][i][t]   (1x1 conv, c x c) */
static void mix_channels(const real *M, int c, const real *x, int T, real *z)
{
#pragma omp parallel for schedule(static)
    for (int tb = 0; tb < T; tb += 256) {
        const int te = imin(tb + 256, T);
        for (int o = 0; o < c; ++o) {
            real *zo = z + (long)o * T;
            for (int t = tb; t < te; ++t) zo[t] = 0;
            for (int i = 0; i < c; ++i) {
                const real m = M[o * c + i];
                const real *xi = x + (long)i * T;
#pragma omp simd
                for (int t = tb; t < te; ++t) zo[t] += m * xi[t];
            }
        }
    }
}

/* ------------------------------------------------------------------------------------------
 * mel upsampler: depthwise ConvTranspose1d + bias, cropped to T   (waveglow.py:126-130,157)
 *   y[c][j] = bias[c] + sum_i h[c][i] * w[c][j + pad - stride*i],  0 <= j+pad-stride*i < K
 * ---------------------------------------------------------------------------------------- */
static void upsample_fwd(const wgo_config *cf, const real *w, const float *bias, const float *h,
                         int F, int T, real *y /* [n_mels][T] */)
{
    const int K = cf->up_kernel, S = cf->up_stride, P = cf->up_pad;
#pragma omp parallel for schedule(static)
    for (int c = 0; c < cf->n_mels; ++c) {
        real *yc = y + (long)c * T;
        for (int j = 0; j < T; ++j) yc[j] = bias ? (real)bias[c] : (real)0;
        for (int i = 0; i < F; ++i) {
            const real hv = (real)h[(long)c * F + i];
            for (int kk = 0; kk < K; ++kk) {
                const int j = S * i + kk - P;
                if (j >= 0 && j < T) yc[j] += hv * w[c * K + kk];
            }
        }
    }
}

static void upsample_bwd(const wgo_config *cf, const real *w, const float *h, int F, int T,
                         const real *dy, real *dw /* += [n_mels][K] */, real *dbias /* += */,
                         real *dh /* nullable, = [n_mels][F] */)
{
    const int K = cf->up_kernel, S = cf->up_stride, P = cf->up_pad;
#pragma omp parallel for schedule(static)
    for (int c = 0; c < cf->n_mels; ++c) {
        const real *dyc = dy + (long)c * T;
        real sb = 0;
        for (int j = 0; j < T; ++j) sb += dyc[j];
        dbias[c] += sb;
        for (int i = 0; i < F; ++i) {
            const real hv = (real)h[(long)c * F + i];
            real acc_h = 0;
            for (int kk = 0; kk < K; ++kk) {
                const int j = S * i + kk - P;
                if (j >= 0 && j < T) {
                    dw[c * K + kk] += hv * dyc[j];
                    acc_h += dyc[j] * w[c * K + kk];
                }
            }
            if (dh) dh[(long)c * F + i] = acc_h;
        }
    }
}

/* ------------------------------------------------------------------------------------------
 * WN transform net   (waveglow.py:49-105)
 * ---------------------------------------------------------------------------------------- */
typedef struct {
    int in_ch, aux, C, Cd, Cs, depth, radix;
    int bias;   /* the table then continues behind `end` with V.bias, start.bias, depth x (W.bias, W_o.bias), end.bias */
} wn_dims;

/* number of parameter-table entries of one WN: V(g,v) start(g,v) depth*(W g,v ; W_o g,v) end  [+ the biases] */
static int wn_entries(int depth, int bias) { return 4 + 4 * depth + 1 + (bias ? 2 + 2 * depth + 1 : 0); }
static int wo_rows(const wn_dims *d, int i) { return i == d->depth - 1 ? d->Cs : d->C + d->Cs; }

typedef struct {
    real *V;      /* [2Cd*depth][aux] */
    real *start;  /* [C][in_ch] */
    real **W;     /* depth x [2Cd][C][radix] */
    real **Wo;    /* depth x [rows_i][Cd] */
    real *end;    /* [2 in_ch][Cs] */
    const float *bV, *bS, *bE;     /* biases (views of the parameter table; NULL without bias): [2Cd*depth], [C], [2 in_ch] */
    const float **bW, **bWo;       /* depth x [2Cd], depth x [rows_i] */
} wn_weights;

static void wn_weights_build(const wn_dims *d, const float *const *p, wn_weights *w)
{
    w->V = ralloc((size_t)2 * d->Cd * d->depth * d->aux);
    weight_norm_fwd(p[0], p[1], 2 * d->Cd * d->depth, d->aux, w->V);
    w->start = ralloc((size_t)d->C * d->in_ch);
    weight_norm_fwd(p[2], p[3], d->C, d->in_ch, w->start);
    w->W = (real **)xmalloc(sizeof(real *) * d->depth);
    w->Wo = (real **)xmalloc(sizeof(real *) * d->depth);
    for (int i = 0; i < d->depth; ++i) {
        w->W[i] = ralloc((size_t)2 * d->Cd * d->C * d->radix);
        weight_norm_fwd(p[4 + 4 * i], p[5 + 4 * i], 2 * d->Cd, d->C * d->radix, w->W[i]);
        w->Wo[i] = ralloc((size_t)wo_rows(d, i) * d->Cd);
        weight_norm_fwd(p[6 + 4 * i], p[7 + 4 * i], wo_rows(d, i), d->Cd, w->Wo[i]);
    }
    const float *e = p[4 + 4 * d->depth];
    w->end = ralloc((size_t)2 * d->in_ch * d->Cs);
    for (long j = 0; j < (long)2 * d->in_ch * d->Cs; ++j) w->end[j] = (real)e[j];
    w->bV = w->bS = w->bE = NULL;
    w->bW = (const float **)xmalloc(sizeof(float *) * d->depth);
    w->bWo = (const float **)xmalloc(sizeof(float *) * d->depth);
    for (int i = 0; i < d->depth; ++i) w->bW[i] = w->bWo[i] = NULL;
    if (d->bias) {
        const float *const *b = p + 4 + 4 * d->depth + 1;
        w->bV = b[0]; w->bS = b[1];
        for (int i = 0; i < d->depth; ++i) { w->bW[i] = b[2 + 2 * i]; w->bWo[i] = b[3 + 2 * i]; }
        w->bE = b[2 + 2 * d->depth];
    }
}
/* out[r][t] += b[r] */
static void add_bias(real *out, int rows, int T, const float *b)
{
    if (!b) return;
    for (int r = 0; r < rows; ++r)
        for (int t = 0; t < T; ++t) out[(long)r * T + t] += (real)b[r];
}
/* db[r] += sum_t a[r][t] */
static void bias_grad(real *db, const real *a, int rows, int T)
{
    if (!db) return;
    for (int r = 0; r < rows; ++r) {
        double s = 0;
        for (int t = 0; t < T; ++t) s += (double)a[(long)r * T + t];
        db[r] += (real)s;
    }
}

static void wn_weights_free(const wn_dims *d, wn_weights *w)
{
    free(w->V); free(w->start); free(w->end);
    for (int i = 0; i < d->depth; ++i) { free(w->W[i]); free(w->Wo[i]); }
    free(w->W); free(w->Wo); free((void *)w->bW); free((void *)w->bWo);
}

/* activations of ONE batch item kept for the backward pass */
typedef struct {
    real *h;     /* (depth) x [C][T]  : input of layer i (h[0] = start(xa)) */
    real *tw;    /* depth x [Cd][T]   : tanh(zw) */
    real *sf;    /* depth x [Cd][T]   : sigmoid(zf) */
    real *skip;  /* [Cs][T]           : cum_skip */
} wn_saved;

static void wn_saved_alloc(const wn_dims *d, int T, wn_saved *s)
{
    s->h = ralloc((size_t)d->depth * d->C * T);
    s->tw = ralloc((size_t)d->depth * d->Cd * T);
    s->sf = ralloc((size_t)d->depth * d->Cd * T);
    s->skip = ralloc((size_t)d->Cs * T);
}
static void wn_saved_free(wn_saved *s) { free(s->h); free(s->tw); free(s->sf); free(s->skip); }

/* (log_s, t) = WN(xa, y) for one batch item.  out is [2*in_ch][T] : rows [0,in_ch) = log_s, rest = t
 * (chunk order of waveglow.py:105).  `s` always receives the activations (callers free it). */
static void wn_forward(const wn_dims *d, const wn_weights *w, const real *xa, const real *y, int T,
                       wn_saved *s, real *out)
{
    const long CT = (long)d->C * T, DT = (long)d->Cd * T;
    real *xy = ralloc((size_t)2 * DT);
    real *o = ralloc((size_t)(d->C + d->Cs) * T);
    real *gate = ralloc((size_t)DT);
    memset(s->h, 0, sizeof(real) * CT);
    conv_tap_acc(s->h, d->C, T, w->start, d->in_ch, 1, xa, d->in_ch, 0);          /* waveglow.py:99 */
    add_bias(s->h, d->C, T, w->bS);
    memset(s->skip, 0, sizeof(real) * d->Cs * T);
    for (int i = 0; i < d->depth; ++i) {
        const int dil = 1 << i;                                                     /* waveglow.py:61 */
        const real *hi = s->h + i * CT;
        memset(xy, 0, sizeof(real) * 2 * DT);
        for (int k = 0; k < d->radix; ++k)                                          /* waveglow.py:28-30,42 */
            conv_tap_acc(xy, 2 * d->Cd, T, w->W[i] + k, (long)d->C * d->radix, d->radix, hi, d->C,
                         (k - (d->radix - 1) / 2) * dil);
        conv_tap_acc(xy, 2 * d->Cd, T, w->V + (long)i * 2 * d->Cd * d->aux, d->aux, 1, y, d->aux, 0); /* :100-102 */
        add_bias(xy, 2 * d->Cd, T, w->bW[i]);
        add_bias(xy, 2 * d->Cd, T, w->bV ? w->bV + (long)i * 2 * d->Cd : NULL);
        real *tw = s->tw + i * DT, *sf = s->sf + i * DT;
#pragma omp parallel for schedule(static)
        for (long j = 0; j < DT; ++j) {                                             /* waveglow.py:13-15,43-44 */
            tw[j] = (real)tanh((double)xy[j]);
            sf[j] = (real)(1.0 / (1.0 + exp(-(double)xy[DT + j])));
            gate[j] = tw[j] * sf[j];
        }
        const int rows = wo_rows(d, i);
        memset(o, 0, sizeof(real) * rows * T);
        conv_tap_acc(o, rows, T, w->Wo[i], d->Cd, 1, gate, d->Cd, 0);               /* waveglow.py:45 */
        add_bias(o, rows, T, w->bWo[i]);
        const real *sk = o;
        if (i < d->depth - 1) {                                                     /* waveglow.py:46 */
            real *hn = s->h + (i + 1) * CT;
            for (long j = 0; j < CT; ++j) hn[j] = o[j] + hi[j];
            sk = o + CT;
        }
        for (long j = 0; j < (long)d->Cs * T; ++j) s->skip[j] += sk[j];             /* waveglow.py:104 */
    }
    memset(out, 0, sizeof(real) * 2 * d->in_ch * T);
    conv_tap_acc(out, 2 * d->in_ch, T, w->end, d->Cs, 1, s->skip, d->Cs, 0);        /* waveglow.py:105 */
    add_bias(out, 2 * d->in_ch, T, w->bE);
    free(xy); free(o); free(gate);
}

/* gradient tables of one WN, in effective-weight space, accumulated over the batch */
typedef struct {
    real *V, *start, **W, **Wo, *end;
    real *bV, *bS, **bW, **bWo, *bE;      /* NULL without bias */
} wn_wgrads;

static void wn_wgrads_alloc(const wn_dims *d, wn_wgrads *g)
{
    g->V = rzalloc((size_t)2 * d->Cd * d->depth * d->aux);
    g->start = rzalloc((size_t)d->C * d->in_ch);
    g->end = rzalloc((size_t)2 * d->in_ch * d->Cs);
    g->W = (real **)xmalloc(sizeof(real *) * d->depth);
    g->Wo = (real **)xmalloc(sizeof(real *) * d->depth);
    for (int i = 0; i < d->depth; ++i) {
        g->W[i] = rzalloc((size_t)2 * d->Cd * d->C * d->radix);
        g->Wo[i] = rzalloc((size_t)wo_rows(d, i) * d->Cd);
    }
    g->bV = g->bS = g->bE = NULL;
    g->bW = (real **)xmalloc(sizeof(real *) * d->depth);
    g->bWo = (real **)xmalloc(sizeof(real *) * d->depth);
    for (int i = 0; i < d->depth; ++i) g->bW[i] = g->bWo[i] = NULL;
    if (d->bias) {
        g->bV = rzalloc((size_t)2 * d->Cd * d->depth); g->bS = rzalloc((size_t)d->C); g->bE = rzalloc((size_t)2 * d->in_ch);
        for (int i = 0; i < d->depth; ++i) { g->bW[i] = rzalloc((size_t)2 * d->Cd); g->bWo[i] = rzalloc((size_t)wo_rows(d, i)); }
    }
}
static void wn_wgrads_free(const wn_dims *d, wn_wgrads *g)
{
    free(g->V); free(g->start); free(g->end);
    for (int i = 0; i < d->depth; ++i) { free(g->W[i]); free(g->Wo[i]); free(g->bW[i]); free(g->bWo[i]); }
    free(g->W); free(g->Wo); free(g->bW); free(g->bWo); free(g->bV); free(g->bS); free(g->bE);
}

static void r2f(const real *a, float *b, long n);
/* map effective-weight gradients through the weight-norm backward into the float grad table */
static void wn_wgrads_emit(const wn_dims *d, const float *const *p, const wn_wgrads *g, float *const *out)
{
    weight_norm_bwd(p[0], p[1], g->V, 2 * d->Cd * d->depth, d->aux, out[0], out[1]);
    weight_norm_bwd(p[2], p[3], g->start, d->C, d->in_ch, out[2], out[3]);
    for (int i = 0; i < d->depth; ++i) {
        weight_norm_bwd(p[4 + 4 * i], p[5 + 4 * i], g->W[i], 2 * d->Cd, d->C * d->radix, out[4 + 4 * i], out[5 + 4 * i]);
        weight_norm_bwd(p[6 + 4 * i], p[7 + 4 * i], g->Wo[i], wo_rows(d, i), d->Cd, out[6 + 4 * i], out[7 + 4 * i]);
    }
    float *e = out[4 + 4 * d->depth];
    for (long j = 0; j < (long)2 * d->in_ch * d->Cs; ++j) e[j] = (float)g->end[j];
    if (d->bias) {
        float *const *b = out + 4 + 4 * d->depth + 1;
        r2f(g->bV, b[0], (long)2 * d->Cd * d->depth);
        r2f(g->bS, b[1], d->C);
        for (int i = 0; i < d->depth; ++i) { r2f(g->bW[i], b[2 + 2 * i], 2 * d->Cd); r2f(g->bWo[i], b[3 + 2 * i], wo_rows(d, i)); }
        r2f(g->bE, b[2 + 2 * d->depth], 2 * d->in_ch);
    }
}

/* What autograd.grad(cat(log_s,t), [xa]+params(+y), grad_outputs=G) evaluates
 * (efficient_modules.py:139-144); G is [2*in_ch][T].  dxa = [in_ch][T] ; dy += [aux][T]. */
static void wn_backward(const wn_dims *d, const wn_weights *w, const wn_saved *s, const real *xa,
                        const real *y, const real *G, int T, wn_wgrads *g, real *dxa, real *dy)
{
    const long CT = (long)d->C * T, DT = (long)d->Cd * T, ST = (long)d->Cs * T;
    real *dS = rzalloc((size_t)ST);
    real *dh = rzalloc((size_t)CT);       /* gradient wrt h_{i+1}, then h_i */
    real *dout = ralloc((size_t)(CT + ST));
    real *dgate = ralloc((size_t)DT);
    real *dxy = ralloc((size_t)2 * DT);
    real *gate = ralloc((size_t)DT);
    /* end: out = W_end . S */
    wgrad_tap_acc(g->end, d->Cs, 1, G, 2 * d->in_ch, s->skip, d->Cs, T, 0);
    bias_grad(g->bE, G, 2 * d->in_ch, T);
    conv_tap_acc(dS, d->Cs, T, w->end, 1, d->Cs, G, 2 * d->in_ch, 0);  /* W_end^T . G */
    for (int i = d->depth - 1; i >= 0; --i) {
        const int dil = 1 << i;
        const int rows = wo_rows(d, i);
        const real *hi = s->h + i * CT;
        const real *tw = s->tw + i * DT, *sf = s->sf + i * DT;
        /* do = (last) ? dS : cat(dh_{i+1}, dS) */
        if (i == d->depth - 1) memcpy(dout, dS, sizeof(real) * ST);
        else { memcpy(dout, dh, sizeof(real) * CT); memcpy(dout + CT, dS, sizeof(real) * ST); }
        for (long j = 0; j < DT; ++j) gate[j] = tw[j] * sf[j];
        wgrad_tap_acc(g->Wo[i], d->Cd, 1, dout, rows, gate, d->Cd, T, 0);
        bias_grad(g->bWo[i], dout, rows, T);
        memset(dgate, 0, sizeof(real) * DT);
        conv_tap_acc(dgate, d->Cd, T, w->Wo[i], 1, d->Cd, dout, rows, 0);  /* W_o^T . do */
#pragma omp parallel for schedule(static)
        for (long j = 0; j < DT; ++j) {
            dxy[j] = dgate[j] * sf[j] * (1 - tw[j] * tw[j]);
            dxy[DT + j] = dgate[j] * tw[j] * sf[j] * (1 - sf[j]);
        }
        for (int k = 0; k < d->radix; ++k)
            wgrad_tap_acc(g->W[i] + k, (long)d->C * d->radix, d->radix, dxy, 2 * d->Cd, hi, d->C, T,
                          (k - (d->radix - 1) / 2) * dil);
        wgrad_tap_acc(g->V + (long)i * 2 * d->Cd * d->aux, d->aux, 1, dxy, 2 * d->Cd, y, d->aux, T, 0);
        bias_grad(g->bW[i], dxy, 2 * d->Cd, T);
        bias_grad(g->bV ? g->bV + (long)i * 2 * d->Cd : NULL, dxy, 2 * d->Cd, T);
        if (dy) conv_tap_acc(dy, d->aux, T, w->V + (long)i * 2 * d->Cd * d->aux, 1, d->aux, dxy, 2 * d->Cd, 0);
        /* dh_i = (i<last ? dh_{i+1} : 0) + sum_k W[:,:,k]^T dxy[t-(k-mid)d]   (residual path waveglow.py:46) */
        if (i == d->depth - 1) memset(dh, 0, sizeof(real) * CT);
        for (int k = 0; k < d->radix; ++k)
            conv_tap_acc(dh, d->C, T, w->W[i] + k, d->radix, (long)d->C * d->radix, dxy, 2 * d->Cd,
                         -(k - (d->radix - 1) / 2) * dil);
    }
    wgrad_tap_acc(g->start, d->in_ch, 1, dh, d->C, xa, d->in_ch, T, 0);
    bias_grad(g->bS, dh, d->C, T);
    memset(dxa, 0, sizeof(real) * d->in_ch * T);
    conv_tap_acc(dxa, d->in_ch, T, w->start, 1, d->in_ch, dh, d->C, 0);
    free(dS); free(dh); free(dout); free(dgate); free(dxy); free(gate);
}

/* ------------------------------------------------------------------------------------------
 * block-level entry points (float I/O) -- mirror efficient_modules.py classes
 * ---------------------------------------------------------------------------------------- */
static void f2r(const float *a, real *b, long n) { for (long i = 0; i < n; ++i) b[i] = (real)a[i]; }
static void r2f(const real *a, float *b, long n) { for (long i = 0; i < n; ++i) b[i] = (float)a[i]; }

/* InvertibleConv1x1.forward_computation (efficient_modules.py:37-41): z = W x ; logdet = T*logdet(W) */
WGO_API int wgo_invconv_forward(const float *W, int c, const float *x, int B, int T, float *z, float *logdet)
{
    if (c > WGO_MAXC) return -1;
    real Wr[WGO_MAXC * WGO_MAXC];
    f2r(W, Wr, (long)c * c);
    real *xb = ralloc((size_t)c * T), *zb = ralloc((size_t)c * T);
    for (int b = 0; b < B; ++b) {
        f2r(x + (long)b * c * T, xb, (long)c * T);
        mix_channels(Wr, c, xb, T, zb);
        r2f(zb, z + (long)b * c * T, (long)c * T);
    }
    *logdet = (float)((real)T * logdet_of(Wr, c));
    free(xb); free(zb);
    return 0;
}

/* InvertibleConv1x1.reverse_computation (efficient_modules.py:49-54): x = W^-1 z ; -T*logdet(W) */
WGO_API int wgo_invconv_reverse(const float *W, int c, const float *z, int B, int T, float *x, float *logdet)
{
    if (c > WGO_MAXC) return -1;
    real Wr[WGO_MAXC * WGO_MAXC], Wi[WGO_MAXC * WGO_MAXC], la;
    int sg;
    f2r(W, Wr, (long)c * c);
    lu_logdet_inverse(Wr, c, &la, &sg, Wi);
    real *xb = ralloc((size_t)c * T), *zb = ralloc((size_t)c * T);
    for (int b = 0; b < B; ++b) {
        f2r(z + (long)b * c * T, zb, (long)c * T);
        mix_channels(Wi, c, zb, T, xb);
        r2f(xb, x + (long)b * c * T, (long)c * T);
    }
    *logdet = (float)(-(real)T * (sg > 0 ? la : (real)NAN));
    free(xb); free(zb);
    return 0;
}

/* core of Conv1x1Func.backward (efficient_modules.py:230-244) on `real` data for one batch item:
 * rebuild x = W^-1 z, dx = W^T dz, dW += dz x^T */
static void invconv_bwd_item(const real *Wr, const real *Wi, int c, const real *z, const real *dz, int T,
                             real *x, real *dx, real *dW)
{
    real Wt[WGO_MAXC * WGO_MAXC];
    for (int i = 0; i < c; ++i) for (int j = 0; j < c; ++j) Wt[i * c + j] = Wr[j * c + i];
    mix_channels(Wi, c, z, T, x);
    mix_channels(Wt, c, dz, T, dx);
    wgrad_tap_acc(dW, c, 1, dz, c, x, c, T, 0);
}

/* Conv1x1Func.backward: given z, dz and the (scalar) grad of log_det_W, rebuild x and return dx, dW */
WGO_API int wgo_invconv_backward(const float *W, int c, const float *z, const float *dz, float dlogdet,
                                 int B, int T, float *x, float *dx, float *dW)
{
    if (c > WGO_MAXC) return -1;
    real Wr[WGO_MAXC * WGO_MAXC], Wi[WGO_MAXC * WGO_MAXC], la;
    int sg;
    f2r(W, Wr, (long)c * c);
    lu_logdet_inverse(Wr, c, &la, &sg, Wi);
    real *dWr = rzalloc((size_t)c * c);
    real *zb = ralloc((size_t)c * T), *dzb = ralloc((size_t)c * T), *xb = ralloc((size_t)c * T), *dxb = ralloc((size_t)c * T);
    for (int b = 0; b < B; ++b) {
        f2r(z + (long)b * c * T, zb, (long)c * T);
        f2r(dz + (long)b * c * T, dzb, (long)c * T);
        invconv_bwd_item(Wr, Wi, c, zb, dzb, T, xb, dxb, dWr);
        r2f(xb, x + (long)b * c * T, (long)c * T);
        r2f(dxb, dx + (long)b * c * T, (long)c * T);
    }
    for (int i = 0; i < c; ++i)
        for (int j = 0; j < c; ++j) dWr[i * c + j] += Wi[j * c + i] * (real)dlogdet * (real)T;  /* :242 */
    r2f(dWr, dW, (long)c * c);
    free(dWr); free(zb); free(dzb); free(xb); free(dxb);
    return 0;
}

/* InvConv1x1Func.backward (efficient_modules.py:262-279): the block ran x_out = W^-1 x_in with
 * log_det = -T logdet W.  Given its output `xo`, grad `dxo` and scalar grad dlogdet:
 * rebuild the input zin = W xo, dzin = W^-T dxo, dWparam = -W^-T (dxo xo^T)... per the reference. */
WGO_API int wgo_invconv_reverse_backward(const float *W, int c, const float *xo, const float *dxo, float dlogdet,
                                         int B, int T, float *zin, float *dzin, float *dW)
{
    if (c > WGO_MAXC) return -1;
    real Wr[WGO_MAXC * WGO_MAXC], Wi[WGO_MAXC * WGO_MAXC], WiT[WGO_MAXC * WGO_MAXC], la;
    int sg;
    f2r(W, Wr, (long)c * c);
    lu_logdet_inverse(Wr, c, &la, &sg, Wi);
    for (int i = 0; i < c; ++i) for (int j = 0; j < c; ++j) WiT[i * c + j] = Wi[j * c + i];
    real *dw = rzalloc((size_t)c * c);
    real *a = ralloc((size_t)c * T), *da = ralloc((size_t)c * T), *r = ralloc((size_t)c * T), *dr = ralloc((size_t)c * T);
    for (int b = 0; b < B; ++b) {
        f2r(xo + (long)b * c * T, a, (long)c * T);
        f2r(dxo + (long)b * c * T, da, (long)c * T);
        mix_channels(Wr, c, a, T, r);        /* x[:] = conv1d(z, inv_weight)   :267 (names swapped upstream) */
        mix_channels(WiT, c, da, T, dr);     /* dx = conv1d(z_grad, weight_T)  :271-273 */
        wgrad_tap_acc(dw, c, 1, da, c, r, c, T, 0);   /* dw = z_grad @ x^T  :274-275 */
        r2f(r, zin + (long)b * c * T, (long)c * T);
        r2f(dr, dzin + (long)b * c * T, (long)c * T);
    }
    /* dinvw = -W^-T dw W^-T - W^-T * dlogdet * T   (:276-277) */
    real tmp[WGO_MAXC * WGO_MAXC], res[WGO_MAXC * WGO_MAXC];
    for (int i = 0; i < c; ++i) for (int j = 0; j < c; ++j) {
        real s = 0;
        for (int k = 0; k < c; ++k) s += WiT[i * c + k] * dw[k * c + j];
        tmp[i * c + j] = s;
    }
    for (int i = 0; i < c; ++i) for (int j = 0; j < c; ++j) {
        real s = 0;
        for (int k = 0; k < c; ++k) s += tmp[i * c + k] * WiT[k * c + j];
        res[i * c + j] = -s - WiT[i * c + j] * (real)dlogdet * (real)T;
    }
    r2f(res, dW, (long)c * c);
    free(dw); free(a); free(da); free(r); free(dr);
    return 0;
}

static void wn_dims_fill(wn_dims *d, int in_ch, int aux, int C, int Cd, int Cs, int depth, int radix)
{
    d->in_ch = in_ch; d->aux = aux; d->C = C; d->Cd = Cd; d->Cs = Cs; d->depth = depth; d->radix = radix; d->bias = 0;
}

/* AffineCouplingBlock forward / reverse (efficient_modules.py:77-96); x is [B][2*in_ch][T].
 * reverse != 0 : x_out_b = (x_b - t)/exp(log_s), returns -log_s.  params: wn_nparams() entries. */
WGO_API int wgo_coupling_apply(int in_ch, int aux, int C, int Cd, int Cs, int depth, int radix,
                               const float *const *params, const float *x, const float *y,
                               int B, int T, int reverse, float *z, float *log_s_out)
{
    wn_dims d; wn_dims_fill(&d, in_ch, aux, C, Cd, Cs, depth, radix);
    wn_weights w; wn_weights_build(&d, params, &w);
    wn_saved s; wn_saved_alloc(&d, T, &s);
    const long IT = (long)in_ch * T;
    real *xa = ralloc((size_t)IT), *yb = ralloc((size_t)aux * T), *out = ralloc((size_t)2 * IT);
    for (int b = 0; b < B; ++b) {
        const float *xb_f = x + (long)b * 2 * IT;
        f2r(xb_f, xa, IT);
        f2r(y + (long)b * aux * T, yb, (long)aux * T);
        wn_forward(&d, &w, xa, yb, T, &s, out);
        float *zb_f = z + (long)b * 2 * IT;
        memcpy(zb_f, xb_f, sizeof(float) * IT);                       /* za = xa */
        for (long j = 0; j < IT; ++j) {
            const real ls = out[j], tt = out[IT + j], xb = (real)xb_f[IT + j];
            if (!reverse) {
                zb_f[IT + j] = (float)(xb * (real)exp((double)ls) + tt);   /* :81 */
                log_s_out[(long)b * IT + j] = (float)ls;
            } else {
                zb_f[IT + j] = (float)((xb - tt) / (real)exp((double)ls)); /* :94 */
                log_s_out[(long)b * IT + j] = (float)(-ls);
            }
        }
    }
    free(xa); free(yb); free(out);
    wn_saved_free(&s); wn_weights_free(&d, &w);
    return 0;
}

/* AffineCouplingFunc.backward (efficient_modules.py:118-154).  Inputs: block output z, y, dz, dlog_s.
 * Outputs: rebuilt x, dx, dy (nullable), grads table (wn_nparams entries, same shapes as params). */
WGO_API int wgo_coupling_backward(int in_ch, int aux, int C, int Cd, int Cs, int depth, int radix,
                                  const float *const *params, const float *z, const float *y,
                                  const float *dz, const float *dlog_s, int B, int T,
                                  float *x, float *dx, float *dy, float *const *grads)
{
    wn_dims d; wn_dims_fill(&d, in_ch, aux, C, Cd, Cs, depth, radix);
    wn_weights w; wn_weights_build(&d, params, &w);
    wn_wgrads g; wn_wgrads_alloc(&d, &g);
    wn_saved s; wn_saved_alloc(&d, T, &s);
    const long IT = (long)in_ch * T;
    real *xa = ralloc((size_t)IT), *yb = ralloc((size_t)aux * T), *out = ralloc((size_t)2 * IT);
    real *G = ralloc((size_t)2 * IT), *dxa = ralloc((size_t)IT), *dyb = dy ? ralloc((size_t)aux * T) : NULL;
    for (int b = 0; b < B; ++b) {
        const float *zf = z + (long)b * 2 * IT, *dzf = dz + (long)b * 2 * IT;
        float *xf = x + (long)b * 2 * IT, *dxf = dx + (long)b * 2 * IT;
        f2r(zf, xa, IT);
        f2r(y + (long)b * aux * T, yb, (long)aux * T);
        wn_forward(&d, &w, xa, yb, T, &s, out);                        /* :127-130 */
        memcpy(xf, zf, sizeof(float) * IT);
        for (long j = 0; j < IT; ++j) {
            const real ls = out[j], tt = out[IT + j];
            const real sc = (real)exp((double)ls);
            const real xb = ((real)zf[IT + j] - tt) / sc;              /* :133-134 */
            const real dzb = (real)dzf[IT + j];
            xf[IT + j] = (float)xb;
            G[j] = dzb * xb * sc + (real)dlog_s[(long)b * IT + j];     /* :143-144 grad_outputs */
            G[IT + j] = dzb;
            dxf[IT + j] = (float)(dzb * sc);                           /* :147 */
        }
        if (dyb) memset(dyb, 0, sizeof(real) * aux * T);
        wn_backward(&d, &w, &s, xa, yb, G, T, &g, dxa, dyb);
        for (long j = 0; j < IT; ++j) dxf[j] = (float)((real)dzf[j] + dxa[j]);   /* :146 */
        if (dyb) r2f(dyb, dy + (long)b * aux * T, (long)aux * T);
    }
    wn_wgrads_emit(&d, params, &g, grads);
    free(xa); free(yb); free(out); free(G); free(dxa); free(dyb);
    wn_saved_free(&s); wn_wgrads_free(&d, &g); wn_weights_free(&d, &w);
    return 0;
}

/* InvAffineCouplingFunc.backward (efficient_modules.py:175-212).  The block computed
 * xo_b = (zi_b - t)/s and returned nls = -log_s.  Inputs: block output xo, y, dxo, dnls.
 * Outputs: rebuilt block input zi, its gradient dzi, dy (nullable), parameter grads. */
WGO_API int wgo_coupling_reverse_backward(int in_ch, int aux, int C, int Cd, int Cs, int depth, int radix,
                                          const float *const *params, const float *xo, const float *y,
                                          const float *dxo, const float *dnls, int B, int T,
                                          float *zi, float *dzi, float *dy, float *const *grads)
{
    wn_dims d; wn_dims_fill(&d, in_ch, aux, C, Cd, Cs, depth, radix);
    wn_weights w; wn_weights_build(&d, params, &w);
    wn_wgrads g; wn_wgrads_alloc(&d, &g);
    wn_saved s; wn_saved_alloc(&d, T, &s);
    const long IT = (long)in_ch * T;
    real *xa = ralloc((size_t)IT), *yb = ralloc((size_t)aux * T), *out = ralloc((size_t)2 * IT);
    real *G = ralloc((size_t)2 * IT), *dza = ralloc((size_t)IT), *dyb = dy ? ralloc((size_t)aux * T) : NULL;
    for (int b = 0; b < B; ++b) {
        const float *xf = xo + (long)b * 2 * IT, *dxf = dxo + (long)b * 2 * IT;
        float *zf = zi + (long)b * 2 * IT, *dzf = dzi + (long)b * 2 * IT;
        f2r(xf, xa, IT);
        f2r(y + (long)b * aux * T, yb, (long)aux * T);
        wn_forward(&d, &w, xa, yb, T, &s, out);                        /* :187-191 */
        memcpy(zf, xf, sizeof(float) * IT);
        for (long j = 0; j < IT; ++j) {
            const real ls = out[j], tt = out[IT + j];
            const real sc = (real)exp((double)ls);
            const real xb = (real)xf[IT + j], dxb = (real)dxf[IT + j];
            const real zb = xb * sc + tt;                              /* :194 */
            zf[IT + j] = (float)zb;
            /* grad(cat(-log_s, -t/s), ..., grad_outputs=cat(dxb*zb/s + dnls, dxb))   :202-203
             *   d/dlog_s = -(dxb*zb/s + dnls) + dxb*t/s ;  d/dt = -dxb/s                        */
            const real go = dxb * zb / sc + (real)dnls[(long)b * IT + j];
            G[j] = -go + dxb * tt / sc;
            G[IT + j] = -dxb / sc;
            dzf[IT + j] = (float)(dxb / sc);                           /* :206 */
        }
        if (dyb) memset(dyb, 0, sizeof(real) * aux * T);
        wn_backward(&d, &w, &s, xa, yb, G, T, &g, dza, dyb);
        for (long j = 0; j < IT; ++j) dzf[j] = (float)((real)dxf[j] + dza[j]);   /* :205 */
        if (dyb) r2f(dyb, dy + (long)b * aux * T, (long)aux * T);
    }
    wn_wgrads_emit(&d, params, &g, grads);
    free(xa); free(yb); free(out); free(G); free(dza); free(dyb);
    wn_saved_free(&s); wn_wgrads_free(&d, &g); wn_weights_free(&d, &w);
    return 0;
}

/* ------------------------------------------------------------------------------------------
 * model level   (waveglow.py:108-212, loss.py:10-15)
 * parameter table order == named_parameters() of the reference model:
 *   upsampler.bias, upsampler.weight_g, upsampler.weight_v,
 *   invconv1x1.{k}.weight (k = 0..flows-1),
 *   for k: WNs.{k}.F.{V.g, V.v, start.g, start.v, layers.{i}.{W.g, W.v, W_o.g, W_o.v}, end.weight}
 * a NULL `*_g` entry means the conv carries a plain weight in the `*_v` slot.
 * ---------------------------------------------------------------------------------------- */
WGO_API int wgo_param_count(const wgo_config *cf) { return 3 + cf->n_flows + cf->n_flows * wn_entries(cf->depth, cf->bias); }

static int flow_channels(const wgo_config *cf, int k)
{
    int c = cf->n_group;
    for (int j = 1; j <= k; ++j)
        if (j % cf->n_early_every == 0) c -= cf->n_early_size;          /* waveglow.py:140-142 */
    return c;
}
static int wn_table_off(const wgo_config *cf, int k) { return 3 + cf->n_flows + k * wn_entries(cf->depth, cf->bias); }

typedef struct {
    const wgo_config *cf;
    real *up_w;                 /* [n_mels][K] */
    real **W, **Wi;             /* per flow c x c */
    real *logdetW;              /* per flow, NaN if det<0 */
    wn_dims *d;
    wn_weights *w;
} model_weights;

static void model_weights_build(const wgo_config *cf, const float *const *p, model_weights *m)
{
    m->cf = cf;
    m->up_w = ralloc((size_t)cf->n_mels * cf->up_kernel);
    weight_norm_fwd(p[1], p[2], cf->n_mels, cf->up_kernel, m->up_w);
    m->W = (real **)xmalloc(sizeof(real *) * cf->n_flows);
    m->Wi = (real **)xmalloc(sizeof(real *) * cf->n_flows);
    m->logdetW = ralloc(cf->n_flows);
    m->d = (wn_dims *)xmalloc(sizeof(wn_dims) * cf->n_flows);
    m->w = (wn_weights *)xmalloc(sizeof(wn_weights) * cf->n_flows);
    for (int k = 0; k < cf->n_flows; ++k) {
        const int c = flow_channels(cf, k);
        m->W[k] = ralloc((size_t)c * c);
        m->Wi[k] = ralloc((size_t)c * c);
        f2r(p[3 + k], m->W[k], (long)c * c);
        real la; int sg;
        lu_logdet_inverse(m->W[k], c, &la, &sg, m->Wi[k]);
        m->logdetW[k] = sg > 0 ? la : (real)NAN;
        wn_dims_fill(&m->d[k], c / 2, cf->n_mels, cf->res_ch, cf->dil_ch, cf->skip_ch, cf->depth, cf->radix);
        m->d[k].bias = cf->bias;
        wn_weights_build(&m->d[k], p + wn_table_off(cf, k), &m->w[k]);
    }
}
static void model_weights_free(model_weights *m)
{
    for (int k = 0; k < m->cf->n_flows; ++k) {
        free(m->W[k]); free(m->Wi[k]);
        wn_weights_free(&m->d[k], &m->w[k]);
    }
    free(m->W); free(m->Wi); free(m->logdetW); free(m->d); free(m->w); free(m->up_w);
}

/* WaveGlow.forward_computation (waveglow.py:150-179) for batch item b.
 * Zs is [n_group][T] "cat(outputs)" layout: early outputs first.  Returns logdet contribution. */
static real model_forward_item(const model_weights *m, const float *audio, const real *y, int T, real *Zs)
{
    const wgo_config *cf = m->cf;
    const int G = cf->n_group;
    real *x = ralloc((size_t)G * T), *tmp = ralloc((size_t)G * T), *out = ralloc((size_t)G * T);
    for (int g = 0; g < G; ++g)
        for (int t = 0; t < T; ++t) x[(long)g * T + t] = (real)audio[(long)t * G + g];   /* squeeze :153 */
    real logdet = 0;
    int base = 0;                       /* channels already emitted */
    real *cur = x;                      /* [c][T] view */
    wn_saved s; wn_saved_alloc(&m->d[0], T, &s);
    for (int k = 0; k < cf->n_flows; ++k) {
        const int c = flow_channels(cf, k);
        if (k % cf->n_early_every == 0 && k) {                                           /* :164-170 */
            memcpy(Zs + (long)base * T, cur, sizeof(real) * cf->n_early_size * T);
            base += cf->n_early_size;
            cur += (long)cf->n_early_size * T;
        }
        mix_channels(m->W[k], c, cur, T, tmp);                                           /* :172 */
        const int ic = c / 2;
        wn_forward(&m->d[k], &m->w[k], tmp, y, T, &s, out);                              /* :173 */
        real ls_sum = 0;
        for (long j = 0; j < (long)ic * T; ++j) {
            const real ls = out[j];
            tmp[(long)ic * T + j] = tmp[(long)ic * T + j] * (real)exp((double)ls) + out[(long)ic * T + j];
            ls_sum += ls;
        }
        logdet += (real)T * m->logdetW[k] + ls_sum;                                      /* :175 */
        memcpy(cur, tmp, sizeof(real) * c * T);
    }
    memcpy(Zs + (long)base * T, cur, sizeof(real) * (G - base) * T);                     /* :178 */
    wn_saved_free(&s);
    free(x); free(tmp); free(out);
    return logdet;
}

static void upsample_item(const model_weights *m, const float *const *p, const float *h, int F, int T, real *y)
{
    upsample_fwd(m->cf, m->up_w, p[0], h, F, T, y);
}

static int check_dims(const wgo_config *cf, int N, int F, int *T)
{
    if (N % cf->n_group) return -2;
    *T = N / cf->n_group;
    const int L = (F - 1) * cf->up_stride - 2 * cf->up_pad + cf->up_kernel;
    if (*T > L) return -3;                                       /* assert x.size(2) <= y.size(2)  :156 */
    if (cf->n_group > WGO_MAXC) return -1;
    return 0;
}

WGO_API int wgo_forward(const wgo_config *cf, const float *const *params, const float *audio, const float *h,
                        int B, int N, int F, float *z, float *logdet)
{
    int T, rc = check_dims(cf, N, F, &T);
    if (rc) return rc;
    model_weights m; model_weights_build(cf, params, &m);
    real *y = ralloc((size_t)cf->n_mels * T), *Zs = ralloc((size_t)cf->n_group * T);
    for (int b = 0; b < B; ++b) {
        upsample_item(&m, params, h + (long)b * cf->n_mels * F, F, T, y);
        const real ld = model_forward_item(&m, audio + (long)b * N, y, T, Zs);
        for (int g = 0; g < cf->n_group; ++g)
            for (int t = 0; t < T; ++t) z[(long)b * N + (long)t * cf->n_group + g] = (float)Zs[(long)g * T + t]; /* :179 */
        logdet[b] = (float)ld;
    }
    free(y); free(Zs);
    model_weights_free(&m);
    return 0;
}

/* WaveGlow.reverse_computation (waveglow.py:181-208) */
WGO_API int wgo_inverse(const wgo_config *cf, const float *const *params, const float *z, const float *h,
                        int B, int N, int F, float *x, float *logdet)
{
    int T, rc = check_dims(cf, N, F, &T);
    if (rc) return rc;
    const int G = cf->n_group;
    model_weights m; model_weights_build(cf, params, &m);
    real *y = ralloc((size_t)cf->n_mels * T), *Zs = ralloc((size_t)G * T);
    real *tmp = ralloc((size_t)G * T), *out = ralloc((size_t)G * T);
    wn_saved s; wn_saved_alloc(&m.d[0], T, &s);
    for (int b = 0; b < B; ++b) {
        upsample_item(&m, params, h + (long)b * cf->n_mels * F, F, T, y);
        for (int g = 0; g < G; ++g)
            for (int t = 0; t < T; ++t) Zs[(long)g * T + t] = (real)z[(long)b * N + (long)t * G + g];
        real ld = 0;
        int base = G - flow_channels(cf, cf->n_flows - 1);
        for (int k = cf->n_flows - 1; k >= 0; --k) {
            const int c = flow_channels(cf, k), ic = c / 2;
            real *cur = Zs + (long)base * T;
            wn_forward(&m.d[k], &m.w[k], cur, y, T, &s, out);                            /* :199 */
            real ls_sum = 0;
            for (long j = 0; j < (long)ic * T; ++j) {
                const real ls = out[j];
                cur[(long)ic * T + j] = (cur[(long)ic * T + j] - out[(long)ic * T + j]) / (real)exp((double)ls);
                ls_sum -= ls;
            }
            mix_channels(m.Wi[k], c, cur, T, tmp);                                       /* :200 */
            memcpy(cur, tmp, sizeof(real) * c * T);
            ld += -(real)T * m.logdetW[k] + ls_sum;                                      /* :202 */
            if (k % cf->n_early_every == 0 && k) base -= cf->n_early_size;               /* :204-205 */
        }
        for (int g = 0; g < G; ++g)
            for (int t = 0; t < T; ++t) x[(long)b * N + (long)t * G + g] = (float)Zs[(long)g * T + t];
        logdet[b] = (float)ld;
    }
    wn_saved_free(&s);
    free(y); free(Zs); free(tmp); free(out);
    model_weights_free(&m);
    return 0;
}

/* WaveGlowLoss.forward (loss.py:10-15), elementwise_mean=True */
WGO_API int wgo_loss(const float *z, const float *logdet, int B, int N, float sigma, float *loss)
{
    real acc = 0;
    for (int b = 0; b < B; ++b) {
        real ss = 0;
        for (int n = 0; n < N; ++n) ss += (real)z[(long)b * N + n] * (real)z[(long)b * N + n];
        acc += (real)0.5 * ss / ((real)sigma * (real)sigma) - (real)logdet[b];
    }
    *loss = (float)(acc / (real)B / (real)N);
    return 0;
}

/* One training step as model/lightning.py:52-56 drives it: z,logdet = model(x,h); loss = NLL(z,logdet);
 * loss.backward().  Returns z, logdet, loss and the gradient of every parameter (table order), plus
 * dh (nullable).  The backward pass walks the flows last to first and rebuilds every block input
 * from the block output (constant-memory protocol). */
WGO_API int wgo_train_step(const wgo_config *cf, const float *const *params, const float *audio,
                           const float *h, int B, int N, int F, float sigma,
                           float *z, float *logdet, float *loss, float *const *grads, float *dh)
{
    int T, rc = check_dims(cf, N, F, &T);
    if (rc) return rc;
    const int G = cf->n_group, nf = cf->n_flows;
    rc = wgo_forward(cf, params, audio, h, B, N, F, z, logdet);
    if (rc) return rc;
    wgo_loss(z, logdet, B, N, sigma, loss);

    model_weights m; model_weights_build(cf, params, &m);
    wn_wgrads *wg = (wn_wgrads *)xmalloc(sizeof(wn_wgrads) * nf);
    real **dW = (real **)xmalloc(sizeof(real *) * nf);
    for (int k = 0; k < nf; ++k) {
        wn_wgrads_alloc(&m.d[k], &wg[k]);
        dW[k] = rzalloc((size_t)G * G);
    }
    real *dup_w = rzalloc((size_t)cf->n_mels * cf->up_kernel), *dbias = rzalloc(cf->n_mels);
    real *y = ralloc((size_t)cf->n_mels * T), *dy = ralloc((size_t)cf->n_mels * T);
    real *Zs = ralloc((size_t)G * T), *dZ = ralloc((size_t)G * T);
    real *out = ralloc((size_t)G * T), *Gr = ralloc((size_t)G * T), *dxa = ralloc((size_t)G * T);
    real *xr = ralloc((size_t)G * T), *dxr = ralloc((size_t)G * T);
    real *dhb = dh ? ralloc((size_t)cf->n_mels * F) : NULL;
    wn_saved s; wn_saved_alloc(&m.d[0], T, &s);
    const real inv_bn = (real)1 / ((real)B * (real)N);
    const real dld = -inv_bn;                         /* d loss / d logdet[b]    (loss.py:11-14) */
    const real dz_scale = inv_bn / ((real)sigma * (real)sigma);

    for (int b = 0; b < B; ++b) {
        upsample_item(&m, params, h + (long)b * cf->n_mels * F, F, T, y);
        memset(dy, 0, sizeof(real) * cf->n_mels * T);
        for (int g = 0; g < G; ++g)
            for (int t = 0; t < T; ++t) {
                const real zv = (real)z[(long)b * N + (long)t * G + g];
                Zs[(long)g * T + t] = zv;
                dZ[(long)g * T + t] = zv * dz_scale;
            }
        int base = G - flow_channels(cf, nf - 1);
        for (int k = nf - 1; k >= 0; --k) {
            const int c = flow_channels(cf, k), ic = c / 2;
            const long IT = (long)ic * T;
            real *cur = Zs + (long)base * T, *dcur = dZ + (long)base * T;
            /* coupling backward (AffineCouplingFunc.backward) */
            wn_forward(&m.d[k], &m.w[k], cur, y, T, &s, out);
            for (long j = 0; j < IT; ++j) {
                const real sc = (real)exp((double)out[j]);
                const real xb = (cur[IT + j] - out[IT + j]) / sc;
                const real dzb = dcur[IT + j];
                cur[IT + j] = xb;
                Gr[j] = dzb * xb * sc + dld;          /* log_s.sum((1,2)) feeds logdet[b]  waveglow.py:175 */
                Gr[IT + j] = dzb;
                dcur[IT + j] = dzb * sc;
            }
            wn_backward(&m.d[k], &m.w[k], &s, cur, y, Gr, T, &wg[k], dxa, dy);
            for (long j = 0; j < IT; ++j) dcur[j] += dxa[j];
            /* invertible 1x1 backward (Conv1x1Func.backward); the scalar log_det_W is broadcast over
             * the batch (waveglow.py:175), so its gradient is sum_b dld -- added once, after the loop */
            invconv_bwd_item(m.W[k], m.Wi[k], c, cur, dcur, T, xr, dxr, dW[k]);
            memcpy(cur, xr, sizeof(real) * c * T);
            memcpy(dcur, dxr, sizeof(real) * c * T);
            if (k % cf->n_early_every == 0 && k) base -= cf->n_early_size;
        }
        upsample_bwd(cf, m.up_w, h + (long)b * cf->n_mels * F, F, T, dy, dup_w, dbias, dhb);
        if (dh) r2f(dhb, dh + (long)b * cf->n_mels * F, (long)cf->n_mels * F);
    }
    /* emit */
    for (int c = 0; c < cf->n_mels; ++c) grads[0][c] = (float)dbias[c];
    weight_norm_bwd(params[1], params[2], dup_w, cf->n_mels, cf->up_kernel, grads[1], grads[2]);
    for (int k = 0; k < nf; ++k) {
        const int c = flow_channels(cf, k);
        const real gl = dld * (real)B * (real)T;      /* sum_b dld, times n_of_groups  (efficient_modules.py:242) */
        for (int i = 0; i < c; ++i)
            for (int j = 0; j < c; ++j) grads[3 + k][i * c + j] = (float)(dW[k][i * c + j] + m.Wi[k][j * c + i] * gl);
        wn_wgrads_emit(&m.d[k], params + wn_table_off(cf, k), &wg[k], grads + wn_table_off(cf, k));
        wn_wgrads_free(&m.d[k], &wg[k]);
        free(dW[k]);
    }
    wn_saved_free(&s);
    free(wg); free(dW); free(dup_w); free(dbias); free(y); free(dy); free(Zs); free(dZ);
    free(out); free(Gr); free(dxa); free(xr); free(dxr); free(dhb);
    model_weights_free(&m);
    return 0;
}

/* ------------------------------------------------------------------------------------------
 * reverse_mode=True architecture (model/base.py:20-28 double swap): model.forward runs the
 * loop of waveglow.py:181-208 while every block's .reverse() resolves to its FORWARD formulas
 * (base.py:25-28).  Per flow k = n-1..0 on the last c_k channels:  coupling forward, then z = W z;
 * early channels are re-attached in front when k % n_early_every == 0.
 * model.reverse / infer runs waveglow.py:150-179 with the blocks' inverse formulas: per flow k = 0..n-1
 * (after the early split) x = W^-1 z, then the coupling inverse.
 * ---------------------------------------------------------------------------------------- */
static real model_forward_item_rm(const model_weights *m, const float *audio, const real *y, int T, real *Zs)
{
    const wgo_config *cf = m->cf;
    const int G = cf->n_group;
    real *tmp = ralloc((size_t)G * T), *out = ralloc((size_t)G * T);
    for (int g = 0; g < G; ++g)
        for (int t = 0; t < T; ++t) Zs[(long)g * T + t] = (real)audio[(long)t * G + g];   /* :183 */
    real logdet = 0;
    int base = G - flow_channels(cf, cf->n_flows - 1);                                    /* split :190-194 */
    wn_saved s; wn_saved_alloc(&m->d[0], T, &s);
    for (int k = cf->n_flows - 1; k >= 0; --k) {
        const int c = flow_channels(cf, k), ic = c / 2;
        real *cur = Zs + (long)base * T;
        wn_forward(&m->d[k], &m->w[k], cur, y, T, &s, out);                               /* :199 -> block forward formulas */
        real ls_sum = 0;
        for (long j = 0; j < (long)ic * T; ++j) {
            cur[(long)ic * T + j] = cur[(long)ic * T + j] * (real)exp((double)out[j]) + out[(long)ic * T + j];
            ls_sum += out[j];
        }
        mix_channels(m->W[k], c, cur, T, tmp);                                            /* :200 */
        memcpy(cur, tmp, sizeof(real) * c * T);
        logdet += (real)T * m->logdetW[k] + ls_sum;                                       /* :202 */
        if (k % cf->n_early_every == 0 && k) base -= cf->n_early_size;                    /* :204-205 */
    }
    wn_saved_free(&s);
    free(tmp); free(out);
    return logdet;
}

WGO_API int wgo_forward_rm(const wgo_config *cf, const float *const *params, const float *audio, const float *h,
                           int B, int N, int F, float *z, float *logdet)
{
    int T, rc = check_dims(cf, N, F, &T);
    if (rc) return rc;
    model_weights m; model_weights_build(cf, params, &m);
    real *y = ralloc((size_t)cf->n_mels * T), *Zs = ralloc((size_t)cf->n_group * T);
    for (int b = 0; b < B; ++b) {
        upsample_item(&m, params, h + (long)b * cf->n_mels * F, F, T, y);
        const real ld = model_forward_item_rm(&m, audio + (long)b * N, y, T, Zs);
        for (int g = 0; g < cf->n_group; ++g)
            for (int t = 0; t < T; ++t) z[(long)b * N + (long)t * cf->n_group + g] = (float)Zs[(long)g * T + t];
        logdet[b] = (float)ld;
    }
    free(y); free(Zs);
    model_weights_free(&m);
    return 0;
}

WGO_API int wgo_inverse_rm(const wgo_config *cf, const float *const *params, const float *z, const float *h,
                           int B, int N, int F, float *x, float *logdet)
{
    int T, rc = check_dims(cf, N, F, &T);
    if (rc) return rc;
    const int G = cf->n_group;
    model_weights m; model_weights_build(cf, params, &m);
    real *y = ralloc((size_t)cf->n_mels * T), *Zs = ralloc((size_t)G * T);
    real *tmp = ralloc((size_t)G * T), *out = ralloc((size_t)G * T);
    wn_saved s; wn_saved_alloc(&m.d[0], T, &s);
    for (int b = 0; b < B; ++b) {
        upsample_item(&m, params, h + (long)b * cf->n_mels * F, F, T, y);
        for (int g = 0; g < G; ++g)
            for (int t = 0; t < T; ++t) Zs[(long)g * T + t] = (real)z[(long)b * N + (long)t * G + g];
        real ld = 0;
        int base = 0;
        for (int k = 0; k < cf->n_flows; ++k) {
            const int c = flow_channels(cf, k), ic = c / 2;
            if (k % cf->n_early_every == 0 && k) base += cf->n_early_size;                /* waveglow.py:164-170 */
            real *cur = Zs + (long)base * T;
            mix_channels(m.Wi[k], c, cur, T, tmp);                                        /* :172 -> block reverse formulas */
            memcpy(cur, tmp, sizeof(real) * c * T);
            wn_forward(&m.d[k], &m.w[k], cur, y, T, &s, out);                             /* :173 */
            real ls_sum = 0;
            for (long j = 0; j < (long)ic * T; ++j) {
                cur[(long)ic * T + j] = (cur[(long)ic * T + j] - out[(long)ic * T + j]) / (real)exp((double)out[j]);
                ls_sum -= out[j];
            }
            ld += -(real)T * m.logdetW[k] + ls_sum;
        }
        for (int g = 0; g < G; ++g)
            for (int t = 0; t < T; ++t) x[(long)b * N + (long)t * G + g] = (float)Zs[(long)g * T + t];
        logdet[b] = (float)ld;
    }
    wn_saved_free(&s);
    free(y); free(Zs); free(tmp); free(out);
    model_weights_free(&m);
    return 0;
}

/* training step of the reverse_mode architecture: forward above, NLL, constant-memory backward that walks the flows in
 * the opposite order (k = 0..n-1): 1x1 backward first (rebuild u = W^-1 w), then the coupling backward. */
WGO_API int wgo_train_step_rm(const wgo_config *cf, const float *const *params, const float *audio,
                              const float *h, int B, int N, int F, float sigma,
                              float *z, float *logdet, float *loss, float *const *grads, float *dh)
{
    int T, rc = check_dims(cf, N, F, &T);
    if (rc) return rc;
    const int G = cf->n_group, nf = cf->n_flows;
    rc = wgo_forward_rm(cf, params, audio, h, B, N, F, z, logdet);
    if (rc) return rc;
    wgo_loss(z, logdet, B, N, sigma, loss);
    model_weights m; model_weights_build(cf, params, &m);
    wn_wgrads *wg = (wn_wgrads *)xmalloc(sizeof(wn_wgrads) * nf);
    real **dW = (real **)xmalloc(sizeof(real *) * nf);
    for (int k = 0; k < nf; ++k) { wn_wgrads_alloc(&m.d[k], &wg[k]); dW[k] = rzalloc((size_t)G * G); }
    real *dup_w = rzalloc((size_t)cf->n_mels * cf->up_kernel), *dbias = rzalloc(cf->n_mels);
    real *y = ralloc((size_t)cf->n_mels * T), *dy = ralloc((size_t)cf->n_mels * T);
    real *Zs = ralloc((size_t)G * T), *dZ = ralloc((size_t)G * T);
    real *out = ralloc((size_t)G * T), *Gr = ralloc((size_t)G * T), *dxa = ralloc((size_t)G * T);
    real *xr = ralloc((size_t)G * T), *dxr = ralloc((size_t)G * T);
    real *dhb = dh ? ralloc((size_t)cf->n_mels * F) : NULL;
    wn_saved s; wn_saved_alloc(&m.d[0], T, &s);
    const real inv_bn = (real)1 / ((real)B * (real)N);
    const real dld = -inv_bn, dz_scale = inv_bn / ((real)sigma * (real)sigma);
    for (int b = 0; b < B; ++b) {
        upsample_item(&m, params, h + (long)b * cf->n_mels * F, F, T, y);
        memset(dy, 0, sizeof(real) * cf->n_mels * T);
        for (int g = 0; g < G; ++g)
            for (int t = 0; t < T; ++t) {
                const real zv = (real)z[(long)b * N + (long)t * G + g];
                Zs[(long)g * T + t] = zv;
                dZ[(long)g * T + t] = zv * dz_scale;
            }
        int base = 0;
        for (int k = 0; k < nf; ++k) {
            const int c = flow_channels(cf, k), ic = c / 2;
            const long IT = (long)ic * T;
            if (k % cf->n_early_every == 0 && k) base += cf->n_early_size;
            real *cur = Zs + (long)base * T, *dcur = dZ + (long)base * T;
            invconv_bwd_item(m.W[k], m.Wi[k], c, cur, dcur, T, xr, dxr, dW[k]);
            memcpy(cur, xr, sizeof(real) * c * T);
            memcpy(dcur, dxr, sizeof(real) * c * T);
            wn_forward(&m.d[k], &m.w[k], cur, y, T, &s, out);
            for (long j = 0; j < IT; ++j) {
                const real sc = (real)exp((double)out[j]);
                const real xb = (cur[IT + j] - out[IT + j]) / sc;
                const real dzb = dcur[IT + j];
                cur[IT + j] = xb;
                Gr[j] = dzb * xb * sc + dld;
                Gr[IT + j] = dzb;
                dcur[IT + j] = dzb * sc;
            }
            wn_backward(&m.d[k], &m.w[k], &s, cur, y, Gr, T, &wg[k], dxa, dy);
            for (long j = 0; j < IT; ++j) dcur[j] += dxa[j];
        }
        upsample_bwd(cf, m.up_w, h + (long)b * cf->n_mels * F, F, T, dy, dup_w, dbias, dhb);
        if (dh) r2f(dhb, dh + (long)b * cf->n_mels * F, (long)cf->n_mels * F);
    }
    for (int c = 0; c < cf->n_mels; ++c) grads[0][c] = (float)dbias[c];
    weight_norm_bwd(params[1], params[2], dup_w, cf->n_mels, cf->up_kernel, grads[1], grads[2]);
    for (int k = 0; k < nf; ++k) {
        const int c = flow_channels(cf, k);
        const real gl = dld * (real)B * (real)T;
        for (int i = 0; i < c; ++i)
            for (int j = 0; j < c; ++j) grads[3 + k][i * c + j] = (float)(dW[k][i * c + j] + m.Wi[k][j * c + i] * gl);
        wn_wgrads_emit(&m.d[k], params + wn_table_off(cf, k), &wg[k], grads + wn_table_off(cf, k));
        wn_wgrads_free(&m.d[k], &wg[k]);
        free(dW[k]);
    }
    wn_saved_free(&s);
    free(wg); free(dW); free(dup_w); free(dbias); free(y); free(dy); free(Zs); free(dZ);
    free(out); free(Gr); free(dxa); free(xr); free(dxr); free(dhb);
    model_weights_free(&m);
    return 0;
}

/* mel upsampler alone (for kernel-level parity tests): y = crop(upsampler(h), T) */
WGO_API int wgo_upsample(const wgo_config *cf, const float *bias, const float *g, const float *v,
                         const float *h, int B, int F, int T, float *y)
{
    real *w = ralloc((size_t)cf->n_mels * cf->up_kernel), *yb = ralloc((size_t)cf->n_mels * T);
    weight_norm_fwd(g, v, cf->n_mels, cf->up_kernel, w);
    for (int b = 0; b < B; ++b) {
        upsample_fwd(cf, w, bias, h + (long)b * cf->n_mels * F, F, T, yb);
        r2f(yb, y + (long)b * cf->n_mels * T, (long)cf->n_mels * T);
    }
    free(w); free(yb);
    return 0;
}

/* ------------------------------------------------------------------------------------------------
 * WSRGlow conditioning front-end: WSRGlow._get_cond, model/wsrglow.py:37-50.
 *   c      = clip(c, -1, 1)                                                              wsrglow.py:38
 *   c_emb  = Embedding(256,400)(MuLawEncoding(256)(c)).view(B,-1,3200).transpose(1,2)     wsrglow.py:27-30,39
 *   spec   = stft(reflect_pad(c,(4,4)), n_fft=16, hop=8, hann(16), center=False)          wsrglow.py:40-46
 *   mag    = |spec| ; phase_emb = AngleEmbedding(120,50)(angle(spec)) as [B, 9*50, F]     wsrglow.py:8-18,47-49
 *   cond   = cat([c_emb, mag, phase_emb], 1)        -> [B, 3200 + 9 + 450 = 3659, F = L/8] wsrglow.py:50
 * MuLawEncoding lives in torchaudio (absent from /root/reference and from this image; the reference pins no
 * version).  Its published algorithm (torchaudio.functional.mu_law_encoding, unchanged since 0.8):
 *   mu = 255 ; x_mu = sign(x) * log1p(mu*|x|) / log1p(mu) ; q = int64((x_mu + 1) / 2 * mu + 0.5)   (truncation)
 * AngleEmbedding: index = int64((angle / pi + 1) * 0.5 * (embed_num - 1))                  wsrglow.py:16-17
 * The two index computations are done in float32 exactly as torch does them, whatever WGO_REAL is: they are
 * quantisers, and a one-ulp difference at a bin edge selects another embedding row.
 * Window: torch.hann_window(16) is periodic: w[n] = 0.5 - 0.5 cos(2 pi n / 16)             wsrglow.py:35
 * ------------------------------------------------------------------------------------------------ */
#define WSR_MU 256
#define WSR_MU_DIM 400
#define WSR_NFFT 16
#define WSR_HOP 8
#define WSR_BINS 9
#define WSR_ANG 120
#define WSR_ANG_DIM 50
#define WSR_COND (8 * WSR_MU_DIM + WSR_BINS * (1 + WSR_ANG_DIM))

static inline float clipf(float x) { return x < -1.f ? -1.f : (x > 1.f ? 1.f : x); }
static inline int wsr_mu_index(float x)
{
    const float mu = 255.f;
    const float sgn = (x > 0.f) - (x < 0.f);
    const float x_mu = sgn * log1pf(mu * fabsf(x)) / log1pf(mu);
    return (int)((x_mu + 1.f) / 2.f * mu + 0.5f);
}
static inline int wsr_angle_index(float ang)
{
    return (int)((ang / 3.14159274101257324f + 1.f) * 0.5f * (float)(WSR_ANG - 1));
}
/* padded signal of F.pad(c, (4,4), 'reflect'): position i in [0, L+8) */
static inline float wsr_padded(const float *c, int L, int i)
{
    int j = i - 4;
    if (j < 0) j = -j;
    if (j >= L) j = 2 * (L - 1) - j;
    return clipf(c[j]);
}
/* one frame of the 16-point STFT: re/im of bins 0..8 (real arithmetic), then the float32 magnitude / angle torch would hold */
static void wsr_frame(const float *c, int L, int f, float *mag, float *ang)
{
    real x[WSR_NFFT];
    for (int n = 0; n < WSR_NFFT; ++n) {
        const real w = (real)0.5 - (real)0.5 * (real)cos(2.0 * M_PI * n / WSR_NFFT);
        x[n] = w * (real)wsr_padded(c, L, WSR_HOP * f + n);
    }
    for (int k = 0; k < WSR_BINS; ++k) {
        real re = 0, im = 0;
        for (int n = 0; n < WSR_NFFT; ++n) {
            const int m = (k * n) % WSR_NFFT;
            re += x[n] * (real)cos(2.0 * M_PI * m / WSR_NFFT);
            im -= x[n] * (real)sin(2.0 * M_PI * m / WSR_NFFT);
        }
        if (k == 0 || k == WSR_NFFT / 2) im = 0;     /* a real FFT returns exactly +0 there (DC, Nyquist) */
        const float fre = (float)re, fim = (float)im;
        mag[k] = hypotf(fre, fim);
        ang[k] = atan2f(fim, fre);
    }
}

/* cond[B][3659][F], F = L/8.  mu_idx[B][L] and ang_idx[B][9][F] are optional outputs (the quantiser decisions). */
WGO_API int wgo_wsr_cond(const float *c, int B, int L, const float *mu_w, const float *ang_w, float *cond,
                         int32_t *mu_idx, int32_t *ang_idx)
{
    if (L % WSR_HOP || L < 8) return -1;
    const int F = L / WSR_HOP;
    for (int b = 0; b < B; ++b) {
        const float *cb = c + (long)b * L;
        float *ob = cond + (long)b * WSR_COND * F;
        for (int f = 0; f < F; ++f) {
            for (int j = 0; j < 8; ++j) {
                const int q = wsr_mu_index(clipf(cb[8 * f + j]));
                if (mu_idx) mu_idx[(long)b * L + 8 * f + j] = q;
                for (int e = 0; e < WSR_MU_DIM; ++e) ob[(long)(j * WSR_MU_DIM + e) * F + f] = mu_w[(long)q * WSR_MU_DIM + e];
            }
            float mag[WSR_BINS], ang[WSR_BINS];
            wsr_frame(cb, L, f, mag, ang);
            for (int k = 0; k < WSR_BINS; ++k) {
                ob[(long)(8 * WSR_MU_DIM + k) * F + f] = mag[k];
                const int q = wsr_angle_index(ang[k]);
                if (ang_idx) ang_idx[((long)b * WSR_BINS + k) * F + f] = q;
                for (int e = 0; e < WSR_ANG_DIM; ++e)
                    ob[(long)(8 * WSR_MU_DIM + WSR_BINS + k * WSR_ANG_DIM + e) * F + f] = ang_w[(long)q * WSR_ANG_DIM + e];
            }
        }
    }
    return 0;
}

/* gradients of the two embedding tables given dcond[B][3659][F] (c itself carries no gradient: both paths from c to cond that
 * have parameters go through integer indices; mag has no parameters).  nn.Embedding backward = scatter-add of rows. */
WGO_API int wgo_wsr_cond_backward(const float *c, int B, int L, const float *dcond, float *dmu_w, float *dang_w)
{
    if (L % WSR_HOP || L < 8) return -1;
    const int F = L / WSR_HOP;
    real *gm = rzalloc((size_t)WSR_MU * WSR_MU_DIM), *ga = rzalloc((size_t)WSR_ANG * WSR_ANG_DIM);
    for (int b = 0; b < B; ++b) {
        const float *cb = c + (long)b * L;
        const float *gb = dcond + (long)b * WSR_COND * F;
        for (int f = 0; f < F; ++f) {
            for (int j = 0; j < 8; ++j) {
                const int q = wsr_mu_index(clipf(cb[8 * f + j]));
                for (int e = 0; e < WSR_MU_DIM; ++e) gm[(long)q * WSR_MU_DIM + e] += (real)gb[(long)(j * WSR_MU_DIM + e) * F + f];
            }
            float mag[WSR_BINS], ang[WSR_BINS];
            wsr_frame(cb, L, f, mag, ang);
            for (int k = 0; k < WSR_BINS; ++k) {
                const int q = wsr_angle_index(ang[k]);
                for (int e = 0; e < WSR_ANG_DIM; ++e)
                    ga[(long)q * WSR_ANG_DIM + e] += (real)gb[(long)(8 * WSR_MU_DIM + WSR_BINS + k * WSR_ANG_DIM + e) * F + f];
            }
        }
    }
    r2f(gm, dmu_w, (long)WSR_MU * WSR_MU_DIM);
    r2f(ga, dang_w, (long)WSR_ANG * WSR_ANG_DIM);
    free(gm); free(ga);
    return 0;
}

WGO_API int wgo_real_bytes(void) { return (int)sizeof(real); }

#ifdef _OPENMP
#include <omp.h>
/* cap the OpenMP team (the loops expose ~64-128 independent row blocks; more threads only add fork/join cost) */
WGO_API int wgo_set_threads(int n) { if (n > 0) omp_set_num_threads(n); return omp_get_max_threads(); }
#else
WGO_API int wgo_set_threads(int n) { (void)n; return 1; }
#endif
