// Log-mel conditioner (model/condition.py:7-19): ReflectionPad1d((n_fft/2 - hop/2, n_fft/2 + hop/2)) -> MelSpectrogram(center=False) ->
// log(x + 1e-7).  MelSpectrogram is torchaudio code (absent from the reference tree and from this image); its published defaults are
// restated here: periodic Hann window of n_fft, power 2, onesided STFT, HTK mel scale, no filterbank normalisation, f_min 0
// (torchaudio.transforms.MelSpectrogram / torchaudio.functional.melscale_fbanks).  PARITY UNPINNED against torchaudio itself.
//
// One workgroup = one frame: the windowed samples and a cos/sin table of 2 pi j / n_fft live in LDS, every thread evaluates its
// bins by the direct n_fft-term DFT (frames are few: 63 per 16 000-sample segment, the whole batch is ~3 GFLOP), the power
// spectrum goes back to LDS and the first n_mels threads apply their triangular filter, evaluated in closed form.
#pragma once
#include <hip/hip_runtime.h>

#define WG_MEL_MAXFFT 2048

struct MelArgs {
    const float *audio;     // [B][N]
    float *mel;             // [B][n_mels][frames]
    float *power;           // optional [B][n_fft/2+1][frames]: |STFT|^2 before the mel filterbank (what torch.stft(...).abs()**2 gives)
    int N, n_fft, hop, n_mels, frames, pad_left, sr_half;
    float m_min, m_max;     // HTK mel of f_min / f_max
};
__device__ __forceinline__ float wg_mel_to_hz(float m) { return 700.0f * (exp10f(m / 2595.0f) - 1.0f); }

__global__ __launch_bounds__(256) void melspec_kernel(const MelArgs a)
{
    __shared__ float xs[WG_MEL_MAXFFT], cs[WG_MEL_MAXFFT], sn[WG_MEL_MAXFFT], pw[WG_MEL_MAXFFT / 2 + 1];
    const int f = blockIdx.x, b = blockIdx.y, tid = threadIdx.x, nf = a.n_fft, nb = nf / 2 + 1;
    const float *x = a.audio + (size_t)b * a.N;
    for (int n = tid; n < nf; n += 256) {
        int j = f * a.hop + n - a.pad_left;                         // reflection (no edge repeat), as nn.ReflectionPad1d
        if (j < 0) j = -j;
        if (j >= a.N) j = 2 * (a.N - 1) - j;
        float s, c;
        sincospif(2.0f * (float)n / (float)nf, &s, &c);
        xs[n] = (0.5f - 0.5f * c) * x[j];                            // periodic Hann
        cs[n] = c;
        sn[n] = s;
    }
    __syncthreads();
    for (int k = tid; k < nb; k += 256) {
        float re = 0.f, im = 0.f;
        int idx = 0;
        for (int n = 0; n < nf; ++n) {
            re = fmaf(xs[n], cs[idx], re);
            im = fmaf(-xs[n], sn[idx], im);
            idx = (idx + k) & (nf - 1);
        }
        pw[k] = re * re + im * im;
        if (a.power) a.power[((size_t)b * nb + k) * a.frames + f] = pw[k];
    }
    __syncthreads();
    if (tid < a.n_mels) {
        // filter tid: triangle over [f_lo, f_c, f_hi] = mel2hz of three consecutive points of linspace(m_min, m_max, n_mels + 2)
        const float dm = (a.m_max - a.m_min) / (float)(a.n_mels + 1);
        const float f_lo = wg_mel_to_hz(a.m_min + dm * tid), f_c = wg_mel_to_hz(a.m_min + dm * (tid + 1)),
                    f_hi = wg_mel_to_hz(a.m_min + dm * (tid + 2));
        float acc = 0.f;
        for (int k = 0; k < nb; ++k) {
            const float fr = (float)a.sr_half * (float)k / (float)(nb - 1);          // linspace(0, sr // 2, n_freqs)
            const float w = fminf((fr - f_lo) / (f_c - f_lo), (f_hi - fr) / (f_hi - f_c));
            if (w > 0.f) acc = fmaf(w, pw[k], acc);
        }
        a.mel[((size_t)b * a.n_mels + tid) * a.frames + f] = logf(acc + 1e-7f);
    }
}

// ------------------------------------------------------------------------------------------------
// LowPass / STFTDecimate (model/condition.py:22-66): zero-pad nfft samples on the right, STFT (center=True: reflect padding of nfft/2,
// periodic Hann, onesided), zero every bin >= cut, ISTFT (overlap-add of windowed inverse frames / overlap-added squared window),
// crop to T, keep every step-th sample.  Frame kernel: one workgroup per frame, forward DFT of the `cut` kept bins and the inverse
// real DFT back to n_fft samples, both direct from LDS tables (the kept band is at most n_fft/2+1 bins; a batch of 12 x 8192
// samples is ~1 GFLOP).  OLA kernel: each output sample gathers the <= n_fft/hop frames that cover it.
// ------------------------------------------------------------------------------------------------
struct LowPassArgs {
    const float *x;         // [B][T]
    float *frames;          // [B][nframes][n_fft]   (workspace)
    float *out;             // [B][ceil(T / step)]
    int T, n_fft, hop, nframes, cut, step, nout;
};
__global__ __launch_bounds__(256) void lowpass_frame_kernel(const LowPassArgs a)
{
    __shared__ float xs[WG_MEL_MAXFFT], cs[WG_MEL_MAXFFT], sn[WG_MEL_MAXFFT], re[WG_MEL_MAXFFT / 2 + 1], im[WG_MEL_MAXFFT / 2 + 1];
    const int f = blockIdx.x, b = blockIdx.y, tid = threadIdx.x, nf = a.n_fft, half = nf / 2;
    const int Lz = a.T + nf;                                        // length after the right zero padding (condition.py:45)
    const float *x = a.x + (size_t)b * a.T;
    for (int n = tid; n < nf; n += 256) {
        int j = f * a.hop + n - half;                               // center=True: reflect padding of the zero-extended signal
        if (j < 0) j = -j;
        if (j >= Lz) j = 2 * (Lz - 1) - j;
        float s, c;
        sincospif(2.0f * (float)n / (float)nf, &s, &c);
        xs[n] = (0.5f - 0.5f * c) * (j < a.T ? x[j] : 0.f);
        cs[n] = c;
        sn[n] = s;
    }
    __syncthreads();
    for (int k = tid; k < a.cut; k += 256) {
        float r = 0.f, i = 0.f;
        int idx = 0;
        for (int n = 0; n < nf; ++n) {
            r = fmaf(xs[n], cs[idx], r);
            i = fmaf(-xs[n], sn[idx], i);
            idx = (idx + k) & (nf - 1);
        }
        re[k] = r;
        im[k] = i;
    }
    __syncthreads();
    float *fr = a.frames + ((size_t)b * a.nframes + f) * nf;
    const int kmax = min(a.cut, half);                              // the Nyquist bin (k = half) is handled apart
    for (int n = tid; n < nf; n += 256) {
        float acc = 0.f;
        int idx = n & (nf - 1);
        for (int k = 1; k < kmax; ++k) {
            acc = fmaf(re[k], cs[idx], acc);
            acc = fmaf(-im[k], sn[idx], acc);
            idx = (idx + n) & (nf - 1);
        }
        float v = re[0] + 2.0f * acc;
        if (a.cut > half) v += (n & 1) ? -re[half] : re[half];
        fr[n] = v / (float)nf * (0.5f - 0.5f * cs[n]);             // irfft, then the synthesis window
    }
}
__global__ void lowpass_ola_kernel(const LowPassArgs a)
{
    const int m = blockIdx.x * blockDim.x + threadIdx.x, b = blockIdx.y;
    if (m >= a.nout) return;
    const int nf = a.n_fft, p = m * a.step + nf / 2;               // position in the overlap-add buffer (center trim = nf/2)
    int f0 = p - nf + 1;
    f0 = f0 <= 0 ? 0 : (f0 + a.hop - 1) / a.hop;
    const int f1 = min(p / a.hop, a.nframes - 1);
    float y = 0.f, env = 0.f;
    for (int f = f0; f <= f1; ++f) {
        const int n = p - f * a.hop;
        float s, c;
        sincospif(2.0f * (float)n / (float)nf, &s, &c);
        const float w = 0.5f - 0.5f * c;
        y += a.frames[((size_t)b * a.nframes + f) * nf + n];
        env = fmaf(w, w, env);
    }
    a.out[(size_t)b * a.nout + m] = y / env;
}
