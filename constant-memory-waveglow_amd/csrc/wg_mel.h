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
