"""numpy restatement of the reference's MelSpec conditioner (model/condition.py:7-19).  TEST INFRASTRUCTURE ONLY.

    ReflectionPad1d((n_fft//2 - hop//2, n_fft//2 + hop//2)) -> torchaudio.transforms.MelSpectrogram(sample_rate, n_fft, hop_length,
    center=False, **kwargs) -> add_(1e-7).log_()

PARITY UNPINNED against torchaudio: it is not part of /root/reference and not installed in this image, and the reference pins no
version.  What is restated is torchaudio's published algorithm and defaults (torchaudio.transforms.MelSpectrogram,
torchaudio.functional.spectrogram / melscale_fbanks): win_length = n_fft, periodic Hann window, power = 2, normalized = False,
onesided, mel_scale = "htk", norm = None, f_min = 0:
    all_freqs = linspace(0, sample_rate // 2, n_fft // 2 + 1)
    m_pts = linspace(hz2mel(f_min), hz2mel(f_max), n_mels + 2), hz2mel(f) = 2595 log10(1 + f / 700) ; f_pts = mel2hz(m_pts)
    fb[f, m] = max(0, min((all_freqs[f] - f_pts[m]) / (f_pts[m+1] - f_pts[m]), (f_pts[m+2] - all_freqs[f]) / (f_pts[m+2] - f_pts[m+1])))
    mel = fb^T |STFT|^2
The arithmetic is float64 (np.fft.rfft); the inputs are float32.
"""
import numpy as np


def mel_filterbank(sr, n_fft, n_mels, f_min=0.0, f_max=None):
    f_max = float(sr // 2) if f_max is None else float(f_max)
    n_freqs = n_fft // 2 + 1
    all_freqs = np.linspace(0.0, sr // 2, n_freqs)
    m_min, m_max = 2595.0 * np.log10(1.0 + f_min / 700.0), 2595.0 * np.log10(1.0 + f_max / 700.0)
    m_pts = np.linspace(m_min, m_max, n_mels + 2)
    f_pts = 700.0 * (10.0 ** (m_pts / 2595.0) - 1.0)
    f_diff = f_pts[1:] - f_pts[:-1]
    slopes = f_pts[None, :] - all_freqs[:, None]
    down = -slopes[:, :-2] / f_diff[:-1]
    up = slopes[:, 2:] / f_diff[1:]
    return np.maximum(0.0, np.minimum(down, up))            # [n_freqs, n_mels]


def melspec(x, sr, n_fft, hop, f_min=0.0, f_max=None, n_mels=128):
    """x [B, N] float32 -> log-mel [B, n_mels, N // hop + 1] (float64)."""
    x = np.asarray(x, np.float32).astype(np.float64)
    B, N = x.shape
    left, right = n_fft // 2 - hop // 2, n_fft // 2 + hop // 2
    xp = np.pad(x, ((0, 0), (left, right)), mode="reflect")
    frames = (xp.shape[1] - n_fft) // hop + 1
    win = 0.5 - 0.5 * np.cos(2.0 * np.pi * np.arange(n_fft) / n_fft)
    idx = hop * np.arange(frames)[:, None] + np.arange(n_fft)[None, :]
    spec = np.abs(np.fft.rfft(xp[:, idx] * win, axis=-1)) ** 2          # [B, frames, n_freqs]
    mel = spec @ mel_filterbank(sr, n_fft, n_mels, f_min, f_max)        # [B, frames, n_mels]
    return np.log(mel.transpose(0, 2, 1) + 1e-7)


# ---- STFTDecimate / LowPass (model/condition.py:22-66): the data-side conditioner of WSRGlow ------------------------------------
def stft_decimate(x, r, nfft=1024, hop=256):
    """x [B, T] float32 -> low-passed, decimated [B, ceil(T / r)] (float64 arithmetic).  Restates, with torch's documented stft / istft
    semantics (center=True, reflect padding, periodic Hann, onesided, no normalisation; istft = overlap-add of windowed inverse
    frames divided by the overlap-added squared window):
        x = pad(x, (0, nfft)) ; S = stft(x) ; S[int((nfft//2+1) / r):] = 0 ; y = istft(S)[:, :T] ; return y[:, ::r]
    Pinned to the reference itself: tests/golden/cond_stftdecimate.npz (make_golden.conditioner_fixture runs condition.py's
    STFTDecimate under ref_shim's legacy-stft wrapper)."""
    x = np.asarray(x, np.float32).astype(np.float64)
    B, T = x.shape
    half = nfft // 2
    xp = np.pad(np.pad(x, ((0, 0), (0, nfft))), ((0, 0), (half, half)), mode="reflect")
    frames = (xp.shape[1] - nfft) // hop + 1
    win = 0.5 - 0.5 * np.cos(2.0 * np.pi * np.arange(nfft) / nfft)
    idx = hop * np.arange(frames)[:, None] + np.arange(nfft)[None, :]
    S = np.fft.rfft(xp[:, idx] * win, axis=-1)                      # [B, frames, nfft/2+1]
    S[:, :, int((half + 1) * (1.0 / r)):] = 0.0
    fr = np.fft.irfft(S, n=nfft, axis=-1) * win
    L = nfft + hop * (frames - 1)
    y = np.zeros((B, L))
    env = np.zeros(L)
    for f in range(frames):
        y[:, f * hop:f * hop + nfft] += fr[:, f]
        env[f * hop:f * hop + nfft] += win * win
    y = y[:, half:L - half] / env[half:L - half]
    return y[:, :T][:, ::r]
