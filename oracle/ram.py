"""oracle.ram -- numpy restatement of Random Amplitude Mixup (RAM).  TEST INFRASTRUCTURE ONLY.

Restates, from the reference checkout:
  extract_amp_spectrum    code/dataset/fundus.py:13-19   (byte-identical twin: code/dataset/prostate.py:10-16)
  low_freq_mutate_np      code/dataset/fundus.py:21-39
  source_to_target_freq   code/dataset/fundus.py:41-61
  RAM call sites          code/dataset/fundus.py:211-225 (clip [0,255], /127.5-1) and
                          code/dataset/prostate.py:186-188 (clip [-1,1])

The reference draws the mix ratio inside ``low_freq_mutate_np`` with ``random.randint(1,10)/10``
(fundus.py:35); here it is an explicit argument ``lam`` so that the GPU path and the oracle can be
fed the same value.  All arithmetic is done in float64 (numpy 1.x semantics of the 2022 reference;
numpy>=2 keeps float32 for float32 input -- the fixtures hold both).
"""
import numpy as np


def window_half_width(h, w, L=0.1):
    """b = floor(min(h,w)*L)  -- fundus.py:26."""
    return int(np.floor(np.amin((h, w)) * L))


def extract_amp_spectrum(img_chw, dtype=np.float64):
    """|fft2(img)| over the last two axes -- fundus.py:13-19."""
    fft = np.fft.fft2(np.asarray(img_chw, dtype=dtype), axes=(-2, -1))
    return np.abs(fft)


def low_freq_mutate(amp_src, amp_trg, lam, L=0.1):
    """Centre-window lerp of two amplitude spectra -- fundus.py:21-39 (ratio == lam)."""
    a_src = np.fft.fftshift(amp_src, axes=(-2, -1)).copy()
    a_trg = np.fft.fftshift(amp_trg, axes=(-2, -1))
    _, h, w = a_src.shape
    b = window_half_width(h, w, L)
    c_h = int(np.floor(h / 2.0))
    c_w = int(np.floor(w / 2.0))
    h1, h2 = c_h - b, c_h + b + 1
    w1, w2 = c_w - b, c_w + b + 1
    a_src[:, h1:h2, w1:w2] = a_src[:, h1:h2, w1:w2] * lam + a_trg[:, h1:h2, w1:w2] * (1 - lam)
    return np.fft.ifftshift(a_src, axes=(-2, -1))


def source_to_target_freq(src_hwc, amp_trg, lam, L=0.1, dtype=np.float64):
    """real(ifft2(A' * exp(j*P))) -- fundus.py:41-61.  src is HWC, result is HWC."""
    src = np.asarray(src_hwc, dtype=dtype).transpose((2, 0, 1))
    fft_src = np.fft.fft2(src, axes=(-2, -1))
    amp_src, pha_src = np.abs(fft_src), np.angle(fft_src)
    amp_src_ = low_freq_mutate(amp_src, amp_trg, lam, L=L)
    fft_src_ = amp_src_ * np.exp(1j * pha_src)
    out = np.real(np.fft.ifft2(fft_src_, axes=(-2, -1)))
    return out.transpose(1, 2, 0)


def ram_fundus(src_hwc, trg_hwc, lam, L=0.1, dtype=np.float64):
    """Fundus call site, fundus.py:211-225: images on the 0..255 scale.
    Returns (img, img_freq) as CHW float32 in [-1, 1]."""
    src = np.asarray(src_hwc, dtype=np.float32)
    amp_trg = extract_amp_spectrum(np.asarray(trg_hwc, dtype=np.float32).transpose(2, 0, 1), dtype=dtype)
    img_freq = source_to_target_freq(src, amp_trg, lam, L=L, dtype=dtype)
    img_freq = np.clip(img_freq, 0, 255).astype(np.float32)
    img = src / np.float32(127.5) - np.float32(1.0)
    img_freq = img_freq / np.float32(127.5) - np.float32(1.0)
    return img.transpose(2, 0, 1).copy(), img_freq.transpose(2, 0, 1).copy()


def ram_prostate(src_hwc, trg_hwc, lam, L=0.1, dtype=np.float64):
    """Prostate call site, prostate.py:186-195: slices already in [-1, 1]; clip to [-1, 1]."""
    src = np.asarray(src_hwc, dtype=np.float32)
    amp_trg = extract_amp_spectrum(np.asarray(trg_hwc, dtype=np.float32).transpose(2, 0, 1), dtype=dtype)
    img_freq = np.clip(source_to_target_freq(src, amp_trg, lam, L=L, dtype=dtype), -1, 1)
    return src.transpose(2, 0, 1).copy(), img_freq.transpose(2, 0, 1).astype(np.float32)


def window_gain_form(src_hwc, trg_hwc, lam, L=0.1):
    """Algebraically equal form used by the HIP kernel: the amplitude lerp keeps
    the source phase, so it is a real gain on the source spectrum,
        g = lam + (1-lam)*|F_trg|/|F_src|   inside |ky|<=b, |kx|<=b,    g = 1 elsewhere,
    i.e. out = src + ifft2( (g-1) * F_src ), a correction supported on the (2b+1)^2 window only.
    Where |F_src| == 0 the reference's angle() is 0, so the bin becomes (1-lam)*|F_trg|."""
    src = np.asarray(src_hwc, dtype=np.float64).transpose(2, 0, 1)
    trg = np.asarray(trg_hwc, dtype=np.float64).transpose(2, 0, 1)
    _, h, w = src.shape
    b = window_half_width(h, w, L)
    fs = np.fft.fft2(src, axes=(-2, -1))
    ft = np.fft.fft2(trg, axes=(-2, -1))
    ky = np.fft.fftfreq(h, 1.0 / h).astype(int)
    kx = np.fft.fftfreq(w, 1.0 / w).astype(int)
    # fftshift centre c=floor(n/2): shifted index c+d holds frequency d for d in [-b, b]
    win = (np.abs(ky)[:, None] <= b) & (np.abs(kx)[None, :] <= b)
    a_s, a_t = np.abs(fs), np.abs(ft)
    corr = np.zeros_like(fs)
    nz = a_s > 0
    m = win[None] & nz
    corr[m] = (1 - lam) * (a_t[m] / a_s[m] - 1.0) * fs[m]
    m0 = win[None] & ~nz
    corr[m0] = (1 - lam) * a_t[m0]
    out = src + np.real(np.fft.ifft2(corr, axes=(-2, -1)))
    return out.transpose(1, 2, 0)
