"""Pixel metrics of the PIE-Bench evaluation (evaluation/matrics_calculator.py:304-383: calculate_psnr / calculate_mse /
calculate_ssim, dispatched by evaluation/evaluate.py:30-84 for the whole image, the unedited part (1 - mask) and the edited part
(mask)), for self-contained sweep reports.  The reference computes them with torchmetrics (not installed offline):
PeakSignalNoiseRatio(data_range=1.0), MeanSquaredError(), StructuralSimilarityIndexMeasure(data_range=1.0) on float images in
[0, 1]; their definitions are restated here in numpy (host work on 512 x 512 x 3 images, not part of the hot path):

  mse  = mean((pred - gt)^2) over every element
  psnr = 10 log10(data_range^2 / mse)
  ssim = torchmetrics' default: 11 x 11 gaussian window (sigma 1.5), K1 = 0.01, K2 = 0.03, reflect padding by 5 and the padded
         border cropped away again, per-channel maps averaged over (C, H, W)

Masks follow the reference: `img * mask` before the metric (so a masked metric still averages over all pixels)."""
import numpy as np


def _prep(img_pred, img_gt, mask_pred=None, mask_gt=None):
    a = np.asarray(img_pred).astype(np.float32) / 255
    b = np.asarray(img_gt).astype(np.float32) / 255
    assert a.shape == b.shape, "Image shapes should be the same."
    if mask_pred is not None:
        a = a * np.asarray(mask_pred).astype(np.float32)
    if mask_gt is not None:
        b = b * np.asarray(mask_gt).astype(np.float32)
    return a, b


def calculate_mse(img_pred, img_gt, mask_pred=None, mask_gt=None):
    """matrics_calculator.py:345-362"""
    a, b = _prep(img_pred, img_gt, mask_pred, mask_gt)
    return float(np.mean((a.astype(np.float64) - b.astype(np.float64)) ** 2))


def calculate_psnr(img_pred, img_gt, mask_pred=None, mask_gt=None):
    """matrics_calculator.py:304-322 (data_range = 1.0); identical images give +inf, as torchmetrics does"""
    mse = calculate_mse(img_pred, img_gt, mask_pred, mask_gt)
    return float("inf") if mse == 0 else float(10.0 * np.log10(1.0 / mse))


def _gauss_1d(size=11, sigma=1.5):
    x = np.arange(size, dtype=np.float64) - (size - 1) / 2.0
    g = np.exp(-(x / sigma) ** 2 / 2)
    return g / g.sum()


def _filter_valid(x, g):
    """separable 'valid' correlation of an [H, W] plane with the 1-D window g along both axes"""
    k = len(g)
    H, W = x.shape
    tmp = np.zeros((H - k + 1, W), dtype=np.float64)
    for i in range(k):
        tmp += g[i] * x[i:i + H - k + 1, :]
    out = np.zeros((H - k + 1, W - k + 1), dtype=np.float64)
    for j in range(k):
        out += g[j] * tmp[:, j:j + W - k + 1]
    return out


def calculate_ssim(img_pred, img_gt, mask_pred=None, mask_gt=None, kernel_size=11, sigma=1.5, k1=0.01, k2=0.03, data_range=1.0):
    """matrics_calculator.py:364-383"""
    a, b = _prep(img_pred, img_gt, mask_pred, mask_gt)
    if a.ndim == 2:
        a, b = a[:, :, None], b[:, :, None]
    pad = (kernel_size - 1) // 2
    g = _gauss_1d(kernel_size, sigma)
    c1, c2 = (k1 * data_range) ** 2, (k2 * data_range) ** 2
    total, count = 0.0, 0
    for ch in range(a.shape[2]):
        p = np.pad(a[:, :, ch].astype(np.float64), pad, mode="reflect")
        t = np.pad(b[:, :, ch].astype(np.float64), pad, mode="reflect")
        mu_p, mu_t = _filter_valid(p, g), _filter_valid(t, g)
        s_pp = _filter_valid(p * p, g) - mu_p * mu_p
        s_tt = _filter_valid(t * t, g) - mu_t * mu_t
        s_pt = _filter_valid(p * t, g) - mu_p * mu_t
        ssim = ((2 * mu_p * mu_t + c1) * (2 * s_pt + c2)) / ((mu_p * mu_p + mu_t * mu_t + c1) * (s_pp + s_tt + c2))
        ssim = ssim[pad:-pad, pad:-pad]          # torchmetrics crops the reflect-padded border off the map again
        total += ssim.sum()
        count += ssim.size
    return float(total / count)


def panel_report(panel, mask=None):
    """Metrics of one 4-panel output image (instruction | source | reconstruction | edit, each S x S; evaluation/evaluate.py:268-273
    crops the same columns): reconstruction vs source (what an inversion method is judged on), edit vs source on the whole image and,
    with the PIE-Bench mask (1 = edited region, run_editing_p2p.mask_decode), on the unedited part."""
    p = np.asarray(panel)
    S = p.shape[0]
    src, rec, edit = p[:, S:2 * S], p[:, 2 * S:3 * S], p[:, 3 * S:4 * S]
    out = {"recon_psnr": calculate_psnr(rec, src), "recon_mse": calculate_mse(rec, src), "recon_ssim": calculate_ssim(rec, src),
           "edit_psnr": calculate_psnr(edit, src), "edit_mse": calculate_mse(edit, src), "edit_ssim": calculate_ssim(edit, src)}
    if mask is not None:
        m = np.asarray(mask).astype(np.float32)
        if m.ndim == 2:
            m = m[:, :, None].repeat(3, 2)
        keep = 1 - m
        out.update(psnr_unedit_part=calculate_psnr(edit, src, keep, keep), mse_unedit_part=calculate_mse(edit, src, keep, keep),
                   ssim_unedit_part=calculate_ssim(edit, src, keep, keep))
    return out
