"""Host logic of the sweep driver (no GPU): RLE mask decode semantics of run_editing_p2p.py:11-27."""
import numpy as np

from run_editing_p2p import mask_decode


def _reference_semantics(encoded, shape=(512, 512)):
    n = shape[0] * shape[1]
    a = np.zeros(n)
    for i in range(0, len(encoded), 2):
        for j in range(min(encoded[i + 1], n - encoded[i])):
            a[encoded[i] + j] = 1
    a = a.reshape(shape)
    a[0, :] = 1; a[-1, :] = 1; a[:, 0] = 1; a[:, -1] = 1
    return a


def test_mask_decode():
    rng = np.random.default_rng(0)
    enc = []
    pos = 0
    while pos < 512 * 512 - 5000:
        pos += int(rng.integers(1, 4000))
        ln = int(rng.integers(1, 3000))
        enc += [pos, ln]
        pos += ln
    enc += [512 * 512 - 10, 500]           # run past the end is clipped
    assert np.array_equal(mask_decode(enc), _reference_semantics(enc))
    assert mask_decode([]).sum() == 4 * 512 - 4


def test_checkpoint_dir_loader_and_weight_flags(tmp_path):
    """--checkpoint_dir reads the diffusers layout the reference downloads (models/p2p_editor.py:23-24); neither flag -> refusal;
    CLIP weights without their tokenizer -> ValueError (hashed stand-in ids into real CLIP weights would be a silently wrong edit)."""
    import argparse
    import json
    import pytest
    import torch
    from safetensors.torch import save_file
    from pnpinversion_amd.checkpoint import add_weight_args, load_checkpoint_dir, resolve_weights
    from pnpinversion_amd.config import TINY16
    from pnpinversion_amd.p2p_editor import P2PEditor
    from pnpinversion_amd.text import ClipBPETokenizer
    for sub in ("unet", "vae", "text_encoder", "tokenizer"):
        (tmp_path / sub).mkdir()
    save_file({"conv_in.weight": torch.ones(2, 2)}, str(tmp_path / "unet" / "diffusion_pytorch_model.safetensors"))
    torch.save({"encoder.conv_in.weight": torch.zeros(3)}, str(tmp_path / "vae" / "diffusion_pytorch_model.bin"))
    save_file({"text_model.embeddings.position_ids": torch.zeros(1, 77), "text_model.final_layer_norm.weight": torch.ones(4)},
              str(tmp_path / "text_encoder" / "model.safetensors"))
    vocab = {"<|startoftext|>": 0, "<|endoftext|>": 1, "a</w>": 2, "c": 3, "at</w>": 4, "cat</w>": 5, "a": 6, "t</w>": 7}
    (tmp_path / "tokenizer" / "vocab.json").write_text(json.dumps(vocab))
    (tmp_path / "tokenizer" / "merges.txt").write_text("#version: 0.2\na t</w>\nc at</w>\n")
    unet, vae, clip, tok = load_checkpoint_dir(str(tmp_path))
    assert list(unet) == ["conv_in.weight"] and list(vae) == ["encoder.conv_in.weight"]
    assert list(clip) == ["final_layer_norm.weight"]                       # prefix stripped, position_ids buffer dropped
    assert isinstance(tok, ClipBPETokenizer) and tok.encode("a cat") == [0, 2, 5, 1]
    ap = argparse.ArgumentParser()
    add_weight_args(ap)
    with pytest.raises(SystemExit, match="--checkpoint_dir"):
        resolve_weights(ap.parse_args([]), TINY16)
    with pytest.raises(SystemExit, match="mutually exclusive"):
        resolve_weights(ap.parse_args(["--checkpoint_dir", str(tmp_path), "--synthetic_weights"]), TINY16)
    got = resolve_weights(ap.parse_args(["--checkpoint_dir", str(tmp_path)]), TINY16, rank=1)
    assert got[0] is None and isinstance(got[3], ClipBPETokenizer)          # other ranks receive the arena by broadcast
    with pytest.raises(ValueError, match="without their tokenizer"):
        P2PEditor(["directinversion+p2p"], "cuda", state_dicts=({}, {}, {"x": torch.zeros(1)}))


def test_word_tokenizer_truncation_keeps_eos():
    from pnpinversion_amd.text import BOS, EOS, WordTokenizer
    tok = WordTokenizer()
    ids = tok.encode(" ".join("w%d" % i for i in range(100)))
    assert len(ids) == 77 and ids[0] == BOS and ids[-1] == EOS and EOS not in ids[1:-1]
    rows = tok([" ".join("w%d" % i for i in range(100)), "a b"], padding="max_length", max_length=77).input_ids
    assert rows.shape == (2, 77) and int(rows[0, -1]) == EOS and int(rows[1, 3]) == EOS


def test_pixel_metrics_against_independent_implementations():
    """pnpinversion_amd.metrics (PSNR / MSE / SSIM of evaluation/matrics_calculator.py:304-383, torchmetrics' definitions) against an
    independent evaluation: scipy.ndimage gaussian correlation with mirror boundaries for SSIM's windowed moments."""
    import scipy.ndimage as ndi
    from pnpinversion_amd import metrics
    rng = np.random.default_rng(3)
    a = rng.integers(0, 256, size=(96, 80, 3)).astype(np.uint8)
    b = np.clip(a.astype(np.int32) + rng.integers(-20, 21, size=a.shape), 0, 255).astype(np.uint8)
    fa, fb = a.astype(np.float64) / 255, b.astype(np.float64) / 255
    mse = np.mean((fa - fb) ** 2)
    assert abs(metrics.calculate_mse(a, b) - mse) < 1e-9
    assert abs(metrics.calculate_psnr(a, b) - 10 * np.log10(1 / mse)) < 1e-6
    assert metrics.calculate_psnr(a, a) == float("inf") and metrics.calculate_ssim(a, a) == 1.0
    # SSIM: full-image gaussian moments with mirror ("reflect" in torch's naming) boundaries == pad-by-5 + valid window + crop-by-5
    # only in the interior; compare on the interior [10:-10] where both formulations see real pixels only
    x = np.arange(11) - 5.0
    g = np.exp(-(x / 1.5) ** 2 / 2); g /= g.sum()
    def blur(z):
        return ndi.correlate1d(ndi.correlate1d(z, g, axis=0, mode="mirror"), g, axis=1, mode="mirror")
    c1, c2 = 0.01 ** 2, 0.03 ** 2
    maps = []
    for ch in range(3):
        p, t = fa[:, :, ch], fb[:, :, ch]
        mp, mt = blur(p), blur(t)
        spp, stt, spt = blur(p * p) - mp * mp, blur(t * t) - mt * mt, blur(p * t) - mp * mt
        maps.append(((2 * mp * mt + c1) * (2 * spt + c2)) / ((mp * mp + mt * mt + c1) * (spp + stt + c2)))
    full = np.stack(maps, -1)
    # torchmetrics' map is the centre crop [5:-5, 5:-5] of the map computed on the reflect-padded image, i.e. the [5:-5] crop of `full`
    want = full[5:-5, 5:-5].mean()
    assert abs(metrics.calculate_ssim(a, b) - want) < 1e-9, (metrics.calculate_ssim(a, b), want)
    m = np.zeros((96, 80, 3), np.float32); m[20:60, 10:50] = 1
    assert metrics.calculate_mse(a, b, 1 - m, 1 - m) < metrics.calculate_mse(a, b)
    panel = np.concatenate([a[:80], a[:80], b[:80], b[:80]], axis=1)
    rep = metrics.panel_report(panel, mask=m[:80, :, 0])
    assert set(rep) >= {"recon_psnr", "edit_ssim", "psnr_unedit_part"} and abs(rep["recon_mse"] - metrics.calculate_mse(b[:80], a[:80])) < 1e-12


def test_proximal_guidance_forward_inversion_guidance_semantics(monkeypatch):
    """models/p2p/proximal_guidance_forward.py:73 reads `mask_edit is not None and inversion_guidance and (recon_t > 0 and t < recon_t) or
    (recon_t < 0 and t > -recon_t)`: the pull towards the inversion trajectory runs with the flag inside a positive recon_t window, and --
    by `and` binding tighter than `or` -- ALWAYS inside a negative one.  Host logic only: what reaches the loop (the `recon` descriptor)."""
    import pytest
    import pnpinversion_amd.p2p.proximal_guidance_forward as pg
    seen = {}

    def fake_loop(**kw):
        seen.clear(); seen.update(kw)
        return "latents", "latent"

    monkeypatch.setattr(pg, "p2p_guidance_forward", fake_loop)
    xs, enc = ["x*0", "x*1", "x*2"], "encoded"
    call = lambda **k: pg.proximal_guidance_forward(model=None, prompt=["a", "b"], controller=None, num_inference_steps=2, prox="l0", **k)
    call(recon_lr=0.5, recon_t=400)                                                      # nothing to pull towards
    assert seen["recon"] is None and seen["prox"] == "l0"
    call(recon_lr=0.5, recon_t=400, x_stars=xs)                                          # positive window, flag off: x_stars is ignored
    assert seen["recon"] is None
    call(recon_lr=0.5, recon_t=400, x_stars=xs, inversion_guidance=True)
    assert seen["recon"]["x_stars"] is xs and seen["recon"]["ref_image"] is None and seen["recon"]["recon_t"] == 400
    call(recon_lr=0.5, recon_t=-600, x_stars=xs)                                         # negative window: the pull runs without the flag
    assert seen["recon"]["x_stars"] is xs and seen["recon"]["recon_t"] == -600
    call(recon_lr=0.5, recon_t=400, image_enc=enc)                                       # reconstruction guidance alone
    assert seen["recon"]["ref_image"] == enc and seen["recon"]["x_stars"] is None
    call(recon_lr=0.0, recon_t=400, image_enc=enc, x_stars=xs, inversion_guidance=True)  # recon_lr = 0 switches both off
    assert seen["recon"] is None
    call(recon_lr=0.5, recon_t=400, x_stars=xs, inversion_guidance=True, edit_stage=False)   # the reconstruction pass: no proximal step at all
    assert seen["recon"] is None and seen["prox"] is None
    # recon_lr < 0: the pred-x0 pull is off (scheduler_dev.py:68 tests recon_lr > 0), the inversion pull is not (line 75 has no test)
    call(recon_lr=-0.5, recon_t=400, image_enc=enc, x_stars=xs, inversion_guidance=True)
    assert seen["recon"]["ref_image"] is None and seen["recon"]["x_stars"] is xs and seen["recon"]["recon_lr"] == -0.5
    call(recon_lr=-0.5, recon_t=400, image_enc=enc)
    assert seen["recon"] is None
    # a negative recon_t without a proximal step: the reference reaches `1 - mask_edit` with mask_edit = None at the first step with t > -recon_t
    from types import SimpleNamespace as NS
    mdl = NS(scheduler=NS(config=NS(num_train_timesteps=1000, steps_offset=0), num_inference_steps=2))
    with pytest.raises(TypeError, match="mask_edit = None"):
        pg.proximal_guidance_forward(model=mdl, prompt=["a", "b"], controller=None, num_inference_steps=2, prox="l0", recon_t=-400, x_stars=xs,
                                     edit_stage=False)
    with pytest.raises(TypeError, match="mask_edit = None"):
        pg.proximal_guidance_forward(model=mdl, prompt=["a", "b"], controller=None, num_inference_steps=2, prox=None, recon_t=-400, x_stars=xs)
    pg.proximal_guidance_forward(model=mdl, prompt=["a", "b"], controller=None, num_inference_steps=2, prox=None, recon_t=-600, x_stars=xs)   # t = 500, 0: never > 600
    assert seen["recon"] is None and seen["prox"] is None
    with pytest.raises(ValueError, match="needs x_stars"):
        call(recon_lr=0.5, recon_t=-600)
    with pytest.raises(ValueError, match="needs x_stars"):
        call(recon_lr=0.5, recon_t=400, inversion_guidance=True)
    with pytest.raises(NotImplementedError):
        pg.proximal_guidance_forward(model=None, prompt=["a", "b"], controller=None, num_inference_steps=2, prox="l2")
