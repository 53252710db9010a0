"""Host utilities with the names / signatures of the reference's utils/utils.py (the subset the P2P path touches).
`latent2image` / `image2latent` take the pipeline's `vae` object exactly like the reference (utils/utils.py:58-80); when that
object is the native VAE they run the fused HIP paths (uint8 in / uint8 out), otherwise they fall back to the generic
encode/decode protocol."""
import numpy as np
import PIL.Image as Image
import torch

from ..p2p.token_align import get_word_inds  # noqa: F401  (re-exported, utils/utils.py:84-102)


def load_512(image_path, left=0, right=0, top=0, bottom=0):
    """Centre-crop to a square and resize to 512x512 RGB uint8 (utils/utils.py:27-46)."""
    image = np.array(Image.open(image_path))[:, :, :3] if type(image_path) is str else image_path
    h, w, _ = image.shape
    left = min(left, w - 1)
    right = min(right, w - left - 1)
    top = min(top, h - left - 1)
    bottom = min(bottom, h - top - 1)
    image = image[top:h - bottom, left:w - right]
    h, w, _ = image.shape
    if h < w:
        off = (w - h) // 2
        image = image[:, off:off + h]
    elif w < h:
        off = (h - w) // 2
        image = image[off:off + w]
    return np.array(Image.fromarray(image).resize((512, 512)))


def slerp(val, low, high):
    """utils/utils.py:7-16: spherical interpolation between the rows of two [n, d] tensors"""
    low_norm = low / torch.norm(low, dim=1, keepdim=True)
    high_norm = high / torch.norm(high, dim=1, keepdim=True)
    omega = torch.acos((low_norm * high_norm).sum(1))
    so = torch.sin(omega)
    return (torch.sin((1.0 - val) * omega) / so).unsqueeze(1) * low + (torch.sin(val * omega) / so).unsqueeze(1) * high


def slerp_tensor(val, low, high):
    """utils/utils.py:19-25 (negative-prompt inversion's npi_interp): slerp of the flattened tensors"""
    shape = low.shape
    return slerp(val, low.flatten(1), high.flatten(1)).reshape(shape)


def init_latent(latent, model, height, width, generator, batch_size):
    """utils/utils.py:48-55"""
    if latent is None:
        latent = torch.randn((1, model.unet.in_channels, height // 8, width // 8), generator=generator)
    latents = latent.expand(batch_size, model.unet.in_channels, height // 8, width // 8).to(model.device)
    return latent, latents


@torch.no_grad()
def latent2image(model, latents, return_type="np"):
    """`model` is the pipeline's vae (utils/utils.py:58-66)."""
    if hasattr(model, "latent2image_u8") and return_type == "np":
        return model.latent2image_u8(latents)
    latents = 1 / 0.18215 * latents.detach()
    image = model.decode(latents)["sample"]
    if return_type == "np":
        image = (image / 2 + 0.5).clamp(0, 1)
        image = image.cpu().permute(0, 2, 3, 1).numpy()
        image = (image * 255).astype(np.uint8)
    return image


@torch.no_grad()
def image2latent(model, image):
    """utils/utils.py:68-80"""
    if isinstance(image, Image.Image):
        image = np.array(image)
    if type(image) is torch.Tensor and image.dim() == 4:
        return image
    if hasattr(model, "image2latent_u8"):
        return model.image2latent_u8(image)
    image = torch.from_numpy(image).float() / 127.5 - 1
    image = image.permute(2, 0, 1).unsqueeze(0).to(model.device)
    return model.encode(image)["latent_dist"].mean * 0.18215


def update_alpha_time_word(alpha, bounds, prompt_ind, word_inds=None):
    """utils/utils.py:104-114 (note: fractions are taken of alpha.shape[0] = num_steps + 1)."""
    if type(bounds) is float:
        bounds = 0, bounds
    start, end = int(bounds[0] * alpha.shape[0]), int(bounds[1] * alpha.shape[0])
    if word_inds is None:
        word_inds = torch.arange(alpha.shape[2])
    alpha[:start, prompt_ind, word_inds] = 0
    alpha[start:end, prompt_ind, word_inds] = 1
    alpha[end:, prompt_ind, word_inds] = 0
    return alpha


def get_time_words_attention_alpha(prompts, num_steps, cross_replace_steps, tokenizer, max_num_words=77):
    """cross_replace_alpha table [num_steps + 1, n_prompts - 1, 1, 1, 77] (utils/utils.py:117-135)."""
    if type(cross_replace_steps) is not dict:
        cross_replace_steps = {"default_": cross_replace_steps}
    if "default_" not in cross_replace_steps:
        cross_replace_steps["default_"] = (0., 1.)
    table = torch.zeros(num_steps + 1, len(prompts) - 1, max_num_words)
    for i in range(len(prompts) - 1):
        table = update_alpha_time_word(table, cross_replace_steps["default_"], i)
    for word, bounds in cross_replace_steps.items():
        if word == "default_":
            continue
        for i in range(1, len(prompts)):
            inds = get_word_inds(prompts[i], word, tokenizer)
            if len(inds) > 0:
                table = update_alpha_time_word(table, bounds, i - 1, inds)
    return table.reshape(num_steps + 1, len(prompts) - 1, 1, 1, max_num_words)


def txt_draw(text, target_size=(512, 512)):
    """Instruction panel (utils/utils.py:137-155).  The reference renders it with matplotlib calls that no longer exist
    (np.fromstring / tostring_argb / Image.ANTIALIAS); it is not part of the numeric path (the evaluation crops it away),
    so it is redrawn with PIL: black text on white, wrapped."""
    from PIL import ImageDraw
    img = Image.new("RGB", (target_size[1], target_size[0]), (255, 255, 255))
    draw = ImageDraw.Draw(img)
    y = 10
    for para in text.split("\n"):
        line = ""
        for word in para.split(" "):
            if len(line) + len(word) + 1 > 70:
                draw.text((10, y), line, fill=(0, 0, 0))
                y += 14
                line = word
            else:
                line = (line + " " + word).strip()
        draw.text((10, y), line, fill=(0, 0, 0))
        y += 18
    return np.asarray(img)[:, :, :3]
