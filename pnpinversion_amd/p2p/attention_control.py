"""Prompt-to-Prompt controllers, declarative form.

Same class names, constructor arguments and error behaviour as the reference's models/p2p/attention_control.py, but instead
of being called back on a materialised [B*8, N, M] probability tensor at each of the 32 attention sites
(attention_control.py:44,178-190) a controller here only *describes* the edit: `tables()` returns the host tables that the
fused HIP attention kernels consume (include/pnpi.h, pnpi_ctrl_desc):

    P_tgt' = c1 * (P_src @ mapper) + c2 * P_tgt,   c1 = a_t * eq * alphas,   c2 = a_t * eq * (1 - alphas) + (1 - a_t)

  AttentionReplace (:301-314)  mapper = replacement mapper [77,77], alphas = 1
  AttentionRefine  (:317-335)  mapper = one-hot gather of mapper[j] (index -1 selects the last column, weight alphas[j] = 0)
  AttentionReweight(:338-363)  eq = equalizer, chained after the previous controller's mapper / alphas
  LocalBlend       (:95-147)   lb_alpha = alpha_layers one-hot rows, start_blend, threshold
  self-attention replacement window num_self_replace (:295-297) and the <= 32^2-token rule (:258-263)
"""
import abc

import numpy as np
import torch

from ..engine import ControllerTables
from ..utils.utils import get_time_words_attention_alpha, get_word_inds
from . import token_align as seq_aligner

MAX_NUM_WORDS = 77
LATENT_SIZE = (64, 64)
LOW_RESOURCE = False


def get_equalizer(text, word_select, values, tokenizer=None):
    """attention_control.py:84-92"""
    if type(word_select) is int or type(word_select) is str:
        word_select = (word_select,)
    equalizer = torch.ones(1, 77)
    for word, val in zip(word_select, values):
        inds = get_word_inds(text, word, tokenizer)
        equalizer[:, inds] = val
    return equalizer


def _native_local_blend(lb, x_t, engine):
    """LocalBlend.__call__ (attention_control.py:108-121) for a step loop that drives model.unet(...) itself (SURVEY 8b level 1):
    `lb` carries the reference's attributes (counter, start_blend); the maps are the ones the cross-attention kernel accumulated over
    the UNet calls of this edit (kept in the library between calls), the blend is the local_blend kernel."""
    if engine is None:
        raise RuntimeError("LocalBlend: the controller is not registered with a native UNet (register_attention_control(model, "
                           "controller) first) -- the 16 x 16 maps it blends with live in the library, not in attention_store")
    lb.counter += 1
    if lb.counter > lb.start_blend:
        x_t = engine.local_blend(x_t, lb.counter - 1)
    return x_t


class LocalBlend:
    """Holds the LocalBlend parameters (attention_control.py:123-147); the blend itself runs in the local_blend HIP kernel
    on the maps accumulated by the cross-attention kernel."""

    def __call__(self, x_t, engine):
        return _native_local_blend(self, x_t, engine)

    def __init__(self, prompts, words, substruct_words=None, start_blend=0.2, th=(.3, .3), tokenizer=None, device="cuda",
                 num_ddim_steps=50):
        alpha_layers = torch.zeros(len(prompts), 1, 1, 1, 1, MAX_NUM_WORDS)
        for i, (prompt, words_) in enumerate(zip(prompts, words)):
            if type(words_) is str:
                words_ = [words_]
            for word in words_:
                ind = get_word_inds(prompt, word, tokenizer)
                alpha_layers[i, :, :, :, :, ind] = 1
        self.alpha_layers = alpha_layers
        self.substruct_layers = None
        if substruct_words is not None:                 # attention_control.py:134-143
            substruct_layers = torch.zeros(len(prompts), 1, 1, 1, 1, MAX_NUM_WORDS)
            for i, (prompt, words_) in enumerate(zip(prompts, substruct_words)):
                if type(words_) is str:
                    words_ = [words_]
                for word in words_:
                    ind = get_word_inds(prompt, word, tokenizer)
                    substruct_layers[i, :, :, :, :, ind] = 1
            self.substruct_layers = substruct_layers
        self.start_blend = int(start_blend * num_ddim_steps)
        self.counter = 0
        self.th = th


class EmptyControl:
    def step_callback(self, x_t):
        return x_t

    def between_steps(self):
        return

    def tables(self):
        return None


class AttentionControl(abc.ABC):
    def __init__(self):
        self.cur_step = 0
        self.num_att_layers = -1
        self.cur_att_layer = 0

    def reset(self):
        self.cur_step = 0
        self.cur_att_layer = 0

    def step_callback(self, x_t):
        return x_t

    def between_steps(self):
        return

    def tables(self):
        return None


class AttentionStore(AttentionControl):
    """AttentionStore (attention_control.py:214-248) has no effect on the denoised latents; natively it is a plain forward.
    (The only stored maps ever consumed are LocalBlend's five 16x16 cross maps, which the edit kernel accumulates itself.)

    keep_maps=True (opt-in; visualisation callers of the reference read them): the controller runs through the call-back path instead
    -- the probabilities of every attention site are materialised and this object is called on them exactly as the reference's is
    (:178-190) -- and `attention_store` / `get_average_attention()` hold the conditional-half maps of the <= 32^2-token sites, summed
    over steps (:221-239).  Slow (no flash attention, one host round trip per site)."""

    def __init__(self, keep_maps=False):
        super().__init__()
        self.keep_maps = bool(keep_maps)
        self._pnpi_force_callback = self.keep_maps
        self.step_store = self.get_empty_store()
        self.attention_store = {}

    @staticmethod
    def get_empty_store():
        return {"down_cross": [], "mid_cross": [], "up_cross": [], "down_self": [], "mid_self": [], "up_self": []}

    def forward(self, attn, is_cross, place_in_unet):
        key = f"{place_in_unet}_{'cross' if is_cross else 'self'}"
        if attn.shape[1] <= 32 ** 2:      # avoid memory overhead (:224)
            self.step_store[key].append(attn.clone())       # the engine re-uses the buffer behind `attn` at the next site
        return attn

    def __call__(self, attn, is_cross, place_in_unet):
        """AttentionControl.__call__ (:178-190, LOW_RESOURCE False): `forward` sees the conditional half; 32 sites make a step."""
        h = attn.shape[0]
        attn[h // 2:] = self.forward(attn[h // 2:], is_cross, place_in_unet)
        self.cur_att_layer += 1
        if self.cur_att_layer == self.num_att_layers:
            self.cur_att_layer = 0
            self.cur_step += 1
            self.between_steps()
        return attn

    def between_steps(self):
        if not self.keep_maps:
            return
        if len(self.attention_store) == 0:
            self.attention_store = self.step_store
        else:
            for key in self.attention_store:
                for i in range(len(self.attention_store[key])):
                    self.attention_store[key][i] += self.step_store[key][i]
        self.step_store = self.get_empty_store()

    def get_average_attention(self):
        return {key: [item / self.cur_step for item in self.attention_store[key]] for key in self.attention_store}

    def reset(self):
        super().reset()
        self.step_store = self.get_empty_store()
        self.attention_store = {}


class AttentionControlEdit(AttentionStore, abc.ABC):
    def __init__(self, prompts, num_steps, cross_replace_steps, self_replace_steps, local_blend, tokenizer=None, device="cuda"):
        super().__init__()
        if len(prompts) != 2:
            raise NotImplementedError("the native controllers handle one (source, target) prompt pair per image")
        self.batch_size = len(prompts)
        self.cross_replace_alpha = get_time_words_attention_alpha(prompts, num_steps, cross_replace_steps, tokenizer)
        if type(self_replace_steps) is float:
            self_replace_steps = 0, self_replace_steps
        self.num_self_replace = int(num_steps * self_replace_steps[0]), int(num_steps * self_replace_steps[1])
        self.local_blend = local_blend
        self.num_steps = num_steps

    def step_callback(self, x_t):
        """attention_control.py:253-256.  Only per-forward (level-1) drivers call this; the device-resident loops blend inside."""
        if self.local_blend is not None:
            x_t = self.local_blend(x_t, self.__dict__.get("_pnpi_engine"))
        return x_t

    # host tables ------------------------------------------------------------------------------------------------------
    def _mapper_alphas(self):
        raise NotImplementedError

    def _equalizer(self):
        return np.ones(MAX_NUM_WORDS, dtype=np.float32)

    def tables(self):
        mapper, alphas = self._mapper_alphas()
        lb = self.local_blend
        return ControllerTables(
            cross_alpha=self.cross_replace_alpha.reshape(self.num_steps + 1, MAX_NUM_WORDS).numpy(),
            mapper=mapper, alphas=alphas, equalizer=self._equalizer(), self_range=self.num_self_replace,
            lb_alpha=lb.alpha_layers.reshape(2, MAX_NUM_WORDS).numpy() if lb is not None else None,
            lb_start=lb.start_blend if lb is not None else 0, lb_threshold=lb.th[0] if lb is not None else 0.3,
            lb_sub_alpha=(lb.substruct_layers.reshape(2, MAX_NUM_WORDS).numpy()
                          if lb is not None and lb.substruct_layers is not None else None),
            lb_threshold_sub=lb.th[1] if lb is not None else 0.3)


class AttentionReplace(AttentionControlEdit):
    def __init__(self, prompts, num_steps, cross_replace_steps, self_replace_steps, local_blend=None, tokenizer=None, device="cuda"):
        super().__init__(prompts, num_steps, cross_replace_steps, self_replace_steps, local_blend, tokenizer, device)
        self.mapper = seq_aligner.get_replacement_mapper(prompts, tokenizer)     # [1, 77, 77], raises ValueError on word-count mismatch

    def _mapper_alphas(self):
        return self.mapper[0].numpy(), np.ones(MAX_NUM_WORDS, dtype=np.float32)


class AttentionRefine(AttentionControlEdit):
    def __init__(self, prompts, num_steps, cross_replace_steps, self_replace_steps, local_blend=None, tokenizer=None, device="cuda"):
        super().__init__(prompts, num_steps, cross_replace_steps, self_replace_steps, local_blend, tokenizer, device)
        self.mapper, alphas = seq_aligner.get_refinement_mapper(prompts, tokenizer)   # [1, 77] int64, [1, 77]
        self.alphas = alphas.reshape(alphas.shape[0], 1, 1, alphas.shape[1])

    def _mapper_alphas(self):
        # attn_base[:, :, mapper] (attention_control.py:320): column j of the result is source column mapper[j];
        # -1 follows Python indexing (last column) and always meets alphas[j] == 0.
        idx = self.mapper[0].numpy()
        m = np.zeros((MAX_NUM_WORDS, MAX_NUM_WORDS), dtype=np.float32)
        for j, w in enumerate(idx):
            m[int(w) % MAX_NUM_WORDS, j] = 1.0
        return m, self.alphas.reshape(-1).numpy()


class AttentionReweight(AttentionControlEdit):
    def __init__(self, prompts, num_steps, cross_replace_steps, self_replace_steps, equalizer, local_blend=None, controller=None,
                 device="cuda", tokenizer=None):
        super().__init__(prompts, num_steps, cross_replace_steps, self_replace_steps, local_blend, tokenizer, device)
        self.equalizer = equalizer
        self.prev_controller = controller

    def _mapper_alphas(self):
        if self.prev_controller is not None:
            return self.prev_controller._mapper_alphas()
        return np.eye(MAX_NUM_WORDS, dtype=np.float32), np.ones(MAX_NUM_WORDS, dtype=np.float32)

    def _equalizer(self):
        return self.equalizer.reshape(-1).numpy().astype(np.float32)


def make_controller(pipeline, prompts, is_replace_controller, cross_replace_steps, self_replace_steps, blend_words=None,
                    equilizer_params=None, num_ddim_steps=50, device="cuda") -> AttentionControlEdit:
    """attention_control.py:366-405"""
    lb = None if blend_words is None else LocalBlend(prompts, blend_words, tokenizer=pipeline.tokenizer, device=device,
                                                      num_ddim_steps=num_ddim_steps)
    cls = AttentionReplace if is_replace_controller else AttentionRefine
    controller = cls(prompts, num_ddim_steps, cross_replace_steps=cross_replace_steps, self_replace_steps=self_replace_steps,
                     local_blend=lb, tokenizer=pipeline.tokenizer)
    if equilizer_params is not None:
        eq = get_equalizer(prompts[1], equilizer_params["words"], equilizer_params["values"], tokenizer=pipeline.tokenizer)
        controller = AttentionReweight(prompts, num_ddim_steps, cross_replace_steps=cross_replace_steps,
                                       self_replace_steps=self_replace_steps, equalizer=eq, local_blend=lb, controller=controller,
                                       tokenizer=pipeline.tokenizer)
    return controller


def controller_tables(controller):
    """Host tables (kernel descriptor) of a registered controller, None for "no edit".  An object without `.tables()` -- e.g. a user
    subclass of the reference's callback classes (models/p2p/attention_control.py:151-363) -- has no descriptor: it runs through the
    level-1 call-back path (NativeUNet.__call__, and the Python step loops of p2p_guidance_forward for callback controllers), never
    through a loop entry point of the library, so asking for its tables is an error."""
    if controller is None:
        return None
    if not hasattr(controller, "tables"):
        raise TypeError("controller %s has no kernel descriptor (.tables()); controllers of the reference's call-back protocol "
                        "`controller(attn, is_cross, place)` are supported through model.unet(...) and the guidance-forward functions "
                        "(level 1), not through the device-resident loop entry points" % type(controller).__name__)
    return controller.tables()


def is_callback_controller(controller):
    """A controller object of the reference's protocol (models/p2p/attention_control.py:151-190: `controller(attn, is_cross,
    place_in_unet)` at every attention site) for which the library has no kernel descriptor: it runs through the
    materialise-and-call-back path of pnpi_set_attention_callback (slow, exact semantics)."""
    if controller is None or not callable(controller):
        return False
    if isinstance(controller, AttentionControlEdit):        # native edit classes: the flag (set by hand) is ignored, see force_callback
        return False
    return getattr(controller, "_pnpi_force_callback", False) or not hasattr(controller, "tables")


class ForeignControllerAdapter:
    """An instance of the REFERENCE's own controller classes (models/p2p/attention_control.py, unmodified) whose edit the kernels know:
    AttentionReplace / AttentionRefine / AttentionReweight, with or without LocalBlend.  Their attributes are the same tensors this module's
    classes build (mapper, alphas, equalizer, cross_replace_alpha, num_self_replace, LocalBlend's alpha_layers / substruct_layers /
    start_blend / th), so the descriptor is read off the object.  Step bookkeeping (cur_step, between_steps, LocalBlend.counter)
    stays on the wrapped object, where the reference's loop code reads it."""

    def __init__(self, wrapped):
        self.__dict__["wrapped"] = wrapped
        if getattr(wrapped, "local_blend", None) is not None:
            # the reference's loop calls `controller.step_callback(latents)` on ITS object (p2p_guidance_forward.py:62,116), whose
            # LocalBlend reads attention_store -- empty here, the kernels keep the five 16 x 16 maps in the library.  The instance
            # gets a step_callback that runs the same blend natively (counter / start_blend stay on the reference's LocalBlend).
            wrapped.step_callback = lambda x_t: _native_local_blend(wrapped.local_blend, x_t, wrapped.__dict__.get("_pnpi_engine"))

    def __getattr__(self, name):
        return getattr(self.wrapped, name)

    def __setattr__(self, name, value):
        setattr(self.wrapped, name, value)

    def tables(self):
        return _tables_from_attributes(self.wrapped)


def _mapper_alphas_from_attributes(c):
    name = type(c).__name__
    ones = np.ones(MAX_NUM_WORDS, dtype=np.float32)
    if name == "AttentionReplace":
        return c.mapper[0].detach().float().cpu().numpy(), ones
    if name == "AttentionRefine":
        idx = c.mapper[0].detach().cpu().numpy()
        m = np.zeros((MAX_NUM_WORDS, MAX_NUM_WORDS), dtype=np.float32)
        for j, w in enumerate(idx):
            m[int(w) % MAX_NUM_WORDS, j] = 1.0
        return m, c.alphas.detach().float().cpu().reshape(-1).numpy()
    if name == "AttentionReweight":
        prev = getattr(c, "prev_controller", None)
        return _mapper_alphas_from_attributes(prev) if prev is not None else (np.eye(MAX_NUM_WORDS, dtype=np.float32), ones)
    raise LookupError(name)


def _tables_from_attributes(c):
    cra = c.cross_replace_alpha.detach().float().cpu()
    steps = cra.shape[0] - 1
    mapper, alphas = _mapper_alphas_from_attributes(c)
    eq = (c.equalizer.detach().float().cpu().reshape(-1).numpy() if type(c).__name__ == "AttentionReweight"
          else np.ones(MAX_NUM_WORDS, dtype=np.float32))
    lb = getattr(c, "local_blend", None)
    lbk = {}
    if lb is not None:
        rows = lambda t: t.detach().float().cpu().reshape(2, MAX_NUM_WORDS).numpy()
        sub = getattr(lb, "substruct_layers", None)
        lbk = dict(lb_alpha=rows(lb.alpha_layers), lb_start=int(lb.start_blend), lb_threshold=float(lb.th[0]),
                   lb_sub_alpha=rows(sub) if sub is not None else None, lb_threshold_sub=float(lb.th[1]))
    return ControllerTables(cross_alpha=cra.reshape(steps + 1, MAX_NUM_WORDS).numpy(), mapper=mapper, alphas=alphas, equalizer=eq,
                            self_range=tuple(int(x) for x in c.num_self_replace), **lbk)


def adapt_foreign_controller(controller):
    """What NativeUNet does with the controller the reference's unmodified register_attention_control closed over:
      * its DummyController (controller=None, attention_control.py:49-56), EmptyControl and a plain AttentionStore: no edit -> None
        (AttentionStore's stored maps are then NOT populated: only visualisation reads them; keep them by passing the object through
        `force_callback(controller)`);
      * this module's own classes: unchanged (they carry `.tables()`);
      * the reference's AttentionReplace / Refine / Reweight, with or without LocalBlend: the kernel descriptor read off their
        attributes (with LocalBlend the instance's step_callback is pointed at the native blend, see ForeignControllerAdapter);
      * anything else callable (user subclasses): the call-back path, exact and slow."""
    if controller is None or getattr(controller, "_pnpi_force_callback", False) or hasattr(controller, "tables"):
        return controller
    name = type(controller).__name__
    if name in ("DummyController", "EmptyControl"):
        return None
    if name == "AttentionStore":
        return _NoEditAdapter(controller)
    if name in ("AttentionReplace", "AttentionRefine", "AttentionReweight"):
        try:
            _tables_from_attributes(controller)
        except (LookupError, AttributeError, IndexError, TypeError, ValueError, RuntimeError):
            return controller
        return ForeignControllerAdapter(controller)
    return controller


class _NoEditAdapter(ForeignControllerAdapter):
    """A foreign AttentionStore: a plain forward, with the step bookkeeping kept on the wrapped object."""

    def tables(self):
        return None


def force_callback(controller):
    """Mark a controller so that it always runs through the call-back path (e.g. an AttentionStore whose stored maps are wanted).
    This module's own AttentionReplace / Refine / Reweight carry their edit as kernel tables, not as a Python `forward`: forcing them
    through the call-back path would silently run a plain, unedited forward, so that is an error."""
    if isinstance(controller, AttentionControlEdit):
        raise TypeError("force_callback: %s edits through the kernel descriptor (tables()) and has no Python forward to call back into; "
                        "force an AttentionStore or a controller object that implements forward()" % type(controller).__name__)
    controller._pnpi_force_callback = True
    return controller


def register_attention_control(model, controller):
    """The reference patches 32 CrossAttention.forward methods here (attention_control.py:12-81).  The native UNet has no
    Python attention modules; registration just hands the controller to the pipeline's UNet, which turns it into the kernel
    descriptor at the next call.  (The reference's OWN unmodified function works too: NativeUNet.named_children() yields markers of
    class CrossAttention whose `.forward` assignment does the same.)"""
    model.unet.set_controller(adapt_foreign_controller(controller))
    if controller is not None and hasattr(controller, "num_att_layers"):
        controller.num_att_layers = model.unet.num_att_layers
