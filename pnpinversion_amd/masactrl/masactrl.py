"""MutualSelfAttentionControl with the constructor of models/masactrl/masactrl.py:14-39.  Semantics (:57-69): at denoising steps
>= start_step and transformer blocks >= start_layer (execution order 0..15), every self-attention row of a CFG half reads the K
and V of the half's FIRST row (the source image) -- in the kernel a row-indirection table of the flash-attention launch.
`layer_idx` / `step_idx` lists (any subsets) replace the windows, as in the reference."""
from ..engine import MasaCtrlTables
from .masactrl_utils import AttentionBase


class MutualSelfAttentionControl(AttentionBase):
    MODEL_TYPE = {"SD": 16, "SDXL": 70}

    def __init__(self, start_step=4, start_layer=10, layer_idx=None, step_idx=None, total_steps=50, model_type="SD"):
        super().__init__()
        self.total_steps = total_steps
        self.total_layers = self.MODEL_TYPE.get(model_type, 16)
        self.start_step = start_step
        self.start_layer = start_layer
        self.layer_idx = layer_idx if layer_idx is not None else list(range(start_layer, self.total_layers))
        self.step_idx = step_idx if step_idx is not None else list(range(start_step, total_steps))

    def tables(self):
        # the reference tests membership (`cur_step not in self.step_idx`, `cur_att_layer // 2 not in self.layer_idx`, masactrl.py:61):
        # the lists travel as a 16-bit block mask and a per-step byte array; the default windows keep their two integers
        win_l = self.layer_idx == list(range(self.start_layer, self.total_layers))
        win_s = self.step_idx == list(range(self.start_step, self.total_steps))
        return MasaCtrlTables(self.start_step, self.start_layer, None if win_l else self.layer_idx, None if win_s else self.step_idx)
