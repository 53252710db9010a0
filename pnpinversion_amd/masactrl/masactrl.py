"""MutualSelfAttentionControl with the constructor of models/masactrl/masactrl.py:14-39.  Semantics (:57-69): at denoising steps
>= start_step and transformer blocks >= start_layer (execution order 0..15), every self-attention row of a CFG half reads the K
and V of the half's FIRST row (the source image) -- in the kernel a row-indirection table of the flash-attention launch."""
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
        if self.layer_idx != list(range(start_layer, self.total_layers)) or self.step_idx != list(range(start_step, total_steps)):
            raise NotImplementedError("the native path takes contiguous [start_step, total) x [start_layer, 16) windows only")

    def tables(self):
        return MasaCtrlTables(self.start_step, self.start_layer)
