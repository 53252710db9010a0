"""AttentionBase / regiter_attention_editor_diffusers with the names of models/masactrl/masactrl_utils.py:14-41,85-144.
On the native pipeline an "attention editor" is not a Python callback inside the UNet: it is translated into the kernel-side
descriptor (engine.MasaCtrlTables -> pnpi_ctrl_desc kind 2); `regiter_attention_editor_diffusers` just hands it to the model."""


class AttentionBase:
    """masactrl_utils.py:14-41: the identity editor (plain attention)."""

    def __init__(self):
        self.cur_step = 0
        self.num_att_layers = -1
        self.cur_att_layer = 0

    def after_step(self):
        pass

    def reset(self):
        self.cur_step = 0
        self.cur_att_layer = 0

    def tables(self):
        return None


def regiter_attention_editor_diffusers(model, editor: AttentionBase):
    """masactrl_utils.py:85-144 (the reference hooks the 32 `Attention` modules; the count is what it stores)."""
    model.masactrl_editor = editor
    editor.num_att_layers = model.engine.cfg.n_attention_layers if hasattr(model.engine.cfg, "n_attention_layers") else 32
