"""Shared helpers for the -m gpu tests: a small-config libpnpi context and pointer plumbing."""
import ctypes as C

import torch

from pnpinversion_amd import _capi


def ptr(t):
    return C.c_void_p(t.data_ptr()) if t is not None else None


def tiny_config(sample_size=16, boc=(32, 64, 64, 64), vae_boc=(32, 32, 64, 64), heads=8, cross_dim=64, layers=1):
    cfg = _capi.ModelConfig()
    _capi.load_library().pnpi_config_sd1(cfg)
    for i in range(4):
        cfg.block_out_channels[i] = boc[i]
        cfg.vae_block_out_channels[i] = vae_boc[i]
    cfg.sample_size = sample_size
    cfg.heads = heads
    cfg.cross_dim = cross_dim
    cfg.layers_per_block = layers
    cfg.vae_layers_per_block = layers
    cfg.clip_layers = 0            # kernel-level tests: no text encoder in the context
    return cfg


class Ctx:
    """RAII wrapper of pnpi_ctx on torch's current HIP stream."""

    def __init__(self, cfg=None, max_rows=4, max_vae=1):
        self.lib = _capi.load_library()
        self.cfg = cfg or tiny_config()
        self.h = C.c_void_p()
        stream = torch.cuda.current_stream().cuda_stream
        st = self.lib.pnpi_create(C.byref(self.h), C.byref(self.cfg), torch.cuda.current_device(), C.c_void_p(stream),
                                  max_rows, max_vae)
        if st != 0:
            msg = self.lib.pnpi_last_error(self.h)
            self.close()
            raise RuntimeError("pnpi_create failed: %d %s" % (st, msg))

    def call(self, name, *args):
        st = getattr(self.lib, name)(self.h, *args)
        _capi.check(self.lib, self.h, st)

    def close(self):
        if self.h:
            self.lib.pnpi_destroy(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def rel_err(a, b):
    a = a.float()
    b = b.float()
    return ((a - b).norm() / (b.norm() + 1e-12)).item()


def max_err(a, b):
    return (a.float() - b.float()).abs().max().item()
