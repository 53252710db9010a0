"""DDIMSchedulerDev with the interface of models/p2p/scheduler_dev.py (+ the diffusers-0.10.0 DDIMScheduler base it extends):
tables, `set_timesteps`, `step`.  `step` runs the fused HIP update (eta = 0, epsilon prediction, no clipping -- the only mode the
reference's P2P path uses, models/p2p_editor.py:18-22).  Error behaviour follows scheduler_dev.py:22-25."""
import numpy as np
import torch


class _Config(dict):
    __getattr__ = dict.__getitem__


class DDIMSchedulerOutput(dict):
    def __init__(self, prev_sample=None, pred_original_sample=None):
        super().__init__(prev_sample=prev_sample, pred_original_sample=pred_original_sample)
        self.prev_sample = prev_sample
        self.pred_original_sample = pred_original_sample


class DDIMSchedulerDev:
    def __init__(self, num_train_timesteps=1000, beta_start=0.0001, beta_end=0.02, beta_schedule="linear", clip_sample=True,
                 set_alpha_to_one=True, steps_offset=0, prediction_type="epsilon"):
        if beta_schedule == "scaled_linear":
            self.betas = torch.linspace(beta_start ** 0.5, beta_end ** 0.5, num_train_timesteps, dtype=torch.float32) ** 2
        elif beta_schedule == "linear":
            self.betas = torch.linspace(beta_start, beta_end, num_train_timesteps, dtype=torch.float32)
        else:
            raise NotImplementedError(f"{beta_schedule} does is not implemented for {self.__class__}")
        self.alphas = 1.0 - self.betas
        self.alphas_cumprod = torch.cumprod(self.alphas, dim=0)
        self.final_alpha_cumprod = torch.tensor(1.0) if set_alpha_to_one else self.alphas_cumprod[0]
        self.init_noise_sigma = 1.0
        self.num_inference_steps = None
        self.timesteps = torch.from_numpy(np.arange(0, num_train_timesteps)[::-1].copy().astype(np.int64))
        self.config = _Config(num_train_timesteps=num_train_timesteps, prediction_type=prediction_type, steps_offset=steps_offset,
                              clip_sample=clip_sample, beta_start=beta_start, beta_end=beta_end, beta_schedule=beta_schedule)
        self._engine = None

    def bind(self, engine):
        self._engine = engine
        engine.set_scheduler(self.alphas_cumprod.numpy(), float(self.final_alpha_cumprod))

    def set_timesteps(self, num_inference_steps, device=None):
        self.num_inference_steps = num_inference_steps
        ratio = self.config.num_train_timesteps // num_inference_steps
        ts = (np.arange(0, num_inference_steps) * ratio).round()[::-1].copy().astype(np.int64)
        self.timesteps = torch.from_numpy(ts + self.config.steps_offset)

    @property
    def step_ratio(self):
        return self.config.num_train_timesteps // self.num_inference_steps

    def step(self, model_output, timestep, sample, eta=0.0, use_clipped_model_output=False, generator=None, variance_noise=None,
             return_dict=True, **kwargs):
        if self.num_inference_steps is None:
            raise ValueError("Number of inference steps is 'None', you need to run 'set_timesteps' after creating the scheduler")
        if eta > 0 or use_clipped_model_output or kwargs.get("clip_sample", False) or self.config.prediction_type != "epsilon":
            raise NotImplementedError("the native scheduler step implements eta = 0 / epsilon prediction / no clipping")
        if self._engine is None:
            raise RuntimeError("scheduler is not bound to a NativeEngine")
        if kwargs.get("ref_image", None) is not None and kwargs.get("recon_lr", 0.0) > 0.0:
            # scheduler_dev.py:68-76: pred_x0 -= recon_lr * (pred_x0 - ref_image) [* recon_mask], one fused launch
            prev, x0 = self._engine.ddim_prev_step_recon(model_output, int(timestep), self.step_ratio, sample, kwargs["ref_image"],
                                                         kwargs["recon_lr"], kwargs.get("recon_mask", None))
            return DDIMSchedulerOutput(prev_sample=prev, pred_original_sample=x0) if return_dict else (prev,)
        prev = self._engine.ddim_prev_step(model_output, int(timestep), self.step_ratio, sample)
        if not return_dict:
            return (prev,)
        return DDIMSchedulerOutput(prev_sample=prev, pred_original_sample=None)
