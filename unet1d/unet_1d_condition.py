"""``UNet1DConditionModel`` — the reference's denoiser API on the MI355X HIP engine.

Boundary kept:
  * ctor kwargs of reference ``unet1d/unet_1d_condition.py:151-203`` as NS2VC passes
    them (``model.py:391-400``); unknown-but-default kwargs are accepted;
  * an ``nn.Module`` whose parameters carry EXACTLY the reference's 701 state-dict
    names/shapes, so ``load_state_dict(strict=True)`` of a reference checkpoint works;
  * ``forward(sample, timestep, encoder_hidden_states, ..., encoder_attention_mask=,
    return_dict=)`` -> object with ``.sample`` (``:743-757, 1034-1037``).

The inference forward is NOT PyTorch: it hands device pointers to libns2vc_hip.so;
without a GPU / with CPU tensors it raises (there is no CPU inference path).  TRAINING
(autograd recording: ``model.py:720`` under ``Trainer.train``) is the one case routed to
plain PyTorch ops (``unet1d/torch_path.py``) so that ``train.py`` stays drop-in; the
counters ``engine_calls`` / ``autograd_calls`` tell which path ran.  The fast path for sampling
is ``ns2vc_amd.pipeline.Denoiser`` (captured loop, condition hoisted once).  Here the
reference API concatenates x and content into a NEW ``sample`` tensor on every solver
step (``model.py:409``), so the content half of conv_in is redone per call, but the
prompt-side hoisting (all 32 cross-attention K/V projections, ``add_embedding``: 92 %
of the step-invariant FLOPs) is cached for as long as the caller keeps passing the same
prompt storage unmodified (``(data_ptr, _version, shape, stride)``; the keyed tensor is
kept alive, so its address cannot be recycled by the allocator while it is the key);
the tiny mask -> bias conversion is refreshed on every call.
Engine precision: ``engine_precision=`` / env ``NS2VC_PRECISION`` (fp32 | fp16 | bf16;
default fp32, the reference's arithmetic).
"""
from __future__ import annotations

import os
from dataclasses import dataclass
from typing import Any, Dict, Optional, Tuple, Union

import torch
from torch import nn

from ns2vc_amd.spec import UNetConfig, param_spec


@dataclass
class UNet1DConditionOutput:
    sample: torch.Tensor

    def __getitem__(self, i):          # tuple-style access like the reference's BaseOutput
        return (self.sample,)[i]


class _Node(nn.Module):
    """bare container so dotted reference names map onto nested modules"""


def _init_(name: str, p: torch.Tensor) -> None:
    with torch.no_grad():
        if ".norm" in name or name.startswith("conv_norm_out"):
            p.fill_(1.0) if name.endswith("weight") else p.zero_()
        elif name.endswith("positional_embedding"):
            p.normal_(0.0, 1.0).div_(p.shape[-1] ** 0.5)
        elif p.ndim >= 2:
            fan_in = p[0].numel()
            p.uniform_(-1.0, 1.0).mul_(fan_in ** -0.5)
        else:
            p.uniform_(-1.0, 1.0).mul_(max(p.numel(), 1) ** -0.5)


class UNet1DConditionModel(nn.Module):
    def __init__(self, sample_size: Optional[int] = None, in_channels: int = 4, out_channels: int = 4,
                 block_out_channels: Tuple[int, ...] = (320, 640, 1280, 1280), layers_per_block: int = 2,
                 norm_num_groups: int = 32, cross_attention_dim: int = 1280, attention_head_dim: int = 8,
                 addition_embed_type: Optional[str] = None, resnet_time_scale_shift: str = "default",
                 engine_precision: Optional[str] = None, **kwargs: Any):
        super().__init__()
        if not isinstance(block_out_channels, (tuple, list)):
            raise ValueError("block_out_channels must be a tuple")
        if isinstance(attention_head_dim, (tuple, list)):
            if len(set(attention_head_dim)) != 1:
                raise ValueError("per-block attention_head_dim is not supported by the HIP engine")
            attention_head_dim = attention_head_dim[0]
        cfg = UNetConfig(in_channels=in_channels, out_channels=out_channels, block_out_channels=tuple(block_out_channels),
                         norm_num_groups=norm_num_groups, cross_attention_dim=cross_attention_dim,
                         attention_head_dim=attention_head_dim, layers_per_block=layers_per_block,
                         addition_embed_type=addition_embed_type or "", resnet_time_scale_shift=resnet_time_scale_shift)
        cfg.validate()          # ValueError on configurations outside NS2VC's (reference :222-255 raises ValueError too)
        self.cfg = cfg
        # the reference exposes its ctor arguments as a dict (Appendix C: self.config[...])
        self.config: Dict[str, Any] = dict(sample_size=sample_size, in_channels=in_channels, out_channels=out_channels,
                                           block_out_channels=tuple(block_out_channels), layers_per_block=layers_per_block,
                                           norm_num_groups=norm_num_groups, cross_attention_dim=cross_attention_dim,
                                           attention_head_dim=attention_head_dim, addition_embed_type=addition_embed_type,
                                           resnet_time_scale_shift=resnet_time_scale_shift, center_input_sample=False,
                                           only_cross_attention=False, **kwargs)
        for name, shape in param_spec(cfg).items():
            node: nn.Module = self
            *path, leaf = name.split(".")
            for part in path:
                if not hasattr(node, part):
                    node.add_module(part, _Node())
                node = getattr(node, part)
            p = nn.Parameter(torch.empty(shape))
            _init_(name, p)
            node.register_parameter(leaf, p)
        self.engine_precision = engine_precision or os.environ.get("NS2VC_PRECISION", "fp32")
        self._engine = None
        self._engine_key = None
        self._engine_shape = None
        self._plist = None              # flat parameter list (the key walk is per call: keep it a list comprehension)
        self._prompt_key = None         # ((data_ptr, _version, shape) of prompt and mask) the engine's prompt half was built from
        self._prompt_hold = None        # ... and the tensors themselves: while they live their storage cannot be re-used
        self.prompt_hoists = 0          # how often the prompt half of the condition was (re)computed: tests / diagnostics
        self.engine_calls = 0           # forwards served by the HIP engine (inference)
        self.autograd_calls = 0         # forwards served by unet1d/torch_path.py (training: autograd was recording)
        self._torch_path = None

    # ---------------------------------------------------------------------------------
    def _weights_key(self):
        if self._plist is None or len(self._plist) != len(self._parameters_flat()):
            self._plist = self._parameters_flat()
        return (self.engine_precision, tuple([(p.data_ptr(), p._version) for p in self._plist]))

    def _parameters_flat(self):
        return list(self.parameters())

    def _apply(self, fn, *a, **kw):      # .to() / .cuda() / .half() may replace parameter storage
        self._plist = None
        self._prompt_key = self._prompt_hold = None
        return super()._apply(fn, *a, **kw)

    def _get_engine(self):
        from ns2vc_amd.engine import Engine
        key = self._weights_key()
        if self._engine is None or self._engine_key != key:
            if self._engine is None or self._engine.precision != self.engine_precision:
                self._engine = Engine(self.cfg, precision=self.engine_precision)
            self._engine.load_state_dict({k: v for k, v in self.state_dict().items()})
            self._engine_key = key
            self._engine_shape = None
            self._prompt_key = self._prompt_hold = None
        return self._engine

    def forward(self, sample: torch.Tensor, timestep: Union[torch.Tensor, float, int], encoder_hidden_states: torch.Tensor,
                class_labels=None, timestep_cond=None, attention_mask=None, cross_attention_kwargs=None, added_cond_kwargs=None,
                down_block_additional_residuals=None, mid_block_additional_residual=None,
                encoder_attention_mask: Optional[torch.Tensor] = None, return_dict: bool = True):
        for nm, v in (("class_labels", class_labels), ("timestep_cond", timestep_cond), ("attention_mask", attention_mask),
                      ("down_block_additional_residuals", down_block_additional_residuals),
                      ("mid_block_additional_residual", mid_block_additional_residual)):
            if v is not None:
                raise NotImplementedError(f"{nm} is not used by NS2VC and not supported by the HIP engine")
        cfg = self.cfg
        if torch.is_grad_enabled() and (sample.requires_grad or any(p.requires_grad for p in self.parameters())):
            # TRAINING: autograd is recording (model.py:720).  Plain PyTorch ops on this module's parameters; never taken
            # under torch.no_grad(), i.e. never by NaturalSpeech2.sample / Svc.infer / the benchmarks.
            if self._torch_path is None:
                from .torch_path import TorchDenoiser
                self._torch_path = TorchDenoiser(self, cfg)
            Bq = sample.shape[0]
            tt = timestep if torch.is_tensor(timestep) else torch.tensor([float(timestep)], device=sample.device)
            tt = tt.to(sample.device).reshape(-1).expand(Bq)
            mask_b = None if encoder_attention_mask is None else encoder_attention_mask.to(sample.device).reshape(Bq, -1).bool()
            out = self._torch_path(sample, tt, encoder_hidden_states, mask_b)
            self.autograd_calls += 1
            return UNet1DConditionOutput(sample=out) if return_dict else (out,)
        if not sample.is_cuda:
            raise RuntimeError("UNet1DConditionModel (HIP engine) needs CUDA/ROCm tensors for inference: there is no CPU inference path "
                               "(autograd / training calls use PyTorch ops: unet1d/torch_path.py)")
        B, Cin, T = sample.shape
        if Cin != cfg.in_channels:
            raise RuntimeError(f"expected {cfg.in_channels} input channels, got {Cin}")
        if encoder_hidden_states.shape[0] != B or encoder_hidden_states.shape[2] != cfg.cross_attention_dim:
            raise RuntimeError("encoder_hidden_states must be (B, Lp, cross_attention_dim)")
        Lp = encoder_hidden_states.shape[1]
        dev = sample.device
        ts = timestep
        if not torch.is_tensor(ts):
            ts = torch.tensor([float(ts)], dtype=torch.float32, device=dev)
        ts = ts.to(device=dev, dtype=torch.float32).reshape(-1).expand(B).contiguous()     # int64 (training / DDIM, model.py:580,714) or float
        x = sample[:, :cfg.latent_channels].to(torch.float32).contiguous()
        content = sample[:, cfg.latent_channels:].to(torch.float32).contiguous()
        prompt = encoder_hidden_states.to(torch.float32).contiguous()
        mask = None
        if encoder_attention_mask is not None:
            mask = encoder_attention_mask.to(device=dev).reshape(B, Lp).to(torch.uint8).contiguous()
        with torch.cuda.device(dev):     # engine creation / weights / workspace / launches all bind to the tensors' device
            eng = self._get_engine()
            if self._engine_shape != (B, T, Lp):
                torch.cuda.synchronize(dev)
                eng.prepare(B, T, Lp)
                self._engine_shape = (B, T, Lp)
                self._prompt_key = self._prompt_hold = None
            out = torch.empty((B, cfg.out_channels, T), dtype=torch.float32, device=dev)
            stream = torch.cuda.current_stream(dev)

            ehs = encoder_hidden_states
            pkey = (ehs.data_ptr(), ehs._version, tuple(ehs.shape), tuple(ehs.stride()), ehs.dtype, stream.cuda_stream)
            if pkey != self._prompt_key:
                eng.set_prompt(prompt, mask, stream=stream)
                self._prompt_key = pkey
                self._prompt_hold = ehs          # alive => its address cannot be handed to another tensor while it is the key
                self.prompt_hoists += 1
            else:                                # the reference rebuilds the (tiny) mask per call (model.py:412): always refresh it
                eng.set_mask(mask, stream=stream)
            eng.set_content(content, stream=stream)
            eng.forward(x, ts, out, stream=stream)
            self.engine_calls += 1
        out = out.to(sample.dtype)
        if not return_dict:
            return (out,)
        return UNet1DConditionOutput(sample=out)
