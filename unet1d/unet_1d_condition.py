"""``UNet1DConditionModel`` — the reference's denoiser API on the MI355X HIP engine.

Boundary kept:
  * ctor kwargs of reference ``unet1d/unet_1d_condition.py:151-203`` as NS2VC passes
    them (``model.py:391-400``); unknown-but-default kwargs are accepted;
  * an ``nn.Module`` whose parameters carry EXACTLY the reference's 701 state-dict
    names/shapes, so ``load_state_dict(strict=True)`` of a reference checkpoint works;
  * ``forward(sample, timestep, encoder_hidden_states, ..., encoder_attention_mask=,
    return_dict=)`` -> object with ``.sample`` (``:743-757, 1034-1037``).

The inference forward on GPU tensors is NOT PyTorch: it hands device pointers to
libns2vc_hip.so (a missing library is an error, never a silent fallback).  Two cases are
routed to plain PyTorch ops on this module's own parameters (``unet1d/torch_path.py``,
product code, never ``oracle/``): TRAINING (autograd recording: ``model.py:720`` under
``Trainer.train``), so that ``train.py`` stays drop-in, and inference on CPU tensors
(``infer.py --device cpu``, BASELINE config 1 "plumbing, no GPU"), which warns once that
it is the slow path.  The counters ``engine_calls`` / ``autograd_calls`` / ``cpu_calls``
tell which path ran.  The fast path for sampling
is ``ns2vc_amd.pipeline.Denoiser`` (captured loop, condition hoisted once).  Here the
reference API concatenates x and content into a NEW ``sample`` tensor on every solver
step (``model.py:409``), so the content half of conv_in is redone per call, but the
prompt-side hoisting (all 32 cross-attention K/V projections, ``add_embedding``: 92 %
of the step-invariant FLOPs) is cached for as long as the caller keeps passing the same
prompt storage unmodified (``(data_ptr, _version, shape, stride)``; the keyed tensor is
kept alive, so its address cannot be recycled by the allocator while it is the key);
the tiny mask -> bias conversion is refreshed on every call.
Engine precision: ``engine_precision=`` / env ``NS2VC_PRECISION`` (auto | fp32 | fp16 | bf16).  Default since round 4: ``auto`` -- the
fp16 engine (7e-4 from the reference's fp32 arithmetic on one evaluation, 3.6x faster than the exact-fp32 engine), MEASURED once per
set of weights against the exact-fp32 engine on the caller's own first inputs (relative L2 over the batch and of the worst utterance,
``precision_error_seen`` / ``precision_error_worst_item``); above ``precision_check`` (1e-3, the parity bar) the module warns and
serves from the fp32 engine from then on.  So the zero-change drop-in is the fast engine where that is inside the bar and the exact one
where it is not -- it used to be the exact-fp32 engine always, 14 ms instead of 3.8 ms per step at the bench shape, while
``ns2vc_amd.pipeline.Denoiser`` defaulted to fp16.  ``fp32`` / ``fp16`` / ``bf16`` select an engine outright (no check).
"""
from __future__ import annotations

import os
import warnings
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
        self.engine_precision = engine_precision or os.environ.get("NS2VC_PRECISION", "auto")
        if self.engine_precision not in ("auto", "fp32", "f32", "fp16", "f16", "bf16"):
            raise ValueError(f"engine_precision must be auto | fp32 | fp16 | bf16, got {self.engine_precision!r}")
        self._auto = self.engine_precision == "auto"
        self._precision = "fp16" if self._auto else self.engine_precision      # what the engine is built in (auto: until the check says otherwise)
        self.precision_check: Optional[float] = 1e-3 if self._auto else None  # auto only: threshold of the one-time fp16-vs-fp32 measurement
        self.precision_error_seen: Optional[float] = None
        self.precision_error_worst_item: Optional[float] = None
        self._precision_checked_key = None                                    # weights key the verdict belongs to
        # r5 (round-4 advice): the first point of a sampling trajectory is not the worst one for the 16-bit engine (ns2vc_amd.pipeline.Denoiser
        # measures three; the last is ~5 % worse), so the same weights are measured ONCE MORE at the first call with a late timestep
        # (max t < late_check_below); and a module whose weights change all the time (evaluation between optimizer steps) does not pay an
        # fp32 engine per change: at most one measurement per `check_min_interval_s` seconds, the last verdict standing in between.
        self.late_check_below: float = 100.0
        self.check_min_interval_s: float = 30.0
        self._late_checked_key = None
        self._last_check_time = None
        self._last_verdict_ok: Optional[bool] = None                         # outcome of the last measurement (None: none yet); r6: a demotion is NOT forgotten inside the interval
        self.precision_checks = 0                                             # fp16-vs-fp32 measurements taken so far (diagnostics / tests)
        self._engine = None
        self._engine_key = None
        self._engine_shape = None
        self._prompt_key = None         # ((data_ptr, _version, shape) of prompt and mask) the engine's prompt half was built from
        self._prompt_hold = None        # ... and the tensors themselves: while they live their storage cannot be re-used
        self.prompt_hoists = 0          # how often the prompt half of the condition was (re)computed: tests / diagnostics
        self.engine_calls = 0           # forwards served by the HIP engine (inference)
        self.autograd_calls = 0         # forwards served by unet1d/torch_path.py (training: autograd was recording)
        self.cpu_calls = 0              # no_grad forwards on CPU tensors, served by unet1d/torch_path.py (plumbing path)
        self._torch_path = None
        self._warned = set()
        # LayerNorm-by-linearity guard of the engine (ns2vc_amd.pipeline.Denoiser has the same one): threshold on the
        # engine's measured max |mean|/std of the LayerNorm rows; None disables.  Checked on the first call of a shape
        # (one wait on the current stream; above it the plan switches to explicit LayerNorm passes and the call is
        # redone), afterwards without blocking (enqueued behind a call, collected at the next one).
        self.ln_guard: Optional[float] = 32.0 if self._precision in ("fp32", "f32") else 8.0
        self.ln_ratio_seen: Optional[float] = None
        self._ln_checked = False
        self._ln_pending = False

    # ---------------------------------------------------------------------------------
    def _weights_key(self):
        # walks the live parameters on every call (4.13 vs 3.97 ms per call measured: tools/dropin_overhead.py): a rebound
        # parameter (m.conv_in.weight = nn.Parameter(...), parametrize, weight_norm) changes data_ptr, an in-place update
        # (optimizer step, .copy_) changes _version -- either reloads the engine's packed weights
        return (self._precision, tuple([(p.data_ptr(), p._version) for p in self.parameters()]))

    def _apply(self, fn, *a, **kw):      # .to() / .cuda() / .half() may replace parameter storage
        self._prompt_key = self._prompt_hold = None
        return super()._apply(fn, *a, **kw)

    def _warn_once(self, tag: str, msg: str) -> None:
        if tag not in self._warned:
            self._warned.add(tag)
            warnings.warn(msg)

    def _run_torch_path(self, sample, timestep, encoder_hidden_states, encoder_attention_mask):
        if self._torch_path is None:
            from .torch_path import TorchDenoiser
            self._torch_path = TorchDenoiser(self, self.cfg)
        Bq = sample.shape[0]
        tt = timestep if torch.is_tensor(timestep) else torch.tensor([float(timestep)], device=sample.device)
        tt = tt.to(sample.device).reshape(-1).expand(Bq)
        mask_b = None if encoder_attention_mask is None else encoder_attention_mask.to(sample.device).reshape(Bq, -1).bool()
        return self._torch_path(sample, tt, encoder_hidden_states, mask_b)

    def _get_engine(self):
        from ns2vc_amd.engine import Engine
        key = self._weights_key()
        if self._engine is None or self._engine_key != key:
            if self._auto and self._precision != "fp16" and self._precision_checked_key is not None and key[1] != self._precision_checked_key \
                    and self._check_due():
                # new weights (an optimizer step, a reloaded checkpoint): the fallback verdict belonged to the old ones -- measure again.  r6 (ADVICE r5,
                # medium): only when a measurement is DUE.  Inside check_min_interval_s the module keeps serving the new weights from the fp32 engine
                # (the last verdict was a demotion: the rate limit must not turn it into unchecked fp16) and returns to fp16 -- and measures -- at the
                # first weight change after the interval.
                self._precision = "fp16"
                self.ln_guard = 8.0
                key = self._weights_key()
            if self._engine is None or self._engine.precision != self._precision:
                self._engine = Engine(self.cfg, precision=self._precision)
            self._engine.load_state_dict({k: v for k, v in self.state_dict().items()})
            self._engine_key = key
            self._engine_shape = None
            self._prompt_key = self._prompt_hold = None
            self._ln_checked = self._ln_pending = False
        return self._engine

    def _check_due(self) -> bool:
        import time as _time
        return self._last_check_time is None or (_time.monotonic() - self._last_check_time) >= self.check_min_interval_s

    def _auto_check(self, eng, out16, x, ts, content, prompt, mask, shape, stream, late: bool = False):
        """engine_precision="auto": the fp16 result of this call against the exact-fp32 engine on the same inputs, once per set of
        weights.  Inside ``precision_check`` (batch figure AND worst utterance): keep fp16, release the fp32 engine.  Outside: warn, keep
        the fp32 engine as THE engine from now on and return its result."""
        from ns2vc_amd.engine import Engine
        import time as _time
        self._precision_checked_key = self._engine_key[1]
        self._last_check_time = _time.monotonic()
        self.precision_checks += 1
        e32 = Engine(self.cfg, precision="fp32")
        e32.load_state_dict({k: v for k, v in self.state_dict().items()})
        torch.cuda.synchronize(x.device)
        e32.prepare(*shape)
        out32 = torch.empty_like(out16)
        e32.set_prompt(prompt, mask, stream=stream)
        e32.set_content(content, stream=stream)
        e32.forward(x, ts, out32, stream=stream)
        num = (out16 - out32).flatten(1).norm(dim=1)
        den = out32.flatten(1).norm(dim=1).clamp_min(1e-30)
        finite = bool(torch.isfinite(out16).all())
        seen = float(num.norm() / den.norm()) if finite else float("inf")
        worst = float((num / den).max()) if finite else float("inf")
        # (the figures reported are the worst over the measurements taken on these weights)
        self.precision_error_seen = max(seen, self.precision_error_seen) if (late and self.precision_error_seen is not None) else seen
        self.precision_error_worst_item = max(worst, self.precision_error_worst_item) if (late and self.precision_error_worst_item is not None) else worst
        self._last_verdict_ok = bool(seen <= self.precision_check and worst <= self.precision_check)
        if self._last_verdict_ok:
            e32.close()
            return out16
        warnings.warn(f"UNet1DConditionModel(engine_precision='auto'): the fp16 engine is {self.precision_error_seen:.2e} (relative L2 over the batch; worst "
                      f"utterance {self.precision_error_worst_item:.2e}) from the exact-fp32 engine on this checkpoint / input (> {self.precision_check:g}): "
                      f"serving from the fp32 engine from now on (engine_precision='fp16' forces the fast engine)")
        eng.close()
        self._precision = "fp32"
        self._engine = e32
        self._engine_key = self._weights_key()
        self._engine_shape = shape
        self._prompt_key = self._prompt_hold = None
        self.ln_guard = 32.0 if self.ln_guard is not None else None
        self._ln_checked = self._ln_pending = False
        return out32

    # ---- LayerNorm-by-linearity guard (same policy as ns2vc_amd.pipeline.Denoiser) ------------------------------------
    def _ln_excess(self, eng, r: float, late: bool) -> bool:
        self.ln_ratio_seen = r if self.ln_ratio_seen is None else max(self.ln_ratio_seen, r)
        if self.ln_guard is None or r <= self.ln_guard:
            return False
        warnings.warn(f"LayerNorm inputs with |mean|/std up to {r:.1f} (> {self.ln_guard}): switching the {eng.precision} engine to "
                      f"explicit LayerNorm passes (ln_linear=0)" + ("; the PREVIOUS result was computed above the threshold" if late else ""))
        eng.set_option("ln_linear", False)
        self._engine_shape = None
        self._prompt_key = self._prompt_hold = None
        self.ln_guard = None
        return True

    def _ln_guard_before(self, eng) -> None:
        if self.ln_guard is None or not self._ln_pending:
            return
        r = eng.ln_ratio_poll()
        if r is not None:
            self._ln_pending = False
            self._ln_excess(eng, r, late=True)

    def _ln_guard_after(self, eng, stream) -> bool:
        if self.ln_guard is None:
            return False
        if not self._ln_checked:
            self._ln_checked = True
            return self._ln_excess(eng, eng.ln_ratio(stream), late=False)
        if not self._ln_pending:
            eng.ln_ratio_post(stream)
            self._ln_pending = True
        return False

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
            if not self.training:        # an eval / benchmark call that forgot torch.no_grad() would silently miss the engine
                self._warn_once("autograd-eval", "UNet1DConditionModel: autograd is recording while the module is in eval() mode -- this call runs "
                                "the PyTorch training path (unet1d/torch_path.py), not the HIP engine; wrap inference in torch.no_grad()")
            out = self._run_torch_path(sample, timestep, encoder_hidden_states, encoder_attention_mask)
            self.autograd_calls += 1
            return UNet1DConditionOutput(sample=out) if return_dict else (out,)
        if not sample.is_cuda:
            # BASELINE config 1 / `infer.py --device cpu` (inference/infer_tool.py:119-135): plumbing on the host through the
            # module's own PyTorch forward (product code, the same one training uses; never oracle/).  Slow by construction.
            self._warn_once("cpu", "UNet1DConditionModel: CPU tensors -- running the PyTorch path (unet1d/torch_path.py), NOT the HIP engine; "
                            "this is the slow plumbing path (move the module and its inputs to a ROCm device for the engine)")
            out = self._run_torch_path(sample.float(), timestep, encoder_hidden_states.float(), encoder_attention_mask).to(sample.dtype)
            self.cpu_calls += 1
            return UNet1DConditionOutput(sample=out) if return_dict else (out,)
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
            self._ln_guard_before(eng)           # (may drop the plan: before the shape check)
            if self._engine_shape != (B, T, Lp):
                torch.cuda.synchronize(dev)
                eng.prepare(B, T, Lp)
                self._engine_shape = (B, T, Lp)
                self._prompt_key = self._prompt_hold = None
                self._ln_checked = self._ln_pending = False
            out = torch.empty((B, cfg.out_channels, T), dtype=torch.float32, device=dev)
            stream = torch.cuda.current_stream(dev)

            ehs = encoder_hidden_states
            # Inference tensors (torch.inference_mode()) carry no version counter: they cannot be keyed, so the prompt half is
            # re-hoisted on every call.  (In-place edits made through `.data` do not bump _version either: callers that
            # rewrite a prompt that way must pass a new tensor.)
            pkey = None if ehs.is_inference() else (ehs.data_ptr(), ehs._version, tuple(ehs.shape), tuple(ehs.stride()), ehs.dtype, stream.cuda_stream)
            if pkey is None or pkey != self._prompt_key:
                eng.set_prompt(prompt, mask, stream=stream)
                self._prompt_key = pkey
                self._prompt_hold = ehs          # alive => its address cannot be handed to another tensor while it is the key
                self.prompt_hoists += 1
            else:                                # the reference rebuilds the (tiny) mask per call (model.py:412): always refresh it
                eng.set_mask(mask, stream=stream)
            eng.set_content(content, stream=stream)
            eng.forward(x, ts, out, stream=stream)
            self.engine_calls += 1
            if self._auto and self._precision == "fp16" and self.precision_check is not None:
                wkey = self._engine_key[1]
                first = self._precision_checked_key != wkey
                due = self._check_due()
                # r6: a measurement that is not due is DEFERRED, not waived -- the weights stay "unchecked" (first stays true) and are measured
                # at the first call after the interval; in between the fp16 engine serves on the strength of the last verdict, which was a pass (after a
                # demotion _get_engine keeps the fp32 engine for the whole interval).  The late re-check obeys the same rate limit, reads the timestep
                # on the host when it can, and costs a device round trip at most once per interval otherwise (never while a stream is capturing).
                late = False
                if not first and due and self._late_checked_key != wkey and not torch.cuda.is_current_stream_capturing():
                    if not torch.is_tensor(timestep):
                        t_max = float(timestep)
                    elif timestep.device.type == "cpu":
                        t_max = float(timestep.max())
                    else:
                        t_max = float(ts.max())
                    late = t_max < self.late_check_below
                if first and not due:
                    self._warn_once("auto-rate", "UNet1DConditionModel(engine_precision='auto'): the weights change more often than once per "
                                    f"{self.check_min_interval_s:g} s; the fp16-vs-fp32 check is rate-limited: new weights are measured at the first call after "
                                    "the interval, and served meanwhile on the last verdict (set engine_precision explicitly for a module under training)")
                elif first or late:
                    if late:
                        self._late_checked_key = wkey
                    out = self._auto_check(eng, out, x, ts, content, prompt, mask, (B, T, Lp), stream, late=late)
                    if self._precision != "fp16":        # demoted: `out` already is the fp32 engine's result
                        out = out.to(sample.dtype)
                        return UNet1DConditionOutput(sample=out) if return_dict else (out,)
            if self._ln_guard_after(eng, stream):      # first call of this plan found LayerNorm rows above the threshold: redo on the explicit plan
                return self.forward(sample, timestep, encoder_hidden_states, encoder_attention_mask=encoder_attention_mask, return_dict=return_dict)
        out = out.to(sample.dtype)
        if not return_dict:
            return (out,)
        return UNet1DConditionOutput(sample=out)
