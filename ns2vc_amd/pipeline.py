"""High-level denoiser sampling API on torch CUDA tensors (torch = memory/stream plumbing).

This is what a maintainer calls from ``NaturalSpeech2.sample`` instead of the
reference's per-step Python loop (``model.py:620-687``): the step-invariant
condition work is hoisted once, the N-step DPM-Solver++ / UniPC loop replays one
captured hipGraph per step, and nothing synchronises with the host until the end.
"""
from __future__ import annotations

import warnings
from typing import Dict, Optional

import numpy as np

from . import dist as _dist
from .engine import DEFAULT_PRECISION, Engine
from .schedule import linear_betas
from .spec import UNetConfig, attention_workgroups_per_forward


# default number of trailing fp32 evaluations of a 16-bit sampling loop, per solver (``Denoiser(tail_fp32=None)``):
# DPM-Solver++(2M) takes its LAST update at second order over the largest log-SNR step of the schedule (no lower_order_final
# for steps >= 10, dpm_solver.py:1198-1201), which amplifies the rounding noise of the last two evaluations ~2.7x (50 steps:
# fp16 1.7e-3 on the sampled latent, with the last two evaluations in fp32 ~2e-4); UniPC-bh2 ends at first order and keeps
# the single-evaluation noise (8.6e-4 at 20 steps), so it stays pure 16-bit unless asked.
DEFAULT_TAIL_FP32 = {"dpmsolver++": 2, "unipc": 0}
# LayerNorm-by-linearity guard thresholds on max |mean|/std (see Denoiser): 16-bit modes lose ~ratio * 2^-11 on a row,
# fp32 loses ~ratio^2 * 2^-24 in the variance E[x^2] - mean^2
LN_GUARD_DEFAULT = {"fp16": 8.0, "bf16": 8.0, "fp32": 32.0}


class Denoiser:
    """``precision``: "fp16" (default: 16-bit MFMA operands, 7e-4 vs the reference fp32 path -- inside the 1e-3 parity
    bar), "fp32" (exact-fp32 MFMA, 1e-6) or "bf16" (same speed as fp16, 5.7e-3; kept for range-critical checkpoints).

    ``tail_fp32``: evaluations at the END of every sampling loop that run on a second, fp32 engine (the solver state is
    handed over on the device, ``Engine.sample(tail=...)``).  None = ``DEFAULT_TAIL_FP32[solver]`` for the 16-bit
    precisions (2 for DPM-Solver++, 0 for UniPC), 0 = never.  The fp32 engine (weights + workspace) is built on first use.

    ``ln_guard``: LayerNorm by linearity (the default plan) lets the 16-bit modes round a LayerNorm's RAW input before
    centring, so their error on a row grows with |mean| / std of that row (and the fp32 variance E[x^2] - mean^2 with its
    square).  The engine records the maximum of that ratio.  It is read after the FIRST evaluation with a new shape /
    set of weights (one wait on the denoiser's stream; above the threshold the plan is switched to explicit LayerNorm
    passes, ``ln_linear`` 0: ~4 % slower, immune, and the evaluation is repeated) and, because the ratio also depends on
    the content, the prompt and the timestep, after EVERY later call without blocking: the read-out is enqueued behind
    the call and collected at the start of the next one (``ln_ratio_seen`` = running maximum); a late excess switches
    the plan for all following calls and warns that the previous result was computed above the threshold.
    ``None`` disables the check; a float sets the threshold (default 8 for the 16-bit modes, 32 for fp32).

    ``precision_check``: the 7e-4 of the fp16 mode was measured on procedural weights; a trained checkpoint with a few hot
    channels can land on either side of the 1e-3 bar (or saturate fp16 operands outright).  So a 16-bit Denoiser MEASURES
    itself ONCE PER SET OF WEIGHTS (the first call; ``recheck_precision()`` forces another): evaluations are run on the 16-bit
    engine and on the exact-fp32 engine (pinned to the reference at 1e-6) on the caller's own inputs.  ``sample`` checks at
    THREE points of its own trajectory -- the first, the middle and the last evaluation point of the loaded table (x_e taken
    from a preliminary 16-bit loop; the last evaluations are the ones that set the sampled latent) --, ``denoise`` at the
    given (x, t).  Two figures are kept and BOTH gated against ``precision_check`` (default 1e-3 for fp16; None disables;
    bf16: 2e-2, its own level): the relative L2 over the batch -- the parity bar's own measure -- in ``precision_error_seen``
    and the worst single utterance in ``precision_error_worst_item`` (maxima over the points; ``precision_errors`` has the
    per-point list).  Above it the Denoiser warns and serves this and all later calls from the fp32 engine (3.6x the step time,
    inside the bar by construction).  The verdict belongs to the weights, not to a shape: new shapes do NOT repeat it (they
    used to: an fp32 re-prepare, two extra forwards and two host waits per group of ``GroupedConverter``), and the fp32 engine is
    released again after a passed check unless a tail needs it (it is rebuilt on demand).

    ``attn_fallback_limit`` (r4): the 16-bit attention kernels first run an optimistic pass without the per-tile maximum (-10 % of their time) and
    repeat a workgroup exactly when a row's scores rose far above its first 64 keys.  With procedural weights that never happens; on a
    checkpoint with sharp attention it can: measured at the bench shape, 1 % of such rows sends 66 % of the workgroups through both passes
    (86 us instead of 45 per launch; the exact pass alone takes 50).  So the first loop / evaluation with a set of weights reads the engine's
    fallback counter (one host wait, once) and, above this share of workgroups (default 0.10, about the break-even), switches the engine to
    ``attn_optimistic`` 0 with a warning; ``attn_fallback_rate_seen`` keeps the measured share.  None disables."""

    def __init__(self, state: Dict[str, object], cfg: UNetConfig = UNetConfig(), precision: str = DEFAULT_PRECISION,
                 betas: Optional[np.ndarray] = None, ln_guard: Optional[float] = -1.0, tail_fp32: Optional[int] = None,
                 precision_check: Optional[float] = -1.0, attn_fallback_limit: Optional[float] = 0.10):
        self.cfg = cfg
        self.attn_fallback_limit = attn_fallback_limit if precision not in ("fp32", "f32") else None
        self.attn_fallback_rate_seen: Optional[float] = None
        self._attn_checked = False
        self.precision = {"f32": "fp32", "f16": "fp16"}.get(precision, precision)
        self.engine = Engine(cfg, precision=precision)
        self.engine.load_state_dict(state)
        self._state = state
        self.betas = linear_betas() if betas is None else np.asarray(betas, dtype=np.float32)
        self._shape = None
        self._table_key = None
        self.ln_guard = LN_GUARD_DEFAULT[self.precision] if (ln_guard is not None and ln_guard < 0) else ln_guard
        self.ln_ratio_seen: Optional[float] = None
        self._ln_checked = False
        self._ln_pending = False
        self._ln_switched = False
        self.tail_fp32 = tail_fp32
        self.tail_engine: Optional[Engine] = None
        if precision_check is not None and precision_check < 0:
            precision_check = {"fp16": 1e-3, "bf16": 2e-2}.get(self.precision)
        self.precision_check = precision_check if self.precision != "fp32" else None
        self.precision_error_seen: Optional[float] = None
        self.precision_error_worst_item: Optional[float] = None
        self.precision_errors: list = []          # [(timestep, batch rel-L2, worst utterance)] of the last self-check
        self._precision_checked = False
        self.serving_fp32 = False        # set by a failed precision check: every call then runs on the fp32 engine
        self._tail_shape = None
        self._tail_table_key = None

    # ---- LayerNorm-by-linearity guard ---------------------------------------------------------------------------------
    def _note_ratio(self, r: float) -> bool:
        self.ln_ratio_seen = r if self.ln_ratio_seen is None else max(self.ln_ratio_seen, r)
        return self.ln_guard is not None and r > self.ln_guard

    def _switch_plan(self, r: float, late: bool) -> None:
        warnings.warn(f"LayerNorm inputs with |mean|/std up to {r:.1f} (> {self.ln_guard}): switching the "
                      f"{self.engine.precision} engine to explicit LayerNorm passes (ln_linear=0)" +
                      ("; the PREVIOUS result was computed above the threshold" if late else ""))
        self.engine.set_option("ln_linear", False)
        if self.tail_engine is not None:     # the fp32 engine loses ~ratio^2 * 2^-24 under the same plan: switch it too
            self.tail_engine.set_option("ln_linear", False)
            self._tail_shape = None
        self._ln_switched = True
        self._shape = None
        self._ln_checked = True          # the explicit plan does not depend on the ratio
        self.ln_guard = None

    def _guard_before(self) -> None:
        """collect the read-out enqueued behind the previous call (non-blocking)"""
        if self.ln_guard is None or not self._ln_pending:
            return
        r = self.engine.ln_ratio_poll()
        if r is None:
            return
        self._ln_pending = False
        if self._note_ratio(r):
            self._switch_plan(r, late=True)

    def _guard_after(self, stream, redo):
        if self.ln_guard is None:
            return None
        if not self._ln_checked:         # first call on this plan: wait for the value, repeat the call if it is too large
            self._ln_checked = True
            r = self.engine.ln_ratio(stream)
            if self._note_ratio(r):
                self._switch_plan(r, late=False)
                return redo()
            return None
        if not self._ln_pending:         # later calls: enqueue, collect next time
            self.engine.ln_ratio_post(stream)
            self._ln_pending = True
        return None

    def _prepare(self, B: int, T: int, Lp: int) -> None:
        if self._shape != (B, T, Lp):
            import torch
            torch.cuda.synchronize()
            self.engine.prepare(B, T, Lp)
            self._shape = (B, T, Lp)
            self._ln_checked = False                # a new shape is a new set of rows: check synchronously once
            self._ln_pending = False

    def set_option(self, name: str, value: bool) -> None:
        """A plan option of the engine (``Engine.set_option``; e.g. ``gn_coop`` off for a pipeline that runs the denoiser on a CU partition).
        The plan and the sampler table are rebuilt by the next ``sample``."""
        self.engine.set_option(name, value)
        self._shape = None
        self._table_key = None

    def _table(self, solver: str, steps: int, order: int) -> None:
        key = (solver, steps, order)
        if self._table_key != key:
            self.engine.load_sampler(solver, steps, self.betas, order)
            self._table_key = key

    def _fp32_engine(self) -> Engine:
        """the exact-fp32 engine (tail of a 16-bit loop, precision self-check, fallback), prepared for the current shape"""
        if self.tail_engine is None:
            self.tail_engine = Engine(self.cfg, precision="fp32")
            self.tail_engine.load_state_dict(self._state)
            if self._ln_switched:
                self.tail_engine.set_option("ln_linear", False)
            self._tail_shape = None
            self._tail_table_key = None
        if self._tail_shape != self._shape:
            import torch
            torch.cuda.synchronize()
            self.tail_engine.prepare(*self._shape)
            self._tail_shape = self._shape
        return self.tail_engine

    def _tail(self, solver: str, steps: int, order: int, n_tail: int) -> Optional[Engine]:
        """the fp32 engine that finishes a 16-bit loop (or runs all of it after a failed precision check), with the table loaded"""
        if (n_tail <= 0 and not self.serving_fp32) or self.precision == "fp32":
            return None
        e = self._fp32_engine()
        key = (solver, steps, order)
        if self._tail_table_key != key:
            e.load_sampler(solver, steps, self.betas, order)
            self._tail_table_key = key
        return e

    def _attn_check(self, evals: int, stream) -> None:
        """once per set of weights: share of attention workgroups that needed the exact fallback during the last ``evals`` evaluations"""
        if self.attn_fallback_limit is None or self._attn_checked or evals <= 0:
            return
        self._attn_checked = True
        B, T, _ = self._shape
        n = self.engine.attn_fallbacks(reset=True, stream=stream)
        self.attn_fallback_rate_seen = n / float(max(1, evals * attention_workgroups_per_forward(self.cfg, B, T)))
        if self.attn_fallback_rate_seen > self.attn_fallback_limit:
            warnings.warn(f"{100 * self.attn_fallback_rate_seen:.0f} % of the attention workgroups needed the exact fallback on this checkpoint / input "
                          f"(> {100 * self.attn_fallback_limit:.0f} %): each of them ran twice -- switching the engine to the exact pass only (attn_optimistic=0)")
            self.engine.set_option("attn_optimistic", False)
            self._shape = None

    def recheck_precision(self) -> None:
        """forget the verdict of the precision self-check (after the weights were changed in place): the next call measures again"""
        self._precision_checked = False
        self.serving_fp32 = False
        self._attn_checked = False

    def _self_check(self, points, c32, p32, mask, stream, keep_fp32: bool) -> None:
        """once per set of weights: the same evaluations on the 16-bit and on the fp32 engine (class docstring, ``precision_check``);
        ``points`` = [(x, t)] with x (B,100,T) fp32 and t (B,) fp32"""
        import torch
        if self.precision_check is None or self._precision_checked:
            return
        self._precision_checked = True
        e32 = self._fp32_engine()
        self.engine.set_condition(c32, p32, mask, stream=stream)
        e32.set_condition(c32, p32, mask, stream=stream)
        self.precision_errors = []
        err, worst = 0.0, 0.0
        for x, t in points:
            outs = []
            for eng in (self.engine, e32):
                o = torch.empty_like(x, dtype=torch.float32)
                eng.forward(x, t, o, stream=stream)
                outs.append(o)
            num = (outs[0] - outs[1]).flatten(1).norm(dim=1)
            den = outs[1].flatten(1).norm(dim=1).clamp_min(1e-30)
            finite = bool(torch.isfinite(outs[0]).all())
            # the parity bar's own measure: relative L2 over the whole batch; the worst single utterance beside it
            e_b = float(num.norm() / den.norm()) if finite else float("inf")
            e_w = float((num / den).max()) if finite else float("inf")
            self.precision_errors.append((float(t.flatten()[0]), e_b, e_w))
            err, worst = max(err, e_b), max(worst, e_w)
        self.precision_error_seen, self.precision_error_worst_item = err, worst
        if not (err <= self.precision_check and worst <= self.precision_check):
            warnings.warn(f"{self.precision} engine is {err:.2e} (relative L2 over the batch; worst utterance {worst:.2e}) from the exact-fp32 "
                          f"engine on this checkpoint / input (> {self.precision_check:g}): serving from the fp32 engine from now on "
                          f"(precision_check=None disables)")
            self.serving_fp32 = True
        elif not keep_fp32:      # passed and no tail wants it: do not keep a second set of weights + workspace resident
            self.tail_engine = None
            self._tail_shape = None
            self._tail_table_key = None

    def _trajectory_points(self, x_T, use_graph, stream):
        """(x_e, t) at the first, middle and last evaluation of the loaded table, from a preliminary loop on the 16-bit engine
        (its condition must be set)"""
        import torch
        tm = np.asarray(self.engine.table.t_model, dtype=np.float32)
        n = len(tm)
        idx = sorted({0, n // 2, n - 1})
        B = x_T.shape[0]
        pts, pos = [], 0
        self.engine.sample_begin(x_T, stream=stream)
        for i in idx:
            self.engine.sample_steps(i - pos, use_graph=use_graph, stream=stream)
            pos = i
            xe = torch.empty_like(x_T)
            self.engine.sample_peek(xe, stream=stream)
            pts.append((xe, torch.full((B,), float(tm[i]), dtype=torch.float32, device=x_T.device)))
        scratch = torch.empty_like(x_T)
        self.engine.sample_end(scratch, stream=stream)
        return pts

    def denoise(self, x, t, content, prompt, prompt_mask=None):
        """One evaluation: x (B,100,T), t (B,), content (B,256,T), prompt (B,Lp,256), mask (B,Lp) bool -> x0_pred."""
        import torch
        B, _, T = x.shape
        self._guard_before()
        self._prepare(B, T, prompt.shape[1])
        s = torch.cuda.current_stream(x.device)
        mask = None if prompt_mask is None else prompt_mask.to(torch.uint8).contiguous()
        c32, p32, x32, t32 = content.float().contiguous(), prompt.float().contiguous(), x.float().contiguous(), t.float().contiguous()
        self._self_check([(x32, t32)], c32, p32, mask, s, keep_fp32=bool(self.tail_fp32))
        eng = self._fp32_engine() if self.serving_fp32 else self.engine
        eng.set_condition(c32, p32, mask, stream=s)
        out = torch.empty_like(x, dtype=torch.float32)
        first_attn = not self.serving_fp32 and not self._attn_checked and self.attn_fallback_limit is not None
        if first_attn:
            eng.attn_fallbacks(reset=True, stream=s)          # count this evaluation alone
        eng.forward(x32, t32, out, stream=s)
        if self.serving_fp32:
            return out
        redone = self._guard_after(s, lambda: self.denoise(x, t, content, prompt, prompt_mask))
        if first_attn and redone is None:
            self._attn_check(1, s)                            # (after the guard's read-out: a switch drops the plan)
        return out if redone is None else redone

    def sample(self, content, prompt, prompt_mask=None, noise=None, solver: str = "unipc", steps: int = 20, order: int = 2,
               use_graph: bool = True, generator=None, tail_fp32: Optional[int] = None):
        """content (B,256,T), prompt (B,Lp,256), mask (B,Lp) bool; ``noise`` (B,100,T) = x_T (drawn with
        torch.randn like model.py:635 if None).  Returns the sampled latent (B,100,T) fp32.
        ``tail_fp32`` overrides the instance's setting for this call (see the class docstring)."""
        import torch
        B, _, T = content.shape
        dev = content.device
        self._guard_before()
        self._prepare(B, T, prompt.shape[1])
        self._table(solver, steps, order)
        n_tail = tail_fp32 if tail_fp32 is not None else self.tail_fp32
        if n_tail is None:
            n_tail = DEFAULT_TAIL_FP32.get(solver, 0) if self.precision != "fp32" else 0
        n_tail = max(0, min(int(n_tail), steps))
        if noise is None:
            noise = torch.randn((B, self.cfg.latent_channels, T), device=dev, generator=generator)
        x = noise.to(device=dev, dtype=torch.float32).contiguous().clone()
        s = torch.cuda.current_stream(dev)
        mask = None if prompt_mask is None else prompt_mask.to(device=dev, dtype=torch.uint8).contiguous()
        c32, p32 = content.float().contiguous(), prompt.float().contiguous()
        if self.precision_check is not None and not self._precision_checked:
            self.engine.set_condition(c32, p32, mask, stream=s)
            self._self_check(self._trajectory_points(x, use_graph, s), c32, p32, mask, s, keep_fp32=n_tail > 0)
        tail = self._tail(solver, steps, order, n_tail)
        if self.serving_fp32:            # a failed precision check: the whole loop on the fp32 engine
            tail.set_condition(c32, p32, mask, stream=s)
            tail.sample(x, use_graph=use_graph, stream=s)
            return x
        self.engine.set_condition(c32, p32, mask, stream=s)
        if tail is not None:
            tail.set_condition(c32, p32, mask, stream=s)
        first_attn = not self._attn_checked and self.attn_fallback_limit is not None
        if first_attn:
            self.engine.attn_fallbacks(reset=True, stream=s)      # count this loop alone (the self-check's evaluations are behind us)
        self.engine.sample(x, use_graph=use_graph, stream=s, tail=tail, tail_steps=n_tail if tail is not None else 0)
        redone = self._guard_after(s, lambda: self.sample(content, prompt, prompt_mask, noise, solver, steps, order, use_graph,
                                                          tail_fp32=tail_fp32))
        if first_attn and redone is None:
            self._attn_check(steps - (n_tail if tail is not None else 0), s)      # (after the guard's read-out: a switch drops the plan)
        return x if redone is None else redone

    def sample_sharded(self, content, prompt, prompt_mask, noise, **kw):
        """Data-parallel: every rank receives the GLOBAL batch description, runs its contiguous slice and the
        finished latents are all-gathered (RCCL).  The noise is drawn for the global batch and sliced, so an utterance's
        result does not depend on the world size beyond the precision's rounding noise (fp32: ~1e-6; 16-bit: a shard of
        3 and a shard of 2 round differently, ~7e-4 -- tests/test_dropin_gpu.py::test_sample_sharded_rccl_*)."""
        import torch
        import torch.distributed as td
        rank = td.get_rank() if td.is_initialized() else 0
        world = td.get_world_size() if td.is_initialized() else 1
        n = content.shape[0]
        lo, hi = _dist.shard_range(n, rank, world)
        if hi == lo:      # fewer utterances than ranks: this rank has nothing to denoise but must still join the collective
            local = torch.zeros((0, self.cfg.latent_channels, content.shape[2]), dtype=torch.float32, device=content.device)
        else:
            pm = None if prompt_mask is None else prompt_mask[lo:hi]
            local = self.sample(content[lo:hi], prompt[lo:hi], pm, noise[lo:hi], **kw)
        return _dist.gather_latents(local, n)


class OverlappedPipeline:
    """Three-stage utterance-batch pipeline on three HIP streams: the PyTorch-ROCm front end (ContentVec / ``Pre_model.infer``,
    reference ``model.py:359-376``) and back end (``vocos.decode``, ``model.py:689-696``) overlap the denoiser instead of
    running before / after it (BASELINE.json north_star):

        pre(k+1)   |   denoise(k)   |   post(k-1)

    ``pre_fn(item)`` runs on the front-end stream and returns a dict with torch CUDA tensors ``content`` (B,256,T),
    ``prompt`` (B,Lp,256) and optionally ``prompt_mask`` (B,Lp) bool and ``noise`` (B,100,T); ``post_fn(latent, item)``
    runs on the back-end stream with the sampled latent (B,100,T) fp32.  Ordering is by events only -- the host never
    blocks until the end of ``run`` -- and the denoiser still replays one captured hipGraph per step on its own stream.
    """

    def __init__(self, denoiser: Denoiser, pre_fn, post_fn, solver: str = "unipc", steps: int = 20, order: int = 2,
                 use_graph: bool = True, pre_device=None, post_device=None, stage_cus=None, denoiser_cus=None):
        """``pre_device`` / ``post_device`` (r4): run the front / back end on ANOTHER ROCm device of the node.  On one GPU the three streams
        serialise (the denoiser's launches hold every CU: 7.8 % of the stages' kernel time overlaps, profiles/r03_overlap_trace.txt); a stage
        on its own device overlaps by construction and only its tensors cross xGMI -- content + prompt 35 MB per 32 x 10 s batch in, the
        latent 12 MB out, stream-ordered peer copies behind the stage's event.  ``pre_fn`` then runs with ``pre_device`` current and must
        return tensors on it; ``post_fn`` receives the latent on ``post_device``.  None = the denoiser's device (the one-GPU pipeline).
        NOT measured on two devices yet (no multi-GPU box this round)."""
        import torch
        if not torch.cuda.is_available():
            raise RuntimeError("OverlappedPipeline needs a ROCm device (there is no CPU path)")
        self.denoiser, self.pre_fn, self.post_fn = denoiser, pre_fn, post_fn
        self.kw = dict(solver=solver, steps=steps, order=order, use_graph=use_graph)
        dev = torch.device("cuda", torch.cuda.current_device())
        self.device = dev
        self.pre_device = torch.device(pre_device) if pre_device is not None else dev
        self.post_device = torch.device(post_device) if post_device is not None else dev
        for d in (self.pre_device, self.post_device):
            if d.type != "cuda" or (d.index is not None and d.index >= torch.cuda.device_count()):
                raise ValueError(f"stage device {d} is not a visible ROCm device")
        self.s_pre = torch.cuda.Stream(self.pre_device)
        self.s_den = torch.cuda.Stream(dev)
        self.s_post = torch.cuda.Stream(self.post_device)
        # r5: a CU partition of ONE device (hipExtStreamCreateWithCUMask through the C ABI, wrapped as torch external streams): the two
        # PyTorch stages on `stage_cus`, the denoiser on `denoiser_cus` (iterables of CU indices; None = the whole chip).  Measured in
        # tools/overlap_partition.py / profiles/r05_overlap_partition.txt.
        self._masked = []
        if stage_cus is not None or denoiser_cus is not None:
            from .engine import Stream as _EStream
            if stage_cus is not None and self.pre_device == dev and self.post_device == dev:
                a, b = _EStream(cu_mask=stage_cus), _EStream(cu_mask=stage_cus)
                self._masked += [a, b]
                self.s_pre, self.s_post = torch.cuda.ExternalStream(a.ptr, device=dev), torch.cuda.ExternalStream(b.ptr, device=dev)
            if denoiser_cus is not None:
                c = _EStream(cu_mask=denoiser_cus)
                self._masked.append(c)
                self.s_den = torch.cuda.ExternalStream(c.ptr, device=dev)

    def _launch_pre(self, item):
        import torch
        with torch.cuda.device(self.pre_device), torch.cuda.stream(self.s_pre):
            cond = self.pre_fn(item)
            ev = torch.cuda.Event()
            ev.record(self.s_pre)
        return item, cond, ev

    def run(self, items):
        """Process an iterable of work items; returns ``[post_fn(latent_k, item_k) for k]`` (all streams drained)."""
        import torch
        it = iter(items)
        first = next(it, None)
        cur = self._launch_pre(first) if first is not None else None
        results = []
        while cur is not None:
            item, cond, ev_pre = cur
            nxt = next(it, None)
            nxt_pre = self._launch_pre(nxt) if nxt is not None else None     # front end of batch k+1 under denoise(k)
            with torch.cuda.device(self.device), torch.cuda.stream(self.s_den):
                self.s_den.wait_event(ev_pre)
                for k_, v in list(cond.items()):
                    if isinstance(v, torch.Tensor):
                        v.record_stream(self.s_den)
                        if v.device != self.device:          # front end on another device: peer copy, ordered behind its event on the denoiser's stream
                            cond[k_] = v.to(self.device, non_blocking=True)
                latent = self.denoiser.sample(cond["content"], cond["prompt"], cond.get("prompt_mask"), cond.get("noise"), **self.kw)
                ev_den = torch.cuda.Event()
                ev_den.record(self.s_den)
            with torch.cuda.device(self.post_device), torch.cuda.stream(self.s_post):
                self.s_post.wait_event(ev_den)
                latent.record_stream(self.s_post)
                if latent.device != self.post_device:        # back end on another device: the latent crosses behind the denoiser's event
                    latent = latent.to(self.post_device, non_blocking=True)
                results.append(self.post_fn(latent, item))                   # back end of batch k under denoise(k+1)
            cur = nxt_pre
        self.s_post.synchronize()
        self.s_den.synchronize()
        return results
