"""The N-step loop on the HIP stepper kernels (reference: Scheduler.propagate,
schedulers.py:48-89, with the four built-in integrators of integrators.py fused into it).

A *source* produces, for the current state, what the step kernels consume:
  ScoreFnSource  -- an arbitrary ``score_fn(x, sigma[B])`` (the public Scheduler API); the state
                    (x, x_e, x_hat) is materialised for every call and the kernels get scores;
  ModuleSource   -- a KarrasModule's network + preconditioner; the kernels get raw network outputs
                    and apply D = c_out*F + c_skip*x, the score, the drift and the update in one
                    pass, and also emit c_in*x for the next evaluation, so x_e is never stored.
Per step (deterministic Heun) the state tensor is read 2x and written 1x and the network input
written 2x: 32 bytes per element per step including the two F reads.
"""
import torch

from ... import ops
from ..._native import DS_IN_NETWORK, DS_IN_SCORE
from .steptable import StepTable


class ScoreFnSource:
    input_kind = DS_IN_SCORE
    wants_xin = False
    guidance = 1.0

    def __init__(self, score_fn, batch, like):
        self.score_fn = score_fn
        self.batch = batch
        self.like = like
        self._ones = torch.ones(batch, dtype=torch.float32, device=like.device)

    def prepare(self, table: StepTable):
        pass

    def evaluate(self, state, xin, row, index, slot):
        sigma = self._ones * row.sigma            # t*ones(B): schedulers.py:254
        s = self.score_fn(state, sigma)
        ops.require_device(s, "score_fn output")
        return s.contiguous(), None


class ModuleSource:
    """model(c_in*x, c_noise[, y]) of KarrasModule.get_denoiser (karrasmodule.py:702-716)."""
    input_kind = DS_IN_NETWORK
    wants_xin = True

    def __init__(self, module, y, guidance, batch, like):
        self.module = module
        self.model = module.model
        self.y = y
        self.guidance = float(guidance)
        self.batch = batch
        self.like = like
        self.conditional = bool(module.conditional) and self.guidance != 0.0
        self.cfg = self.conditional and self.guidance != 1.0
        self.planned = bool(getattr(self.model, "forward_with_shifts", None)) and (
            like.dim() == 4 or (like.dim() == 5 and getattr(self.model, "dim", 2) == 3))
        self._out = {}

    def prepare(self, table: StepTable):
        evals = table.evals
        dev = self.like.device
        cn = torch.tensor([e.c_noise for e in evals], dtype=torch.float32)
        if self.planned:
            # sigma is shared by the batch, so the whole time-conditioning path (Fourier features,
            # conditional embedding, 14 time MLPs) is evaluated once for all evaluations of the run.
            m = self.model
            cn = cn.to(dev)
            ye = m.embed_condition(self.y) if self.conditional else None
            if ye is not None and ye.shape[0] != 1:
                raise NotImplementedError("per-sample conditions in the planned sampler (y is un-batched in sample())")
            self.shifts_c = m.time_shifts(m.embed_time(cn, ye))
            self.shifts_u = m.time_shifts(m.embed_time(cn, None)) if (self.cfg or not self.conditional) else None
            if not self.conditional:
                self.shifts_c = self.shifts_u
        else:
            self.cnoise = cn[:, None].expand(len(evals), self.batch).contiguous().to(dev)

    def _buf(self, slot, name):
        key = (slot, name)
        if key not in self._out:
            self._out[key] = torch.empty_like(self.like)
        return self._out[key]

    def evaluate(self, state, xin, row, index, slot):
        if self.planned:
            f = self.model.forward_with_shifts(xin, self.shifts_c, row=index, out=self._buf(slot, "c"))
            fu = None
            if self.cfg:
                fu = self.model.forward_with_shifts(xin, self.shifts_u, row=index, out=self._buf(slot, "u"))
            return f, fu
        cn = self.cnoise[index]
        if self.conditional:
            f = self.model(xin, cn, self.y)
            fu = self.model(xin, cn) if self.cfg else None
        else:
            f, fu = self.model(xin, cn), None
        ops.require_device(f, "model output")
        return f.contiguous(), (None if fu is None else fu.contiguous())


class Loop:
    """Buffers + launch sequence of one tabulated run.  Construction allocates everything and lets
    the source precompute its per-evaluation tables; ``launch`` only enqueues kernels (plus
    whatever the source's evaluation does), so for a planned ModuleSource it can be captured into
    a hipGraph and replayed.

    Usage: loop.load(x0[, scale]); loop.launch(); loop.result()."""

    def __init__(self, table: StepTable, source, like, record_history=False):
        ops.require_device(like, "x")
        self.table, self.source, self.record_history = table, source, record_history
        n = len(table.rows)
        shape, dev = tuple(like.shape), like.device
        new = lambda: torch.empty(shape, dtype=torch.float32, device=dev)  # noqa: E731
        if record_history:
            self.history = torch.zeros((n + 1,) + shape, dtype=torch.float32, device=dev)   # schedulers.py:68-69
            self.x = self.history[0]
        else:
            self.history = None
            self.x = new()
        self.xin = new() if source.wants_xin else None
        self.tmp = new() if (not source.wants_xin or table.kind == "karras") else None
        self.tmp2 = new() if (not source.wants_xin and table.kind == "karras") else None
        self.eps = (torch.empty((n,) + shape, dtype=torch.float32, device=dev) if table.needs_noise else None)
        source.prepare(table)

    def load(self, x0, scale=None):
        """state <- x0 (or scale*x0: x*maximum_scale of karrasmodule.py:881)."""
        ops.require_device(x0, "x")
        x0 = x0.contiguous()
        if tuple(x0.shape) != tuple(self.x.shape):
            raise ValueError("x shape does not match the loop")
        if scale is None:
            self.x.copy_(x0)
        else:
            ops.scale(x0, scale, out=self.x)

    def set_noise(self, eps=None):
        """eps [nsteps, *shape]: injected noise; None draws it on the device generator, one tensor
        per step in step order (the reference calls randn_like once per step)."""
        if self.eps is None:
            return
        if eps is None:
            self.eps.normal_()
        else:
            if eps.shape[0] < self.eps.shape[0] or tuple(eps.shape[1:]) != tuple(self.eps.shape[1:]):
                raise ValueError("eps must be [nsteps, *x.shape]")
            self.eps.copy_(eps[:self.eps.shape[0]])

    def launch(self):
        table, source = self.table, self.source
        n = len(table.rows)
        kind, g = source.input_kind, source.guidance
        xin, tmp, eps = self.xin, self.tmp, self.eps
        karras = table.kind == "karras"
        cur = self.x
        e = 0
        if source.wants_xin and n > 0 and not karras:
            ops.scale(cur, table.rows[0].first.c_in, out=xin)       # c_in*x, karrasmodule.py:702
        for i, row in enumerate(table.rows):
            nxt = self.history[i + 1] if self.record_history else cur
            nxt_row = table.rows[i + 1] if i + 1 < n else None
            chain = source.wants_xin and nxt_row is not None and not karras
            c_in_next = nxt_row.first.c_in if chain else 1.0
            xin_next = xin if chain else None
            base = cur
            if karras:
                base = tmp                                           # x_hat, integrators.py:104-105
                ops.churn(cur, eps[i], row.churn_coef, xhat_out=base, xin_out=xin, c_in=row.first.c_in)
            k1 = row.first.coef(kind, g)
            f1, f1u = source.evaluate(base, xin, row.first, e, 0)
            e += 1
            if row.second is None:
                ops.euler(base, f1, k1, row.dt, fu=f1u, x_out=nxt, xin_out=xin_next, c_in_next=c_in_next,
                          eps=eps[i] if table.kind == "euler-maruyama" else None,
                          noise_coef=row.noise_coef, sqrt_abs_dt=row.sqrt_abs_dt)
            else:
                k2 = row.second.coef(kind, g)
                xe = None
                if not source.wants_xin:
                    xe = self.tmp2 if karras else tmp
                ops.euler(base, f1, k1, row.dt, fu=f1u, x_out=xe, xin_out=xin, c_in_next=row.second.c_in)
                f2, f2u = source.evaluate(xe, xin, row.second, e, 1)
                e += 1
                ops.heun(base, f1, k1, f2, k2, row.dt, f1u=f1u, f2u=f2u, x_out=nxt, xin_out=xin_next,
                         c_in_next=c_in_next)
            cur = nxt
        self._final = cur

    def result(self):
        return self.history if self.record_history else self._final


def run_table(table: StepTable, source, x, record_history=False, eps=None):
    """One eager pass: returns the final state (a new tensor) or the history [len(rows)+1, *x.shape]."""
    loop = Loop(table, source, x, record_history)
    loop.load(x)
    loop.set_noise(eps)
    loop.launch()
    return loop.result()
