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
from ..nets import precision
from .steptable import StepTable


class ScoreFnSource:
    input_kind = DS_IN_SCORE
    wants_xin = False
    xin_copies = 1
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
        if row.scaled:                            # score_fn(x / s, sigma), schedulers.py:287
            state = ops.div_scalar(state.contiguous(), row.scale)
        s = self.score_fn(state, sigma)
        ops.require_device(s, "score_fn output")
        return s.contiguous(), None


def condition_signature(y):
    """Structure of a condition (keys, shapes, dtypes) -- what a captured plan depends on.  The VALUES of its tensors never
    enter a plan key: everything derived from them lives in plan-owned buffers that `ModuleSource.refresh` rewrites before
    every replay (an address- or checksum-based key cannot tell two one-hot labels, or a new tensor that reuses a
    freed block, apart).  Leaves that are NOT tensors (Python scalars, strings, None inside a container) cannot be rewritten
    in a captured graph, so their values ARE part of the key: a different scalar is a different plan.  Lists and tuples are
    containers like dicts."""
    if y is None:
        return None
    if isinstance(y, dict):
        return ("dict",) + tuple((k, condition_signature(v)) for k, v in sorted(y.items()))
    if isinstance(y, (list, tuple)):
        return (type(y).__name__,) + tuple(condition_signature(v) for v in y)
    if torch.is_tensor(y):
        return (tuple(y.shape), str(y.dtype), str(y.device))
    if isinstance(y, (bool, int, float, str, bytes)):
        return ("value", type(y).__name__, y)
    return ("object", repr(type(y)), id(y))             # anything else: the very object (a plan cannot follow what it cannot read)


def clone_condition(y):
    """Plan-owned copy of a condition's tensors (same structure; lists and tuples are rebuilt around the copies)."""
    if isinstance(y, dict):
        return {k: clone_condition(v) for k, v in y.items()}
    if isinstance(y, (list, tuple)):
        return type(y)(clone_condition(v) for v in y)
    return y.clone() if torch.is_tensor(y) else y


def copy_condition(dst, src):
    """Write src's values into dst's tensors (the structure, and the value of every non-tensor leaf, is part of the plan key)."""
    if isinstance(dst, dict):
        for k in dst:
            copy_condition(dst[k], src[k])
    elif isinstance(dst, (list, tuple)):
        for d, v in zip(dst, src):
            copy_condition(d, v)
    elif torch.is_tensor(dst):
        dst.copy_(src)


MODEL_SWITCHES = ("conv_precision", "fuse_norm", "fuse_max_cot", "direct_out", "upsample_parity", "norm_images", "tile_stats_norms",
                  "exact_input_layer", "capturable")


def _tree_slots(model):
    mods = list(model.modules())
    shape = tuple((d, len(d)) for m in mods for d in (m._modules, m._parameters, m._buffers))
    children = tuple((m._modules, name, c) for m in mods for name, c in m._modules.items())
    # parameters AND buffers: a plan's tables are computed from buffers too (the Fourier projection's W is one)
    params = tuple((m._parameters, name) for m in mods for name, p in m._parameters.items() if p is not None) + \
        tuple((m._buffers, name) for m in mods for name, p in m._buffers.items() if p is not None)
    return shape, children, params


def model_signature(model):
    """What a captured plan bakes in of the network: the parameters' and buffers' addresses and versions and every attribute that selects
    kernels (the documented A/B switches, the precision the guards may have moved it to).  Shared by KarrasModule and SIModule.
    Called once per run, so the walk over the module tree (0.26 ms of the 0.32 ms this took for PUNetG-64's 212 parameters) is
    done once per tree: the cached list holds the modules' own `_parameters` / `_modules` dicts, so a parameter assigned anew
    is read through its dict, and a submodule replaced, added or removed fails the identity / length check and rebuilds the
    list (shared parameters then appear once per owner: still a function of the same tensors)."""
    slots = model.__dict__.get("_signature_slots")
    if slots is not None:
        shape, children, params = slots
        if not (all(len(d) == n for d, n in shape) and all(d.get(name) is c for d, name, c in children)):
            slots = None
    if slots is None:
        slots = model.__dict__["_signature_slots"] = _tree_slots(model)
    out = []
    for d, name in slots[2]:
        p = d[name]
        out.append((p.data_ptr(), p._version) if p is not None else None)
    return tuple(out), tuple(getattr(model, a, None) for a in MODEL_SWITCHES)


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
        self.planned = bool(getattr(self.model, "forward_with_shifts", None)) and getattr(self.model, "capturable", True) and (
            like.dim() == 4 or (like.dim() == 5 and getattr(self.model, "dim", 2) == 3))
        # A field-valued conditional embedding (punetg.py:405-407): the time shifts are per-pixel MLPs of te(sigma) + ye, so they are
        # computed inside every evaluation -- from the tabulated te row and a plan-owned copy of ye, out of the network's workspace
        # (PUNetG.field_shifts), which keeps the run capturable
        self.field = bool(self.planned and self.conditional and getattr(self.model, "condition_is_field", None)
                          and self.model.condition_is_field(y))
        if self.field and not getattr(self.model, "field_shifts", None):
            self.planned = self.field = False
        # Classifier-free guidance evaluates the network twice on the same state; with tabulated conditioning the two
        # evaluations differ only in their time-shift rows, so they run as ONE evaluation of batch 2B (rows B.. are the
        # unconditional half): half the launches (config 5 at 16 samples per GPU: 9.05 -> 9.00 ms per pair sustained, 9.24 -> 8.84 on a cool chip).
        # Bit-identical to two evaluations: every kernel treats samples independently.  Not for PUNetGCond-style networks,
        # whose channel condition cannot be dropped (the reference's cannot run the unconditional branch either).
        self.batched_cfg = (self.planned and self.cfg and getattr(module, "batch_cfg", True)
                            and not hasattr(self.model, "_split_condition") and not self.field)
        # ... and the step kernels write the network input into both halves of the [2B, ...] buffer themselves (ds_eval_coef.xin_copies)
        self.xin_copies = 2 if self.batched_cfg else 1
        # an evaluated-as-given network inside a captured run (KarrasModule.capture_eager): the condition the captured calls read
        # is a plan-owned copy that refresh() rewrites
        self.static_condition = False
        # the range guard's result check rides on the run's last step kernel (nets/precision.py)
        self.nonfinite_word = precision.result_word(self.model, like.device) if like.is_cuda else None
        self._out = {}
        self.shifts_c = self.shifts_u = self.shifts_cu = None

    def _tables(self, ye):
        """Time-conditioning rows of every evaluation: [n_evals, C] per block, or [n_evals, B, C] when the embedded
        condition differs per sample (punetg.py:400-410 adds ye to the time embedding row by row)."""
        m, cn, n = self.model, self._cn, self._cn.numel()
        if ye is not None and ye.shape[0] != 1:
            if ye.shape[0] != self.batch:
                raise ValueError(f"conditional embedding batch {ye.shape[0]} does not match the {self.batch} samples")
            B = self.batch
            te = m.embed_time(cn.repeat_interleave(B), ye.repeat(n, 1))          # row e*B + b = (eval e, sample b)
            return [s.view(n, B, s.shape[1]) for s in m.time_shifts(te)]
        return m.time_shifts(m.embed_time(cn, ye))

    def prepare(self, table: StepTable):
        evals = table.evals
        dev = self.like.device
        cn = torch.tensor([e.c_noise for e in evals], dtype=torch.float32)
        if self.planned:
            # sigma is shared by the batch, so the whole time-conditioning path (Fourier features,
            # conditional embedding, 14 time MLPs) is evaluated once for all evaluations of the run.
            m = self.model
            self._cn = cn.to(dev)
            ye = m.embed_condition(self.y) if self.conditional else None
            if self.field:
                self.te_rows = m.embed_time(self._cn)              # [n_evals, C]: the Fourier features of every evaluation
                self.ye_field = ye.clone()
                self.shifts_c = []
            else:
                self.shifts_c = self._tables(ye)
            self.shifts_u = self._tables(None) if (self.cfg or not self.conditional) else None
            if not self.conditional:
                self.shifts_c = self.shifts_u
            if self.batched_cfg:
                B = self.batch
                self.shifts_cu = [torch.empty((c.shape[0], 2 * B, c.shape[-1]), dtype=torch.float32, device=dev)
                                  for c in self.shifts_c]
                for cu, u in zip(self.shifts_cu, self.shifts_u):
                    cu[:, B:].copy_(u[:, None, :].expand(-1, B, -1))
                self._fill_conditional_half()
        else:
            self.cnoise = cn[:, None].expand(len(evals), self.batch).contiguous().to(dev)
            if self.static_condition:
                self.y = clone_condition(self.y)

    def _fill_conditional_half(self):
        B = self.batch
        for cu, c in zip(self.shifts_cu, self.shifts_c):
            cu[:, :B].copy_(c if c.dim() == 3 else c[:, None, :].expand(-1, B, -1))

    def refresh(self, y):
        """A new condition of the same structure for an existing (possibly captured) run: recompute what depends on
        y INTO the buffers the launch sequence already reads.  The unconditional tables depend only on the step
        table and the weights, both part of the plan key."""
        if self.static_condition and not self.planned:
            copy_condition(self.y, y)
            return
        self.y = y
        if not (self.planned and self.conditional):
            return
        with torch.inference_mode():                 # the tables may have been created under inference_mode
            ye = self.model.embed_condition(y)       # PUNetGCond also refreshes its channel-field buffer here
            if self.field:
                if tuple(ye.shape) != tuple(self.ye_field.shape):
                    raise RuntimeError("the condition changed shape under a captured plan (plan key out of date)")
                self.ye_field.copy_(ye)
                return
            new = self._tables(ye)
            if len(new) != len(self.shifts_c) or any(a.shape != b.shape for a, b in zip(new, self.shifts_c)):
                raise RuntimeError("the condition changed shape under a captured plan (plan key out of date)")
            for dst, src in zip(self.shifts_c, new):
                dst.copy_(src)
            if self.batched_cfg:
                self._fill_conditional_half()

    def _buf(self, slot, name):
        key = (slot, name)
        if key not in self._out:
            self._out[key] = torch.empty_like(self.like)
        return self._out[key]

    def evaluate(self, state, xin, row, index, slot):
        if self.planned and self.batched_cfg:
            B = self.batch                      # xin is [2B, ...]: both halves written by the step kernel that produced it
            key = (slot, "cu")
            if key not in self._out:
                self._out[key] = torch.empty_like(xin)
            f2 = self.model.forward_with_shifts(xin, self.shifts_cu, row=index, out=self._out[key])
            return f2[:B], f2[B:]
        if self.planned:
            if self.field:
                shifts = self.model.field_shifts(self.te_rows[index:index + 1], self.ye_field, self.batch)
                f = self.model.forward_with_shifts(xin, shifts, row=None, out=self._buf(slot, "c"))
            else:
                f = self.model.forward_with_shifts(xin, self.shifts_c, row=index, out=self._buf(slot, "c"))
            fu = None
            if self.cfg:
                fu = self.model.forward_with_shifts(xin, self.shifts_u, row=index, out=self._buf(slot, "u"))
            return f, fu
        cn = self.cnoise[index]
        # our networks' domain guards (nets/precision.py) cost a host read: the run checks once at its end, not per evaluation
        net = getattr(self.model, "forward_unguarded", self.model)
        if self.conditional:
            f = net(xin, cn, self.y)
            fu = net(xin, cn) if self.cfg else None
        else:
            f, fu = net(xin, cn), None
        ops.require_device(f, "model output")
        return f.contiguous(), (None if fu is None else fu.contiguous())


def device_generator_state(device, counters):
    """(seed, offset) for one stochastic run from torch's CUDA generator of `device` -- so torch.manual_seed(s)
    makes the in-kernel noise reproducible, and consecutive runs draw disjoint counter ranges -- advancing the
    generator by what the run consumes (torch keeps its Philox offset a multiple of 4)."""
    gen = torch.cuda.default_generators[device.index if device.index is not None else torch.cuda.current_device()]
    seed, offset = int(gen.initial_seed()), int(gen.get_offset())
    gen.set_offset(offset + (int(counters) + 3) // 4 * 4)
    return seed, offset


class Loop:
    """Buffers + launch sequence of one tabulated run.  Construction allocates everything and lets
    the source precompute its per-evaluation tables; ``launch`` only enqueues kernels (plus
    whatever the source's evaluation does), so for a planned ModuleSource it can be captured into
    a hipGraph and replayed.

    Noise of the stochastic integrators (reference: one torch.randn_like(x) per step, integrators.py:66-69,103-104):
      injected_noise=True   eps [nsteps, *shape] supplied by the caller (parity runs replay the reference's draws);
      injected_noise=False  generated inside the churn / Euler-Maruyama kernels by Philox from a 16-byte device
                            state (seed, base offset) that set_noise() rewrites -- no eps buffer, graph-capturable.

    Usage: loop.load(x0[, scale]); loop.set_noise(eps); loop.launch(); loop.result()."""

    def __init__(self, table: StepTable, source, like, record_history=False, injected_noise=False, noise_shard=None):
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
        copies = getattr(source, "xin_copies", 1)
        self.nonfinite_word = getattr(source, "nonfinite_word", None)
        self.xin = (new() if copies == 1 else torch.empty((copies * shape[0],) + shape[1:], dtype=torch.float32, device=dev)) \
            if source.wants_xin else None
        self.tmp = new() if (not source.wants_xin or table.kind == "karras") else None
        self.tmp2 = new() if (not source.wants_xin and table.kind == "karras") else None
        self.eps = self.rng = None
        # noise_shard = (first element, total elements): this state is rows [lo, hi) of a larger batch sampled by several
        # ranks (parallel.sample_sharded).  The in-kernel stream is then addressed as the single process would address it --
        # element e of step i reads counter base + i * ceil(total / 4) + (first + e) / 4 -- so the shards of a stochastic
        # run draw disjoint noise from one seed and reproduce the unsharded run.
        first, total = (0, like.numel()) if noise_shard is None else (int(noise_shard[0]), int(noise_shard[1]))
        if first % 4 or first < 0 or first + like.numel() > total:
            raise ValueError("noise_shard: the shard must start at a multiple of 4 elements and lie inside the total")
        self.counters_per_step = ops.philox_counters(total)
        self.noise_base = first // 4
        if table.needs_noise:
            if injected_noise:
                self.eps = torch.empty((n,) + shape, dtype=torch.float32, device=dev)
            else:
                self.rng = torch.zeros(2, dtype=torch.int64, device=dev)
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

    def set_noise(self, eps=None, seed_offset=None):
        """Injected mode: eps [>= nsteps, *shape] is copied in.  Generator mode: (seed, offset) -- given, or taken from
        (and advancing) torch's CUDA generator -- is written to the device state the kernels read."""
        if not self.table.needs_noise:
            return
        n = len(self.table.rows)
        if self.eps is not None:
            if eps is None:
                raise ValueError("this loop was built for injected noise: pass eps [nsteps, *x.shape]")
            if eps.shape[0] < n or tuple(eps.shape[1:]) != tuple(self.eps.shape[1:]):
                raise ValueError("eps must be [nsteps, *x.shape]")
            self.eps.copy_(eps[:n])
            return
        if eps is not None:
            raise ValueError("this loop generates its noise in the kernels: build it with injected_noise=True to pass eps")
        if seed_offset is None:
            seed_offset = device_generator_state(self.x.device, n * self.counters_per_step)
        seed, offset = seed_offset
        self.seed_offset = (int(seed), int(offset))
        as_i64 = lambda v: v - (1 << 64) if v >= (1 << 63) else v      # noqa: E731  (uint64 bit pattern in an int64 tensor)
        self.rng.copy_(torch.tensor([as_i64(int(seed) & (2**64 - 1)), as_i64(int(offset) & (2**64 - 1))], dtype=torch.int64))

    def step_noise(self, i):
        """The eps of step i as a tensor (generator mode regenerates it from the counters: tests / diagnostics)."""
        if self.eps is not None:
            return self.eps[i]
        return ops.philox_normal(self.rng, i * self.counters_per_step + self.noise_base, self.x.shape)

    def launch(self, max_rows=None):
        """Enqueue the run.  max_rows: only the first rows (the warm-up before a capture: one step touches every buffer the
        network and the source allocate lazily -- both evaluation slots, every workspace shape -- at a fraction of the run)."""
        table, source = self.table, self.source
        n = len(table.rows)
        kind, g = source.input_kind, source.guidance
        xin, tmp, eps = self.xin, self.tmp, self.eps
        karras = table.kind == "karras"
        em = table.kind == "euler-maruyama"
        cur = self.x
        e = 0
        copies = getattr(source, "xin_copies", 1)
        if source.wants_xin and n > 0 and not karras:
            r0 = table.rows[0].first
            for half in (xin.view((copies,) + tuple(cur.shape)) if copies > 1 else (xin,)):
                if r0.scaled:                                        # c_in * (x / s): schedulers.py:287, karrasmodule.py:702
                    ops.scale(ops.div_scalar(cur, r0.scale, out=half), r0.c_in, out=half)
                else:
                    ops.scale(cur, r0.c_in, out=half)                # c_in*x, karrasmodule.py:702
        for i, row in enumerate(table.rows):
            if max_rows is not None and i >= max_rows:
                break
            nxt = self.history[i + 1] if self.record_history else cur
            nxt_row = table.rows[i + 1] if i + 1 < n else None
            chain = source.wants_xin and nxt_row is not None and not karras
            c_in_next = nxt_row.first.c_in if chain else 1.0
            s_next = nxt_row.first.scale if chain else 1.0
            xin_next = xin if chain else None
            base = cur
            eps_i = eps[i] if eps is not None else None
            philox_i = (self.rng, i * self.counters_per_step + self.noise_base) if (self.rng is not None) else None
            if karras:
                base = tmp                                           # x_hat, integrators.py:104-105
                ops.churn(cur, eps_i, row.churn_coef, xhat_out=base, xin_out=xin, c_in=row.first.c_in, philox=philox_i,
                          ratio=row.churn_ratio, scale=row.first.scale, xin_copies=copies)
            word = self.nonfinite_word if i == n - 1 else None      # the last step's result is the run's
            k1 = row.first.coef(kind, g, next_scale=s_next if row.second is None else row.second.scale, xin_copies=copies,
                                nonfinite=word if row.second is None else None)
            f1, f1u = source.evaluate(base, xin, row.first, e, 0)
            e += 1
            if row.second is None:
                ops.euler(base, f1, k1, row.dt, fu=f1u, x_out=nxt, xin_out=xin_next, c_in_next=c_in_next,
                          eps=eps_i if em else None, philox=philox_i if em else None,
                          noise_coef=row.noise_coef, sqrt_abs_dt=row.sqrt_abs_dt)
            else:
                k2 = row.second.coef(kind, g, next_scale=s_next, xin_copies=copies, nonfinite=word)
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


class _TorchGraph:
    """loop.launch() captured by torch (see PlanCache.run)."""

    def __init__(self, loop, stream):
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph, stream=stream):
            loop.launch()
        self.nodes = -1

    def launch(self):
        self.graph.replay()


class PlanCache:
    """Captured runs (static buffers + hipGraph), keyed by everything a capture bakes in; shared by KarrasModule and
    SIModule.  hipGraph capture needs a non-default stream: planned runs live on a side stream that is ordered after
    the caller's stream on entry and before it on exit."""

    def __init__(self, capacity=8):
        self.capacity = capacity
        self.plans = {}
        self.stream = None

    def clear(self):
        self.plans = {}

    def run(self, key, make_loop, x, y=None, scale=None, eps=None, torch_graph=False):
        """torch_graph: capture with torch.cuda.CUDAGraph instead of ops.Graph -- torch's allocator then serves allocations made
        inside the captured region from a pool the graph owns, which is what a run through user torch modules (extra_residual, an
        evaluated-as-given network) needs; our own capture refuses allocations instead (ops.Graph)."""
        if self.stream is None or self.stream.device != x.device:
            self.stream = torch.cuda.Stream(device=x.device)
        caller = torch.cuda.current_stream(x.device)
        self.stream.wait_stream(caller)
        with torch.cuda.stream(self.stream):
            plan = self.plans.pop(key, None)
            if plan is not None:
                self.plans[key] = plan                       # least recently USED goes first: a hit moves the plan to the back
            if plan is None:
                loop = make_loop()
                loop.load(x, scale)
                loop.set_noise(eps)
                # eager warm-up: allocates the workspace, packs the weights, validates shapes.  Two steps, not the run (round 3: the
                # first call of config 3 took 6.9 s, 3.2 s of them this pass): the first step visits both evaluation slots and every
                # buffer shape, the second the chained form of the first evaluation
                loop.launch(max_rows=2)
                self.stream.synchronize()
                if torch_graph:
                    try:
                        g = _TorchGraph(loop, self.stream)
                    except Exception as e:
                        raise RuntimeError("capture_eager: this run could not be captured by torch.cuda.CUDAGraph -- a module evaluated "
                                           "as given must be capture-safe (no host reads such as .item(), no data-dependent control "
                                           "flow); set capture_eager = False to run it step by step") from e
                else:
                    with ops.Graph() as g:
                        loop.launch()
                plan = (loop, g)
                if len(self.plans) >= self.capacity:
                    self.plans.pop(next(iter(self.plans)))
                self.plans[key] = plan
                if loop.table.needs_noise and loop.rng is not None:
                    loop.set_noise(None, seed_offset=loop.seed_offset)    # the replay below repeats the eager pass's draw
                    replay_same_noise = True
                else:
                    replay_same_noise = False
            else:
                replay_same_noise = False
                refresh = getattr(plan[0].source, "refresh", None)
                if refresh is not None:
                    refresh(y)
            loop, g = plan
            loop.load(x, scale)
            if not replay_same_noise:
                loop.set_noise(eps)
            g.launch()
            out = loop.result().clone()
        caller.wait_stream(self.stream)
        return out


def run_table(table: StepTable, source, x, record_history=False, eps=None):
    """One eager pass: returns the final state (a new tensor) or the history [len(rows)+1, *x.shape]."""
    loop = Loop(table, source, x, record_history, injected_noise=eps is not None)
    loop.load(x)
    loop.set_noise(eps)
    loop.launch()
    return loop.result()
