"""Multi-GPU sampling: one process per GPU, the batch sharded by rows, no collective inside the
loop, one RCCL all-gather of the finished samples.

Samples are independent (every norm and the attention are per sample), so rank r simply owns
rows [r*G/W, (r+1)*G/W) of the global batch.  The white noise is defined exactly as the
single-process reference defines it -- ``torch.manual_seed(seed); torch.randn(G, *shape)`` on the
CPU generator (karrasmodule.py:837) -- and each rank keeps its rows, so the gathered result does
not depend on the number of ranks."""
import torch
import torch.distributed as dist


def shard_rows(total, world, rank):
    """Contiguous row range of `rank`; the first total % world ranks get one extra row."""
    base, extra = divmod(total, world)
    start = rank * base + min(rank, extra)
    return start, start + base + (1 if rank < extra else 0)


_CHUNK_FLOATS = 1 << 24        # 64 MiB of throw-away noise at a time while skipping other ranks' rows
_SKIP_AHEAD_OK = None          # once per process: does this torch's CPU normal fill skip ahead the way the chunked draw assumes?


def skip_ahead_matches_full_draw():
    """The chunked draw below leans on an implementation detail of the installed torch (CPU normal_fill: one uniform per
    element, Box-Muller in aligned blocks of 16); a torch upgrade that changes it would hand every rank VALID but DIFFERENT
    noise than the unsharded draw, silently.  So the first use in a process draws a small case both ways (rows of 16, 32 and
    48 floats, skipped in one and in several chunks) and the chunked path is used only if they agree bit for bit."""
    global _SKIP_AHEAD_OK
    if _SKIP_AHEAD_OK is None:
        ok = True
        for per_row, total, lo, hi, chunk in ((16, 5, 2, 4, 1), (32, 7, 3, 7, 2), (48, 4, 1, 2, 8)):
            g = torch.Generator()
            g.manual_seed(12345 + per_row)
            full = torch.randn(total, per_row, generator=g)[lo:hi]
            g.manual_seed(12345 + per_row)
            r = 0
            while r < lo:
                n = min(chunk, lo - r)
                torch.randn(n, per_row, generator=g)
                r += n
            ok = ok and torch.equal(torch.randn(hi - lo, per_row, generator=g), full)
        _SKIP_AHEAD_OK = bool(ok)
        if not ok:
            import warnings
            warnings.warn("diffsci_amd.parallel: this torch's CPU normal fill does not skip ahead row by row; every rank draws the "
                          "full global noise tensor and keeps its rows (same values, more host memory)", RuntimeWarning)
    return _SKIP_AHEAD_OK


def global_white_noise(total, shape, seed, rows=None):
    """Rows [rows[0], rows[1]) of torch.randn(total, *shape) under manual_seed(seed) (CPU
    generator), without disturbing the global generator.

    A rank needs only its own rows.  torch's CPU normal fill consumes one uniform per element, in order, and turns
    them into normals in aligned blocks of 16, so when a row holds a multiple of 16 floats the rows before `lo` can
    be drawn (and dropped) chunk by chunk and the rows after `hi` not at all -- same values as the full draw
    (tests/test_host_logic.py::test_global_noise_rows_without_the_full_tensor), without materialising G x shape
    floats on every rank (config 5: 134 MB per rank per run)."""
    g = torch.Generator()
    g.manual_seed(seed)
    if rows is None:
        return torch.randn(total, *shape, generator=g)
    lo, hi = rows
    per_row = 1
    for d in shape:
        per_row *= int(d)
    if per_row % 16 or per_row == 0 or total * per_row < 16 or not skip_ahead_matches_full_draw():
        return torch.randn(total, *shape, generator=g)[lo:hi].contiguous()
    chunk_rows = max(1, _CHUNK_FLOATS // per_row)
    r = 0
    while r < lo:                                             # advance the generator past the rows of lower ranks
        n = min(chunk_rows, lo - r)
        torch.randn(n, per_row, generator=g)
        r += n
    return torch.randn(hi - lo, *shape, generator=g)


def sample_sharded(module, nsamples, shape, nsteps=100, seed=0, y=None, guidance=1.0, integrator=None,
                   gather=True, white_noise=None):
    """Global batch of `nsamples` across the ranks of the default process group.
    Returns all samples on every rank (gather=True) or the local shard."""
    world = dist.get_world_size() if dist.is_initialized() else 1
    rank = dist.get_rank() if dist.is_initialized() else 0
    lo, hi = shard_rows(nsamples, world, rank)
    if white_noise is None:
        local = global_white_noise(nsamples, list(shape), seed, rows=(lo, hi))
    else:
        local = white_noise[lo:hi]
    local = local.to(module.device)
    # stochastic integrators: every rank addresses the in-kernel noise stream as the single process would (engine.Loop), so with
    # the same torch.manual_seed on every rank the shards draw disjoint noise and the gathered batch is the unsharded run's
    per_row = local[0].numel() if local.shape[0] else 1
    shard = (lo * per_row, nsamples * per_row) if (world > 1 and per_row % 4 == 0 and hasattr(module, "noise_shard")) else None
    if shard is not None:
        module.noise_shard = shard
    try:
        out = module.propagate_white_noise(local, y=y, guidance=guidance, nsteps=nsteps, integrator=integrator)
    finally:
        if shard is not None:
            module.noise_shard = None
    if not gather or not dist.is_initialized():
        return out
    return gather_samples(out, nsamples)


def gather_samples(out, nsamples=None):
    """All-gather the per-rank sample shards (RCCL over xGMI; backend "nccl" is RCCL on ROCm)."""
    if not dist.is_initialized():
        return out
    world = dist.get_world_size()          # a one-rank group still goes through the collective (RCCL on the GPU)
    if nsamples is None:
        nsamples = out.shape[0] * world
    if out.is_cuda and dist.get_backend() == "gloo":
        # a rehearsal of the N > 1 path with several ranks on ONE GPU (RCCL refuses two ranks on a device): gloo moves host memory
        return gather_samples(out.cpu(), nsamples).to(out.device)
    if nsamples % world == 0:
        full = torch.empty((nsamples,) + tuple(out.shape[1:]), dtype=out.dtype, device=out.device)
        dist.all_gather_into_tensor(full, out.contiguous())
        return full
    # ragged split: pad every shard to the largest one, gather, drop the padding rows
    sizes = [shard_rows(nsamples, world, r) for r in range(world)]
    most = max(b - a for a, b in sizes)
    padded = torch.zeros((most,) + tuple(out.shape[1:]), dtype=out.dtype, device=out.device)
    padded[:out.shape[0]] = out
    full = torch.empty((world * most,) + tuple(out.shape[1:]), dtype=out.dtype, device=out.device)
    dist.all_gather_into_tensor(full, padded)
    return torch.cat([full[r * most:r * most + (b - a)] for r, (a, b) in enumerate(sizes)], dim=0)
