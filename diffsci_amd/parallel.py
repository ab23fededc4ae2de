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
    if per_row % 16 or per_row == 0 or total * per_row < 16:
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
