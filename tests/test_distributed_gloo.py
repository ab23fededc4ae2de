"""N>1 path on CPU: world_size-2 gloo processes exercise the batch sharding and the all-gather
of samples.  The sampler itself needs the GPU, so a stand-in module with a deterministic
per-sample map takes its place; what is tested is that shard + gather reproduces the
single-process batch for even and ragged splits."""
import os
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


class _StandIn:
    device = torch.device("cpu")
    noise_shard = None          # KarrasModule's attribute: (first element, total elements) of this rank's rows
    seen = None

    def propagate_white_noise(self, x, y=None, guidance=1.0, nsteps=100, integrator=None):
        type(self).seen = self.noise_shard
        return torch.tanh(x) * nsteps + x.flatten(1).sum(1).view(-1, 1, 1, 1)   # per-sample, deterministic


def _worker(rank, world, port, nsamples, q):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from diffsci_amd.parallel import sample_sharded
    out = sample_sharded(_StandIn(), nsamples, [1, 4, 4], nsteps=3, seed=7, gather=True)
    m = _StandIn()
    local = sample_sharded(m, nsamples, [1, 4, 4], nsteps=3, seed=7, gather=False)
    # the in-kernel noise of the stochastic integrators is addressed as the unsharded run's: the module is told where
    # this rank's rows start and how long the whole batch is, and the setting does not outlive the call
    from diffsci_amd.parallel import shard_rows
    lo, _ = shard_rows(nsamples, world, rank)
    assert _StandIn.seen == (lo * 16, nsamples * 16) and m.noise_shard is None
    q.put((rank, out.numpy().copy(), local.numpy().copy()))     # plain bytes: a tensor travels as a shared-memory handle the parent must
                                                                # open while this process is still alive (an EOFError under load, once)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("nsamples", [8, 5])
def test_two_rank_shard_and_gather(nsamples):
    from diffsci_amd.parallel import global_white_noise, shard_rows
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    import socket
    with socket.socket() as sock:                       # a port the kernel says is free (a pid-derived one collided once)
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    procs = [ctx.Process(target=_worker, args=(r, 2, port, nsamples, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    want = _StandIn().propagate_white_noise(global_white_noise(nsamples, [1, 4, 4], 7), nsteps=3)
    for rank, full, local in got:
        full, local = torch.from_numpy(full), torch.from_numpy(local)
        assert torch.equal(full, want)
        lo, hi = shard_rows(nsamples, 2, rank)
        assert torch.equal(local, want[lo:hi])
