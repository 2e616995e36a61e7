"""Environment-axis sharding across the GPUs of a node: one process per GPU, no collective on the
step path.

Every environment is independent (no cross-env data, no shared RNG), so rank r simply owns the
contiguous slice [lo, hi) of the global environment axis and seeds it with the *global* index
(environment i always gets seed base_seed + i, whatever the number of GPUs: results are bitwise
independent of the sharding).  step() launches only local kernels.  The only communication is the
optional host-side hand-off of results to a learner: `gather_env_axis` (all_gather over
torch.distributed: RCCL on GPUs, gloo in the CPU tests) and `scatter_env_axis` for actions that a
central policy produced on rank 0.
"""
import torch
import torch.distributed as dist


def shard_range(total, world_size, rank):
    """Contiguous partition of range(total): the first (total % world_size) ranks get one extra."""
    if not (0 <= rank < world_size) or total < 0:
        raise ValueError("bad shard request: total=%r world_size=%r rank=%r" % (total, world_size, rank))
    base, extra = divmod(total, world_size)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def shard_sizes(total, world_size):
    return [shard_range(total, world_size, r)[1] - shard_range(total, world_size, r)[0] for r in range(world_size)]


def _world():
    if dist.is_available() and dist.is_initialized():
        return dist.get_world_size(), dist.get_rank()
    return 1, 0


def gather_env_axis(local, total, group=None):
    """Concatenate per-rank tensors [n_r, ...] into [total, ...] on every rank (uneven shards allowed)."""
    world, rank = _world()
    if world == 1:
        return local
    sizes = shard_sizes(total, world)
    assert local.shape[0] == sizes[rank], (local.shape, sizes, rank)
    pad = max(sizes)
    buf = local
    if local.shape[0] < pad:
        buf = torch.zeros((pad,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
        buf[:local.shape[0]] = local
    out = [torch.empty_like(buf) for _ in range(world)]
    dist.all_gather(out, buf.contiguous(), group=group)
    return torch.cat([o[:n] for o, n in zip(out, sizes)], 0)


def scatter_env_axis(full, total, src=0, like=None, group=None):
    """Rank `src` holds [total, ...]; every rank receives its own slice [n_r, ...]."""
    world, rank = _world()
    if world == 1:
        return full
    sizes = shard_sizes(total, world)
    pad = max(sizes)
    ref = full if rank == src else like
    recv = torch.empty((pad,) + tuple(ref.shape[1:]), dtype=ref.dtype, device=ref.device)
    chunks = None
    if rank == src:
        chunks = []
        for r in range(world):
            lo, hi = shard_range(total, world, r)
            c = torch.zeros_like(recv)
            c[:hi - lo] = full[lo:hi]
            chunks.append(c)
    dist.scatter(recv, chunks, src=src, group=group)
    return recv[:sizes[rank]]


class ShardedPcgrlEnv:
    """The rank-local shard of a `total_envs` batch.  Same surface as BatchedPcgrlEnv; step() and reset()
    touch only this rank's GPU.  Construct it after torch.distributed.init_process_group (or alone)."""

    def __init__(self, prob="binary", rep="narrow", total_envs=1, base_seed=0, device=None, auto_reset=True):
        from .envs import BatchedPcgrlEnv
        self.world_size, self.rank = _world()
        self.total_envs = int(total_envs)
        self.lo, self.hi = shard_range(self.total_envs, self.world_size, self.rank)
        if device is None:
            device = "cuda:%d" % (self.rank % max(torch.cuda.device_count(), 1))
        self.env = BatchedPcgrlEnv(prob=prob, rep=rep, num_envs=self.hi - self.lo, device=device,
                                   seed=base_seed + self.lo, auto_reset=auto_reset)
        self.num_envs = self.env.num_envs

    def __getattr__(self, name):
        return getattr(self.env, name)

    def reset(self):
        return self.env.reset()

    def step(self, actions):
        return self.env.step(actions)

    def gather(self, tensor):
        return gather_env_axis(tensor, self.total_envs)

    def scatter_actions(self, actions_on_rank0, like=None):
        return scatter_env_axis(actions_on_rank0, self.total_envs, src=0, like=like)
