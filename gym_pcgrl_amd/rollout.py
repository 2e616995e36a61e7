"""Device-resident rollout collection for an on-policy trainer (the role of stable-baselines' runner behind the
reference's `train.py:86-94`: `model.learn` steps a VecEnv `n_steps` at a time and hands PPO the stacked
observations, actions, rewards and episode starts).

Everything stays on the GPU: the step writes the wrapped observation of step t+1 straight into row t+1 of the preallocated
`[T, N, ...]` storage (the observation target of the environment is moved from row to row: no copy of the image, which is
the bulk of a rollout's bytes); rewards / dones are small and copied from the environment's own tensors.
No policy or optimiser lives here (out of scope): `policy(obs) -> actions` is any callable on device
tensors.
"""
from collections import OrderedDict


class RolloutBuffer:
    """Preallocated `[T, N, ...]` device storage of one rollout."""

    def __init__(self, torch, n_steps, num_envs, obs_shape, action_shape, device):
        z = lambda shape, dt: torch.zeros(shape, dtype=dt, device=device)
        self.n_steps, self.num_envs = n_steps, num_envs
        self.obs = z((n_steps, num_envs) + tuple(obs_shape), torch.uint8)
        self.actions = z((n_steps, num_envs) + tuple(action_shape), torch.int64)
        self.rewards = z((n_steps, num_envs), torch.float64)
        self.dones = z((n_steps, num_envs), torch.bool)
        self.episode_starts = z((n_steps, num_envs), torch.bool)     # obs[t] is the first observation of an episode
        self.last_obs = z((num_envs,) + tuple(obs_shape), torch.uint8)
        # asynchronous collection (collect(policy, pop_budget=...)): row t is a TICK.  took[t, e]: environment e took actions[t, e]
        # (from obs[t, e]); fresh[t, e]: a step of e completed in tick t -- rewards[t, e] / dones[t, e] are the outcome of the LAST
        # action it took (at tick t or earlier) and the next row's observation is the one that follows it; where fresh is false the
        # collector writes reward 0 and done False.  Lockstep: all true.
        self.took = torch.ones((n_steps, num_envs), dtype=torch.bool, device=device)
        self.fresh = torch.ones((n_steps, num_envs), dtype=torch.bool, device=device)

    def as_dict(self):
        return OrderedDict(obs=self.obs, actions=self.actions, rewards=self.rewards, dones=self.dones,
                           episode_starts=self.episode_starts, last_obs=self.last_obs, took=self.took, fresh=self.fresh)


class RolloutCollector:
    """Steps a `BatchedVecEnv` (utils.make_vec_envs) with `policy` and fills a RolloutBuffer.

    collect() continues from where the previous call stopped (like a SB runner: the environment is only reset
    once), so consecutive rollouts tile the same trajectories.  With `vec_env.monitor` the finished episodes'
    returns/lengths accumulate in `episode_returns` / `episode_lengths` (device tensors, no host sync)."""

    def __init__(self, vec_env, n_steps):
        self.env = vec_env
        self.torch = vec_env.env.pcgrl_env._torch
        dev = vec_env.env.pcgrl_env.device
        a = vec_env.action_space
        ashape = () if hasattr(a, "n") else (len(a.nvec),)
        self.buffer = RolloutBuffer(self.torch, n_steps, vec_env.num_envs, vec_env.observation_space.shape, ashape, dev)
        self._obs = None
        # The step writes its image straight into a buffer row when every row starts 16-byte aligned (pcgrl_bind_observation
        # wants that); a row is num_envs * h * w * depth bytes, so e.g. three 10 x 10 x 5 environments do not qualify -- the
        # wrapper's own tensor then stays bound and each image is copied into its row (what every collector did before round 3).
        row_bytes = int(self.buffer.last_obs.numel())
        self.direct = (row_bytes % 16 == 0 and self.buffer.obs.data_ptr() % 16 == 0 and self.buffer.last_obs.data_ptr() % 16 == 0)
        self._start = self.torch.ones(vec_env.num_envs, dtype=self.torch.bool, device=dev)
        self.episode_returns, self.episode_lengths = [], []      # per step, the last `keep_steps` steps (default: one rollout)
        self.keep_steps = int(n_steps)

    def collect(self, policy, pop_budget=None):
        """pop_budget: collect with asynchronous ticks (BatchedPcgrlEnv.tick; sokoban / mdungeon / ddave) -- a row of the buffer is then a
        tick, `took` / `fresh` say which environments acted in it and which completed a step (an environment whose search is
        suspended sits ticks out; the policy's action for it is ignored; its reward and done in those rows are zero).  Per environment the rows with `took` / `fresh` set, in
        order, are the transitions a lockstep rollout holds.  No host synchronisation either way."""
        self.begin(pop_budget)
        for t in range(self.buffer.n_steps):
            self.step(t, policy, pop_budget)
        return self.buffer

    def begin(self, pop_budget=None):
        """The head of a rollout: the first observation in row 0 (a reset the first time, afterwards where the previous rollout stopped).
        collect() = begin() + step(t) for every row; DoubleBufferedCollector calls the two itself, sub-batch by sub-batch."""
        torch, b = self.torch, self.buffer
        if pop_budget is not None and getattr(self, "_pending", None) is None:
            self._pending = torch.zeros(self.env.num_envs, dtype=torch.bool, device=b.obs.device)
        w = self.env.env                                  # the image wrapper below the Monitor layer: no host sync
        if self._obs is None:
            if self.direct:
                w.set_observation_target(b.obs[0])        # the first observation lands in row 0
            self._obs = self.env.reset()
            if not self.direct:
                b.obs[0].copy_(self._obs)
                self._obs = b.obs[0]
        else:
            b.obs[0].copy_(b.last_obs)                    # one copy per rollout: where the previous one stopped
            self._obs = b.obs[0]

    def step(self, t, policy, pop_budget=None):
        """Row t of the rollout: the policy's actions for obs[t], one step (or tick) of the environment, its outcome."""
        torch, b = self.torch, self.buffer
        asynchronous = pop_budget is not None
        w = self.env.env
        direct = self.direct
        b.episode_starts[t].copy_(self._start)
        actions = policy(self._obs)
        b.actions[t].copy_(actions)
        nxt = b.obs[t + 1] if t + 1 < b.n_steps else b.last_obs
        if direct:
            w.set_observation_target(nxt)             # the step writes the next row itself
        if asynchronous:
            b.took[t].copy_(~self._pending)
            self._obs, rew, done, _, pend = w.tick(actions, pop_budget=pop_budget)
            self._pending = pend != 0
            b.fresh[t].copy_(~self._pending)
        else:
            self._obs, rew, done, _ = w.step(actions)
        if not direct:
            nxt.copy_(self._obs)
            self._obs = nxt
        b.rewards[t].copy_(rew)
        b.dones[t].copy_(done)
        if asynchronous:       # an episode starts where a step completed in this tick with done set; pending rows keep their flag
            self._start = torch.where(b.fresh[t], done.to(torch.bool), self._start)
            b.dones[t] &= b.fresh[t]
            # ... and no reward: the environment's reward row still holds that of its last completed step, which an earlier row of
            # the buffer already carries -- a consumer that sums rewards over rows (returns, GAE) without looking at `fresh` would
            # count it once per tick sat out.  (The obs rows of such ticks are in-flight images: pair `took` rows with the `fresh` rows that follow them.)
            b.rewards[t].masked_fill_(~b.fresh[t], 0)
        else:
            self._start = done.to(torch.bool).clone()
        if self.env.monitor:
            st = self.env.episode_stats()
            m = b.dones[t]
            self.episode_returns.append(torch.where(m, st["last_return"], torch.full_like(st["last_return"], float("nan"))))
            self.episode_lengths.append(torch.where(m, st["last_length"], torch.zeros_like(st["last_length"])))
            if len(self.episode_returns) > self.keep_steps:      # a bounded window (a trainer reads it once per rollout)
                del self.episode_returns[0], self.episode_lengths[0]


class DoubleBufferedCollector:
    """A rollout of K sub-batches of ONE GPU (K = 2: double buffering), a RolloutCollector and a stream each: row t of sub-batch A --
    policy, step, bookkeeping -- is issued on A's stream, then row t of B on B's, and so on.  Nothing orders A's row t + 1 behind B's
    row t, so the device runs the policy of one sub-batch next to the environment step of the other, and the tail of one sub-batch's
    step kernel (single wavefronts finishing the longest tasks, most SIMDs idle) under the front of the other's: with no policy
    at all 65 536 environments step in 26.4 us (binary-narrow) / 24.8 us (zelda-wide 11x16) as two halves against 28.3 / 27.5 us as
    one batch (bench.py configs.C2_sub2 / C3_sub2).  The reference's counterpart is SubprocVecEnv's worker processes stepping
    while the learner thinks (utils.py:60-71); here it is two streams.

    `vec_envs`: BatchedVecEnv's of the same kind on the same device -- for the trajectories of one batch of N environments seeded
    `seed`, make sub-batch k with `seed + k * N / K` (environment i of a batch is seeded seed + i).  `buffers[k]` / collect()'s
    return value: the sub-batches' RolloutBuffer's ([T, N / K, ...] each; concatenate along axis 1 where one tensor is wanted).
    `policy(obs)` is called per sub-batch, on that sub-batch's stream."""

    def __init__(self, vec_envs, n_steps):
        from .node import side_by_side_streams
        self.parts = [RolloutCollector(v, n_steps) for v in vec_envs]
        self.torch = self.parts[0].torch
        self.device = self.parts[0].buffer.obs.device
        for p in self.parts:
            if p.buffer.obs.device != self.device:
                raise ValueError("the sub-batches of a DoubleBufferedCollector live on one device (several GPUs: node.MultiGpuPcgrlEnv)")
        self.streams = side_by_side_streams(self.torch, self.device, len(self.parts))
        self.buffers = [p.buffer for p in self.parts]

    def collect(self, policy, pop_budget=None):
        torch = self.torch
        cur = torch.cuda.current_stream(self.device)
        for s in self.streams:
            s.wait_stream(cur)                            # what the caller queued (parameters of the policy, the previous rollout's consumer)
        for p, s in zip(self.parts, self.streams):
            with torch.cuda.stream(s):
                p.begin(pop_budget)
        for t in range(self.parts[0].buffer.n_steps):
            for p, s in zip(self.parts, self.streams):
                with torch.cuda.stream(s):
                    p.step(t, policy, pop_budget)
        for s in self.streams:
            cur.wait_stream(s)                            # the rollout is complete for whatever the caller queues next
        return self.buffers
