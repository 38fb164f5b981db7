"""Index-sharded multi-GPU stepping: one process per GPU, `torch.distributed` (backend "nccl" = RCCL over xGMI).

The reference never shards bodies (each of its 2 GPUs holds the whole system and runs a different scenario,
hw5.cu:564-567,587-588); this is the new data-parallel scheme the north star asks for:

  rank r of P owns targets [r*N/P, (r+1)*N/P): their velocities live only on rank r;
  every rank holds ALL positions twice (ping-pong float4[N] {x,y,z,G*m}) because every target needs every source;
  per step: (1) force on own targets from gathered array `cur`, fused kick-drift written straight into this
  rank's slot of array `nxt`; (2) ONE in-place all-gather of `nxt` (send = own slot, recv = whole array);
  (3) swap.  Masses never change, so G*m travels inside the float4 and is never re-sent separately.

xGMI is point-to-point (7 links/GPU): the all-gather moves each 16*N/P-byte shard once per peer link —
2 MiB per rank at N=2^20/P=8, 8 MiB at N=2^22 — tens of microseconds against tens of milliseconds of compute.
`overlap=True` hides even that: a step is cut into phases over disjoint source ranges — the own shard
first (final as soon as this rank's previous launch has written it) while the asynchronous all-gather of the other
shards is still in flight, the remote shards after it — with the running sums kept in the kernel's workspace between
the phases.  Off by default: bench.py reports both on the multi-GPU node (`overlap_ab`).

`exchange="ring"` is the memory-scalable alternative (the ring-attention analogue): no rank ever holds all N
positions.  A rank keeps its own shard (ping-pong) and two travelling blocks of N/P sources; a step is P phases — the own
shard, then the block that has arrived from rank r-1 while the previous phase computed and the block before it was already
on its way to rank r+1 — so the footprint is 4·N/P records instead of 2·N and every transfer overlaps a phase.  With
288 GB per GPU the all-gather form never needs it at the BASELINE sizes (256 MiB at N = 2^24); it is here for systems
whose positions do not fit twice on one device, parity-tested against the all-gather form.

`shared_pairs` (default: on whenever it applies): the ranks share the UNORDERED pairs of the system instead — every pair is
evaluated once, by the rank that owns the earlier of its two 4096-body superblocks (cyclically), with the symmetric kernel
K1s (capi.launch_pair_forces_f32).  A rank then holds a partial force on ALL N bodies: per step ONE reduce-scatter (sum) of
float4[N] hands every shard owner its total, the owner kicks and drifts (capi.launch_kick_drift_f32), and the all-gather of
positions follows as before.  Needs whole superblocks per shard (N % (4096 P) == 0) and N >= 28672; K1s reaches ~0.79 of
the fp32 peak per GPU where the ordered-pair kernel K1 reaches ~0.59.

torch is used for device memory, the stream and the collective only; the arithmetic is the HIP kernel behind
`capi.launch_f32`.  `compute` is injectable so the sharding/exchange logic can be exercised on CPU with gloo
(tests/test_distributed_gloo.py passes the oracle there — test infrastructure, not a product fallback).
"""
import torch
import torch.distributed as dist

from . import capi


def shard_range(n, rank, world):
    """Contiguous index shard of rank `rank`; requires world | n (the configs use powers of two)."""
    if n % world:
        raise ValueError(f"n={n} must be divisible by world size {world}")
    per = n // world
    return rank * per, (rank + 1) * per


def workspace_bytes(n_src, n_tgt, acc64=False, targets_per_lane=0, j_split=0, source_path=0, wg_size=0, cap=8 << 30):
    """Bytes of the source-slice workspace for one rank's launches: running sum + compensation + one partial-sum slot per
    source slice of a launch.  The documented minimum is 16 slots (capi.workspace_bytes_f32); a shard whose plan cuts the
    sources into more slices gets up to 64 slots, so that the whole step is ONE force launch + ONE reducer instead of
    js/16 of each (N = 2^20 over 8 ranks, 64 slices: 29.7 instead of 30.5 ms per step and rank,
    profiles/r03_shard_slots.txt), within `cap` bytes."""
    base = capi.workspace_bytes_f32(n_tgt, acc64)
    rec = base // 18
    # (asked with room for 64 slices: a small shard's slice count is chosen within what ONE launch's workspace holds)
    _, js, _ = capi.plan_f32(n_src, n_tgt, acc64, targets_per_lane, j_split, 66 * rec, source_path, wg_size)
    slots = max(16, min(64, js, cap // rec - 2))
    sliced = (slots + 2) * rec
    if n_src == n_tgt and source_path in (0, 3):  # one rank holds the whole system: room for K1s' pair slots
        return max(sliced, capi.workspace_bytes_sym_f32(n_src, acc64))
    return sliced


def hip_compute(acc64=False, targets_per_lane=0, j_split=0, source_path=0, wg_size=0):
    """The product compute step: nb_launch_step_f32 on torch's current HIP stream.  The j-split workspace
    (partial sums when a shard's targets alone cannot fill the chip) is a torch tensor allocated once."""
    ws = {}

    def compute(src, out, vel, off, n_tgt, eps2, dt, pos64=None, vel64=None, src_range=None, phase=capi.NB_PHASE_WHOLE,
                tgt=None):
        if not src.is_cuda:
            raise RuntimeError("nbody_amd has no CPU compute path: tensors must live on a HIP device")
        key = (src.device, src.shape[0], n_tgt)
        if key not in ws:
            ws[key] = torch.empty(workspace_bytes(src.shape[0], n_tgt, acc64, targets_per_lane, j_split, source_path,
                                                  wg_size), dtype=torch.uint8, device=src.device)
        w = ws[key]
        stream = torch.cuda.current_stream(src.device).cuda_stream
        capi.launch_f32(src.data_ptr(), out.data_ptr(), src.shape[0], off, n_tgt, eps2, dt, stream,
                        vel_ptr=vel.data_ptr() if vel is not None else 0,
                        pos64_ptr=pos64.data_ptr() if pos64 is not None else 0,
                        vel64_ptr=vel64.data_ptr() if vel64 is not None else 0,
                        acc64=acc64, targets_per_lane=targets_per_lane, j_split=j_split, source_path=source_path,
                        wg_size=wg_size,
                        workspace_ptr=w.data_ptr() if w is not None else 0,
                        workspace_bytes=w.numel() if w is not None else 0, phase=phase,
                        src_begin=src_range[0] if src_range else 0, src_end=src_range[1] if src_range else 0,
                        tgt_ptr=tgt.data_ptr() if tgt is not None else 0)

    compute.is_hip_compute = True
    compute.forced = bool(targets_per_lane or j_split or wg_size or source_path in (1, 2))  # asks for the ordered kernel K1
    return compute


class _StagedRecv:
    """gloo rehearsal with device tensors: the receive lands in host memory and is copied to the device on wait()."""

    def __init__(self, works, host, dev):
        self.works, self.host, self.dev = works, host, dev

    def wait(self):
        for w in self.works:
            w.wait()
        self.dev.copy_(self.host)


class ShardedSystem:
    """N bodies sharded by index over the ranks of the default process group (or unsharded when world == 1)."""

    def __init__(self, n, pos_shard, vel_shard, eps, dt, device, compute=None, acc64=False, group=None, trace=False,
                 exchange="in_place", overlap=False, shared_pairs=None, pair_steps=None):
        self.dist_on = dist.is_initialized()  # a one-rank group still runs the collective (tests the in-place call)
        self.world = dist.get_world_size(group) if self.dist_on else 1
        self.rank = dist.get_rank(group) if self.dist_on else 0
        self.group = group
        self.n = n
        self.lo, self.hi = shard_range(n, self.rank, self.world)
        self.n_tgt = self.hi - self.lo
        assert tuple(pos_shard.shape) == (self.n_tgt, 4) and tuple(vel_shard.shape) == (self.n_tgt, 4)
        self.eps, self.eps2, self.dt = float(eps), float(eps) * float(eps), float(dt)
        self.acc64 = acc64
        if exchange not in ("in_place", "staged", "ring"):
            raise ValueError(f"exchange={exchange!r}: expected 'in_place', 'staged' or 'ring'")
        self.exchange = exchange
        self.ring = exchange == "ring" and self.world > 1
        # two-phase step: sources are cut at shard boundaries, which must be whole 256-body tiles of the kernel
        self.overlap = bool(overlap) and self.world > 1
        if self.overlap and self.n_tgt % 256:
            raise ValueError(f"overlap needs n/world = {self.n_tgt} to be a multiple of 256")
        self._pending = None  # the asynchronous all-gather of pos[cur], if one is in flight
        self.kernel_events = None  # a list: step() appends (start, end) torch.cuda.Event pairs around its launches
        self.trace = trace and torch.cuda.is_available()  # roctx ranges (torch.cuda.nvtx -> roctx on ROCm) for rocprofv3
        self.compute = compute or hip_compute(acc64)
        self.vel = vel_shard.to(device=device, dtype=torch.float32).contiguous()
        # share the unordered pairs of the system among the ranks (K1s + reduce-scatter of partial forces)?  Default: yes
        # where it applies — several ranks, the all-gather form, no two-phase step, the product's HIP compute (an injected
        # `compute`, as the CPU tests use, keeps the ordered form), device tensors, whole superblocks per shard
        # `pair_steps` = (pair_forces, kick_drift) injects the two launches of that step, as `compute` injects the ordered one:
        # the CPU tests pass oracle stand-ins (tests/test_distributed_gloo.py) — test infrastructure, not a product fallback
        self._pair_steps = pair_steps
        can = (self.world > 1 and not self.ring and not self.overlap and
               (pair_steps is not None or
                (torch.device(device).type == "cuda"
                 and (compute is None or (getattr(compute, "is_hip_compute", False) and not compute.forced))
                 and capi.workspace_bytes_shared_pairs_f32(n, self.world, acc64) > 0)))
        # `can` holds device-local facts (this rank's compute-unit count sizes the workspace): the ranks must AGREE, or some
        # would enqueue reduce-scatter + all-gather per step and others only the all-gather — mismatched collectives, a hang
        can = self._all_ranks(can, device)
        if shared_pairs and not can:
            raise ValueError("shared_pairs needs >= 2 ranks, HIP tensors, the all-gather exchange without overlap, "
                             "n % (4096 * world) == 0 and n >= 28672 — on EVERY rank")
        self.shared_pairs = can if shared_pairs is None else bool(shared_pairs)
        self.shared_pairs_note = None
        if self.shared_pairs and shared_pairs is None and dist.get_backend(group) == "nccl":
            # the step will need a second collective, reduce_scatter_tensor: try it once on four numbers per rank, so that a
            # communicator that cannot do it (an argument error raised before anything is enqueued) costs this run the
            # shared pairs — said in `shared_pairs_note`, which bench.py prints — and not the whole measurement.  Asked for
            # explicitly (shared_pairs=True) nothing is tried and nothing is caught.
            try:
                probe = torch.ones((self.world, 4), dtype=torch.float64 if acc64 else torch.float32, device=device)
                got = torch.empty((1, 4), dtype=probe.dtype, device=device)
                dist.reduce_scatter_tensor(got, probe, op=dist.ReduceOp.SUM, group=group)
                if float(got.sum().item()) != 4.0 * self.world:
                    raise RuntimeError(f"reduce_scatter_tensor of ones gave {got.tolist()}")
            except Exception as e:  # noqa: BLE001
                self.shared_pairs = False
                self.shared_pairs_note = f"shared pairs off, every rank runs K1 on its own targets: {type(e).__name__}: {e}"
            # the probe's verdict is per rank (an exception caught here was raised on this rank): one refusal switches every
            # rank to the ordered form
            if not self._all_ranks(self.shared_pairs, device) and self.shared_pairs:
                self.shared_pairs = False
                self.shared_pairs_note = "shared pairs off on every rank: another rank's reduce_scatter_tensor probe failed"
        self._fpart = self._facc = self._pair_ws = None
        if self.ring:
            if overlap:
                raise ValueError("the ring pass overlaps every transfer by construction: overlap=True has no meaning")
            if acc64 and self.n_tgt % 256:
                raise ValueError("ring pass: n/world must be a multiple of 256")
            # own shard (ping-pong) + two travelling blocks: 4*N/P records, never all N
            self.own = [pos_shard.to(device=device, dtype=torch.float32).contiguous().clone() for _ in range(2)]
            self.blk = [torch.empty_like(self.own[0]) for _ in range(2)]
            self.pos = None
            self.pos64 = self.vel64 = None
            if acc64:
                self.pos64 = pos_shard.to(device=device, dtype=torch.float64).contiguous()
                self.vel64 = vel_shard.to(device=device, dtype=torch.float64).contiguous()
            self.cur = 0
            self.overlap = False
            self._pending = None
            self.kernel_events = None
            return
        self.pos = [torch.zeros((n, 4), dtype=torch.float32, device=device) for _ in range(2)]
        self.pos64 = self.vel64 = None
        if acc64:
            self.pos64 = pos_shard.to(device=device, dtype=torch.float64).contiguous()
            self.vel64 = vel_shard.to(device=device, dtype=torch.float64).contiguous()
        self.cur = 0
        self.pos[0][self.lo:self.hi] = pos_shard.to(device=device, dtype=torch.float32)
        self._exchange(self.pos[0])
        self.pos[1].copy_(self.pos[0])  # G*m column of the other buffer for slots this rank never writes
        if self.shared_pairs and self._pair_steps is None:
            self._alloc_shared_pairs(self.pos[0].device)  # not inside somebody's timed first step

    def _all_ranks(self, flag, device):
        """True iff `flag` holds on every rank of the group (all_reduce MIN of 0/1); the flag itself without a group."""
        if not self.dist_on or self.world == 1:
            return bool(flag)
        on_gpu = dist.get_backend(self.group) == "nccl"
        t = torch.tensor([1 if flag else 0], dtype=torch.int32, device=device if on_gpu else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MIN, group=self.group)
        return bool(int(t.item()))

    @property
    def exchange_mode(self):
        """How positions travel between ranks: "none" (one rank), "in_place" (RCCL/NCCL: send buffer = this rank's slot
        of the receive buffer), "staged" (RCCL with the shard cloned first; only when asked for with exchange="staged"),
        "list" (gloo rehearsal / CPU tests)."""
        if not self.dist_on:
            return "none"
        if self.ring:
            return "ring"
        if dist.get_backend(self.group) == "nccl":
            return self.exchange
        return "list"

    def _exchange(self, buf, async_op=False):
        """In-place all-gather: every rank contributes its own slot of `buf`.  Errors of the
        collective propagate: a failed RCCL call must never be retried on the same communicator, and every rank
        must issue the same collective sequence.  async_op -> the Work handle (wait() orders the current stream after
        the collective without blocking the host, for RCCL)."""
        if not self.dist_on:
            return None
        if dist.get_backend(self.group) == "nccl":
            # RCCL: in place, send buffer = this rank's slot of the receive buffer (ncclAllGather's in-place form)
            shard = buf[self.lo:self.hi]
            return dist.all_gather_into_tensor(buf, shard.clone() if self.exchange == "staged" else shard,
                                               group=self.group, async_op=async_op)
        # gloo (CPU tests / single-GPU rehearsal): same exchange through the list form
        per = self.n_tgt
        return dist.all_gather([buf[r * per:(r + 1) * per] for r in range(self.world)],
                               buf[self.lo:self.hi].clone(), group=self.group, async_op=async_op)

    def _wait_gather(self):
        if self._pending is not None:
            self._pending.wait()
            self._pending = None

    def _pass_block(self, send, recv):
        """Start moving `send` to rank r+1 and the next block from rank r-1 into `recv`; -> handles to wait on."""
        nxt, prv = (self.rank + 1) % self.world, (self.rank - 1) % self.world
        if dist.get_backend(self.group) == "nccl":
            return dist.batch_isend_irecv([dist.P2POp(dist.isend, send, nxt, self.group),
                                           dist.P2POp(dist.irecv, recv, prv, self.group)])
        # gloo (rehearsal / CPU tests): point-to-point on host memory
        host_s = send if not send.is_cuda else send.cpu()
        host_r = recv if not recv.is_cuda else torch.empty_like(recv, device="cpu")
        works = [dist.isend(host_s, nxt, self.group), dist.irecv(host_r, prv, self.group)]
        return [_StagedRecv(works, host_r, recv)] if recv.is_cuda else works

    def _step_ring(self):
        """P phases over the travelling blocks; block k (k = 0: own shard) holds the bodies of rank (r - k) mod P."""
        tgt, out, P = self.own[self.cur], self.own[self.cur ^ 1], self.world
        common = (out, self.vel, 0, self.n_tgt, self.eps2, self.dt, self.pos64, self.vel64)
        pending = self._pass_block(tgt, self.blk[0])  # the own shard starts its round trip
        self.compute(tgt, *common, phase=capi.NB_PHASE_FIRST, tgt=tgt)
        for k in range(1, P):
            have = self.blk[(k - 1) & 1]
            for w in pending:
                w.wait()  # block k is here (and the block sent before has left its buffer)
            if k < P - 1:  # hand it on while it is being consumed; the other buffer is free since phase k - 1 computed
                pending = self._pass_block(have, self.blk[k & 1])
            self.compute(have, *common, phase=capi.NB_PHASE_LAST if k == P - 1 else capi.NB_PHASE_MIDDLE, tgt=tgt)
        self.cur ^= 1

    def step(self):
        if self.ring:
            if self.kernel_events is not None:
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
            self._step_ring()
            if self.kernel_events is not None:
                e1.record()
                self.kernel_events.append((e0, e1))
            return
        src, out = self.pos[self.cur], self.pos[self.cur ^ 1]
        if self.shared_pairs:
            self._step_shared_pairs(src, out)
            self.cur ^= 1
            return
        args = (src, out, self.vel, self.lo, self.n_tgt, self.eps2, self.dt, self.pos64, self.vel64)
        if self.trace:
            torch.cuda.nvtx.range_push("nbody.force_kick_drift")
        if self.kernel_events is not None:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
        if self.overlap:
            lo, hi, n = self.lo, self.hi, self.n
            # own shard: written by this rank's previous launch, so its all-gather need not have finished
            self.compute(*args, src_range=(lo, hi), phase=capi.NB_PHASE_FIRST)
            self._wait_gather()  # the other shards of `src` have landed
            if lo > 0:
                self.compute(*args, src_range=(0, lo), phase=capi.NB_PHASE_MIDDLE if hi < n else capi.NB_PHASE_LAST)
            if hi < n:
                self.compute(*args, src_range=(hi, n), phase=capi.NB_PHASE_LAST)
        else:
            self.compute(*args)
        if self.kernel_events is not None:
            e1.record()
            self.kernel_events.append((e0, e1))
        if self.trace:
            torch.cuda.nvtx.range_pop()
            torch.cuda.nvtx.range_push("nbody.allgather_positions")
        if self.overlap:
            self._pending = self._exchange(out, async_op=True)
        else:
            self._exchange(out)
        if self.trace:
            torch.cuda.nvtx.range_pop()
        self.cur ^= 1

    def _alloc_shared_pairs(self, dev):
        """Partial force on all N bodies, the summed force on the own shard, the pair-slot workspace (once)."""
        if self._fpart is None:
            fdt = torch.float64 if self.acc64 else torch.float32
            self._fpart = torch.empty((self.n, 4), dtype=fdt, device=dev)
            self._facc = torch.empty((self.n_tgt, 4), dtype=fdt, device=dev)
            self._pair_ws = torch.empty(capi.workspace_bytes_shared_pairs_f32(self.n, self.world, self.acc64),
                                        dtype=torch.uint8, device=dev)

    def _reduce_scatter_forces(self):
        """Sum of the ranks' partial forces, every rank keeping its own shard of it."""
        if dist.get_backend(self.group) == "nccl":
            dist.reduce_scatter_tensor(self._facc, self._fpart, op=dist.ReduceOp.SUM, group=self.group)
        else:  # gloo (CPU tests, rehearsal on one GPU): the same sum through host memory
            host = self._fpart.cpu()
            dist.all_reduce(host, op=dist.ReduceOp.SUM, group=self.group)
            self._facc.copy_(host[self.lo:self.hi])

    def _step_shared_pairs(self, src, out):
        """K1s on this rank's share of the unordered pairs -> partial force on all N bodies -> reduce-scatter -> kick-drift
        of the own shard -> all-gather of the positions."""
        dev = src.device
        if self._pair_steps is not None:  # injected stand-ins (CPU tests): same sequence, same collectives
            pair_forces, kick_drift = self._pair_steps
            fdt = torch.float64 if self.acc64 else torch.float32
            if self._fpart is None:
                self._fpart = torch.zeros((self.n, 4), dtype=fdt, device=dev)
                self._facc = torch.zeros((self.n_tgt, 4), dtype=fdt, device=dev)
            pair_forces(src, self.lo, self.n_tgt, self.eps2, self._fpart)
            self._reduce_scatter_forces()
            kick_drift(src, out, self.vel, self.lo, self.n_tgt, self.dt, self._facc, self.pos64, self.vel64)
            self._exchange(out)
            return
        self._alloc_shared_pairs(dev)
        stream = torch.cuda.current_stream(dev).cuda_stream
        if self.kernel_events is not None:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
        capi.launch_pair_forces_f32(src.data_ptr(), self.n, self.lo, self.n_tgt, self.eps2, stream, self._fpart.data_ptr(),
                                    self._pair_ws.data_ptr(), self._pair_ws.numel(), acc64=self.acc64)
        if self.kernel_events is not None:
            e1.record()
            self.kernel_events.append((e0, e1))
        self._reduce_scatter_forces()
        capi.launch_kick_drift_f32(src.data_ptr(), out.data_ptr(), self.n, self.lo, self.n_tgt, self.dt, stream,
                                   self._facc.data_ptr(), parts=1, vel_ptr=self.vel.data_ptr(),
                                   pos64_ptr=self.pos64.data_ptr() if self.pos64 is not None else 0,
                                   vel64_ptr=self.vel64.data_ptr() if self.vel64 is not None else 0, acc64=self.acc64)
        self._exchange(out)

    @property
    def positions(self):
        """All N positions on this rank.  The ring pass never holds them: there they are gathered on demand (inspection
        and checks only, not part of a step) — a COLLECTIVE in that mode: every rank must read the property."""
        if self.ring:
            full = torch.empty((self.n, 4), dtype=torch.float32, device=self.own[0].device)
            mine = self.own[self.cur]
            if dist.get_backend(self.group) == "nccl":
                dist.all_gather_into_tensor(full, mine, group=self.group)
            else:
                dist.all_gather([full[r * self.n_tgt:(r + 1) * self.n_tgt] for r in range(self.world)], mine.clone(),
                                group=self.group)
            return full
        self._wait_gather()
        return self.pos[self.cur]

    # ---- checkpoints of a sharded run (the 1000-step N = 2^24 run takes hours) ----
    def _gather_rows(self, shard):
        """(n_tgt, k) shard of every rank -> (N, k) on every rank (a collective)."""
        if not self.dist_on or self.world == 1:
            return shard.clone()
        full = torch.empty((self.n,) + tuple(shard.shape[1:]), dtype=shard.dtype, device=shard.device)
        if dist.get_backend(self.group) == "nccl":
            dist.all_gather_into_tensor(full, shard.contiguous(), group=self.group)
        else:
            dist.all_gather([full[r * self.n_tgt:(r + 1) * self.n_tgt] for r in range(self.world)], shard.contiguous(),
                            group=self.group)
        return full

    def save_checkpoint(self, path, step, G):
        """One NBODYST2 state file (include/nbody_amd.h) for the whole system, written by rank 0: q, v as the fp64 masters
        (NB_F32_ACC64) or the widened fp32 state (NB_F32), m = (G*m)/G from the fp32 records.  A COLLECTIVE: every rank
        calls it (velocities, and in the fp64-master and ring modes positions, live only on their owners)."""
        self._wait_gather()
        if self.acc64:
            pos = self._gather_rows(self.pos64)
            vel = self._gather_rows(self.vel64)
        else:
            pos = self.positions if self.ring or self.pos is None else self.pos[self.cur]
            vel = self._gather_rows(self.vel)
        if self.rank == 0:
            p = pos.detach().cpu().numpy().astype("float64")
            v = vel.detach().cpu().numpy().astype("float64")
            capi.write_state_file(path, p[:, :3].T.copy(), v[:, :3].T.copy(), p[:, 3] / G,
                                  precision=capi.NB_F32_ACC64 if self.acc64 else capi.NB_F32, step=step, G=G,
                                  eps=self.eps, dt=self.dt)
        if self.dist_on:
            dist.barrier(group=self.group)  # the file is complete when any rank returns

    @staticmethod
    def load_checkpoint_shard(path, rank, world):
        """-> (header, pos_shard (n_tgt,4) float64 {x,y,z,G*m}, vel_shard (n_tgt,4) float64) of this rank, ready for the
        constructor; every rank reads the file itself (no collective)."""
        import numpy as np
        hdr, q, v, m, _ = capi.read_state_file(path)
        lo, hi = shard_range(hdr["n"], rank, world)
        pos = np.ascontiguousarray(np.concatenate([q[:, lo:hi].T, (hdr["G"] * m[lo:hi])[:, None]], axis=1))
        vel = np.ascontiguousarray(np.concatenate([v[:, lo:hi].T, np.zeros((hi - lo, 1))], axis=1))
        return hdr, pos, vel

    def pairs_per_step(self):
        """Interactions the whole job evaluates per step, counted as the reference does (nbody.cc:57-60): N(N-1)."""
        return self.n * (self.n - 1)
