"""PPOLearner -- drop-in for rlgym_ppo/ppo/ppo_learner.py:10-271 with the update running in librlppo.so.

Same constructor, attributes (`policy`, `value_net`, `policy_optimizer`, `value_optimizer`,
`cumulative_model_updates`), `learn(exp) -> 8-key report`, `save_to` / `load_from` (same four .pt files with stock
state_dict layouts).  What changed underneath:

  * one call to rlppo_ppo_minibatch per minibatch replaces ~60 eager ATen launches, 5 H2D copies and 5 .item()
    syncs (ppo_learner.py:139-185); the report statistics accumulate in device doubles and are read back once;
  * the epoch permutation is drawn on the host (bit-identical numpy legacy stream) while the GPU is still busy
    with the previous epoch, and only the index vector crosses PCIe;
  * clip_grad_norm_ + Adam.step() are one fused pass over a flat parameter arena (rlppo_clip_adam);
  * the tail of learn() (ppo_learner.py:213-236: update magnitudes, report means) is ONE launch that writes pinned memory and
    releases a completion word (rlppo_learn_report); nothing is enqueued in front of the first pass that the pass does not need;
  * data-parallel: the minibatch slices of a batch are dealt round-robin to the ranks of the default
    torch.distributed group and ONE RCCL all-reduce of the flat gradient arena precedes clipping -- algebraically
    the reference's own gradient accumulation (ppo_learner.py:134-193) with slice j living on rank j % world.
"""
import ctypes
import os
import time

import numpy as np
import torch

from .. import _native as N
from ..dp import all_reduce_sum, dist_info, fuse_runs, slices_for_rank
from ..engine import Workspace, ptr, require_gpu, stream_ptr
from .continuous_policy import ContinuousPolicy
from .discrete_policy import DiscreteFF
from .multi_discrete_policy import MultiDiscreteFF
from .value_estimator import ValueEstimator

MAX_GRAD_NORM = 0.5  # ppo_learner.py:187-190


class FusedAdam(torch.optim.Adam):
    """torch.optim.Adam whose step() is librlppo's fused clip+Adam kernel on the module's flat arena.  State lives
    in two flat tensors; `state[p]` exposes views of them in the stock layout {step, exp_avg, exp_avg_sq}, so
    state_dict()/load_state_dict() and checkpoints interchange with the reference's optimisers."""

    def __init__(self, module, lr):
        self.arena = module.arena
        super().__init__(list(module.arena.params()), lr=lr)
        a = self.arena
        self.exp_avg = torch.zeros(a.n_flat, dtype=torch.float32, device=a.device)
        self.exp_avg_sq = torch.zeros(a.n_flat, dtype=torch.float32, device=a.device)
        self.gnorm2 = torch.zeros(1, dtype=torch.float64, device=a.device)
        self.step_count = 0

    def _expose_state(self):
        o = 0
        for p in self.param_groups[0]["params"]:
            n = p.numel()
            self.state[p] = {"step": torch.tensor(float(self.step_count)),
                             "exp_avg": self.exp_avg[o:o + n].view(p.shape),
                             "exp_avg_sq": self.exp_avg_sq[o:o + n].view(p.shape)}
            o += n

    def zero_grad(self, set_to_none=True):
        self.arena.grad.zero_()  # keep the .grad views alive: the kernels accumulate into the arena

    @torch.no_grad()
    def step(self, closure=None, max_norm=None):
        g = self.param_groups[0]
        if not self.arena.is_bound():
            self.arena.bind()
        self.step_count += 1
        b1, b2 = g["betas"]
        N.check(N.lib().rlppo_clip_adam(stream_ptr(), ptr(self.arena.flat), ptr(self.arena.grad), ptr(self.exp_avg),
                                        ptr(self.exp_avg_sq), self.arena.n_flat,
                                        float("inf") if max_norm is None else float(max_norm), float(g["lr"]), float(b1),
                                        float(b2), float(g["eps"]), self.step_count, ptr(self.gnorm2)))
        self.arena.native_epoch += 1

    def fused_descriptor(self, max_norm):
        """rlppo_opt_net of THIS update (advances the step count): one half of rlppo_clip_adam_pack2."""
        g = self.param_groups[0]
        a = self.arena
        if not a.is_bound():
            a.bind()
        self.step_count += 1
        b1, b2 = g["betas"]
        d = N.OptNet()
        d.dims, d.n_layers = ctypes.cast(a.dims_c, ctypes.POINTER(ctypes.c_int32)), a.n_layers
        d.params, d.grads = a.flat.data_ptr(), a.grad.data_ptr()
        d.exp_avg, d.exp_avg_sq = self.exp_avg.data_ptr(), self.exp_avg_sq.data_ptr()
        d.packed, d.gnorm2 = a.packed.data_ptr(), self.gnorm2.data_ptr()
        d.max_norm = float("inf") if max_norm is None else float(max_norm)
        d.lr, d.beta1, d.beta2, d.eps, d.step = float(g["lr"]), float(b1), float(b2), float(g["eps"]), self.step_count
        return d

    def state_dict(self):
        if self.step_count > 0:
            self._expose_state()
        return super().state_dict()

    def load_state_dict(self, state_dict):
        super().load_state_dict(state_dict)
        o, step = 0, 0
        for p in self.param_groups[0]["params"]:
            n = p.numel()
            st = self.state.get(p)
            if st:
                self.exp_avg[o:o + n].copy_(st["exp_avg"].reshape(-1))
                self.exp_avg_sq[o:o + n].copy_(st["exp_avg_sq"].reshape(-1))
                step = int(float(st["step"]))
            o += n
        self.step_count = step
        if step > 0:
            self._expose_state()


class OptimizerBarrierTimeout(RuntimeError):
    """The one-launch optimiser tail's grid barrier gave up; see PPOLearner.learn (the learner has already recovered)."""


class PPOLearner(object):
    def __init__(self, obs_space_size, act_space_size, policy_type, policy_layer_sizes, critic_layer_sizes,
                 continuous_var_range, batch_size, n_epochs, policy_lr, critic_lr, clip_range, ent_coef, mini_batch_size,
                 device):
        self.device = device
        self._dev = require_gpu(device)
        N.lib()  # fail here, loudly, if the HIP library is missing

        assert batch_size % mini_batch_size == 0, "MINIBATCH SIZE MUST BE AN INTEGER MULTIPLE OF BATCH SIZE"

        obs_space_size = int(obs_space_size)
        self.policy_type = int(policy_type)
        # construction order (policy, then critic) fixes the CPU-generator consumption and therefore the initial
        # weights (ppo_learner.py:34-53)
        if policy_type == 2:
            self.policy = ContinuousPolicy(obs_space_size, act_space_size * 2, policy_layer_sizes, device,
                                           var_min=continuous_var_range[0], var_max=continuous_var_range[1]).to(device)
            self._act_dim = int(act_space_size)
        elif policy_type == 1:
            self.policy = MultiDiscreteFF(obs_space_size, policy_layer_sizes, device).to(device)
            self._act_dim = 8
        else:
            self.policy = DiscreteFF(obs_space_size, act_space_size, policy_layer_sizes, device).to(device)
            self._act_dim = 1
        self.value_net = ValueEstimator(obs_space_size, critic_layer_sizes, device).to(device)
        self.mini_batch_size = mini_batch_size

        self.policy_optimizer = FusedAdam(self.policy, lr=policy_lr)
        self.value_optimizer = FusedAdam(self.value_net, lr=critic_lr)
        # both squared-norm accumulators side by side: rlppo_clip_adam_pack2 then clears them with one fill
        self._gnorm2 = torch.zeros(2, dtype=torch.float64, device=self._dev)
        self.policy_optimizer.gnorm2, self.value_optimizer.gnorm2 = self._gnorm2[0:1], self._gnorm2[1:2]
        # grid-barrier state of the one-launch optimiser tail (include/rlppo.h, RLPPO_OPT_SYNC_BYTES): zeroed once, here
        self._opt_sync = torch.zeros(N.OPT_SYNC_BYTES // 4, dtype=torch.int32, device=self._dev)

        n_pol = sum(p.numel() for p in self.policy.parameters() if p.requires_grad)
        n_val = sum(p.numel() for p in self.value_net.parameters() if p.requires_grad)
        print("Trainable Parameters:")
        print(f"{'Component':<10} {'Count':<10}")
        print("-" * 20)
        print(f"{'Policy':<10} {n_pol:<10}")
        print(f"{'Critic':<10} {n_val:<10}")
        print("-" * 20)
        print(f"{'Total':<10} {n_pol + n_val:<10}")
        print(f"Current Policy Learning Rate: {policy_lr}")
        print(f"Current Critic Learning Rate: {critic_lr}")

        self.n_epochs = n_epochs
        self.batch_size = batch_size
        self.clip_range = clip_range
        self.ent_coef = ent_coef
        self.cumulative_model_updates = 0

        # one contiguous gradient buffer [policy | critic] so that data-parallel needs a single all-reduce
        pa, va = self.policy.arena, self.value_net.arena
        self._grad_all = torch.zeros(pa.n_flat + va.n_flat, dtype=torch.float32, device=self._dev)
        pa.grad = self._grad_all[:pa.n_flat]
        va.grad = self._grad_all[pa.n_flat:]
        pa.bind()
        va.bind()
        self._stats = torch.zeros(N.N_STATS, dtype=torch.float64, device=self._dev)
        self._stats_clean = True   # rlppo_learn_report leaves the accumulators zeroed for the next learn()
        # [r6] the tail of learn() as one launch (rlppo_learn_report): the parameters before the first optimiser step (copied behind the
        # first pass's launches, where the copy is free), the kernel's ticket block, its pinned output and completion word
        self._before = (torch.empty_like(pa.flat), torch.empty_like(va.flat))
        self._report_ws = torch.zeros(N.REPORT_WS_BYTES, dtype=torch.uint8, device=self._dev)
        self._report_out = torch.zeros(N.REPORT_OUT_DOUBLES, dtype=torch.float64).pin_memory()
        self._report_done = torch.zeros(1, dtype=torch.int32).pin_memory()
        self._report_out_np, self._report_seq = self._report_out.numpy(), 0
        self._ws = Workspace(self._dev)
        self.n_slots = int(os.environ.get("RLPPO_SLOTS", 1))  # minibatches of a batch kept in flight concurrently
        # Minibatch fusion.  The reference sums the gradients of a batch's minibatches, each the MB/B-scaled mean over its
        # rows, before ONE clip + Adam (ppo_learner.py:134-193): the minibatches are independent given the parameters and
        # their sum is the 1/B-scaled sum over all their rows.  They only exist to bound activation memory; with 288 GB of
        # HBM a rank evaluates up to `max_fused_minibatches` CONSECUTIVE minibatches in one pass of rlppo_ppo_minibatch
        # (mb = k MB rows, mb_ratio = k MB / B): the same gradient and the same report means up to fp32 summation order,
        # in launches that are k times larger.  RLPPO_FUSE=1 keeps one pass per minibatch.
        self.max_fused_minibatches = max(1, int(os.environ.get("RLPPO_FUSE", 8)))
        self.fused_optimizer_step = os.environ.get("RLPPO_FUSED_OPT", "1") != "0"  # rlppo_clip_adam_pack2 (False: FusedAdam.step x2)
        self.grad_probe = None  # callable(flat [grad_policy | grad_value]) before every optimiser step (tests)
        self.one_launch_optimizer = os.environ.get("RLPPO_OPT_ONE_LAUNCH", "1") != "0"  # ... as ONE launch with a grid barrier
        # [r5] update precision of THIS learner (rlppo_minibatch_args.precision): None = the process default chosen with
        # engine.set_update_precision, else "fp32" / "bf16" / "x3" -- two learners of one process may differ
        self.update_precision = None

    # --------------------------------------------------------------------------------------------- learn
    def _minibatch_args(self, exp, rank=0, world=1):
        pa, va = self.policy.arena, self.value_net.arena
        a = N.MinibatchArgs()
        a.head = self.policy_type
        a.pol_layers, a.val_layers = pa.n_layers, va.n_layers
        a.act_dim = self._act_dim
        a.pol_dims = ctypes.cast(pa.dims_c, ctypes.POINTER(ctypes.c_int32))
        a.val_dims = ctypes.cast(va.dims_c, ctypes.POINTER(ctypes.c_int32))
        a.pol_packed, a.val_packed = pa.packed.data_ptr(), va.packed.data_ptr()
        if self.update_precision is None:
            prec, a.precision = int(N.lib().rlppo_get_update_precision()), N.PRECISION_DEFAULT
        else:
            prec = {"fp32": 0, "bf16": 1, "x3": 2}[self.update_precision]
            a.precision = 1 + prec
        self._bf16 = prec == 1  # bf16-operand forward: the rounded weight images travel too
        self._x3 = prec == 2    # [r4] fp32 update with split-bf16 hidden forward / dX products: the three-plane images travel
        if self._x3:
            a.pol_wb16, a.val_wb16 = pa.ensure_packed_x3().data_ptr(), va.ensure_packed_x3().data_ptr()
        if self._bf16:
            (pr, pw), (vr, vw) = pa.ensure_packed_bf16(), va.ensure_packed_bf16()
            a.pol_packed_r, a.val_packed_r, a.pol_wb16, a.val_wb16 = pr.data_ptr(), vr.data_ptr(), pw.data_ptr(), vw.data_ptr()
        a.pol_grad, a.val_grad = pa.grad.data_ptr(), va.grad.data_ptr()
        st, a.ring_base, a.ring_cap = exp.ring()  # physical rows; the kernels map the permutation's logical rows onto them
        a.states, a.ld_states, a.n_rows = st["states"].data_ptr(), st["states"].shape[1], st["states"].shape[0]
        a.actions = st["actions"].data_ptr()
        a.old_logp = st["log_probs"].data_ptr()
        a.targets = st["values"].data_ptr()
        a.advantages = st["advantages"].data_ptr()
        a.clip_range, a.ent_coef = float(self.clip_range), float(self.ent_coef)
        a.mb_ratio = float(self.mini_batch_size / self.batch_size)
        if self.policy_type == 2:
            a.var_m, a.var_b = float(self.policy.affine_map.m), float(self.policy.affine_map.b)
        a.stats = self._stats.data_ptr()
        # one activation workspace per slot: minibatches in different slots overlap on the GPU (rlppo_ppo_join)
        # rows of the largest pass THIS rank launches: its share of a batch's slices, fused (8 ranks x 8 slices: one 65,536-row pass)
        n_slices = self.batch_size // self.mini_batch_size
        runs = fuse_runs(slices_for_rank(max(1, n_slices), rank, world), self.max_fused_minibatches)
        self._fused_rows = self.mini_batch_size * max([cnt for _, cnt in runs] or [1])
        nbytes = int(N.lib().rlppo_minibatch_workspace_bytes_for(pa.dims_c, pa.n_layers, va.dims_c, va.n_layers, self._fused_rows, a.precision))
        nbytes = (nbytes + 255) // 256 * 256
        ws = self._ws.get(nbytes * self.n_slots)
        self._slot_ws = [ws.data_ptr() + i * nbytes for i in range(self.n_slots)]
        a.workspace, a.ws_bytes = self._slot_ws[0], nbytes
        return a

    def learn(self, exp):
        """Compute PPO updates with an experience buffer; returns the reference's report dictionary
        (ppo_learner.py:225-234)."""
        dist, rank, world = dist_info()
        steps = self.learn_steps(exp, rank, world)
        try:
            buf = next(steps)
            while True:  # one exchange per optimiser step + one for the report statistics (dp.py)
                all_reduce_sum(buf, dist)
                buf = steps.send(None)
        except StopIteration as done:
            return done.value

    def learn_steps(self, exp, rank=0, world=1):
        """learn() as a generator that stops at every data-parallel exchange point: it yields the flat tensor whose sum over the
        ranks it needs (the [grad_policy | grad_value] arena after a batch's last backward pass, the report statistics at the end)
        and continues once the caller has put that sum into it; the report dictionary is the generator's return value.  learn()
        drives it with torch.distributed's all-reduce; dp.run_virtual_ranks drives the generators of N replicas in ONE process
        (the 8-way partition of BASELINE configs[3] on a one-GPU box).  With world == 1 it never yields."""
        L = N.lib()
        pa, va = self.policy.arena, self.value_net.arena
        B, MB = self.batch_size, self.mini_batch_size
        n_slices = B // MB

        # [r6] nothing is enqueued before the first pass that the pass does not need: the report sums were zeroed by the previous
        # learn()'s report kernel, and the "before" copies of the parameters (ppo_learner.py:111-116) follow the first pass's launches
        # in stream order -- the first optimiser step is still behind them
        if not self._stats_clean:
            self._stats.zero_()
        self._stats_clean = False
        have_before = False
        n_iterations = 0
        n_minibatch_iterations = 0
        n_passes = 0  # launches of rlppo_ppo_minibatch on this rank (each adds one mean to every report statistic)
        total = len(exp)
        n_batches = total // B if B > 0 else 0

        t1 = time.time()
        grads_zero = False
        if n_batches > 0 and total > 0:
            if exp.ring()[0]["actions"][:1].reshape(1, -1).shape[1] != self._act_dim:
                raise ValueError("experience buffer action width does not match the policy head")
            args = self._minibatch_args(exp, rank, world)
            st = stream_ptr()
            # The legacy-MT19937 permutation is inherently serial host work.  The buffer's shuffle pipeline draws it on
            # helper threads several epochs ahead (also across learn() calls) and uploads every index vector on its own
            # stream (engine.LegacyPermutation / DeviceIndexRing): here an epoch only orders the stream after that copy.
            # With 8 ranks the GPU share of an epoch is ~1.1 ms; the serial stream phase (~0.8 ms per 512k indices) is
            # the only part of the shuffle that cannot be spread over threads.
            for epoch in range(self.n_epochs):
                idx_dev = exp.epoch_indices_device(refill=False)   # (the look-ahead is topped up behind the epoch's first launches)
                refilled = False
                for b in range(n_batches):
                    if not grads_zero:
                        self._grad_all.zero_()
                    grads_zero = False
                    pa.ensure_packed()
                    va.ensure_packed()
                    if self._bf16:  # re-round the master weights the optimiser step just moved (one small launch per net)
                        pa.ensure_packed_bf16()
                        va.ensure_packed_bf16()
                    if self._x3:    # re-split them (two small launches per covered layer)
                        pa.ensure_packed_x3()
                        va.ensure_packed_x3()
                    for k, (j, cnt) in enumerate(fuse_runs(slices_for_rank(n_slices, rank, world), self.max_fused_minibatches)):
                        args.slot = k % self.n_slots
                        args.workspace = self._slot_ws[args.slot]
                        off = b * B + j * MB
                        args.idx = idx_dev.data_ptr() + 8 * off
                        args.mb = cnt * MB                      # cnt consecutive minibatches in one pass
                        args.mb_ratio = float(cnt * MB / B)
                        N.check(L.rlppo_ppo_minibatch(st, ctypes.byref(args)))
                        n_passes += 1
                    N.check(L.rlppo_ppo_join(st))
                    if not refilled:
                        exp.refill_shuffle()
                        refilled = True
                    if not have_before:
                        self._before[0].copy_(pa.flat)
                        self._before[1].copy_(va.flat)
                        have_before = True
                    n_minibatch_iterations += n_slices
                    if world > 1:
                        yield self._grad_all  # summed over the ranks (RCCL over xGMI) before clipping
                    if self.grad_probe is not None:  # test hook: the batch gradient clip_grad_norm_ / Adam are about to see
                        self.grad_probe(self._grad_all)
                    if self.fused_optimizer_step:
                        # both clip + Adam steps, the re-pack of both weight copies and the next batch's zero_grad: 3 stream
                        # operations instead of 9 (csrc/optim.hip)
                        dv = self.value_optimizer.fused_descriptor(MAX_GRAD_NORM)
                        dp_ = self.policy_optimizer.fused_descriptor(MAX_GRAD_NORM)
                        N.check(L.rlppo_clip_adam_pack2(st, ctypes.byref(dv), ctypes.byref(dp_),
                                                        ptr(self._opt_sync) if self.one_launch_optimizer else None))
                        va.mark_repacked()
                        pa.mark_repacked()
                        grads_zero = True
                    else:
                        self.value_optimizer.step(max_norm=MAX_GRAD_NORM)
                        self.policy_optimizer.step(max_norm=MAX_GRAD_NORM)
                    n_iterations += 1
                if not refilled:
                    exp.refill_shuffle()
        else:
            for _ in range(self.n_epochs):
                exp.epoch_indices()  # the reference consumes one permutation per epoch even if no batch fits

        # every pass added one mean to each report statistic; the number of passes travels with the sums, so the report is the
        # mean over the passes of ALL ranks even when the slices do not divide evenly over them (3 slices on 2 ranks)
        if not have_before:   # (no batch fitted: nothing moved)
            self._before[0].copy_(pa.flat)
            self._before[1].copy_(va.flat)
        rep = N.ReportArgs()
        rep.pol_before, rep.pol_now, rep.n_pol = self._before[0].data_ptr(), pa.flat.data_ptr(), pa.n_flat
        rep.val_before, rep.val_now, rep.n_val = self._before[1].data_ptr(), va.flat.data_ptr(), va.n_flat
        rep.stats, rep.add_passes = self._stats.data_ptr(), float(n_passes)
        rep.timeout_word = self._opt_sync.data_ptr() + 4 * N.OPT_SYNC_TIMEOUT_WORD   # optimiser steps THIS rank's barrier skipped
        to_all = None
        if world > 1:
            # the give-up count travels with the statistics: a give-up on ANY rank is known to EVERY rank after this exchange
            self._stats[N.STAT_PASSES] += float(n_passes)
            rep.add_passes = 0.0
            ex = torch.cat((self._stats, self._opt_sync[N.OPT_SYNC_TIMEOUT_WORD:N.OPT_SYNC_TIMEOUT_WORD + 1].double()))
            yield ex
            self._stats.copy_(ex[:N.N_STATS])
            to_all = ex[N.N_STATS:].contiguous()
            rep.extra = to_all.data_ptr()
        # [r6] update magnitudes (ppo_learner.py:214-222), the statistics and the give-up words in ONE launch that writes pinned memory
        # and releases a completion word (rlppo_learn_report): one device->host hand-over per learn(), no eager tail
        self._report_seq = self._report_seq % 0x7FFFFFFF + 1
        rep.out, rep.done_word, rep.done_value, rep.ws = (self._report_out.data_ptr(), self._report_done.data_ptr(), self._report_seq,
                                                          self._report_ws.data_ptr())
        N.check(L.rlppo_learn_report(stream_ptr(), ctypes.byref(rep)))
        if L.rlppo_host_wait_words(self._report_done.data_ptr(), 1, self._report_seq, 5000) != 0:
            torch.cuda.current_stream(self._dev).synchronize()   # (a long learn(): sleep instead of spinning)
            if L.rlppo_host_wait_words(self._report_done.data_ptr(), 1, self._report_seq, 1000000) != 0:
                raise RuntimeError("rlppo_learn_report: the completion word did not arrive")
        self._stats_clean = True
        stats = self._report_out_np.copy()
        if stats[N.N_STATS + 3] != 0:
            # A grid-barrier wait of the one-launch optimiser step gave up on some rank (GPU shared with a kernel that never yields,
            # a partitioned device, or a defect).  Giving up is all or nothing (include/rlppo.h): on the rank it happened, that step
            # and every later one of this call were SKIPPED -- its parameters and Adam moments are those of its last completed step,
            # finite -- and its gradient arena still holds the skipped batches' sums.  Recover the state a caller can continue from
            # (zero gradients, a re-armed block, the three-operation form from now on) and report loudly.
            n_to = int(stats[N.N_STATS + 2])   # skipped optimiser steps of THIS rank (a launch counts itself once: csrc/optim.hip)
            for opt in (self.policy_optimizer, self.value_optimizer):
                opt.step_count = max(0, opt.step_count - n_to)   # Adam's bias corrections must not count steps that never happened
            self.cumulative_model_updates += max(0, n_iterations - n_to)
            self._grad_all.zero_()
            self._opt_sync.zero_()
            self.one_launch_optimizer = False
            where = "the affected optimiser steps of this learn() were skipped -- parameters and Adam state are those of the last completed step, no NaN was written"
            if world > 1:
                # [r5, advisor] Data-parallel: the rank that gave up skipped steps the others applied, so the replicas have DIVERGED
                # (parameters, moments, step counts).  The outcome is made collective: every rank takes this branch (the count was
                # summed over the ranks above) and adopts rank 0's state -- one more exchange in which only rank 0 contributes, so
                # the sum IS its state, bit for bit -- before raising on every rank.
                opts = (self.policy_optimizer, self.value_optimizer)
                pieces = [pa.flat, va.flat] + [t for o in opts for t in (o.exp_avg, o.exp_avg_sq)]
                sync = torch.cat([t.reshape(-1) for t in pieces])
                counts = torch.tensor([float(o.step_count) for o in opts] + [float(self.cumulative_model_updates)], dtype=torch.float64,
                                      device=self._dev)
                if rank != 0:
                    sync.zero_()
                    counts.zero_()
                yield sync
                yield counts
                off = 0
                for t in pieces:
                    t.copy_(sync[off:off + t.numel()].view_as(t))
                    off += t.numel()
                c = counts.cpu().numpy()
                self.policy_optimizer.step_count, self.value_optimizer.step_count = int(c[0]), int(c[1])
                self.cumulative_model_updates = int(c[2])
                pa.native_epoch += 1  # `flat` was rewritten behind torch's back: re-pack before the next kernel reads the weights
                va.native_epoch += 1
                pa.invalidate()
                va.invalidate()
                where = ("the rank(s) it happened on skipped optimiser steps the others applied; every rank has now adopted rank 0's "
                         "parameters, Adam moments and step counts (bit-identical replicas again), no NaN was written")
            raise OptimizerBarrierTimeout(
                "rlppo_clip_adam_pack2: the optimiser's grid barrier gave up (%d skipped step(s) over all ranks); %s -- gradients are "
                "zeroed and this learner now uses the three-operation optimiser tail: calling learn() again is safe"
                % (int(stats[N.N_STATS + 3]), where))
        elapsed = time.time() - t1
        n_iter_r = max(n_iterations, 1)
        n_mb_r = max(float(stats[N.STAT_PASSES]), 1.0)
        policy_update_magnitude, critic_update_magnitude = float(stats[N.N_STATS]), float(stats[N.N_STATS + 1])
        self.cumulative_model_updates += n_iter_r

        report = {
            "PPO Batch Consumption Time": elapsed / n_iter_r,
            "Cumulative Model Updates": self.cumulative_model_updates,
            "Policy Entropy": float(stats[N.STAT_ENTROPY]) / n_mb_r,
            "Mean KL Divergence": float(stats[N.STAT_KL]) / n_mb_r,
            "Value Function Loss": float(stats[N.STAT_VLOSS]) / n_mb_r,
            "SB3 Clip Fraction": float(stats[N.STAT_CLIPFRAC]) / n_mb_r if n_minibatch_iterations else 0,
            "Policy Update Magnitude": policy_update_magnitude,
            "Value Function Update Magnitude": critic_update_magnitude,
        }
        if not grads_zero:  # the fused optimiser step leaves the arena zeroed
            self._grad_all.zero_()
        return report

    # ---------------------------------------------------------------------------------------- checkpoints
    def save_to(self, folder_path):
        os.makedirs(folder_path, exist_ok=True)
        torch.save(self.policy.state_dict(), os.path.join(folder_path, "PPO_POLICY.pt"))
        torch.save(self.value_net.state_dict(), os.path.join(folder_path, "PPO_VALUE_NET.pt"))
        torch.save(self.policy_optimizer.state_dict(), os.path.join(folder_path, "PPO_POLICY_OPTIMIZER.pt"))
        torch.save(self.value_optimizer.state_dict(), os.path.join(folder_path, "PPO_VALUE_NET_OPTIMIZER.pt"))

    def load_from(self, folder_path):
        assert os.path.exists(folder_path), "PPO LEARNER CANNOT FIND FOLDER {}".format(folder_path)
        self.policy.load_state_dict(torch.load(os.path.join(folder_path, "PPO_POLICY.pt")))
        self.value_net.load_state_dict(torch.load(os.path.join(folder_path, "PPO_VALUE_NET.pt")))
        self.policy_optimizer.load_state_dict(torch.load(os.path.join(folder_path, "PPO_POLICY_OPTIMIZER.pt")))
        self.value_optimizer.load_state_dict(torch.load(os.path.join(folder_path, "PPO_VALUE_NET_OPTIMIZER.pt")))
