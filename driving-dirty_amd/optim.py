"""``HipAdam``: torch.optim.Adam semantics (reference autoencoder.py:119-120, roadmap_bce_v2.py:154-157)
executed by ``dd_adam_step``, one fused read-modify-write pass per parameter tensor.

``grad_scale`` folds the 1/world_size of data-parallel gradient averaging into the same pass, so the
all-reduced SUM never needs a separate divide kernel.

Rank-B mode (``fuse_linear_wgrad(module)``; round 5): the weight gradient of a big ``nn.Linear`` is never written.  ``ops.Linear.backward``
hands its factors -- the layer's input X [rows, in] and output gradient dY [rows, out] -- to ``linear_factors`` instead of launching
``dd_linear_wgrad``, ``weight.grad`` stays None, and ``dd_adam_step_rankb`` forms every gradient element dW[o][i] = sum_b dY[b][o] X[b][i]
in MFMA accumulators inside the optimizer pass (bias: the column sum of dY, updated by the same launch): six passes over the tensor
instead of eight (reference call sites: components.py:105, roadmap_bce_v2.py:75,154-157).  Used when the gradient does not have to
travel as a tensor: one GPU, or ``GradSync`` factor mode (the gathered factors feed the same kernel).  With an all-reduce or a sharded
``GradSync`` the layer keeps its materialised gradient.

Shard mode (``attach(grad_sync)`` / ``overlap_with_backward(grad_sync=...)`` with ``GradSync(shard_optimizer=True)``): a tensor
that travels as reduce-scatter + all-gather is updated only in the slices this rank owns (``grad_sync.shards(p)``), its moments
exist only for those slices (``state[p]["shards"][piece] = (exp_avg, exp_avg_sq)``), and each updated slice is handed straight
back to ``grad_sync.gather_shard`` -- piece k's all-gather is on the links while piece k+1's update runs.  Elementwise, so the
owned elements get bit for bit the update of the whole-tensor pass.
"""
import os

import torch

from . import lightning, ops


class HipAdam(torch.optim.Optimizer):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8):
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps))
        self._side = None
        self._pending = []
        self._early = set()
        self._hooks = []
        self._hooked = set()
        self._big_numel = 1 << 20
        self._scale = 1.0
        self._sync = None
        self._factored = []        # parameters whose factors are on the links (ddp.GradSync, factor mode)
        self._fgrad = {}           # p -> persistent buffer of the global-batch gradient formed from gathered factors
        self._rankb = {}           # weight -> bias (or None): Linear layers whose gradient is formed inside the Adam pass
        self._rankb_keys = {}      # data_ptr -> weight
        self._rankb_now = {}       # weight -> (x, dy) of this step's backward, until the pass has been launched
        self._held = []            # factors read by launches on the side stream: kept until step() has joined it
        self._fac_now = {}         # weight -> gathered Factors that have arrived (factor mode of ddp.GradSync + rank-B)
        self._last = False         # passes_last(): c2's data gradient first, the queued passes beside its weight gradient

    SMALL_NUMEL = 1 << 16      # tensors up to this size go into one multi-tensor launch (0: never; tests set it per instance)

    def _state_of(self, p):
        st = self.state[p]
        if not st:
            st["step"] = 0
            st["exp_avg"] = torch.zeros_like(p, memory_format=torch.contiguous_format)
            st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.contiguous_format)
        return st

    def _launch(self, p, g, m, v, group, step, grad_scale):
        """The one place the elementwise kernel is called (flat fp32 device tensors of equal length)."""
        b1, b2 = group["betas"]
        ops.adam_step_flat(p, g, m, v, group["lr"], b1, b2, group["eps"], step, grad_scale)

    # ---- rank-B mode -----------------------------------------------------------------------------------------------------------
    def fuse_linear_wgrad(self, module, min_numel=None):
        """Register every ``nn.Linear`` of ``module`` whose weight has >= ``min_numel`` elements (default: the big-tensor threshold)
        and belongs to this optimizer: from now on its weight gradient is formed inside its Adam pass (``dd_adam_step_rankb``) and
        ``weight.grad`` / ``bias.grad`` stay None after a backward.  Returns the registered weights."""
        owned = {id(p) for group in self.param_groups for p in group["params"]}
        limit = self._big_numel if min_numel is None else min_numel
        done = []
        for mod in module.modules():
            w = getattr(mod, "weight", None)
            if not isinstance(mod, torch.nn.Linear) or w is None or id(w) not in owned or w.numel() < limit:
                continue
            if not (w.is_cuda and w.dtype == torch.float32 and w.data.is_contiguous() and w.shape[1] % 4 == 0 and w.shape[0] % 4 == 0):
                continue
            b = mod.bias if (mod.bias is not None and id(mod.bias) in owned and mod.bias.dtype == torch.float32) else None
            self._rankb[w] = b
            self._rankb_keys[w.data_ptr()] = w
            ops.RANKB[w.data_ptr()] = self
            done.append(w)
        return done

    def _sync_blocks_rankb(self):
        """A gradient that has to travel as a tensor (all-reduce or reduce-scatter) must exist as one."""
        s = self._sync
        return s is not None and (getattr(s, "active", False) or getattr(s, "shard", False)) and not getattr(s, "factor", False)

    # Beside the backward a rank-B pass needs 72 registers: it fits beside the c2 WEIGHT gradient (440) and beside the small kernels in
    # front of the conv stack's backward, not beside the fused data gradient (475), which cannot start on a CU until the optimizer's
    # workgroup there has drained (the plain 48-register dd_adam_step runs beside all of them).  With the passes last (passes_last) there are
    # two places for a rank-B pass, both sized per batch row (config 2 at 32 rows in brackets):
    #   * the WINDOW beside the weight gradient (~1.2 ms; a pass moves ~2.5 GB/ms there): the largest registered tensor whose p / m / v
    #     bytes fit WINDOW_BYTES_PER_ROW (3.2 GB: fc1's 2.9 GB, 1.26 ms);
    #   * EARLY, on the side stream the moment the factors exist, beside the FC tail's and c3's HBM-bound kernels (1.5-2 ms until the data
    #     gradient is dispatched): the other tensors, smallest first, while they fit EARLY_BYTES_PER_ROW (4 GB: the head's 1.0 GB, 0.37 ms
    #     there; the autoencoder's decoder fc2, 3.9 GB, whose factors exist before the encoder's backward starts: 11.17 -> 10.96 ms).
    # Everything else keeps its materialised gradient and the plain pass.  A/B on one box each (profiles/r05_ab_rankb_schedule.txt): config 2
    # both in the window 7.40, head early 7.29 ms; autoencoder bs 32 with both of its tensors (6.7 GB) queued for the window 12.1 ms -- the data
    # gradient waited 1.6 ms for them --, fc1 in the window and the decoder's fc2 plain 11.0-11.2, fc2 early 10.9-11.0 (no rank-B: 11.1-11.4);
    # hidden 256: the head rides, fc1 (5.8 GB) stays plain.
    WINDOW_BYTES_PER_ROW = 1.0e8
    EARLY_BYTES_PER_ROW = 1.25e8

    def _rankb_plan(self, rows):
        """{tensor: "window" | "early"} for the registered tensors that take the rank-B pass at this batch size (overlap mode)."""
        live = sorted((q for q in self._rankb if q.requires_grad), key=lambda q: q.numel())
        key = (rows, self._last, tuple(id(q) for q in live))
        if getattr(self, "_plan_key", None) == key:
            return self._plan
        cost = lambda q: 24.0 * q.numel()
        plan = {}
        fits = [q for q in live if cost(q) <= self.WINDOW_BYTES_PER_ROW * rows]
        if fits:
            plan[fits[-1]] = "window"
        if self._last:
            budget = self.EARLY_BYTES_PER_ROW * rows
            for q in live:
                if q not in plan and q.numel() >= self._big_numel and cost(q) <= budget:
                    plan[q] = "early"
                    budget -= cost(q)
        else:      # passes in front of the weight gradient (no rank-B window of their own): what fits it together, smallest first
            budget = 1.25 * self.WINDOW_BYTES_PER_ROW * rows
            plan = {}
            for q in live:
                budget -= cost(q)
                if budget < 0:
                    break
                plan[q] = "window"
        self._plan_key, self._plan = key, plan
        return plan

    def _rankb_fits(self, p, rows):
        if self._side is None:
            return True                                       # passes after the backward, by themselves: no window to fit
        return p in self._rankb_plan(rows)

    def _goes_early(self, p, rows):
        return self._rankb_plan(rows).get(p) == "early"

    def linear_factors(self, weight, x, dy):
        """Called by ``ops.Linear.backward`` (on the backward's stream) for a weight registered in ``ops.RANKB``.  Returns 0: declined,
        the caller forms dW and db as usual; 1: the weight's gradient will be formed from (x, dy) inside its Adam pass, the caller
        still owes db; 2: the bias is updated by that pass too."""
        p = self._rankb_keys.get(weight.data_ptr())
        if p is None or not p.requires_grad or p.grad is not None or self._sync_blocks_rankb():
            return 0
        if x.shape[0] > self.MAX_ROWS or (p not in self._rankb_now and not self._rankb_fits(p, x.shape[0])):
            return 0
        if not (x.is_contiguous() and dy.is_contiguous() and x.dtype == torch.float32 and dy.dtype == torch.float32):
            return 0
        bias = self._rankb[p]
        if bias is not None and (not bias.requires_grad or bias.grad is not None):
            bias = None
        if p in self._early:
            raise RuntimeError("HipAdam (rank-B mode): a Linear layer ran a second backward after its optimizer pass of this step was "
                               "launched; call step() between the backwards or build TrainStep(fuse_linear_wgrad=False)")
        prev = self._rankb_now.get(p)
        if prev is not None:      # a second backward through the layer before the step (shared weight, retained graph): the gradients add
            x, dy = torch.cat([prev[0], x]), torch.cat([prev[1], dy])
        else:
            self._queue_rankb(p)
        self._rankb_now[p] = (x, dy, bias)
        if prev is None and self._side is not None and self._last and self._goes_early(p, x.shape[0]):
            # the window beside c2's weight gradient belongs to the LARGEST rank-B tensor (it fills it: fc1's pass 1.26 ms beside a 1.21 ms
            # kernel); a smaller one queued in front of it would push it past the kernel's end.  It goes to the side stream now, beside the
            # small HBM-bound kernels between here and the conv stack's backward (head: 0.37 ms there; config 2 7.40 -> 7.29 ms, same box)
            # One CU per XCD stays free of the pass: a single-workgroup kernel of the backward that needs nearly a whole CU's LDS
            # (dd_mlp_tail_bwd, 154 KB) cannot start beside a workgroup of the pass (16 KB) and, with one on every CU, waited for the first
            # of them to retire -- 0.6 ms of the autoencoder's step, whose decoder fc2 pass (1.0 ms) is in flight when the encoder's FC tail
            # comes up (profiles/r05_ae_bs32_kernel_stats.csv: mlp_tail_bwd 626 us there, 40 us by itself)
            ev = torch.cuda.current_stream().record_event()
            with torch.no_grad(), torch.cuda.stream(self._side):
                self._side.wait_event(ev)
                ops.check(ops._lib.lib().dd_set_adam_spare_cus(self.EARLY_SPARE_CUS), "dd_set_adam_spare_cus")
                try:
                    self._take_rankb(p, self._group_of(p), self._scale)
                finally:
                    ops.check(ops._lib.lib().dd_set_adam_spare_cus(0), "dd_set_adam_spare_cus")
            self._pending = [(q, g) for q, g in self._pending if q is not p]
        return 2 if bias is not None else 1

    MAX_ROWS = 64      # dd_adam_step_rankb: batch rows (world x batch for gathered factors)
    EARLY_SPARE_CUS = 8      # CUs an "early" pass leaves free of its workgroups (dd_set_adam_spare_cus)

    def factor_bias(self, weight, rows):
        """Factor mode of ddp.GradSync: whether the pass over the gathered factors (``rows`` = world x batch of them) will update this
        layer's bias too (then the caller owes no bias gradient)."""
        p = self._rankb_keys.get(weight.data_ptr())
        bias = self._rankb.get(p) if p is not None else None
        return (bias is not None and bias.requires_grad and bias.grad is None and rows <= self.MAX_ROWS and not self._sync_blocks_rankb())

    def _queue_rankb(self, p):
        if self._side is not None and p.numel() >= self._big_numel:
            self._pending.append((p, self._group_of(p)))      # launched on the side stream when backward reaches its MFMA-bound stretch

    def _update_rankb(self, p, group, grad_scale, x, dy, bias):
        """The one place ``dd_adam_step_rankb`` is called: Adam on ``p`` (and ``bias``) with the gradient dy^T x formed on the fly."""
        st = self._state_of(p)
        if "shards" in st:
            raise RuntimeError("HipAdam: a sharded tensor cannot take a rank-B update")
        st["step"] += 1
        bst = None
        if bias is not None:
            bst = self._state_of(bias)
            bst["step"] += 1
            if bst["step"] != st["step"]:
                raise RuntimeError("HipAdam: a Linear layer's weight and bias have taken different numbers of steps")
            self._early.add(bias)
        b1, b2 = group["betas"]
        ops.adam_step_rankb(p.data, st["exp_avg"], st["exp_avg_sq"], dy, x, None if bias is None else bias.data,
                            None if bst is None else bst["exp_avg"], None if bst is None else bst["exp_avg_sq"],
                            group["lr"], b1, b2, group["eps"], st["step"], grad_scale)
        self._held.append((x, dy))
        self._early.add(p)

    def _take_rankb(self, p, group, grad_scale):
        now = self._rankb_now.pop(p, None)
        if now is None:
            return False
        self._update_rankb(p, group, grad_scale, *now)
        return True

    def attach(self, grad_sync):
        """Use ``grad_sync`` (ddp.GradSync) for per-piece waits and, in shard mode, for the shards and their all-gathers."""
        self._sync = grad_sync
        if getattr(grad_sync, "factor", False):
            grad_sync.on_factors = self._factored.append

    def _group_of(self, p):
        for group in self.param_groups:
            if any(q is p for q in group["params"]):
                return group
        raise RuntimeError("HipAdam: factors arrived for a tensor this optimizer does not own")

    def _form_factored(self, p):
        """Factor mode: wait (on the CURRENT stream) for the gathers of the layer's input and output gradient and form the
        global-batch weight gradient dy_all^T x_all into ``p.grad``."""
        fac = self._sync.take_factors(p)
        if fac is None:
            return False
        for work in fac.works:
            work.wait()
        if p in self._rankb and fac.rows <= self.MAX_ROWS:      # rank-B mode: no gradient tensor, the gathered factors go straight into the Adam pass
            self._fac_now[p] = fac
            return True
        g = self._fgrad.get(p)
        if g is None or g.shape != p.shape or g.device != p.device:
            g = self._fgrad[p] = torch.empty_like(p, memory_format=torch.contiguous_format)
        ops.linear_wgrad(fac.dy_all, fac.x_all, g)
        p.grad = g
        return True

    def _update_formed(self, p, group, grad_scale):
        """The Adam pass behind ``_form_factored``: from ``p.grad``, or -- rank-B mode -- straight from the gathered factors."""
        fac = self._fac_now.pop(p, None)
        if fac is None:
            return self._update(p, group, grad_scale)
        bias = self._rankb[p]
        if bias is not None and (not bias.requires_grad or bias.grad is not None):
            bias = None
        self._update_rankb(p, group, grad_scale, fac.x_all, fac.dy_all, bias)

    def _update_factored(self, p, group, grad_scale):
        if not self._form_factored(p):
            return False
        self._update_formed(p, group, grad_scale)
        return True

    @torch.no_grad()
    def _flush_factored(self):
        """Between c2's weight and data gradient (ops.MFMA_PHASE2_HOOKS): the factors gathered since the top of the backward have
        arrived (N = 2: 202 MB over one link in ~3 ms).  The weight-gradient kernels run HERE, on the backward's stream -- on the side
        stream, beside conv kernels that fill every CU with 440-register waves, they starve (measured on a one-rank communicator:
        81 us -> 1.2 ms) -- and the Adam passes, built to run beside those kernels, go to the side stream behind them."""
        if not self._factored:
            return
        done = [p for p in self._factored if self._form_factored(p)]
        self._factored.clear()
        ev = torch.cuda.current_stream().record_event()
        with torch.cuda.stream(self._side):
            self._side.wait_event(ev)
            for p in done:
                self._update_formed(p, self._group_of(p), self._scale)
                self._early.add(p)

    def _update_shards(self, p, group, grad_scale, shards):
        """The update of the slices of ``p`` this rank owns: wait (on the current stream) for piece k's reduce-scatter, update the
        slice, start its all-gather behind the update, go on.  Whole moments found in the state (a checkpoint in torch.optim.Adam's
        layout -- the reference's -- or steps taken in all-reduce mode before) are cut into this rank's slices on the way."""
        st = self.state[p]
        if not st:
            st["step"] = 0
            st["shards"] = {}
        whole = None
        if "shards" not in st:
            whole = (st.pop("exp_avg").reshape(-1), st.pop("exp_avg_sq").reshape(-1))
            st["shards"] = {}
        st["step"] += 1
        for sh in shards:
            if sh.work is not None:
                sh.work.wait()
            mv = st["shards"].get(sh.index)
            if mv is None or mv[0].numel() != sh.param.numel():
                if whole is not None:
                    mv = (whole[0][sh.lo:sh.hi].clone(), whole[1][sh.lo:sh.hi].clone())
                else:
                    mv = (torch.zeros_like(sh.param), torch.zeros_like(sh.param))
                st["shards"][sh.index] = mv
            self._launch(sh.param, sh.grad, mv[0], mv[1], group, st["step"], grad_scale)
            self._sync.gather_shard(p, sh)

    # ---- checkpoints of the optimizer ------------------------------------------------------------------------------------------
    def _whole_moments(self, p, st):
        """(exp_avg, exp_avg_sq) of ``p`` as whole tensors: the sharded moments all-gathered piece by piece (a COLLECTIVE when the
        GradSync is live: every rank calls it, for the same tensors in the same order)."""
        sync = self._sync
        if sync is None or not getattr(sync, "shard", False):
            raise RuntimeError("HipAdam: sharded moments but no sharded GradSync attached to put them together")
        import torch.distributed as dist
        m = torch.zeros_like(p, memory_format=torch.contiguous_format)
        v = torch.zeros_like(p, memory_format=torch.contiguous_format)
        mf, vf = m.view(-1), v.view(-1)
        w, r = sync.shard_world, sync.rank
        for k, (a, b) in enumerate(sync._pieces_of(p.numel())):
            n = (b - a) // w
            mv = st["shards"].get(k)
            if mv is None:
                mv = (torch.zeros(n, device=p.device, dtype=p.dtype), torch.zeros(n, device=p.device, dtype=p.dtype))
            if sync.active:
                dist.all_gather_into_tensor(mf[a:b], mv[0], group=sync.group)
                dist.all_gather_into_tensor(vf[a:b], mv[1], group=sync.group)
            else:                                 # simulate_world: one process holds rank 0's slices only
                mf[a + r * n:a + (r + 1) * n].copy_(mv[0])
                vf[a + r * n:a + (r + 1) * n].copy_(mv[1])
        return m, v

    def consolidated_state_dict(self):
        """``state_dict()`` in torch.optim.Adam's layout (``step``, ``exp_avg``, ``exp_avg_sq`` per parameter: what the reference's
        ``configure_optimizers()`` optimizer writes, autoencoder.py:119-120 / roadmap_bce_v2.py:154-157) whatever mode the steps were
        taken in.  In shard mode each rank holds 1 / world of the moments of a sharded tensor: they are all-gathered here, so EVERY rank
        must call this (the result is the same on all of them; save rank 0's).  Loads into ``torch.optim.Adam`` and into a ``HipAdam``
        of any mode (``_update_shards`` cuts whole moments into slices again)."""
        if self._sync is not None and hasattr(self._sync, "wait_gathers"):
            self._sync.wait_gathers()
        if self._side is not None:
            torch.cuda.current_stream().wait_stream(self._side)
        sd = super().state_dict()
        flat = [p for group in self.param_groups for p in group["params"]]
        for idx, p in enumerate(flat):
            st = sd["state"].get(idx)
            if st is None:
                continue
            st = dict(st)
            if "shards" in st:
                m, v = self._whole_moments(p, st)
                del st["shards"]
                st["exp_avg"], st["exp_avg_sq"] = m, v
            sd["state"][idx] = st
        return sd

    def load_state_dict(self, state_dict):
        """Takes torch.optim.Adam's state (whole moments; ``step`` possibly a tensor) as well as this class's own."""
        super().load_state_dict(state_dict)
        for st in self.state.values():
            if "step" in st and isinstance(st["step"], torch.Tensor):
                st["step"] = int(st["step"].item())

    def _update_pieces(self, p, group, grad_scale, pieces):
        """The update of a tensor whose all-reduce travels in pieces: wait (on the current stream) for piece k, update that
        slice, go on -- the pass over piece k runs while piece k+1 is still on the links."""
        st = self._state_of(p)
        st["step"] += 1
        b1, b2 = group["betas"]
        if not (p.grad.is_contiguous() and p.data.is_contiguous()):
            for work, _, _ in pieces:
                work.wait()
            st["step"] -= 1
            return self._update(p, group, grad_scale)
        pf, gf, mf, vf = p.data.view(-1), p.grad.view(-1), st["exp_avg"].view(-1), st["exp_avg_sq"].view(-1)
        for work, off, n in pieces:
            work.wait()
            self._launch(pf[off:off + n], gf[off:off + n], mf[off:off + n], vf[off:off + n], group, st["step"], grad_scale)

    def _update(self, p, group, grad_scale):
        st = self._state_of(p)
        if "shards" in st:
            if self._sync is None or not getattr(self._sync, "shard", False):
                raise RuntimeError("HipAdam: a tensor with sharded moments is updated whole and no sharded GradSync is attached to put "
                                   "them together (GradSync removed while its optimizer lives on?): save with consolidated_state_dict() "
                                   "before leaving shard mode")
            st["exp_avg"], st["exp_avg_sq"] = self._whole_moments(p, st)      # a collective: every rank is here for the same tensor
            del st["shards"]
        st["step"] += 1
        g = p.grad if p.grad.is_contiguous() else p.grad.contiguous()
        self._launch(p.data.view(-1), g.view(-1), st["exp_avg"].view(-1), st["exp_avg_sq"].view(-1), group, st["step"], grad_scale)

    def overlap_with_backward(self, big_numel=1 << 20, grad_scale=1.0, grad_sync=None):
        """Run the Adam pass of every parameter of >= ``big_numel`` elements on a side stream the moment autograd has
        finished its gradient (post-accumulate-grad hook) instead of after backward.

        Same arithmetic, different timing: the 481 MB encoder ``fc1`` weight and the 164 MB head weight get their
        gradients at the START of backward, and their optimizer pass is pure HBM streaming (7 passes over the
        tensor), while the conv data/weight-gradient kernels that follow are MFMA-bound and leave most of the HBM
        bandwidth idle -- so ~0.8 of the 1.0 ms Adam time disappears under them.  With ``grad_sync`` (data parallel)
        the side stream first waits for that parameter's all-reduce.  ``step()`` then only handles the small
        parameters and joins the side stream."""
        self._side = torch.cuda.Stream()
        ops.check(ops._lib.lib().dd_set_adam_blocks_per_cu(1), "dd_set_adam_blocks_per_cu")      # beside conv kernels: nothing queued ahead of them
        self._scale = grad_scale
        if grad_sync is not None:
            self._sync = grad_sync
        self._pending = []
        self._big_numel = big_numel
        self._hooked = set()
        # (c2's data gradient first, the passes beside its weight gradient last: same step time, round 5 A/B 7.51 / 7.51 ms)
        ops.MFMA_PHASE_HOOKS.append(self._flush_pending)
        ops.MFMA_PHASE2_HOOKS.append(self._flush_factored)
        if getattr(self._sync, "factor", False):
            ops.C2_DGRAD_FIRST = True             # the Adam passes behind the gathered factors run beside c2's weight gradient: it goes last
        self.refresh()
        lightning.on_unfreeze(self)

    def passes_last(self, on=True):
        """Where the queued optimizer passes are launched in the encoder's backward: at the first MFMA hook (in front of c2's weight
        gradient, then its data gradient: the round-1..4 order) or, ``on``, with c2's DATA gradient first and the passes beside its
        weight gradient, which then goes last (``ops.C2_DGRAD_FIRST``).  The rank-B passes (72 registers) fit beside the weight gradient
        only: launched in front of it they are still resident when the 475-register data gradient is dispatched, and that kernel waits
        for them CU by CU (1.9 ms from dispatch to end for 1.4 ms of work).  Same step time either way on one GPU (round 5 A/B: 7.51 /
        7.51 ms), but with the passes last the step's longest kernel runs -- and is timed -- by itself."""
        if self._side is None:
            return
        here, there = (ops.MFMA_PHASE2_HOOKS, ops.MFMA_PHASE_HOOKS) if on else (ops.MFMA_PHASE_HOOKS, ops.MFMA_PHASE2_HOOKS)
        if self._flush_pending in there:
            there.remove(self._flush_pending)
        if self._flush_pending not in here:
            here.insert(0, self._flush_pending)
        self._last = bool(on)
        ops.C2_DGRAD_FIRST = bool(on) or bool(getattr(self._sync, "factor", False))

    def refresh(self):
        """Hook the big parameters that require a gradient and are not hooked yet: a frozen feature extractor
        (roadmap_bce_v2.py:45-47) gets its hooks when ``LightningModule.unfreeze()`` switches it on.  A big parameter
        that was never hooked is simply updated in ``step()`` with the small ones."""
        if self._side is None:
            return
        for group in self.param_groups:
            for p in group["params"]:
                if p.requires_grad and p.numel() >= self._big_numel and p not in self._hooked:
                    self._hooks.append(p.register_post_accumulate_grad_hook(lambda q, g=group: self._early_step(q, g)))
                    self._hooked.add(p)

    def close(self):
        """Undo overlap_with_backward: remove the gradient hooks and the backward-phase callback (an optimizer that is dropped
        while another model is trained in the same process would otherwise stay registered)."""
        for h in self._hooks:
            h.remove()
        self._hooks = []
        self._hooked = set()
        for key in list(self._rankb_keys):
            if ops.RANKB.get(key) is self:
                del ops.RANKB[key]
        self._rankb, self._rankb_keys, self._rankb_now, self._fac_now = {}, {}, {}, {}
        if self._flush_pending in ops.MFMA_PHASE_HOOKS:
            ops.MFMA_PHASE_HOOKS.remove(self._flush_pending)
        if self._flush_pending in ops.MFMA_PHASE2_HOOKS:
            ops.MFMA_PHASE2_HOOKS.remove(self._flush_pending)
        if self._flush_factored in ops.MFMA_PHASE2_HOOKS:
            ops.MFMA_PHASE2_HOOKS.remove(self._flush_factored)
            ops.C2_DGRAD_FIRST = False
        if self._side is not None:
            torch.cuda.current_stream().wait_stream(self._side)
        self._side = None
        self._sync = None

    def _early_step(self, p, group):
        if p.grad is not None:
            self._pending.append((p, group))      # launched when backward reaches its MFMA-bound stretch

    @torch.no_grad()
    def _flush_pending(self):
        if not self._pending:
            return
        ev = torch.cuda.current_stream().record_event()
        with torch.cuda.stream(self._side):
            self._side.wait_event(ev)
            # rank-B passes first: they only fit beside the weight gradient, the plain passes behind them also run beside the data gradient
            for p, group in self._pending:
                if p in self._rankb_now:
                    self._take_rankb(p, group, self._scale)
            for p, group in self._pending:
                if p in self._early:
                    continue
                shards = self._sync.shards(p) if self._sync is not None else None
                pieces = self._sync.pieces(p) if self._sync is not None else None
                if shards:                        # sharded optimizer: this rank's slices only, each all-gathered behind its update
                    self._update_shards(p, group, self._scale, shards)
                elif pieces:                      # the side stream (not the host) waits for the all-reduce, piece by piece
                    self._update_pieces(p, group, self._scale, pieces)
                else:
                    self._update(p, group, self._scale)
                self._early.add(p)
        self._pending = []

    @torch.no_grad()
    def step(self, grad_scale=1.0):
        if self._side is not None:
            self._flush_pending()                 # backward never reached an MFMA phase hook (other models)
        self._factored.clear()
        for group in self.param_groups:
            small = {}                            # step count -> [(p, g, m, v)]: one launch for all the small tensors
            for p in group["params"]:
                if p in self._rankb_now and p not in self._early:      # rank-B mode without the side stream: here, after the backward
                    self._take_rankb(p, group, grad_scale)
                    continue
                if p not in self._early and self._sync is not None and getattr(self._sync, "factor", False) and self._sync.has_factors(p):
                    self._update_factored(p, group, grad_scale)      # no side stream: here, after the backward
                    continue
                if p.grad is None or p in self._early:
                    continue
                shards = self._sync.shards(p) if self._sync is not None else None
                if shards:
                    self._update_shards(p, group, grad_scale, shards)
                    continue
                if p.numel() > self.SMALL_NUMEL or not p.grad.is_contiguous() or not p.data.is_contiguous():
                    self._update(p, group, grad_scale)
                    continue
                st = self._state_of(p)
                st["step"] += 1
                small.setdefault(st["step"], []).append((p.data.view(-1), p.grad.view(-1), st["exp_avg"].view(-1),
                                                          st["exp_avg_sq"].view(-1)))
            b1, b2 = group["betas"]
            for step, quads in small.items():
                ops.adam_step_multi(quads, group["lr"], b1, b2, group["eps"], step, grad_scale)
        if self._side is not None:
            torch.cuda.current_stream().wait_stream(self._side)
        self._early.clear()
        self._held.clear()
        self._rankb_now.clear()
        self._fac_now.clear()
