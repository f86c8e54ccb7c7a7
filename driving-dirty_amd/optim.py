"""``HipAdam``: torch.optim.Adam semantics (reference autoencoder.py:119-120, roadmap_bce_v2.py:154-157)
executed by ``dd_adam_step``, one fused read-modify-write pass per parameter tensor.

``grad_scale`` folds the 1/world_size of data-parallel gradient averaging into the same pass, so the
all-reduced SUM never needs a separate divide kernel.

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

    SMALL_NUMEL = int(os.environ.get("DD_ADAM_MULTI_NUMEL", 1 << 16))      # tensors up to this size go into one multi-tensor launch (0: never)

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
        g = self._fgrad.get(p)
        if g is None or g.shape != p.shape or g.device != p.device:
            g = self._fgrad[p] = torch.empty_like(p, memory_format=torch.contiguous_format)
        ops.linear_wgrad(fac.dy_all, fac.x_all, g)
        p.grad = g
        return True

    def _update_factored(self, p, group, grad_scale):
        if not self._form_factored(p):
            return False
        self._update(p, group, grad_scale)
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
                self._update(p, self._group_of(p), self._scale)
                self._early.add(p)

    def _update_shards(self, p, group, grad_scale, shards):
        """The update of the slices of ``p`` this rank owns: wait (on the current stream) for piece k's reduce-scatter, update the
        slice, start its all-gather behind the update, go on."""
        st = self.state[p]
        if not st:
            st["step"] = 0
            st["shards"] = {}
        if "shards" not in st:
            raise RuntimeError("HipAdam: a tensor that was updated whole is now sharded (GradSync(shard_optimizer=True) must be "
                               "attached before the first step)")
        st["step"] += 1
        for sh in shards:
            if sh.work is not None:
                sh.work.wait()
            mv = st["shards"].get(sh.index)
            if mv is None or mv[0].numel() != sh.param.numel():
                mv = st["shards"][sh.index] = (torch.zeros_like(sh.param), torch.zeros_like(sh.param))
            self._launch(sh.param, sh.grad, mv[0], mv[1], group, st["step"], grad_scale)
            self._sync.gather_shard(p, sh)

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
            raise RuntimeError("HipAdam: a sharded tensor arrived without shards (GradSync.finish() of this step not reached, or the "
                               "GradSync was removed while its optimizer lives on)")
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
        self._scale = grad_scale
        if grad_sync is not None:
            self._sync = grad_sync
        self._pending = []
        self._big_numel = big_numel
        self._hooked = set()
        ops.MFMA_PHASE_HOOKS.append(self._flush_pending)
        ops.MFMA_PHASE2_HOOKS.append(self._flush_factored)
        if getattr(self._sync, "factor", False):
            ops.C2_DGRAD_FIRST = True             # the Adam passes behind the gathered factors run beside c2's weight gradient: it goes last
        self.refresh()
        lightning.on_unfreeze(self)

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
        if self._flush_pending in ops.MFMA_PHASE_HOOKS:
            ops.MFMA_PHASE_HOOKS.remove(self._flush_pending)
        if self._flush_factored in ops.MFMA_PHASE2_HOOKS:
            ops.MFMA_PHASE2_HOOKS.remove(self._flush_factored)
            if getattr(self._sync, "factor", False):
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
            for p, group in self._pending:
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
