"""Data-parallel gradient synchronisation: one process per GPU, RCCL over xGMI via torch.distributed.

The reference has no distributed code of its own; multi-GPU is whatever Lightning's ``--gpus N
--distributed_backend ddp`` does: DDP gradient averaging.  Here it is explicit and shaped
for this model: 99.6 % of the 648 MB gradient is two tensors -- the head ``fc1.weight`` (164 MB) and the
encoder's ``fc1.fc1.weight`` (481 MB) -- and both are produced EARLY in backward (they sit next to the
loss), long before the conv data/weight-gradient kernels where the FLOPs are.  So:

  * a gradient of >= ``big_numel`` elements is all-reduced by itself, asynchronously, the moment autograd
    has accumulated it (post-accumulate-grad hook): no flat copy, no bucket fill, and the collective runs
    on RCCL's stream underneath the remaining backward kernels.  It goes out in pieces of ``chunk_numel``
    elements (128 MB), in order, so that the optimizer pass of piece k (HipAdam waits per piece on its side
    stream) runs while piece k+1 is still on the links: only the last piece's update is left after the
    last byte has arrived, not the whole tensor's;
  * everything else (a few hundred KB) is flattened into one buffer and reduced once at ``finish()``;
  * the sum is NOT divided here: ``HipAdam.step(grad_scale=1/world)`` folds the average into its pass.

Budget at 8 GPUs for the 7.8 ms step of round 1 (forward 2.8 ms, backward 5.0 ms of which the MFMA-bound c2 / c1 kernels
are the last 3.3 ms): the two big gradients exist 0.5 ms into the backward, so their all-reduce has ~4.5 ms of backward to
hide under.  xGMI is point-to-point (7 links x ~153 GB/s per GPU): ONE ring moves 2 x 7/8 x 648 MB through a single link
= 7.4 ms (not hidden: +2.9 ms, 5.8x at 8 GPUs); RCCL's multi-ring / direct algorithms over all seven links need
2 x 81 MB per link and phase = 1.1 ms at wire speed, ~3 ms at the ~330 GB/s bus bandwidth RCCL typically reaches: hidden.
What is then left after the last byte is one 128 MB piece's Adam pass (0.35 ms) plus the small tensors: ~8.3 ms per
step = 7.5x.  >= 6x needs the 8-GPU step <= 10.4 ms, i.e. at most 2.6 ms of exposed communication.  Constructing a
GradSync broadcasts rank 0's parameters and buffers, so ranks cannot start from different weights.

Frozen parameters (the reference's fine-tuning modules start with ``self.ae.freeze()`` and call ``self.ae.unfreeze()`` at
``unfreeze_epoch_no``: roadmap_bce_v2.py:45-47,127-129, spatial_w_rm.py:45-48,148-150, roadmap_pretrain_ae.py:131): torch
refuses a gradient hook on a tensor that does not require gradients, so only trainable parameters are hooked, and the
others are hooked the moment they become trainable -- ``LightningModule.unfreeze()`` calls ``refresh()`` on every live
GradSync (``lightning.on_unfreeze``), and ``finish()`` catches whatever was switched on by hand (``p.requires_grad_(True)``)
since: such a gradient is reduced there, synchronously, and hooked from then on.  Every rank takes the same decisions (the
epoch counter is the same everywhere), so the collectives still pair up.
"""
import torch
import torch.distributed as dist

from . import lightning


class GradSync:
    def __init__(self, module, process_group=None, big_numel=1 << 20, chunk_numel=1 << 25, reserve_cus=0, force_collectives=False):
        self.group = process_group
        self.world = dist.get_world_size(process_group) if dist.is_initialized() else 1
        # force_collectives: issue every collective even in a 1-rank group -- a rehearsal of the N > 1 call pattern (async
        # all-reduce of gradient pieces from autograd hooks, the optimizer's per-piece waits, the CU hand-over) on the real
        # backend when only one GPU is at hand; results are unchanged (a 1-rank all-reduce is the identity)
        self.active = self.world > 1 or (bool(force_collectives) and dist.is_initialized())
        self.big_numel = big_numel
        self.chunk_numel = max(4, chunk_numel - chunk_numel % 4)      # pieces start on 16-byte boundaries
        # RCCL's workgroups need compute units the resident conv grids do not leave: while collectives are in flight (from
        # the first big gradient to finish()) the conv kernels are launched on 256 - reserve_cus units; the forward, which
        # runs beside no collective, keeps the whole GPU.
        self.reserve_cus = int(reserve_cus) if self.active else 0
        self._reserved = False
        self.params = [p for p in module.parameters()]
        if self.active:      # every replica starts from rank 0's weights and BatchNorm statistics (DDP's contract)
            with torch.no_grad():
                for t in list(module.parameters()) + list(module.buffers()):
                    dist.broadcast(t.data, src=0, group=self.group)
        self._handles = []
        self._by_param = {}
        self._small = []
        self._hooks = []
        self._hooked = set()
        self.refresh()
        lightning.on_unfreeze(self)

    def refresh(self):
        """Hook every parameter that requires a gradient and has no hook yet (called again after an ``unfreeze()``)."""
        if not self.active:
            return
        for p in self.params:
            if p.requires_grad and p not in self._hooked:
                self._hooks.append(p.register_post_accumulate_grad_hook(self._on_grad))
                self._hooked.add(p)

    @property
    def grad_scale(self):
        return 1.0 / self.world

    def _on_grad(self, p):
        if p.grad is None:
            return
        if p.grad.numel() >= self.big_numel and self.reserve_cus and not self._reserved:
            self._set_budget(256 - self.reserve_cus)
            self._reserved = True
        if p.grad.numel() >= self.big_numel and p.grad.is_contiguous():
            flat = p.grad.view(-1)
            pieces = []
            for off in range(0, flat.numel(), self.chunk_numel):
                n = min(self.chunk_numel, flat.numel() - off)
                work = dist.all_reduce(flat[off:off + n], op=dist.ReduceOp.SUM, group=self.group, async_op=True)
                self._handles.append(work)
                pieces.append((work, off, n))
            self._by_param[p] = pieces
        elif p.grad.numel() >= self.big_numel:
            work = dist.all_reduce(p.grad, op=dist.ReduceOp.SUM, group=self.group, async_op=True)
            self._handles.append(work)
            self._by_param[p] = [(work, 0, p.grad.numel())]
        else:
            self._small.append(p)

    def pieces(self, p):
        """[(work, offset, numel), ...] of the in-flight all-reduce of ``p`` in issue order (None: p went the small-tensor way)."""
        return self._by_param.get(p)

    def wait_param(self, p):
        """Make the CURRENT stream wait for the whole all-reduce of ``p`` (no-op when p went the small-tensor way)."""
        for work, _, _ in self._by_param.get(p) or ():
            work.wait()

    def finish(self):
        """Reduce the small gradients in one message and wait for everything in flight."""
        if self.active:
            # parameters switched to requires_grad by hand since the last refresh(): their hooks did not exist during this
            # backward, so their gradients are still local -- reduce them here (with the small tensors, or by themselves when
            # big) and hook them for the steps to come
            late = [p for p in self.params if p.requires_grad and p not in self._hooked and p.grad is not None]
            for p in late:
                if p.grad.numel() >= self.big_numel:
                    self._handles.append(dist.all_reduce(p.grad, op=dist.ReduceOp.SUM, group=self.group, async_op=True))
                else:
                    self._small.append(p)
            if late:
                self.refresh()
        if self.active and self._small:
            flat = torch.cat([p.grad.reshape(-1) for p in self._small])
            dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=self.group)
            off = 0
            for p in self._small:
                n = p.grad.numel()
                p.grad.copy_(flat[off:off + n].view_as(p.grad))
                off += n
        for h in self._handles:
            h.wait()
        self._handles, self._small, self._by_param = [], [], {}
        if self._reserved:
            self._set_budget(256)
            self._reserved = False

    @staticmethod
    def _set_budget(cus):
        from . import _lib
        _lib.check(_lib.lib().dd_set_cu_budget(cus), "dd_set_cu_budget")

    def remove(self):
        for h in self._hooks:
            h.remove()
        self._hooks = []
        self._hooked = set()
        self.active = False
