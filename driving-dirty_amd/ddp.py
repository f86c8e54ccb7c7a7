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

Sharded optimizer (``shard_optimizer=True``; round 4).  Configs 4 and 5 of BASELINE.json exist only on 8 GPUs, and for config 5
(2.09 GB of gradients, 14.6 GB of Adam traffic per step and GPU) the replicated optimizer is the largest single consumer of HBM
bytes in the step.  In shard mode a big gradient is REDUCE-SCATTERED instead of all-reduced: the flat tensor is cut into the same
in-order pieces, every piece into ``world`` equal slices, and rank r receives the sum of slice r of every piece (a persistent
buffer of numel / world elements per tensor).  The optimizer then updates only the slices this rank owns -- Adam state exists only
for them: 1/world of the moments' memory and of the pass's HBM traffic -- and every updated slice is ALL-GATHERED in place into the
parameter (input = the owned slice of the output: no staging copy), asynchronously: the gather runs on RCCL's stream underneath the
NEXT step's forward, and the first kernel that touches the parameter waits for it (``PARAM_WAITS``, consulted where the shims take a
tensor's device pointer).  Same bytes on the links per step (RS + AG = AR), but each phase has its own compute to hide under, and
the arithmetic is elementwise: the replicas hold bit-for-bit the parameters of the all-reduce path whenever the backend's
reduce-scatter adds in the order of its all-reduce (gloo: always; tests/test_ddp_gloo.py, world sizes 2 and 4).  In shard mode
``p.grad`` of a sharded tensor keeps THIS rank's local gradient; the reduced gradient exists only as ``shards(p)``.
Factor gather (``factor_linear=True``; round 4, for 2 <= world <= 4).  The weight gradient of a Linear layer over an m-row batch is a
rank-m product, dW = dY^T X, and for the two tensors that ARE the message here the factors are far smaller than the product: the
encoder's ``fc1.fc1`` has 120 MB of pooled activations X (32 x 940032) and 16 KB of dY against a 481 MB gradient, the head 8 KB of X
and 82 MB of dlogits against 164 MB.  ``ops.Linear.backward`` hands X and dY to ``linear_factors`` instead of forming its local dW;
both are ALL-GATHERED over the batch dimension (asynchronously, under the rest of the backward) and every rank forms the
global-batch gradient itself (``HipAdam``: the weight-gradient kernel over world x m rows, then the replicated Adam pass, on its side
stream behind the gathers).  Bytes received per rank: (world - 1) x 202 MB instead of 2 x (world - 1) / world x 648 MB -- 202 against
648 MB at world 2, where ONE xGMI link carries everything and no schedule hides 648 MB under a 7.6 ms step; 606 against 972 at 4;
1414 against 1134 at 8, where the reduce-scatter wins.  Same products as the all-reduce path, summed in a
different order (one GEMM over the global batch instead of a sum of per-rank GEMMs); every rank computes from identical gathered
factors with a deterministic kernel, so the replicas stay bit-identical to EACH OTHER.  The bias gradients and everything else travel
as before.  Every rank must bring the SAME number of rows (``all_gather_into_tensor``: use ``drop_last`` on the loader -- the all-reduce
path has no such condition), and a registered layer runs ONE backward per optimizer step (a second one raises).

``simulate_world=N`` (one process, no communicator) cuts the shards of rank 0 of an N-rank job out of the local gradient and skips
the gather: the COMPUTE side of an N-GPU sharded step on one GPU -- a timing aid (`bench.py --simulate-shard N`), the parameters
it leaves are meaningless.
"""
import torch
import torch.distributed as dist

from . import lightning

# data_ptr of a parameter -> callable that makes the CURRENT stream wait for the all-gather still writing into it (shard mode).
# ops._p / gconv._p pop and call the entry the first time a kernel operand with that address is handed to the C ABI.
PARAM_WAITS = {}


def param_ready(t):
    """Wait (on the current stream) for an in-flight all-gather into ``t``; no-op otherwise."""
    if PARAM_WAITS:
        wait = PARAM_WAITS.pop(t.data_ptr(), None)
        if wait is not None:
            wait()


# data_ptr of a Linear weight -> GradSync in factor mode: ops.Linear.backward hands its factors over instead of forming dW
FACTOR_SYNC = {}


class Factors:
    """The gathered factors of one Linear weight gradient: dW = dy_all^T x_all over ``rows`` = world x m rows, valid once every
    handle in ``works`` has been waited for."""
    __slots__ = ("works", "x_all", "dy_all", "rows")

    def __init__(self, works, x_all, dy_all, rows):
        self.works, self.x_all, self.dy_all, self.rows = works, x_all, dy_all, rows


class Shard:
    """One piece's slice owned by this rank: ``param`` = flat view of p.data[lo:hi] (updated in place by the optimizer),
    ``grad`` = the summed gradient of exactly those elements, valid once ``work`` (None: already there) has been waited for."""
    __slots__ = ("work", "index", "lo", "hi", "piece_lo", "piece_hi", "param", "grad")

    def __init__(self, work, index, lo, hi, piece_lo, piece_hi, param, grad):
        self.work, self.index, self.lo, self.hi, self.piece_lo, self.piece_hi, self.param, self.grad = \
            work, index, lo, hi, piece_lo, piece_hi, param, grad


class GradSync:
    def __init__(self, module, process_group=None, big_numel=1 << 20, chunk_numel=1 << 25, reserve_cus=0, force_collectives=False,
                 shard_optimizer=False, simulate_world=0, factor_linear=False):
        self.group = process_group
        self.world = dist.get_world_size(process_group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(process_group) if dist.is_initialized() else 0
        # force_collectives: issue every collective even in a 1-rank group -- a rehearsal of the N > 1 call pattern (async
        # all-reduce / reduce-scatter of gradient pieces from autograd hooks, the optimizer's per-piece waits, the all-gathers,
        # the CU hand-over) on the real backend when only one GPU is at hand; results are unchanged (1-rank collectives are copies)
        self.active = self.world > 1 or (bool(force_collectives) and dist.is_initialized())
        self.simulate_world = int(simulate_world) if (simulate_world and not self.active) else 0
        self.shard = bool(shard_optimizer) and (self.active or self.simulate_world > 0)
        self.shard_world = self.simulate_world if self.simulate_world else self.world
        self.big_numel = big_numel
        # factor mode: big Linear weights send their factors (all-gather over the batch) instead of their gradients; not with shards
        self.factor = bool(factor_linear) and self.active and not self.shard
        self._factors = {}         # p -> Factors of the latest backward, until the optimizer takes them
        self._fbuf = {}            # p -> (x_all, dy_all): persistent gather targets
        self._fkeys = {}           # data_ptr -> p
        self._xwork = {}           # p -> (work, data_ptr, rows) of an input gather started in the forward
        self.on_factors = None     # callback(p): the optimizer queues its early update (HipAdam.attach)
        self.chunk_numel = max(4, chunk_numel - chunk_numel % 4)      # pieces start on 16-byte boundaries
        if self.shard:      # every piece splits into shard_world slices of whole 16-byte groups
            q = 4 * self.shard_world
            self.chunk_numel = max(q, chunk_numel - chunk_numel % q)
        # RCCL's workgroups need compute units the resident conv grids do not leave: while collectives are in flight (from
        # the first big gradient to finish()) the conv kernels are launched on 256 - reserve_cus units; the forward, which
        # runs beside no collective (shard mode: beside the tail of the all-gathers), keeps the whole GPU.
        self.reserve_cus = int(reserve_cus) if self.active else 0
        self._reserved = False
        self.params = [p for p in module.parameters()]
        # the module that owns each parameter directly: a consumer that CALLS that module (plain torch layers) waits for an
        # in-flight all-gather through a forward pre-hook; the hot path's shims, which take weights by pointer, through PARAM_WAITS
        self._owner = {p: m for m in module.modules() for p in m.parameters(recurse=False)}
        self._pre_hooked = set()
        if self.active:      # every replica starts from rank 0's weights and BatchNorm statistics (DDP's contract)
            with torch.no_grad():
                for t in list(module.parameters()) + list(module.buffers()):
                    dist.broadcast(t.data, src=0, group=self.group)
        self.early_input_gathers = 0      # factor mode: input gathers started from the forward and picked up by the backward (tests)
        self._handles = []
        self._by_param = {}
        self._shards = {}          # p -> [Shard]: kept until the next backward replaces them (the optimizer reads them after finish())
        self._gshard = {}          # p -> persistent numel / shard_world buffer the reduce-scatters write into
        self._gathers = {}         # p -> [work]: all-gathers of updated slices still in flight
        self._small = []
        self._hooks = []
        self._hooked = set()
        if self.shard and self.active:
            # shard mode leaves all-gathers writing into the parameters when a step returns: whoever reads the parameters as a whole --
            # state_dict(), a checkpoint (LightningModule.save goes through state_dict), an EMA copy made from it -- first waits for them
            self._hooks.append(module.register_state_dict_pre_hook(lambda _m, _prefix, _keep: self.wait_gathers()))
        self.refresh()
        lightning.on_unfreeze(self)

    def refresh(self):
        """Hook every parameter that requires a gradient and has no hook yet (called again after an ``unfreeze()``)."""
        if not (self.active or self.shard):
            return
        for p in self.params:
            if p.requires_grad and p not in self._hooked:
                self._hooks.append(p.register_post_accumulate_grad_hook(self._on_grad))
                self._hooked.add(p)
                if self.factor and p.dim() == 2 and p.numel() >= self.big_numel and p.data.is_contiguous():
                    FACTOR_SYNC[p.data_ptr()] = self
                    self._fkeys[p.data_ptr()] = p

    # ---- factor mode ------------------------------------------------------------------------------------------------------
    def _factor_bufs(self, p, m, k, n, like):
        w = self.world
        bufs = self._fbuf.get(p)
        if bufs is None or bufs[0].shape != (w * m, k) or bufs[1].shape != (w * m, n) or bufs[0].device != like.device:
            bufs = self._fbuf[p] = (torch.empty((w * m, k), device=like.device, dtype=like.dtype),
                                    torch.empty((w * m, n), device=like.device, dtype=like.dtype))
        return bufs

    def linear_input(self, weight, x):
        """Called by ``ops.Linear.forward`` (training, factor mode): an input of >= big_numel elements starts its all-gather right
        away -- it is known a forward-tail and a backward-head earlier than the layer's output gradient."""
        p = self._fkeys.get(weight.data_ptr())
        if p is None or not self.factor or x.numel() < self.big_numel or not x.is_contiguous():
            return
        bufs = self._factor_bufs(p, x.shape[0], x.shape[1], weight.shape[0], x)
        if self.reserve_cus and not self._reserved:
            self._set_budget(256 - self.reserve_cus)
            self._reserved = True
        with torch.no_grad():      # called from the forward, in grad mode; the waits happen in the backward / optimizer (no grad): gloo splits
            work = self._all_gather(bufs[0], x.detach())      # the output into views, which must be made and written in ONE grad mode
        self._handles.append(work)
        self._xwork[p] = (work, x.data_ptr(), x.shape[0])

    def linear_factors(self, weight, x, dy):
        """Called by ``ops.Linear.backward`` (on the backward's stream) for a weight registered in FACTOR_SYNC: start the all-gathers
        of the layer's input ``x`` [m, k] and output gradient ``dy`` [m, n] over the batch dimension and return True -- the caller
        then forms NO local weight gradient (``weight.grad`` stays None; the optimizer computes the global-batch gradient from
        ``take_factors``).  False: not in factor mode for this tensor, the caller proceeds as usual."""
        p = self._fkeys.get(weight.data_ptr())
        if p is None or not self.factor:
            return False
        if p in self._factors:
            # the gather buffers of p are persistent and its earlier factors may still be on the links: a second backward through the
            # layer before the optimizer has taken them (gradient accumulation, a shared weight, retain_graph) would overwrite them
            raise RuntimeError("GradSync(factor_linear=True): a registered Linear layer ran a second backward before the optimizer step; "
                               "step between the backwards, or use the all-reduce / sharded modes, which accumulate into .grad")
        m, w = x.shape[0], self.world
        x, dy = x.contiguous(), dy.contiguous()
        bufs = self._factor_bufs(p, m, x.shape[1], dy.shape[1], x)
        if self.reserve_cus and not self._reserved:
            self._set_budget(256 - self.reserve_cus)
            self._reserved = True
        early = self._xwork.pop(p, None)
        if early is not None and early[1] == x.data_ptr() and early[2] == m:      # this x is already on the links (linear_input)
            self.early_input_gathers += 1
            works = [early[0], self._all_gather(bufs[1], dy)]
            self._handles.append(works[1])
        else:
            works = [self._all_gather(bufs[0], x), self._all_gather(bufs[1], dy)]
            self._handles.extend(works)
        self._factors[p] = Factors(works, bufs[0], bufs[1], w * m)
        if self.on_factors is not None:
            self.on_factors(p)
        return True

    def has_factors(self, p):
        return p in self._factors

    def take_factors(self, p):
        """The gathered factors of ``p`` from the latest backward (None: p did not travel as factors); the caller waits for
        ``works`` on the stream that will read them."""
        return self._factors.pop(p, None)

    @property
    def grad_scale(self):
        return 1.0 / self.world

    # ---- the collectives (one place each: a backend without one of them can be served by overriding these) ----------------
    def _all_reduce(self, t, async_op=True):
        return dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.group, async_op=async_op)

    def _reduce_scatter(self, out, inp):
        return dist.reduce_scatter_tensor(out, inp, op=dist.ReduceOp.SUM, group=self.group, async_op=True)

    def _all_gather(self, out, inp):
        return dist.all_gather_into_tensor(out, inp, group=self.group, async_op=True)

    def shardable(self, p):
        """Whether ``p`` travels as reduce-scatter + all-gather: shard mode, a big contiguous tensor that splits evenly."""
        return (self.shard and p.numel() >= self.big_numel and p.numel() % (4 * self.shard_world) == 0 and p.data.is_contiguous()
                and (p.grad is None or p.grad.is_contiguous()))

    def _pieces_of(self, n):
        return [(off, min(off + self.chunk_numel, n)) for off in range(0, n, self.chunk_numel)]

    def _reduce_big(self, p):
        """Start the reduction of one big gradient: reduce-scatter per piece (shard mode) or all-reduce per piece."""
        flat = p.grad.view(-1) if p.grad.is_contiguous() else None
        if flat is not None and self.shardable(p):
            w, r = self.shard_world, self.rank
            gs = self._gshard.get(p)
            if not self.simulate_world and (gs is None or gs.numel() != flat.numel() // w or gs.device != flat.device):
                gs = self._gshard[p] = torch.empty(flat.numel() // w, device=flat.device, dtype=flat.dtype)
            pflat = p.data.view(-1)
            shards = []
            for k, (a, b) in enumerate(self._pieces_of(flat.numel())):
                n = (b - a) // w
                lo = a + r * n
                if self.simulate_world:          # rank 0's slice of the LOCAL gradient: no communicator to sum over
                    work, out = None, flat[lo:lo + n]
                else:
                    out = gs[a // w:a // w + n]
                    work = self._reduce_scatter(out, flat[a:b])
                    self._handles.append(work)
                shards.append(Shard(work, k, lo, lo + n, a, b, pflat[lo:lo + n], out))
            self._shards[p] = shards
            self._by_param.pop(p, None)
            owner = self._owner.get(p)
            if owner is not None and p not in self._pre_hooked:
                self._hooks.append(owner.register_forward_pre_hook(lambda _m, _a, q=p: self.wait_param_gather(q)))
                self._pre_hooked.add(p)
            return
        self._shards.pop(p, None)
        if flat is not None:
            pieces = []
            for a, b in self._pieces_of(flat.numel()):
                work = self._all_reduce(flat[a:b])
                self._handles.append(work)
                pieces.append((work, a, b - a))
            self._by_param[p] = pieces
        else:
            work = self._all_reduce(p.grad)
            self._handles.append(work)
            self._by_param[p] = [(work, 0, p.grad.numel())]

    def _on_grad(self, p):
        if p.grad is None:
            return
        big = p.grad.numel() >= self.big_numel
        if big and self.reserve_cus and not self._reserved:
            self._set_budget(256 - self.reserve_cus)
            self._reserved = True
        if big and (self.active or self.shardable(p)):
            self._reduce_big(p)
        elif self.active:
            self._small.append(p)
            # every reduction started so far has finished and no all-gather is on the links: the conv grids get their compute
            # units back for the rest of the backward (config 4: the big gradients are done tens of milliseconds before it ends)
            if self._reserved and not self._gathers and all(h.is_completed() for h in self._handles):
                self._set_budget(256)
                self._reserved = False

    def pieces(self, p):
        """[(work, offset, numel), ...] of the in-flight all-reduce of ``p`` in issue order (None: p went the small-tensor way,
        or travels as shards)."""
        return self._by_param.get(p)

    def shards(self, p):
        """[Shard, ...] of ``p`` from the latest backward (shard mode, big tensors), None otherwise.  Still valid after
        ``finish()``: the optimizer may run after it."""
        return self._shards.get(p) if self.shard else None

    def gather_shard(self, p, shard):
        """All-gather the updated slice ``shard.param`` into its piece of ``p``, in place and asynchronously, issued behind the
        CURRENT stream's work (the optimizer pass that wrote the slice).  Whoever reads ``p`` next waits through PARAM_WAITS."""
        if self.simulate_world or not self.active:
            return
        flat = p.data.view(-1)
        if self.reserve_cus and not self._reserved:      # the gathers run under the NEXT forward: RCCL keeps its compute units until the last one is waited for
            self._set_budget(256 - self.reserve_cus)
            self._reserved = True
        work = self._all_gather(flat[shard.piece_lo:shard.piece_hi], shard.param)
        self._gathers.setdefault(p, []).append(work)
        PARAM_WAITS[p.data_ptr()] = lambda q=p: self.wait_param_gather(q)

    def wait_param_gather(self, p):
        PARAM_WAITS.pop(p.data_ptr(), None)
        works = self._gathers.pop(p, ())
        for work in works:
            work.wait()
        if works and p.is_cuda:
            # the wait above orders the CURRENT stream behind the gathers.  The first toucher may be a side stream (the weight-pack
            # stream, the optimizer's); the entry is gone after this call, so the main stream has to be ordered here as well
            cur, main = torch.cuda.current_stream(p.device), torch.cuda.default_stream(p.device)
            if cur != main:
                main.wait_event(cur.record_event())
        if self._reserved and not self._gathers and not self._handles:
            self._set_budget(256)                        # the last all-gather has been waited for: the whole GPU for the kernels behind it
            self._reserved = False

    def wait_gathers(self):
        """Make the current stream wait for every all-gather in flight: before validation, ``state_dict()``, a checkpoint, or
        anything else that reads parameters outside the hot path's kernels."""
        for p in list(self._gathers):
            self.wait_param_gather(p)

    def wait_param(self, p):
        """Make the CURRENT stream wait for the whole reduction of ``p`` (no-op when p went the small-tensor way)."""
        for work, _, _ in self._by_param.get(p) or ():
            work.wait()
        for sh in self._shards.get(p) or ():
            if sh.work is not None:
                sh.work.wait()

    def finish(self):
        """Reduce the small gradients in one message and wait for everything in flight."""
        if self.active:
            # parameters switched to requires_grad by hand since the last refresh(): their hooks did not exist during this
            # backward, so their gradients are still local -- reduce them here (with the small tensors, or by themselves when
            # big: as shards in shard mode, so that the optimizer sees one layout per tensor for good) and hook them
            late = [p for p in self.params if p.requires_grad and p not in self._hooked and p.grad is not None]
            for p in late:
                if p.grad.numel() >= self.big_numel:
                    self._reduce_big(p)
                else:
                    self._small.append(p)
            if late:
                self.refresh()
        elif self.shard:
            late = [p for p in self.params if p.requires_grad and p not in self._hooked and p.grad is not None]
            for p in late:
                if self.shardable(p):
                    self._reduce_big(p)
            if late:
                self.refresh()
        if self.active and self._small:
            flat = torch.cat([p.grad.reshape(-1) for p in self._small])
            self._all_reduce(flat, async_op=False)
            off = 0
            for p in self._small:
                n = p.grad.numel()
                p.grad.copy_(flat[off:off + n].view_as(p.grad))
                off += n
        for h in self._handles:
            h.wait()
        self._handles, self._small, self._by_param = [], [], {}
        if self._reserved and not self._gathers:         # shard mode: the all-gathers keep RCCL's compute units (wait_param_gather returns them)
            self._set_budget(256)
            self._reserved = False

    @staticmethod
    def _set_budget(cus):
        from . import _lib
        _lib.check(_lib.lib().dd_set_cu_budget(cus), "dd_set_cu_budget")

    def remove(self):
        self.wait_gathers()
        for h in self._hooks:
            h.remove()
        self._hooks = []
        self._hooked = set()
        self._pre_hooked = set()
        self._shards, self._gshard = {}, {}
        for key in list(self._fkeys):
            if FACTOR_SYNC.get(key) is self:
                del FACTOR_SYNC[key]
        self._fkeys, self._factors, self._fbuf, self._xwork = {}, {}, {}, {}
        self.factor = False
        self.active = False
        self.shard = False
