"""``TrainStep``: the step ``bench.py`` measures, as one callable over any of the drop-in modules.

The reference hands its modules to ``pytorch_lightning.Trainer.fit`` (autoencoder.py:192-193, submit.py:40-46), which per batch does
``optimizer.zero_grad(); out = model.training_step(batch, i); out['loss'].backward(); optimizer.step()`` with the optimizer (and
scheduler) ``configure_optimizers()`` returned -- ``Adam(lr)`` for every module, plus ``ReduceLROnPlateau(patience=10)`` on the
validation loss for the road-map modules (roadmap_bce_v2.py:154-157) -- and, under
``--distributed_backend ddp``, averages gradients across processes.  ``configure_optimizers()`` of the drop-in modules still returns
exactly that (torch.optim.Adam): a caller with its own loop keeps working.  This helper is the SAME step arranged for the MI355X:

  * ``optim.HipAdam`` (one fused pass per tensor, torch.optim.Adam's arithmetic) with the pass of the big tensors on a side stream
    beside the MFMA-bound stretch of the backward (``overlap_with_backward``); the big ``nn.Linear`` weights (encoder fc1, head,
    decoder fc2) take a rank-B update: their gradient is formed inside the Adam pass from the layer's input and output gradient and
    never written (``fuse_linear_wgrad``; ``weight.grad`` of those layers stays None -- pass ``fuse_linear_wgrad=False`` to keep it);
  * ``ddp.GradSync`` when ``torch.distributed`` is initialised: per-tensor asynchronous all-reduce from autograd hooks, or -- with
    ``shard_optimizer=True`` -- reduce-scatter, Adam on the owned 1/N, in-place all-gather under the next forward; or -- with
    ``factor_linear=True``, for 2-4 ranks -- the big Linear layers send their factors (input, output gradient) instead of their
    weight gradients and every rank forms the global-batch gradient itself (ddp.py);
  * frozen feature extractors (``self.ae.freeze()`` in the fine-tuning modules) are handled: both objects are built over the model
    as constructed and re-arm when ``training_step`` unfreezes it (lightning.on_unfreeze);
  * ``validation_epoch_end(val_loss)`` steps ``ReduceLROnPlateau`` the way Lightning does for a scheduler returned beside the
    optimizer (monitor = ``val_loss``), for the modules whose ``configure_optimizers`` has one.

It is a helper, not a Trainer: no data loading, logging, checkpoint policy or CLI (out of scope).
"""
import torch
import torch.distributed as dist

from .ddp import GradSync
from .optim import HipAdam


class TrainStep:
    def __init__(self, model, lr=None, adam_overlap="auto", shard_optimizer=False, reserve_cus=None, process_group=None,
                 force_collectives=False, simulate_world=0, scheduler="auto", big_numel=1 << 20, chunk_numel=1 << 25, factor_linear=False,
                 fuse_linear_wgrad=True, passes_last="auto"):
        self.model = model
        hp = getattr(model, "hparams", None)
        if lr is None:
            lr = getattr(hp, "learning_rate", None)
            if lr is None:
                raise ValueError("TrainStep: no lr given and model.hparams has no learning_rate")
        distributed = dist.is_available() and dist.is_initialized()
        world = dist.get_world_size(process_group) if distributed else 1
        comm = distributed and (world > 1 or force_collectives)
        if reserve_cus is None:      # RCCL's workgroups need LDS the resident conv grids do not leave free
            reserve_cus = 16 if comm and next(model.parameters()).is_cuda and dist.get_backend(process_group) == "nccl" else 0
        # over the model AS CONSTRUCTED (extractor possibly frozen): every parameter is in the optimizer from the start, as with
        # the reference's Adam(self.parameters()); a parameter without a gradient is skipped until it has one
        self.optimizer = self._make_optimizer(model.parameters(), lr)
        self.sync = GradSync(model, process_group=process_group, big_numel=big_numel, chunk_numel=chunk_numel, reserve_cus=reserve_cus,
                             force_collectives=force_collectives, shard_optimizer=shard_optimizer, simulate_world=simulate_world,
                             factor_linear=factor_linear)
        self.optimizer.attach(self.sync)
        if adam_overlap == "auto":
            # fp32 models: the passes of the big tensors ride beside the MFMA-bound c2 weight gradient.  A bf16 encoder has no MFMA-bound
            # stretch -- its conv kernels are HBM-bound, and an 11.6 GB optimizer stream beside them only time-shares the same HBM
            # (config 5: the c2 weight gradient at 0.19-0.28 of the HBM peak beside it, 0.62 alone; same step either way,
            # profiles/r05_config5_adam_overlap_ab.txt): there the optimizer runs after the backward, four workgroups per CU
            adam_overlap = not any(getattr(m, "precision", None) == "bf16" for m in model.modules())
        self.overlap = bool(adam_overlap)
        if not self.overlap and next(model.parameters()).is_cuda:
            from . import _lib
            _lib.check(_lib.lib().dd_set_adam_blocks_per_cu(4), "dd_set_adam_blocks_per_cu")
        if self.overlap:
            self.optimizer.overlap_with_backward(big_numel=big_numel, grad_scale=self.sync.grad_scale,
                                                 grad_sync=self.sync if (self.sync.active or self.sync.shard) else None)
        # rank-B mode: the weight gradient of the big Linear layers is formed inside their Adam pass, never written (optim.py); the
        # optimizer declines by itself where the gradient has to travel as a tensor (all-reduce / sharded GradSync)
        self.fused = self.optimizer.fuse_linear_wgrad(model, min_numel=big_numel) if fuse_linear_wgrad else []
        # rank-B passes only fit beside c2's weight gradient: that kernel goes last, the data gradient runs by itself in front of it
        # (HipAdam.passes_last).  Under an all-reduce / sharded GradSync the passes are the plain 48-register kernel, ready at the top of
        # the backward, and keep the old order (they run beside both conv kernels).
        if passes_last == "auto":
            passes_last = bool(self.fused) and not ((self.sync.active or self.sync.shard) and not self.sync.factor)
        if self.overlap and passes_last:
            self.optimizer.passes_last(True)
        if scheduler == "auto":      # the modules that return ([optimizer], [scheduler]) from configure_optimizers
            scheduler = self._reference_has_scheduler(model)
        self.scheduler = torch.optim.lr_scheduler.ReduceLROnPlateau(self.optimizer, patience=10) if scheduler else None
        self.last = None
        self._one = None

    @staticmethod
    def _make_optimizer(params, lr):
        return HipAdam(params, lr=lr)

    @staticmethod
    def _reference_has_scheduler(model):
        """Whether the module's own ``configure_optimizers()`` returns ``([optimizer], [scheduler])`` (the road-map modules,
        roadmap_bce_v2.py:154-157) rather than a bare optimizer (autoencoder.py:119-120, spatial_w_rm.py:166-167)."""
        make = getattr(model, "configure_optimizers", None)
        if make is None:
            return False
        out = make()                                        # a torch.optim.Adam over the parameters: no state until it steps
        return isinstance(out, (tuple, list)) and len(out) == 2 and bool(out[1])

    def __call__(self, batch, batch_idx):
        """One training step; returns ``training_step``'s dict (``out['loss']`` is the step's loss, still on the device)."""
        self.model.zero_grad(set_to_none=True)
        out = self.model.training_step(batch, batch_idx)
        loss = out["loss"]
        if self._one is None or self._one.device != loss.device or self._one.dtype != loss.dtype:
            self._one = torch.ones((), device=loss.device, dtype=loss.dtype)
        loss.backward(self._one)                 # the root gradient is a kept tensor: no ones_like fill launch per step
        self.sync.finish()
        self.optimizer.step(grad_scale=self.sync.grad_scale)
        self.last = out
        return out

    def sync_params(self):
        """Sharded optimizer: wait for the all-gathers still writing into parameters.  Call before validation, ``state_dict()`` or
        a checkpoint -- anything that reads parameters other than through the next ``training_step``."""
        self.sync.wait_gathers()

    def validation_epoch_end(self, val_loss):
        """``scheduler.step(val_loss)`` once per validation epoch (Lightning 0.7.5's handling of ReduceLROnPlateau)."""
        self.sync_params()
        if self.scheduler is not None:
            self.scheduler.step(float(val_loss))

    @property
    def lr(self):
        return self.optimizer.param_groups[0]["lr"]

    def close(self):
        self.sync_params()
        self.optimizer.close()
        self.sync.remove()
