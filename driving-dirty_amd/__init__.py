"""driving-dirty on MI355X: the multi-camera -> BEV training hot path, hand-written for gfx950.

Layout:
  csrc/            HIP kernels + the C-ABI shared library (``include/dd_hotpath.h``)
  _lib.py          ctypes binding of that library (fails loudly when it is missing)
  ops.py           torch.autograd.Function shims over the C-ABI (own all tensors; kernels never allocate)
  components.py    Encoder / Decoder / DenseBlock with the reference's names and state_dict keys
  lightning.py     minimal self-hosted LightningModule surface (the image has no pytorch_lightning)
  autoencoder.py   BasicAE;  roadmap.py  RoadMapBCE / RoadMap;  ddp.py  bucketed RCCL gradient all-reduce
  synth.py         closed-form synthetic tensors for tests and the bench
"""
__version__ = "0.1.0"
