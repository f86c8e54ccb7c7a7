"""CPU oracle for the multi-camera -> BEV training hot path.

TEST INFRASTRUCTURE ONLY.  Nothing in the product package may import this
directory: only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline``
leg of ``bench.py`` do, and only as the checker.

What it is: a plain-PyTorch (CPU, fp32 or fp64) restatement of the reference's
arithmetic for the path, module-for-module with identical ``state_dict`` keys:

* ``oracle.ae_parts``       <- reference ``src/autoencoder/components.py:6-109``
* ``oracle.spatial_parts``  <- reference ``src/bounding_box_model/spatial_bb/components.py:6-170``
* ``oracle.steps``          <- the LightningModule glue: ``src/autoencoder/autoencoder.py:53-120``,
  ``src/roadmap_model/roadmap_bce_v2.py:53-157``, ``src/roadmap_model/roadmap_pretrain_ae.py:67-110``,
  ``src/bounding_box_model/spatial_bb/spatial_w_rm.py:54-154``, ``src/utils/helper.py:22-23,74-77``

How it is pinned: the reference has no tests and no golden vectors (SURVEY.md
section 4), so parity is pinned by fixtures generated HERE from the imported
reference modules (``tests/golden/make_golden.py`` imports
``src.autoencoder.components`` and ``src.bounding_box_model.spatial_bb.components``
from /root/reference, which depend on torch only) and committed under
``tests/golden/*.npz``.  ``tests/test_oracle_golden.py`` checks this restatement
against those fixtures.  The Lightning-level glue files cannot be imported in
this image (pytorch_lightning / torchvision / test_tube are absent: an ordinary
ModuleNotFoundError, not a permission denial), so the glue in ``oracle.steps``
is a restatement read from the source text; its arithmetic reduces to the
pinned component modules plus index permutations that are pinned by
closed-form fixtures.
"""
