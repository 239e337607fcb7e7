"""CPU oracle for the segmentation training hot path.  TEST INFRASTRUCTURE ONLY.

This package is the checker, never the product: only ``tests/``,
``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import it.
Nothing under ``segmentation_training_pipeline_amd/`` or ``segmentation_pipeline/``
imports it, and the product path raises when the HIP extension is missing.

PARITY UNPINNED.  The reference (musket-ml/segmentation_training_pipeline) keeps
every FLOP of this path in un-vendored third-party packages that are absent from
``/root/reference`` and cannot be installed here (no network):

* ``musket_core`` (unpinned, reference ``setup.py:24``) - fit loop, datasets,
  losses (``segmentation_pipeline/segmentation.py:15-22`` registers them by name),
* ``segmentation_models==0.2.1`` (``requires.txt:15``) - Unet/FPN/Linknet/PSPNet graphs
  (call site ``segmentation_pipeline/segmentation.py:113,155``),
* ``classification_models`` (``segmentation_pipeline/segmentation.py:5,114``) - ResNet/VGG encoders,
* ``keras>=2.2.4`` / ``tensorflow==1.15`` (``requires.txt:8,12``) - layers, BN, losses, optimizers,
* ``imgaug==0.3.0`` (``requires.txt:11``) - augmentation
  (catalogue ``segmentation_pipeline/schemas/augmenters.raml:43-133``).

The reference tree holds no tests, fixtures or golden vectors for this path
(only ``.travis.yml:15-16`` runs a bare ``pytest`` on another repo), so this oracle
restates the *published* algorithms of those packages (Keras 2.2.4 layer/optimizer
semantics, segmentation_models 0.2.1 U-Net graph, classification_models ResNet
graph) and is anchored on the reference's call sites and schema defaults
(``segmentation_pipeline/schemas/segmentation.raml:26-249``).  The one piece of the
reference that *does* import here (``segmentation_pipeline/impl/rle.py``) is pinned
by golden vectors generated from it (``tests/golden/make_rle_golden.py``).

Arithmetic is plain PyTorch-CPU fp32 (conv/pool/interpolate primitives + autograd),
cross-checked by the naive numpy loops in :mod:`oracle.np_ops` on small shapes.
"""
