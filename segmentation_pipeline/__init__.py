"""Drop-in surface of musket-ml/segmentation_training_pipeline backed by the MI355X-native
HIP training path in :mod:`segmentation_training_pipeline_amd`.

Same import paths as the reference (``README.md:120-126``):

    from segmentation_pipeline.impl.datasets import SimplePNGMaskDataSet
    from segmentation_pipeline import segmentation
    cfg = segmentation.parse("config.yaml"); cfg.fit(ds)
"""
