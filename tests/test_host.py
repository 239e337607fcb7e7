"""CPU tests of the host-side mirror of the reference interface (no GPU compute)."""
import os

import numpy as np
import torch
import pytest
import yaml

from oracle import augment as oaug
from segmentation_pipeline import segmentation
from segmentation_pipeline.impl import datasets
from segmentation_training_pipeline_amd import augment, distributed, graph, nets, pipeline
from segmentation_training_pipeline_amd.backend import parse_loss

BASE = {"architecture": "Unet", "backbone": "ResNet34", "classes": 1, "activation": "sigmoid", "shape": [64, 64, 3],
        "optimizer": "Adam", "batch": 2, "loss": "binary_crossentropy+0.1*dice_loss", "stages": [{"epochs": 1}],
        "augmentation": {"Fliplr": 0.5, "Flipud": 0.5}, "metrics": ["binary_accuracy", "iou"],
        "primary_metric": "val_binary_accuracy", "freeze_encoder": True,
        "callbacks": {"EarlyStopping": {"patience": 3, "monitor": "val_binary_accuracy"},
                      "ReduceLROnPlateau": {"patience": 2, "factor": 0.5, "monitor": "val_binary_accuracy", "cooldown": 1}}}


def write_cfg(tmp_path, **over):
    c = dict(BASE)
    c.update(over)
    p = str(tmp_path / "config.yaml")
    with open(p, "w") as f:
        yaml.safe_dump(c, f)
    return p


def test_parse_sets_path_and_createnet_kwarg_rules(tmp_path):
    cfg = segmentation.parse(write_cfg(tmp_path))
    assert cfg.path.endswith("config.yaml") and cfg.batch == 2 and cfg.folds_count == 5
    m = cfg.createNet()
    # alias renaming + lower-casing + pipeline-level keys filtered out (reference segmentation.py:119-129)
    assert m.backbone_name == "resnet34" and m.input_shape == (64, 64, 3) and m.freeze_encoder is True
    assert cfg.all["backbone"] == "resnet34"
    cfg2 = segmentation.parse(write_cfg(tmp_path, activation="none"))
    assert cfg2.createNet().activation is None                       # reference :97-101
    cfg3 = segmentation.parse(write_cfg(tmp_path, crops=2))
    assert cfg3.createNet().input_shape == (32, 32, 3)               # reference :135-136


def test_unknown_names_raise_the_reference_errors(tmp_path, capsys):
    cfg4 = segmentation.parse(write_cfg(tmp_path, architecture="Linknet", backbone="resnet50"))
    assert cfg4.createNet().architecture == "Linknet" and cfg4.createNet().backbone_name == "resnet50"
    cfg6 = segmentation.parse(write_cfg(tmp_path, architecture="FPN", backbone="resnet50", classes=3, activation="softmax"))
    assert cfg6.createNet().architecture == "FPN" and cfg6.createNet().classes == 3
    cfg7 = segmentation.parse(write_cfg(tmp_path, architecture="PSPNet", backbone="resnet101", classes=20, activation="softmax", shape=[96, 96, 3]))
    assert cfg7.createNet().architecture == "PSPNet" and cfg7.createNet().input_shape == (96, 96, 3)
    cfg5 = segmentation.parse(write_cfg(tmp_path, decoder_block_type="transpose"))         # schemas/segmentation.raml:166-169
    assert cfg5.createNet().decoder_block_type == "transpose"
    with pytest.raises(ValueError, match="Unknown architecture"):
        segmentation.parse(write_cfg(tmp_path, architecture="Nope")).createNet()
    with pytest.raises(ValueError, match="Unknown backbone"):
        segmentation.parse(write_cfg(tmp_path, backbone="vgg11")).createNet()
    assert "Known backbones" in capsys.readouterr().out


def test_custom_models_registry_takes_precedence(tmp_path):
    seen = {}

    def mine(backbone_name="x", input_shape=None, classes=1):
        seen.update(backbone_name=backbone_name, input_shape=input_shape, classes=classes)
        return "model"
    segmentation.custom_models["Mine"] = mine
    try:
        assert segmentation.parse(write_cfg(tmp_path, architecture="Mine")).createNet() == "model"
        assert seen == {"backbone_name": "resnet34", "input_shape": [64, 64, 3], "classes": 1}
    finally:
        del segmentation.custom_models["Mine"]


def test_stages_and_unfreeze(tmp_path):
    cfg = segmentation.parse(write_cfg(tmp_path, stages=[{"epochs": 2}, {"epochs": 3, "unfreeze_encoder": True, "lr": 1e-4,
                                                                       "loss": "dice_loss", "negatives": "none"}]))
    assert [s.epochs for s in cfg.stages] == [2, 3]
    assert isinstance(cfg.stages[0], segmentation.SegmentationStage)
    m = cfg.createNet()
    cfg.stages[1].unfreeze(m)
    assert m.freeze_encoder is False and cfg.stages[1].negatives == "none"
    assert cfg.weightsPath(0, 1).endswith(os.path.join("weights", "best-0.1.weights"))


def test_loss_grammar():
    assert parse_loss("binary_crossentropy+0.1*dice_loss") == (1.0, 0.1)
    assert parse_loss("dice_loss") == (0.0, 1.0)
    assert parse_loss("lovasz_loss") == (0.0, 0.0, 0.0, 0.0, 0.0, 1.0)
    assert parse_loss("binary_crossentropy+0.5*iou_loss") == (1.0, 0.0, 0.5, 0.0, 0.0, 0.0)
    with pytest.raises(ValueError):
        parse_loss("hinge_loss")
    with pytest.raises(ValueError):
        parse_loss("lovasz_loss", classes=3)


def test_kfold_is_deterministic_disjoint_and_respects_test_split():
    class DS:
        def __len__(self): return 23
        def isPositive(self, i): return i % 3 != 0
    a = pipeline.KFoldedDataSet(DS(), range(23), 5, 33, 0.2)
    b = pipeline.KFoldedDataSet(DS(), range(23), 5, 33, 0.2)
    assert len(a.test_indexes) == 5
    allv = np.concatenate([v for _, v in a.folds])
    assert sorted(allv.tolist() + a.test_indexes.tolist()) == list(range(23))
    for (t1, v1), (t2, v2) in zip(a.folds, b.folds):
        assert t1.tolist() == t2.tolist() and v1.tolist() == v2.tolist() and not set(t1) & set(v1)
    pos = a.sampledIndexes(0, True, "none")
    assert all(i % 3 != 0 for i in pos)
    one = a.sampledIndexes(0, True, 1)
    assert len(one) <= 2 * len(pos)


def test_callbacks_lr_policies():
    class M:
        lr = 1.0
        def get_lr(self): return self.lr
        def set_lr(self, v): self.lr = v
    class T:
        model = M()
    t = T()
    es = pipeline.EarlyStopping(patience=2, monitor="val_loss")
    for v in (1.0, 0.9, 0.95, 0.96):
        es.on_epoch_end(t, 0, {"val_loss": v})
    assert es.stop
    rl = pipeline.ReduceLROnPlateau(patience=1, factor=0.5, monitor="val_acc", cooldown=0)
    for v in (0.5, 0.4, 0.3):
        rl.on_epoch_end(t, 0, {"val_acc": v})
    assert t.model.lr == 0.25
    c = pipeline.CyclicLR(base_lr=0.1, max_lr=0.5, step_size=4)
    lrs = []
    for _ in range(8):
        c.on_batch_end(t)
        lrs.append(t.model.lr)
    assert abs(max(lrs) - 0.5) < 1e-9 and abs(lrs[-1] - 0.1) < 1e-9
    with pytest.raises(ValueError):
        pipeline.make_callbacks({"TensorBoard": {}})


def test_augment_matrices_match_the_oracle_and_reject_unknown():
    rng = np.random.RandomState(0)
    # augmenters compose in the listed order (imgaug Sequential): flip first, then the affine about the centre
    spec = [{"Fliplr": 1.0}, {"Affine": {"scale": 1.3, "translate_percent": {"x": 0.1, "y": -0.05}, "rotate": 12, "shear": -7}},
            {"Add": 7}, {"Multiply": 1.1}]
    prm = augment.sample_batch(spec, rng, 1, 40, 52, (32, 48))
    assert prm.shape == (1, augment.AUG_RECORD) == (1, oaug.AUG_RECORD)
    ref = oaug.compose(40, 52, [("fliplr",), ("affine", 1.3, (0.1, -0.05), 12.0, -7.0)], (32, 48))
    np.testing.assert_allclose(prm[0, :6].reshape(2, 3), ref[:2], rtol=1e-6, atol=1e-5)
    assert list(prm[0, 6:9]) == [7, 7, 7] and np.allclose(prm[0, 9:12], 1.1, atol=1e-6) and prm[0, 12] == 0
    ident = augment.identity_batch(2, 64, 64, (64, 64))
    np.testing.assert_allclose(ident[0, :6], [1, 0, 0, 0, 1, 0], atol=1e-12)
    # crop / pad family: deterministic cases against the oracle's canvas algebra
    prm = augment.sample_batch([{"CropToFixedSize": {"width": 52, "height": 40}}, {"PadToFixedSize": {"width": 52, "height": 40}},
                                {"Pad": {"px": [3, 1, 2, 4]}}, {"Flipud": 1.0}], rng, 1, 40, 52, (20, 26))
    ref = oaug.compose(40, 52, [("crop", -3, -4, 45, 57), ("resize", 40, 52), ("flipud",)], (20, 26))
    np.testing.assert_allclose(prm[0, :6].reshape(2, 3), ref[:2], rtol=1e-6, atol=1e-5)
    prm = augment.sample_batch([{"CropAndPad": {"percent": -0.1}}], rng, 1, 40, 50, (40, 50))      # crop 10 % per side, keep size
    ref = oaug.compose(40, 50, [("crop", 4, 5, 32, 40), ("resize", 40, 50)], (40, 50))
    np.testing.assert_allclose(prm[0, :6].reshape(2, 3), ref[:2], rtol=1e-6, atol=1e-5)
    # Rotate90 (examples/people/ds_1.yaml:6): exactly the four quarter turns of np.rot90
    sq = np.random.RandomState(1).randint(0, 256, (1, 8, 8, 3)).astype(np.uint8)
    turns = set()
    for sd in range(24):
        m = augment.sample_batch([{"Rotate90": True}], np.random.RandomState(sd), 1, 8, 8, (8, 8))
        out, _ = oaug.warp_u8(sq, None, m, (8, 8))
        ks = [k for k in range(4) if np.array_equal(out[0], np.rot90(sq[0], k))]
        assert len(ks) == 1
        turns.add(ks[0])
    assert turns == {0, 1, 2, 3}
    # callbacks may monitor Keras' function-name spelling of a metric (examples/people/ds_1.yaml:19-27)
    assert pipeline.log_value({"val_iou": 0.5, "loss": 1.0}, "val_iou_coef") == 0.5 and pipeline.log_value({"loss": 1.0}, "loss") == 1.0
    assert pipeline.log_value({"dice_loss": 0.25}, "dice_coef_loss") == 0.25 and pipeline.log_value({}, "val_iou_coef") is None
    # point operations land in the record (include/stp_hip.h layout)
    prm = augment.sample_batch([{"Invert": 1.0}, {"Grayscale": {"alpha": 0.5}}, {"AdditiveGaussianNoise": {"scale": 12.75, "per_channel": True}},
                                {"Dropout": {"p": 0.25}}, {"AddElementwise": [-10, 10]}, {"MultiplyElementwise": {"mul": [0.9, 1.1], "per_channel": True}},
                                {"Add": {"value": [-5, 5], "per_channel": True}}], rng, 1, 8, 8, (8, 8))[0]
    F = augment
    assert int(prm[12]) == F.F_INVERT | F.F_NOISE_PC | F.F_ADDE | F.F_MULE | F.F_MULE_PC
    assert prm[13] == 128 and prm[14] == round(12.75 * 65536 / 147.8) and prm[15] == (1 << 22)
    assert (prm[16], prm[17]) == (-10, 10) and np.allclose(prm[18:20], [0.9, 1.1]) and 0 <= prm[20] < (1 << 24)
    assert len(set(prm[6:9])) > 1                                   # per-channel Add drew three values
    picks = {int(augment.sample_batch([{"OneOf": [{"Invert": 1.0}, {"Dropout": 0.5}]}], rng, 1, 8, 8, (8, 8))[0, 12]) for _ in range(20)}
    assert picks == {0, 1}                                          # OneOf takes exactly one child
    for bad in ("PiecewiseAffine", "ElasticTransformation", "BackgroundReplacer"):     # (not in the MERGED single-pass sampler)
        with pytest.raises(ValueError, match=bad):
            augment.sample_batch([{bad: 1.0}], rng, 1, 8, 8, (8, 8))
    # neighbourhood filters come back as a second record set (stp_filter_u8), at most MAX_FILTERS per image
    prm, filt = augment.sample_batch_ex([{"OneOf": [{"GaussianBlur": {"sigma": [0.5, 1.0]}}, {"MedianBlur": {"k": 3}}]}, {"Sharpen": 0.5}],
                                        rng, 3, 8, 8, (8, 8))
    assert prm.shape == (3, 24) and filt.shape == (2, 3, augment.FILTER_RECORD) and set(filt[1, :, 0]) == {3}
    with pytest.raises(ValueError, match="use sample_batch_ex"):
        augment.sample_batch([{"AverageBlur": 3}], rng, 1, 8, 8, (8, 8))
    with pytest.raises(ValueError, match="more than 2"):
        augment.sample_batch_ex([{"AverageBlur": 3}] * 3, rng, 1, 8, 8, (8, 8))
    assert pipeline.aug_list({"Fliplr": 0.5, "Flipud": 0.5}) == [{"Fliplr": 0.5}, {"Flipud": 0.5}]
    # DirectedEdgeDetect (schemas/augmenters.raml:118-121): neighbours weighted towards the direction, summing to -1 around a centre 1
    _, f = augment.sample_batch_ex([{"DirectedEdgeDetect": {"alpha": 1.0, "direction": 0.0}}], rng, 1, 8, 8, (8, 8))
    k = f[0, 0, 4:13].reshape(3, 3).astype(np.float64) / 16384.0
    assert f[0, 0, 0] == 3 and abs(k[1, 1] - 1.0) < 1e-3 and abs(k.sum()) < 1e-3 and k[0, 1] < -0.5 < k[2, 1] <= 0     # direction 0 = "up": the cell above weighs most
    _, f2 = augment.sample_batch_ex([{"DirectedEdgeDetect": {"alpha": 0.5, "direction": 0.25}}], rng, 1, 8, 8, (8, 8))
    k2 = f2[0, 0, 4:13].reshape(3, 3).astype(np.float64) / 16384.0
    assert abs(k2.sum() - 0.5) < 1e-3 and k2[1, 2] < -0.25 < k2[1, 0] <= 0                                                 # a quarter turn: the cell to the right


def test_augmentation_passes_follow_the_listed_order():
    """imgaug Sequential semantics (README.md:247-268): augmenters apply in the YAML order.  One device pass = [warp, point
    operations in the kernel's order, filters]; the sampler opens a new pass exactly where the listed order needs it."""
    rng = np.random.RandomState(4)
    stage = lambda spec, h=40, w=52, out=(40, 52): augment.sample_staged(spec, rng, h, w, out)
    # the common shape (geometry, colour, blur; item size = network size): ONE pass, identical to the merged record
    one = stage([{"Fliplr": 1.0}, {"Affine": {"rotate": 10}}, {"Add": 5}, {"Multiply": 1.1}, {"GaussianBlur": {"sigma": 1.0}}])
    assert len(one) == 1 and len(one[0][1]) == 1 and one[0][2] == (40, 52)
    merged, _ = augment.sample_batch_ex([{"Fliplr": 1.0}, {"Affine": {"rotate": 10}}, {"Add": 5}, {"Multiply": 1.1}], rng, 1, 40, 52, (40, 52))
    np.testing.assert_allclose(one[0][0][:12], merged[0][:12], rtol=1e-6, atol=1e-6)
    # colour BEFORE geometry, blur between two warps, Add after Multiply, a repeated Add: each boundary is a pass
    p = stage([{"Add": 5}, {"Affine": {"rotate": 10}}, {"GaussianBlur": {"sigma": 1.0}}, {"Fliplr": 1.0}, {"Multiply": 1.1}, {"Add": 3}, {"Add": 2}])
    assert len(p) == 5
    assert list(p[0][0][6:9]) == [5, 5, 5] and np.allclose(p[0][0][:6], [1, 0, 0, 0, 1, 0])          # pass 1: Add on the raw pixels
    assert not np.allclose(p[1][0][:6], [1, 0, 0, 0, 1, 0]) and len(p[1][1]) == 1 and p[1][0][6] == 0   # pass 2: rotate, then blur
    assert p[2][0][0] == -1 and np.allclose(p[2][0][9:12], 1.1) and p[2][0][6] == 0                     # pass 3: flip, Multiply
    assert list(p[3][0][6:9]) == [3, 3, 3] and list(p[4][0][6:9]) == [2, 2, 2]                         # passes 4, 5: the Adds (clipped between)
    # the trailing Resize joins the last pass only when that pass is pure geometry
    assert len(stage([{"Fliplr": 1.0}], out=(20, 26))) == 1
    r = stage([{"Fliplr": 1.0}, {"Add": 5}], out=(20, 26))
    assert len(r) == 2 and r[0][2] == (40, 52) and r[1][2] == (20, 26) and r[1][0][6] == 0 and np.isclose(r[1][0][0], 2.0)
    # canvas-changing geometry between passes: the intermediate buffers take the canvas of their pass
    c = stage([{"CropToFixedSize": {"width": 30, "height": 20}}, {"Invert": 1.0}, {"PadToFixedSize": {"width": 64, "height": 48}}], out=(24, 32))
    assert [q[2] for q in c] == [(20, 30), (24, 32)] and int(c[0][0][12]) == augment.F_INVERT
    # three filters: two per pass
    f = stage([{"AverageBlur": 3}] * 3)
    assert [len(q[1]) for q in f] == [2, 1]
    # batches: equal structure -> batch passes, otherwise per image
    bp, per = augment.sample_batch_staged([{"Add": 5}, {"Fliplr": 1.0}], rng, 3, 16, 16, (16, 16))
    assert per is None and len(bp) == 2 and bp[0][0].shape == (3, 24) and bp[1][2] == (16, 16)
    bp, per = augment.sample_batch_staged([{"Fliplr": 1.0}, {"Sometimes": {"p": 0.5, "then_list": [{"Add": 5}, {"Flipud": 1.0}]}}], np.random.RandomState(1), 8, 16, 16, (16, 16))
    assert bp is None and sorted({len(q) for q in per}) == [1, 2]


def test_displacement_augmenters_sample_into_their_own_passes_and_the_oracle_fields_behave():
    """PiecewiseAffine / ElasticTransformation (schemas/augmenters.raml:126-133): a pass carries ONE displacement field after its
    matrix geometry; the oracle's integer fields: zero jitter = zero field, constant jitter = constant field, the elastic
    field scales with alpha and is bounded by it."""
    from oracle import augment as oaug
    rng = np.random.RandomState(3)
    spec = [{"Fliplr": 1.0}, {"PiecewiseAffine": {"scale": [0.02, 0.05]}}, {"Add": 5},
            {"ElasticTransformation": {"alpha": [20, 40], "sigma": [3, 5]}}, {"PiecewiseAffine": {"scale": 0.03, "nb_rows": 3, "nb_cols": 5}}]
    p = augment.sample_staged(spec, rng, 40, 52, (40, 52))
    assert [len(q) for q in p] == [4, 4, 4] and [q[3][0] for q in p] == ["piecewise", "elastic", "piecewise"]
    assert p[0][0][0] == -1 and list(p[0][0][6:9]) == [5, 5, 5] and p[1][0][6] == 0          # flip + field + Add | field | field
    assert p[0][3][1:3] == (4, 4) and p[0][3][3].shape == (32,) and p[2][3][1:3] == (3, 5)
    rec = p[1][3][1]
    assert rec.shape == (augment.ELASTIC_RECORD,) and 20 * 64 <= rec[1] <= 40 * 64 and 12 <= rec[2] <= 20
    assert int(rec[4]) + 2 * int(rec[5:5 + rec[2]].sum()) == 32768                              # exact DC gain
    # scale 0 / alpha 0: no field, no extra pass; the merged sampler refuses fields
    assert [len(q) for q in augment.sample_staged([{"PiecewiseAffine": 0.0}, {"ElasticTransformation": {"alpha": 0}}], rng, 8, 8, (8, 8))] == [3]
    with pytest.raises(ValueError):
        augment.sample_batch_ex([{"PiecewiseAffine": 0.05}], rng, 1, 8, 8, (8, 8))
    with pytest.raises(ValueError):
        augment.sample_staged([{"BackgroundReplacer": {"path": "no-such-folder"}}], rng, 8, 8, (8, 8))
    with pytest.raises(ValueError):
        augment.sample_staged([{"CoarseDropout": 0.1}], rng, 8, 8, (8, 8))
    # a resize after a field is a pass of its own
    r = augment.sample_staged([{"PiecewiseAffine": 0.05}], rng, 40, 52, (20, 26))
    assert [len(q) for q in r] == [4, 3] and r[0][2] == (40, 52) and r[1][2] == (20, 26)
    # batches
    bp, per = augment.sample_batch_staged([{"ElasticTransformation": {"alpha": 30, "sigma": 4}}], rng, 3, 16, 16, (16, 16))
    assert per is None and bp[0][3][0] == "elastic" and bp[0][3][1].shape == (3, augment.ELASTIC_RECORD)
    bp, per = augment.sample_batch_staged([{"OneOf": [{"PiecewiseAffine": 0.05}, {"ElasticTransformation": {"alpha": 30, "sigma": 4}}]}],
                                          np.random.RandomState(0), 8, 16, 16, (16, 16))
    assert bp is None and {q[0][3][0] for q in per} == {"piecewise", "elastic"}
    # oracle fields
    assert not oaug.field_piecewise(np.zeros((1, 4, 4, 2), np.int32), 20, 30).any()
    assert set(np.unique(oaug.field_piecewise(np.full((2, 3, 5, 2), -64), 20, 30))) == {-64}
    g = np.zeros((1, 2, 2, 2), np.int32)
    g[0, :, 1, 0] = 640                                                     # right edge moves 10 px in x: a linear ramp in x
    f = oaug.field_piecewise(g, 8, 10)
    assert (f[..., 1] == 0).all() and (np.diff(f[0, 0, :, 0]) >= 0).all() and f[0, 0, 0, 0] == 0 and (f[0, :, :, 0] == f[0, :1, :, 0]).all()
    e1 = oaug.field_elastic(p[1][3][1][None], 40, 52)
    half = p[1][3][1].copy(); half[1] //= 2
    e2 = oaug.field_elastic(half[None], 40, 52)
    assert np.abs(e1).max() <= rec[1] and e1.std() > 8 and np.abs(e1 - 2 * e2).max() <= 2


def test_background_replacer_sampling_and_oracle(tmp_path):
    """BackgroundReplacer (reference README.md:270-278, FAQ.md:24-38): `augmenters` first, then - with probability 1 - rate - a
    random folder image replaces the pixels outside the eroded mask, at the head of a new pass; `./bg` resolves against the
    experiment's directory."""
    from PIL import Image
    from oracle import augment as oaug
    os.makedirs(str(tmp_path / "bg"))
    for i in range(3):
        Image.fromarray(np.full((20 + i, 30, 3), 50 * i + 10, np.uint8)).save(str(tmp_path / "bg" / ("b%d.png" % i)))
    rng = np.random.RandomState(1)
    raw = [{"BackgroundReplacer": {"path": "./bg", "rate": 0.0, "erosion": [0, 2], "augmenters": {"Fliplr": 1.0}}}, {"Add": 5}]
    spec = augment.resolve_paths(raw, str(tmp_path))
    assert os.path.isabs(spec[0]["BackgroundReplacer"]["path"]) and raw[0]["BackgroundReplacer"]["path"] == "./bg"
    p = augment.sample_staged(spec, rng, 16, 24, (16, 24))
    assert [len(q) for q in p] == [3, 5] and p[0][0][0] == -1 and p[1][3] is None and p[1][0][6] == 5
    bg, er = p[1][4]
    assert bg.dtype == np.uint8 and bg.shape[1:] == (30, 3) and 0 <= er <= 2
    assert [len(q) for q in augment.sample_staged([{"BackgroundReplacer": {"path": spec[0]["BackgroundReplacer"]["path"], "rate": 1.0}}], rng, 16, 24, (16, 24))] == [3]
    bp, per = augment.sample_batch_staged(spec, rng, 2, 16, 24, (16, 24))
    assert bp is None and len(per) == 2                       # a background image is per image: items run one by one
    b1 = augment.batch_of_one(p[1])
    assert len(b1) == 5 and b1[0].shape == (1, 24) and b1[3] is None and b1[4][1] == er
    # first in the list, nothing before it: no extra pass
    assert [len(q) for q in augment.sample_staged([{"BackgroundReplacer": {"path": spec[0]["BackgroundReplacer"]["path"], "rate": 0.0}}], rng, 16, 24, (16, 24))] == [5]
    m = np.zeros((1, 8, 8), np.uint8)
    m[0, 2:6, 2:6] = 1
    m[0, 0:3, 6:8] = 1                                         # touches the border: the border itself does not erode
    img, bgi = np.full((1, 8, 8, 3), 200, np.uint8), np.full((1, 8, 8, 3), 7, np.uint8)
    o0, o1 = oaug.background_replace_u8(img, m, bgi, 0), oaug.background_replace_u8(img, m, bgi, 1)
    assert ((o0[0, :, :, 0] == 200) == (m[0] != 0)).all()
    assert (o1[0, :, :, 0] == 200).sum() == 4 + 2 and o1[0, 3, 3, 0] == 200 and o1[0, 0, 7, 0] == 200 and o1[0, 2, 2, 0] == 7


def test_cfg_gpus_without_torchrun_explains_how_to_launch(tmp_path, monkeypatch):
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    cfg = segmentation.parse(write_cfg(tmp_path))
    cfg.gpus = 4
    with pytest.raises(RuntimeError, match="nproc-per-node 4"):
        cfg.fit(object())                     # a live dataset object cannot be shipped to the ranks


def test_crops_dataset_splits_items_into_cells():
    """`crops: N` (README.md:476-491): N x N cells per item, row-major, boundaries floor(k * size / N); the cells tile the image."""
    from segmentation_pipeline.impl.datasets import PredictionItem

    class DS(object):
        def __len__(self):
            return 2

        def __getitem__(self, i):
            x = np.arange(7 * 10 * 3, dtype=np.uint8).reshape(7, 10, 3) + i
            y = np.zeros((7, 10, 1), np.uint8); y[5:, 7:] = 1
            return PredictionItem("im%d" % i, x, y)

    cd = pipeline.CropsDataSet(DS(), 3)
    assert len(cd) == 18 and pipeline.crop_bounds(7, 3) == [0, 2, 4, 7] and pipeline.crop_bounds(10, 3) == [0, 3, 6, 10]
    it = cd[9 + 5]                                                   # base item 1, cell (row 1, col 2)
    assert it.id == "im1.5" and it.x.shape == (2, 4, 3) and np.array_equal(it.x, DS()[1].x[2:4, 6:10]) and it.y.shape == (2, 4, 1)
    rows = [np.concatenate([cd[r * 3 + c].x for c in range(3)], axis=1) for r in range(3)]
    assert np.array_equal(np.concatenate(rows, axis=0), DS()[0].x)
    assert [cd.isPositive(k) for k in range(9)] == [False] * 8 + [True]


def test_lr_finder_schedule_and_stop_rule():
    f = pipeline.LRFinder(1e-5, 1.0, 100)
    assert abs(1e-5 * f.mult ** 100 - 1.0) < 1e-9                       # geometric sweep start_lr -> end_lr over the batches
    lr = 1e-5
    for loss in (2.0, 1.5, 1.0, 1.2, 3.9):
        assert not f.record(lr, loss); lr *= f.mult
    assert f.record(lr, 4.1) and len(f.lrs) == len(f.losses) == 6        # > 4 x best (1.0): stop
    assert pipeline.LRFinder(1e-5, 1.0, 10).record(1e-5, float("nan"))
    assert f.get_derivatives(2)[:2] == [0.0, 0.0] and abs(f.get_derivatives(2)[2] - (1.0 - 2.0) / 2) < 1e-12


def test_host_prefetcher_prepares_batches_in_order_and_reports_errors():
    """The loader thread that replaces the reference's augmentation worker processes: items come out batched, in order,
    labels reduced to one uint8 plane (binary / class index / arg-max of one-hot); a failing dataset raises in the consumer."""
    class DS(object):
        def __init__(self, n, fail_at=None):
            self.n, self.fail_at = n, fail_at

        def __len__(self):
            return self.n

        def __getitem__(self, i):
            if i == self.fail_at:
                raise IOError("broken file %d" % i)
            x = np.full((6, 5, 4), i, np.uint8)                       # RGBA: the alpha plane is dropped
            y = np.zeros((6, 5, 3), np.uint8); y[:, :, i % 3] = 1     # one-hot maps
            from segmentation_pipeline.impl.datasets import PredictionItem
            return PredictionItem("id%d" % i, x, y)

    got = list(pipeline.HostPrefetcher(DS(7), list(range(7)), 3, classes=3, pin=False))
    assert [len(b) for b in got] == [3, 3, 1] and [it.id for b in got for it in b] == ["id%d" % i for i in range(7)]
    it = got[1][2]
    assert tuple(it.x.shape) == (6, 5, 3) and it.x.dtype == torch.uint8 and int(it.x[0, 0, 0]) == 5
    assert tuple(it.y.shape) == (6, 5) and int(it.y[0, 0]) == 5 % 3
    # equal-size batches are packed into two blocks (wrapped around to the plan's batch) with the sampled records
    smp = lambda n, h, w: augment.sample_batch_staged([{"Fliplr": 0.5}], np.random.RandomState(0), n, h, w, (8, 8))
    hb = list(pipeline.HostPrefetcher(DS(5), list(range(5)), 4, classes=3, pin=False, sampler=smp))
    assert isinstance(hb[0], pipeline.HostBatch) and tuple(hb[1].X.shape) == (4, 6, 5, 3) and len(hb[1]) == 1
    assert [int(v) for v in hb[1].X[:, 0, 0, 0]] == [4, 4, 4, 4] and hb[0].per_item is None and len(hb[0].passes) == 1
    assert hb[0].passes[0][0].shape == (4, 24) and hb[0].passes[0][1] is None and hb[0].passes[0][2] == (8, 8)
    b = pipeline.prepare_item(DS(1)[0], 1, False)                     # 1-class head: any non-zero label is foreground
    assert set(np.unique(b.y.numpy())) == {1}
    with pytest.raises(IOError, match="broken file 4"):
        list(pipeline.HostPrefetcher(DS(7, fail_at=4), list(range(7)), 3, classes=1, pin=False))


def test_fit_launcher_honours_num_gpus_and_gpus_per_net(tmp_path):
    """`musket fit --num_gpus/--gpus_per_net` (README.md:45-57): experiments -> waves of torchrun jobs on disjoint devices."""
    from segmentation_training_pipeline_amd import fit
    for n in ("a", "b", "c"):
        os.makedirs(str(tmp_path / "experiments" / n))
        (tmp_path / "experiments" / n / "config.yaml").write_text("architecture: Unet\n")
    exps = fit.find_experiments(str(tmp_path))
    assert [n for n, _ in exps] == ["a", "b", "c"] and exps[0][1].endswith(os.path.join("experiments", "a", "config.yaml"))
    waves = fit.plan_launches(exps, num_gpus=8, gpus_per_net=4)
    assert [[j["name"] for j in w] for w in waves] == [["a", "b"], ["c"]]
    assert waves[0][0]["devices"] == [0, 1, 2, 3] and waves[0][1]["devices"] == [4, 5, 6, 7] and waves[1][0]["devices"] == [0, 1, 2, 3]
    assert len({j["port"] for w in waves for j in w}) == 3
    cmd = fit.command(waves[0][1], allow_resume=True, folds=[0, 2])
    assert cmd[1:4] == ["-m", "torch.distributed.run", "--nnodes=1"] and cmd[cmd.index("--nproc-per-node") + 1] == "4"
    assert "127.0.0.1" in cmd and cmd[-3:] == ["--allow_resume", "--folds", "0,2"] and "--worker" in cmd
    assert fit.plan_launches(exps[:1], 8, 8)[0][0]["devices"] == list(range(8))
    with pytest.raises(ValueError, match="exceeds"):
        fit.plan_launches(exps, 2, 4)
    with pytest.raises(FileNotFoundError):
        fit.find_experiments(str(tmp_path), ["nope"])


def test_plan_structure_matches_unet_resnet34():
    plan = graph.Plan(2, "bf16", "cpu", training=True)
    plan.define(lambda p: nets.unet_resnet(p, "resnet34", 64, 64))
    kernels = sum(i.numel for i in plan.params.values() if i.kind == "kernel")
    assert kernels == 24421456                                         # same count as the oracle graph
    names = [n for _, _, n, _ in plan.fwd]
    # the raw-image BN and the BN after the max-pool reduce on their own; the other 43 take their batch
    # statistics from the epilogue of the conv that produces their input
    # (where the partial table is small the finalize rides in the apply pass: stp_bn_finalize_apply)
    assert names.count("stp_conv2d") == 48 and names.count("stp_bn_stats") == 2
    assert names.count("stp_bn_finalize") + names.count("stp_bn_finalize_apply") == 43 and names.count("stp_bn_finalize_apply") >= 10
    bnames = [n for _, _, n, _ in plan.bwd]
    # weight gradients: the row-of-taps layers (3x3 / stride 1, 64-channel blocks, maps of 16+ columns) are collected into
    # grouped launches (stp_wgrad_group_*: consecutive eligible layers of one class), the others are launched alone
    grouped = sum(len(names_) for names_, _ in plan.wgroups)
    assert grouped == 9 and [c for _, c in plan.wgroups] == [32, 64, 64]       # at 64 px only the 16- / 32- / 64-pixel maps qualify
    assert bnames.count("stp_wgrad_group_partial") == bnames.count("stp_wgrad_group_reduce") == len(plan.wgroups) == 3
    # data gradients: none for the stem; a stride-2 projection shortcut's rides in its sibling conv1's launch where the parity-class
    # path takes the shape (stp_conv_params.fold_*: at 64 px stage 2 only - the later maps are too small for a pixel tile per class)
    folded = sum(1 for t in plan.tensors.values() if t.meta.get("dgrad_folded"))
    assert folded == 1 and plan.tensors["stage2_unit1_sc"].meta.get("dgrad_folded")
    assert bnames.count("stp_conv2d_wgrad") == 48 - grouped and bnames.count("stp_conv2d") == 47 - folded
    assert bnames.count("stp_conv2d_wgrad_reduce") == 48 - grouped
    # BatchNormalization outputs read by exactly one convolution get their backward sums from that convolution's
    # data-gradient epilogue: 16 bn2 + 12 bn1 of the non-first units + 5 decoder bn1 + the last decoder bn2, plus
    # decoder_stage3_bn2 whose consumer's data-gradient folds the UpSampling2D gradient (dst_sum2x2, small-channel kernel);
    # and the bn1 of the four stage-first units (read by the shortcut, conv1 and - stages 2..4 - a decoder concat): the last
    # data gradient to arrive accumulates on top of the others and reduces the sums of the complete gradient
    assert bnames.count("stp_bn_backward_fused") + bnames.count("stp_bn_backward_fused_add") + bnames.count("stp_bn_backward") == 44
    # ... and the four BN outputs whose gradient is completed by an UpSampling2D gradient (bn1, decoder_stage0..2_bn2: their
    # consumer concatenates a skip, so the fold does not apply): stp_upsample2x_bwd_bn masks and reduces in the same pass.
    # bn0 is completed by the max-pool gradient; stp_maxpool3x3s2_bwd_bn can fuse that too but measured slower than the pair
    # (opt-in: STP_FUSE_POOL_BN=1), so bn0 keeps the two-pass stp_bn_backward.
    # (stp_bn_backward_fused_add: the bn1 of a non-first unit accumulates onto the residual gradient, which is the dY of a convolution
    # whose weight gradient waits in a pending group - the sum goes to a fresh buffer, that dY stays intact)
    assert bnames.count("stp_bn_backward_fused") + bnames.count("stp_bn_backward_fused_add") == 43 and bnames.count("stp_maxpool3x3s2_bwd_bn") == 0
    assert bnames.count("stp_bn_backward_fused_add") == 2              # stage1_unit3 / unit2 bn1 (the 16-pixel maps of stage 1 are grouped)
    assert bnames.count("stp_upsample2x_bwd_bn") == 3 and bnames.count("stp_upsample2x_bwd") == 0    # 5 decoder stages, two folded
    # stream markers (honoured only with STP_SIDE_STREAM_WGRAD=1): one fork per per-layer chain and per group, a join at every convolution
    assert bnames.count("fork") == 48 - grouped + 3 and bnames.count("join") == 48
    assert "stp_add_inplace" not in bnames                             # every residual gradient aliases
    fl = sum(m["flops"] for _, _, _, m in plan.fwd if m)
    assert abs(fl / 2 / (2 * 1e6) - 31323 * (64 * 64) / (512 * 512)) < 2.0   # 31.3 GMAC/img at 512^2
    frozen = graph.Plan(2, "fp32", "cpu", training=True)
    frozen.frozen_prefixes = nets.ENCODER_PREFIXES
    frozen.define(lambda p: nets.unet_resnet(p, "resnet18", 64, 64))
    fb = [m["layer"] for _, _, n, m in frozen.bwd if n == "stp_conv2d_wgrad"] + [l for names_, _ in frozen.wgroups for l in names_]
    assert fb and all(l.startswith("decoder_") or l.startswith("final_") for l in fb)


def test_weight_gradient_groups_of_the_headline_workload():
    """U-Net/ResNet34 512x512 bs16 (BASELINE configs[1]): 35 of the 48 weight gradients run in 6 grouped launches - one per stage
    (consecutive row-of-taps layers of one class, cut at the stride-2 / 1x1 layers) - the arena fills from its tail as before."""
    plan = graph.Plan(16, "bf16", "cpu", training=True)
    plan.define(lambda p: nets.unet_resnet(p, "resnet34", 512, 512))
    got = [(c, len(n), n[0], n[-1]) for n, c in plan.wgroups]
    assert got == [(32, 1, "decoder_stage3_conv1", "decoder_stage3_conv1"), (64, 2, "decoder_stage2_conv2", "decoder_stage2_conv1"),
                   (128, 9, "decoder_stage1_conv2", "stage4_unit1_conv2"), (128, 11, "stage3_unit6_conv2", "stage3_unit1_conv2"),
                   (128, 7, "stage2_unit4_conv2", "stage2_unit1_conv2"), (64, 6, "stage1_unit3_conv2", "stage1_unit1_conv1")]
    bnames = [n for _, _, n, _ in plan.bwd]
    assert bnames.count("stp_conv2d_wgrad") == 12 and bnames.count("stp_wgrad_group_partial") == 6
    # every projection shortcut's data gradient rides in its sibling's launch (stride 2: space-to-depth form; stage 1: a second source)
    assert sorted(n for n, t in plan.tensors.items() if t.meta.get("dgrad_folded")) == ["stage1_unit1_sc", "stage2_unit1_sc", "stage3_unit1_sc", "stage4_unit1_sc"]
    dg = {m["layer"]: m for _, _, n, m in plan.bwd if m and m.get("pass") == "dgrad"}
    assert all(dg["stage%d_unit1_conv1" % k]["s2d"] for k in (2, 3, 4)) and dg["stage1_unit1_conv1"]["fold1"] and "stage1_unit1_sc" not in dg
    assert bnames.count("stp_conv2d") == 47 - 4
    assert bnames.count("stp_bn_backward_fused_add") == 12 and "stp_add_inplace" not in bnames      # 12 non-first units
    assert plan.bwd_monotone
    lows = [low for _, low in plan.bwd_marks]
    assert lows == sorted(lows, reverse=True) and lows[-1] == 0
    # every group's table: all workgroup slots used, 1-3 partial slabs per tile
    import ctypes
    for fn, args, name, meta in plan.bwd:
        if name == "stp_wgrad_group_partial":
            magic, bm, nl, nseg, nwg, ntile = list((ctypes.c_int32 * 6).from_address(args[0]))
            assert bm == meta["bm"] and nl == len(meta["layers"]) and ntile <= nseg <= 3 * ntile + nwg
            assert nwg <= 256 * {128: 2, 64: 3, 32: 4}[bm]


def test_bottleneck_shortcut_data_gradients_run_at_low_resolution():
    """FPN/ResNet50 (BASELINE configs[3]): the 1x1 / stride-2 projection shortcuts of stages 2-4 - their data gradient is a 1x1 / stride-1
    launch at LOW resolution + stp_scatter2x_bwd (round 5; the zero-inserted form ran the GEMM over four parity classes).  nets.py issues
    the shortcut AFTER conv1, so in the backward pass it arrives first: the scatter writes a fresh buffer (no accumulate, no mask) and
    conv1's dense 1x1 data gradient completes the gradient of bn1's output in its epilogue (stp_bn_backward_fused follows)."""
    plan = graph.Plan(4, "bf16", "cpu", training=True)
    plan.define(lambda p: nets.fpn_resnet(p, "resnet50", 256, 256, classes=3))
    bnames = [n for _, _, n, _ in plan.bwd]
    assert bnames.count("stp_scatter2x_bwd") == 3 and bnames.count("stp_scatter2x_bwd_bn") == 0
    for k in (2, 3, 4):
        i_sc = next(i for i, (_, _, n, m) in enumerate(plan.bwd) if n == "stp_conv2d" and m and m.get("layer") == "stage%d_unit1_sc" % k)
        i_c1 = next(i for i, (_, _, n, m) in enumerate(plan.bwd) if n == "stp_conv2d" and m and m.get("layer") == "stage%d_unit1_conv1" % k)
        assert i_sc < i_c1 and plan.bwd[i_sc + 1][2] == "stp_scatter2x_bwd"
    dg = {m["layer"]: m for _, _, n, m in plan.bwd if n == "stp_conv2d" and m and m.get("pass") == "dgrad"}
    for k, (cin, cout, hw) in {2: (256, 512, 32), 3: (512, 1024, 16), 4: (1024, 2048, 8)}.items():
        m = dg["stage%d_unit1_sc" % k]
        assert abs(m["flops"] - 2.0 * 4 * hw * hw * cin * cout) < 1.0          # the GEMM of ONE parity class


def test_class_heads_of_fpn_and_pspnet_take_the_tap_channel_form():
    """final_conv of FPN (512 -> 3) and PSPNet (512 -> 20): a 1x1 launch into 9 x classes tap channels + stp_tapsum_fwd; the parameters keep
    the layer's names and shapes (checkpoints, set_weights / get_weights and the oracle's weight mapping do not change); U-Net's 16 -> 1 head
    stays a 3x3 launch."""
    for build, classes, cin in ((lambda p: nets.fpn_resnet(p, "resnet50", 256, 256, classes=3), 3, 512),
                                (lambda p: nets.pspnet_resnet(p, "resnet50", 192, 192, classes=20), 20, 512)):
        plan = graph.Plan(2, "bf16", "cpu", training=True)
        plan.define(build)
        assert plan.params["final_conv/kernel"].shape == (classes, 3, 3, cin) and plan.params["final_conv/bias"].shape == (classes,)
        assert plan.tensors["final_conv_taps"].C == 9 * classes and plan.tensors["final_conv"].C == classes
        fn = [n for _, _, n, _ in plan.fwd]
        bn = [n for _, _, n, _ in plan.bwd]
        assert fn.count("stp_tapsum_fwd") == 1 and bn.count("stp_tapsum_bwd") == 1
        head = [m for _, _, n, m in plan.fwd if n == "stp_conv2d" and m and m.get("layer") == "final_conv_taps"]
        assert len(head) == 1 and abs(head[0]["flops"] - 2.0 * 2 * plan.tensors["final_conv"].H ** 2 * 9 * classes * cin) < 1.0
    plan = graph.Plan(2, "bf16", "cpu", training=True)
    plan.define(lambda p: nets.unet_resnet(p, "resnet18", 64, 64))
    assert "final_conv_taps" not in plan.tensors


def test_pspnet_head_without_its_concatenation_keeps_the_reference_parameters(monkeypatch):
    """segmentation_models' PSPNet head (schemas/segmentation.raml:226-249) is planned WITHOUT its 2560-channel concatenation (round 6,
    nets.pspnet_resnet): the parameters keep the reference's names and shapes (one `psp_final/kernel` over all C + 4 F input channels),
    the roofline bookkeeping keeps the reference layer's FLOP, the gradient arena is still written in descending order (the data-parallel
    overlap needs it), and STP_PSP_SPLIT=0 restores the concatenated plan.  Host logic only."""
    from oracle import nets as onets

    def plan_for(split):
        monkeypatch.setenv("STP_PSP_SPLIT", split)
        p = graph.Plan(2, "bf16", "cpu", training=True)
        p.define(lambda q_: nets.pspnet_resnet(q_, "resnet50", 96, 96, classes=5))
        return p
    new, old = plan_for("1"), plan_for("0")
    P = onets.init_pspnet_resnet("resnet50", classes=5, seed=1)
    shapes = lambda pl: {k: tuple(v.shape) for k, v in pl.params.items()}
    assert shapes(new) == shapes(old) and set(shapes(new)) == {k for k in P if not k.endswith(("moving_mean", "moving_variance"))}
    assert shapes(new)["psp_final/kernel"] == (512, 1, 1, 512 + 4 * 512)              # OHWI of Keras' (1, 1, 2560, 512)
    f = lambda pl, n: [x for x in pl.prep + pl.fwd + pl.bwd if x[2] == n]
    assert len(f(new, "stp_upsample_sum")) == 1 and not f(old, "stp_upsample_sum")
    # no resize launch is left: the pyramid levels enter through stp_upsample_sum, the logits' resize lives inside the loss (see below)
    assert len(f(new, "stp_resize_bilinear")) == 0 and len(f(old, "stp_resize_bilinear")) == 5
    assert len(f(new, "stp_softmax_cce_dice_up")) == 1 and len(f(old, "stp_softmax_cce_dice_up")) == 1
    assert len(f(new, "stp_copy_cols_f32")) == 10 and new.prep[-1][2] == "stp_weight_prepare_batched"  # five ranges in, five gradients out
    flops = lambda pl: sum(m["flops"] for _, _, _, m in pl.fwd if m and "flops" in m)
    assert abs(flops(new) - flops(old)) < 1e-6 * flops(old)
    assert new.bwd_monotone and old.bwd_monotone
    assert "psp_concat" not in new.tensors and "psp_pyramid_sum" in new.tensors and new.tensors["psp_pyramid_sum"].C == 512


def test_loss_on_resized_logits_is_planned_without_the_resized_tensor(monkeypatch):
    """PSPNet / FPN end in Conv -> bilinear resize -> softmax loss (segmentation_models 0.2.1: `final_interpolation`, FPN's last
    UpSampling2D): a TRAINING plan replaces resize + loss + (dynamic loss scale) + resize gradient by stp_softmax_cce_dice_up, which reads
    the low-resolution logits and writes their gradient; the resize launch is kept as the deferred record of the "logits" tensor.  The
    1-class (sigmoid) heads, nearest interpolation, inference plans and STP_UP_LOSS=0 keep the unfused launches.  Host logic only."""
    def plan_for(net, training=True, **kw):
        p = graph.Plan(2, "bf16", "cpu", training=training)
        p.define(lambda q_: net(q_, "resnet18", 96, 96, **kw))
        return p
    names = lambda pl, lst=None: [x[2] for x in (pl.prep + pl.fwd + pl.bwd if lst is None else lst)]
    for net, f in ((nets.pspnet_resnet, 8), (nets.fpn_resnet, 4)):
        p = plan_for(net, classes=4)
        n = names(p)
        assert n.count("stp_softmax_cce_dice_up") == 1 and "stp_softmax_cce_dice" not in n and "fused:resize->loss" in n
        assert not [a for a in p.bwd if a[2] == "stp_resize_bilinear_bwd" and a[1][5] == 8 and a[1][6] == f]      # (the 8 padded class channels)
        logits = p.tensors["logits"]
        assert logits.meta.get("fused_into_loss") and logits.meta["deferred"][2] == "stp_resize_bilinear" and logits.grad is None
        a = [x for x in p.fwd if x[2] == "stp_softmax_cce_dice_up"][0][1]
        assert a[2:8] == (2, 96 // f, 96 // f, f, 4, 4) and a[13] == 8                   # N, H, W, factor, classes, row stride; padded gradient rows
        # inference plan, 1-class head, nearest interpolation: nothing changes
        assert "stp_softmax_cce_dice_up" not in names(plan_for(net, training=False, classes=4, with_loss=False))
        assert "stp_softmax_cce_dice_up" not in names(plan_for(net, classes=1))
    assert "stp_softmax_cce_dice_up" not in names(plan_for(nets.pspnet_resnet, classes=4, final_interpolation="nearest"))
    monkeypatch.setenv("STP_UP_LOSS", "0")
    n = names(plan_for(nets.pspnet_resnet, classes=4))
    assert "stp_softmax_cce_dice_up" not in n and n.count("stp_softmax_cce_dice") == 1 and "stp_resize_bilinear" in n


def test_pyramid_pooling_is_planned_as_one_pass(monkeypatch):
    """PSPNet's four AveragePooling2D of the feature map (levels 1, 2, 3, 6: windows that nest) are ONE stp_avgpool_pyramid launch and one
    stp_avgpool_pyramid_bwd in a training plan, the pooled tensors keep their names and shapes; max pooling and STP_POOL_PYRAMID=0 keep the
    separate launches.  Host logic only."""
    def plan_for(**kw):
        p = graph.Plan(1, "bf16", "cpu", training=True)
        p.define(lambda q_: nets.pspnet_resnet(q_, "resnet18", 384, 384, classes=4, **kw))      # (1/8 feature 48 x 48: windows 48 / 24 / 16 / 8)
        return p
    count = lambda pl, n: len([x for x in pl.prep + pl.fwd + pl.bwd if x[2] == n])
    p = plan_for()
    assert count(p, "stp_avgpool_pyramid") == 1 and count(p, "stp_avgpool_pyramid_bwd") == 1 and count(p, "stp_avgpool") == 0 and count(p, "stp_avgpool_bwd") == 0
    a = [x for x in p.fwd if x[2] == "stp_avgpool_pyramid"][0][1]
    assert a[5:9] == (8, 16, 24, 48) and a[9:13] == (1, 48, 48, 128)                       # finest window first; N, H, W, C of the 1/8 feature
    for level in (1, 2, 3, 6):
        t = p.tensors["psp_level%d_pool" % level]
        assert (t.H, t.W, t.C) == (level, level, 128)
    pm = plan_for(psp_pooling_type="max")
    assert count(pm, "stp_avgpool_pyramid") == 0 and count(pm, "stp_maxpool_k") == 4
    monkeypatch.setenv("STP_POOL_PYRAMID", "0")
    po = plan_for()
    assert count(po, "stp_avgpool_pyramid") == 0 and count(po, "stp_avgpool") == 4 and count(po, "stp_avgpool_bwd") == 4


def test_pointwise_kernel_sizing_queries_do_not_depend_on_the_table_pointer():
    """stp_conv2d_stats_floats is asked BEFORE the table of fused sums is allocated (graph.Plan.conv) and again when the BatchNormalization
    that reads it is planned: both answers - and the kernel the launch gets - must agree, with and without stats_partial (round 6: the
    pointwise kernel writes one column per workgroup; a BatchNormalization-backward launch sized as the per-tap kernel's 16 tile columns
    and run as 32 workgroups wrote past its table at 2 x 32 x 32 pixels).  Host logic only: runs without a GPU."""
    import ctypes as C
    from segmentation_training_pipeline_amd import _lib
    lib = _lib.load()
    for (cin, cout) in ((64, 64), (64, 256), (256, 64), (256, 128), (128, 256), (128, 512), (512, 128), (256, 256), (512, 256), (64, 512)):
        for (n, h, w) in ((2, 32, 32), (1, 16, 64), (8, 192, 192)):
            for bnb in (0, 1):
                p = _lib.ConvParams()
                p.src0 = p.weight = p.dst0 = 16
                p.N, p.Hs0, p.Ws0, p.Hv, p.Wv, p.C0, p.C1 = n, h, w, h, w, cin, 0
                p.src0_mode, p.KH, p.KW, p.stride, p.pad, p.Ho, p.Wo, p.Cout, p.Cd0, p.dtype = 0, 1, 1, 1, 0, h, w, cout, cout, _lib.BF16
                if bnb:
                    p.bnb_x = p.bnb_mean = p.bnb_rstd = 16
                    p.bnb_relu = 1
                t0, f0 = lib.stp_conv2d_tile_for(C.byref(p)), lib.stp_conv2d_stats_floats(C.byref(p))
                p.stats_partial = 16
                t1, f1 = lib.stp_conv2d_tile_for(C.byref(p)), lib.stp_conv2d_stats_floats(C.byref(p))
                assert (t0, f0) == (t1, f1), ((cin, cout), (n, h, w), bnb, (t0, f0), (t1, f1))
                assert t1 == 800 and f1 == 2 * cout * lib.stp_conv2d_pw_cols(C.byref(p)) and 0 < lib.stp_conv2d_pw_cols(C.byref(p)) <= 512


def test_bench_names_every_launch_of_the_headline_plan():
    """bench.py's instrumented pass maps every launch of the step to the kernel the library runs for it (kernel_key): a launch kind
    it does not know (a new tile id) must fail here, on CPU, not in the driver's bench run."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    for dtype in ("bf16", "fp16", "fp32"):
        plan = graph.Plan(16, dtype, "cpu", training=True)
        plan.define(lambda p: nets.unet_resnet(p, "resnet34", 512, 512))
        keys = set()
        for lst in (plan.prep, plan.fwd, plan.bwd):
            for fn, args, name, meta in lst:
                if fn is not None:
                    keys.add(bench.kernel_key(name, meta or {}, dtype))
        assert any(k.startswith("conv_halo_kernel") for k in keys) == (dtype != "fp32")
        if dtype != "fp32":
            assert "conv_scw_stream_kernel<unsigned short>" in keys
            # the 128-channel groups run the all-taps kernel (round 4), the 64 / 32-channel groups the row-of-taps one
            assert "conv_wgrad_taps9_group_kernel<128, 3>" in keys and "conv_wgrad_row_group_kernel<64, 1, 4, 3>" in keys
        if dtype == "bf16":
            # roofline_hbm: algorithmic bytes of the streaming entry points from their argument lists (include/stp_hip.h order)
            byt = {}
            for lst in (plan.prep, plan.fwd, plan.bwd, plan.opt):
                for fn, args, name, meta in lst:
                    b = bench.hbm_bytes(name, args) if fn is not None else None
                    if b is not None:
                        assert b > 0, name
                        byt.setdefault(name, []).append(b)
            assert set(byt) >= {"stp_bn_apply", "stp_bn_finalize_apply", "stp_bn_backward_fused", "stp_bn_apply_maxpool3x3s2", "stp_maxpool3x3s2_bwd"}
            assert max(byt["stp_bn_backward_fused"]) == 3 * 16 * 512 * 512 * 16 * 2         # decoder_stage4: x, g read + dx written, bf16
            assert bench.hbm_bytes("stp_adam", (0, 0, 0, 0, 1000)) == 28000          # (the optimizer launch is the backend's, not the plan's)
            assert byt["stp_bn_apply_maxpool3x3s2"] == [2 * 16 * 256 * 256 * 64 * 2 + 16 * 128 * 128 * 64 * 3]      # bn0 + relu0 + pooling0 in one launch
            assert {e for f in bench.HBM_FAMILIES.values() for e in f["entries"]} >= set(byt)


def test_simple_png_mask_dataset(tmp_path):
    from PIL import Image
    img_dir, msk_dir = tmp_path / "img", tmp_path / "msk"
    img_dir.mkdir(); msk_dir.mkdir()
    rng = np.random.RandomState(0)
    for i in range(3):
        Image.fromarray(rng.randint(0, 256, (20, 24, 3)).astype(np.uint8)).save(str(img_dir / ("im%d.jpg" % i)))
        m = np.zeros((20, 24), np.uint8); m[5:10, 3:9] = 255
        Image.fromarray(m).save(str(msk_dir / ("im%d.png" % i)))
    ds = datasets.SimplePNGMaskDataSet(str(img_dir), str(msk_dir))
    assert len(ds) == 3
    it = ds[1]
    assert isinstance(it, datasets.PredictionItem) and it.id == "im1"
    assert it.x.shape == (20, 24, 3) and it.x.dtype == np.uint8
    assert it.y.shape == (20, 24, 1) and set(np.unique(it.y)) == {0, 1} and it.y.sum() == 30
    assert ds.isPositive(0) and datasets.AUGMENTER_QUEUE_LIMIT == 50


def test_bucketing_and_sharding():
    b = distributed.bucket_bounds(1003, 256)
    assert b[0] == (0, 256) and b[-1][1] == 1003 and all(s % 4 == 0 for s, _ in b)
    seen = []
    for r in range(4):
        idx = distributed.shard_indices(10, r, 4, epoch=3, seed=1)
        assert len(idx) == 3
        seen += idx
    assert set(seen) == set(range(10))
    assert distributed.shard_indices(10, 0, 4, 3, 1) == distributed.shard_indices(10, 0, 4, 3, 1)
    assert distributed.shard_indices(10, 0, 4, 4, 1) != distributed.shard_indices(10, 0, 4, 3, 1)


def test_nchannel_adaptation_of_pretrained_weights():
    """models.adapt_nchannel (reference segmentation.py:138-153 -> musket_core adaptNet, restated): equal shapes copied, the input-
    channel axis of the first kernel / the input BatchNormalization vectors widened, copyWeights seeds channel 3 from channel 2."""
    from segmentation_training_pipeline_amd.models import adapt_nchannel
    rng = np.random.RandomState(0)
    pre = {"conv0/kernel": rng.randn(7, 7, 3, 8).astype(np.float32), "bn_data/beta": rng.randn(3).astype(np.float32),
           "bn_data/moving_variance": (1 + rng.rand(3)).astype(np.float32), "stage1/kernel": rng.randn(3, 3, 8, 8).astype(np.float32),
           "fc/kernel": rng.randn(8, 10).astype(np.float32)}
    cur = {"conv0/kernel": np.zeros((7, 7, 5, 8), np.float32), "bn_data/beta": np.zeros(5, np.float32),
           "bn_data/moving_variance": np.ones(5, np.float32), "stage1/kernel": np.zeros((3, 3, 8, 8), np.float32)}
    out = adapt_nchannel(pre, cur, copy=False)
    assert sorted(out) == sorted(cur)                                    # the classifier head of the pretrained file is dropped
    assert np.array_equal(out["conv0/kernel"][:, :, :3], pre["conv0/kernel"]) and not out["conv0/kernel"][:, :, 3:].any()
    assert np.array_equal(out["bn_data/moving_variance"], np.concatenate([pre["bn_data/moving_variance"], [1, 1]]).astype(np.float32))
    assert np.array_equal(out["stage1/kernel"], pre["stage1/kernel"])
    out = adapt_nchannel(pre, cur, copy=True)
    assert np.array_equal(out["conv0/kernel"][:, :, 3], pre["conv0/kernel"][:, :, 2]) and not out["conv0/kernel"][:, :, 4].any()
    assert out["bn_data/beta"][3] == pre["bn_data/beta"][2] and out["bn_data/beta"][4] == 0
    with pytest.raises(ValueError):
        adapt_nchannel({"stage1/kernel": np.zeros((3, 3, 4, 8), np.float32)}, cur)


def test_extra_train_data_joins_every_folds_training_set_only(tmp_path):
    """``extra_train_data: name`` + ``segmentation.extra_train[name] = ds`` (reference segmentation.py:29, README.md:698-709): the
    extra items are appended to the TRAINING indexes of every fold, never to a validation split; an unregistered name raises."""
    class DS(object):
        def __init__(self, n, tag):
            self.n, self.tag = n, tag

        def __len__(self):
            return self.n

        def __getitem__(self, i):
            return (self.tag, int(i))

    base, extra = DS(10, "base"), DS(4, "extra")
    cat = pipeline.ConcatDataSet(base, extra)
    assert len(cat) == 14 and cat[9] == ("base", 9) and cat[10] == ("extra", 0) and cat[13] == ("extra", 3)
    kf = pipeline.KFoldedDataSet(cat, range(len(base)), folds_count=5, random_state=33)
    kf.extra_train_indexes = range(10, 14)
    for f in range(5):
        tr, va = kf.sampledIndexes(f, True), kf.sampledIndexes(f, False)
        assert set(range(10, 14)) <= set(tr.tolist()) and not (set(va.tolist()) & set(range(10, 14)))
        assert len(tr) == 8 + 4 and len(va) == 2 and not (set(tr.tolist()) & set(va.tolist()))
    assert segmentation.extra_train is pipeline.extra_train
    cfg_path = str(tmp_path / "c.yaml")
    with open(cfg_path, "w") as f:
        yaml.safe_dump({"architecture": "Unet", "backbone": "resnet18", "classes": 1, "activation": "sigmoid", "encoder_weights": None,
                        "shape": [64, 64, 3], "optimizer": "Adam", "lr": 0.001, "batch": 2, "folds_count": 2, "loss": "binary_crossentropy",
                        "extra_train_data": "people_not_registered", "stages": [{"epochs": 1}]}, f)
    cfg = segmentation.parse(cfg_path)
    with pytest.raises(ValueError, match="not registered"):
        cfg.fit(base)


def test_writeable_prediction_datasets_and_update(tmp_path):
    """reference segmentation.py:58-60, :196-208: ``update`` attaches predictions; ``create_/load_writeable_dataset`` keep
    predictions in a folder, uint8 (scale <= 255) or uint16-compressed per ``compressPredictionsAsInts`` / ``compressScale``."""
    class DS(datasets.DataSet):
        name = "val"
        def __len__(self): return 3
        def __getitem__(self, i): return datasets.PredictionItem("id%d" % i, np.full((4, 4, 3), i, np.uint8), None)
    rng = np.random.RandomState(0)
    preds = [rng.rand(4, 4, 1).astype(np.float32) for _ in range(3)]
    for over, tol, dt in (({}, 1.0 / 255, np.uint8), ({"compressScale": 1000}, 1e-3, np.uint16), ({"compressPredictionsAsInts": False}, 0.0, np.float32)):
        sub = tmp_path / ("p%d" % len(over) if not over else "p_" + list(over)[0])
        sub.mkdir()
        cfg = segmentation.parse(write_cfg(sub, **over))
        w = cfg.create_writeable_dataset(DS(), str(sub / "out"))
        assert isinstance(w, datasets.WriteableDataSet) and len(w) == 0 and w.name == "val_predictions"
        for p in preds:
            w.append(p)
        w.commit()
        with np.load(str(sub / "out" / "1.npy.npz")) as z:
            assert z["arr"].dtype == dt
        r = cfg.load_writeable_dataset(DS(), str(sub / "out"))
        assert len(r) == 3
        for i in range(3):
            it = r[i]
            assert it.id == "id%d" % i and it.x[0, 0, 0] == i and np.abs(it.y - preds[i]).max() <= tol + 1e-7
        with pytest.raises(IndexError):
            r[3]
    cfg = segmentation.parse(write_cfg(tmp_path))
    z = pipeline.ItemBatch([None], [None], ["a"])
    cfg.update(z, [preds[0]])
    assert len(z.segmentation_maps_aug) == 1 and z.segmentation_maps_aug[0].arr is preds[0] and z.segmentation_maps_aug[0].shape == (4, 4, 1)


def test_folds_load_returns_the_first_items_of_the_fold():
    """reference segmentation.py:41, 228: ``folds.load(fold, isTrain, negatives, limit)``."""
    class DS(object):
        def __len__(self): return 10
        def __getitem__(self, i): return datasets.PredictionItem("i%d" % i, np.full((2, 2, 3), i, np.uint8), np.full((2, 2, 1), i % 2, np.uint8))
        def isPositive(self, i): return i % 2 == 1
    cfg = pipeline.GenericTaskConfig(folds_count=2, shape=[8, 8, 3])
    kf = cfg.kfold(DS())
    b = kf.load(0, False, "all", 3)
    want = [int(i) for i in kf.sampledIndexes(0, False, "all")[:3]]
    assert b.data == ["i%d" % i for i in want] and [int(x[0, 0, 0]) for x in b.images] == want and len(b.segmentation_maps) == 3
    pos = kf.load(0, False, "none", 16)
    assert all(int(x[0, 0, 0]) % 2 == 1 for x in pos.images)
    assert kf.cfg is cfg
    with pytest.raises(ValueError):
        pipeline.KFoldedDataSet(DS(), range(10)).augmentor(True)
