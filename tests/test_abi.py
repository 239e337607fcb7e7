"""CPU tests: the C-ABI library builds/loads and exports every symbol include/stp_hip.h declares
(no compute calls - there is no GPU here), and the product has no CPU fallback."""
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    with open(os.path.join(ROOT, "include", "stp_hip.h")) as f:
        src = f.read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(stp_[a-z0-9_]+)\s*\(", src)))


def test_library_exports_every_declared_symbol():
    from segmentation_training_pipeline_amd import _lib, build
    build.build()
    lib = _lib.load()
    names = declared_symbols()
    assert len(names) >= 25
    for n in names:
        assert hasattr(lib, n), "libstp_hip.so does not export %s" % n
    assert lib.stp_abi_version() == 1
    assert lib.stp_storage_dtype() == _lib.BF16
    # the IEEE-half build of the same sources: same symbols, STP_F16 storage
    f16 = _lib.load("fp16")
    for n in names:
        assert hasattr(f16, n), "libstp_hip_f16.so does not export %s" % n
    assert f16.stp_storage_dtype() == _lib.F16 == 3 and f16.stp_abi_version() == 1


def test_ctypes_signatures_cover_the_header():
    from segmentation_training_pipeline_amd import _lib
    assert sorted(_lib.SIGNATURES) == declared_symbols()


def test_param_structs_match_header_field_order():
    from segmentation_training_pipeline_amd import _lib
    with open(os.path.join(ROOT, "include", "stp_hip.h")) as f:
        src = f.read()
    for struct, cls in (("stp_conv_params", _lib.ConvParams), ("stp_wgrad_params", _lib.WgradParams)):
        body = re.search(r"typedef struct %s \{(.*?)\} %s;" % (struct, struct), src, flags=re.S).group(1)
        body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
        fields = []
        for decl in body.split(";"):
            decl = decl.strip()
            if not decl:
                continue
            names = decl.split()[-1] if "," not in decl else None
            if names is None:
                first, rest = decl.split(",", 1)
                fields.append(first.split()[-1].lstrip("*"))
                fields += [r.strip().lstrip("*") for r in rest.split(",")]
            else:
                fields.append(names.lstrip("*"))
        assert fields == [f[0] for f in cls._fields_], struct


def test_no_cpu_fallback():
    import torch
    from segmentation_training_pipeline_amd import _lib, graph, nets, ops
    plan = graph.Plan(2, "bf16", "cpu", training=True)
    plan.define(lambda p: nets.unet_resnet(p, "resnet18", 64, 64))
    with pytest.raises(_lib.StpError):
        plan.run(plan.fwd)                       # a plan can be built for inspection, never run, off-GPU
    with pytest.raises(_lib.StpError):
        ops.bn_stats(torch.zeros(4, 4), 4, 4, 1e-3, 0.99, torch.zeros(4), torch.zeros(4), None, None, torch.zeros(8192))
    if not torch.cuda.is_available():
        with pytest.raises(_lib.StpError):
            graph.Plan(2, "bf16", "cuda")
