"""CPU-side checks of the drop-in boundary: the C-ABI library builds, loads and
exports every symbol include/skf.h declares (no compute calls without a GPU)."""
import ctypes as C
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    from sketchformer_amd import build, _lib
    build.build_library(verbose=False)
    return _lib.load()


def _declared():
    text = open(os.path.join(ROOT, "include", "skf.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(skf_[a-z0-9_]+)\s*\(", text)))


def test_every_declared_symbol_is_exported(lib):
    names = _declared()
    assert len(names) >= 30
    for n in names:
        assert hasattr(lib, n), "libskf.so does not export %s" % n


def test_binding_table_matches_header(lib):
    from sketchformer_amd import _lib
    assert sorted(_lib.SIGNATURES) == _declared()


def test_host_only_entry_points(lib):
    from sketchformer_amd import _lib, engine
    assert lib.skf_version() >= 100
    cfg = engine.make_config(batch=4, seq_len=24, d_model=64, num_heads=4, dff=128, num_layers=2, vocab_size=52,
                             n_classes=7, lowerdim=32)
    assert lib.skf_config_validate(C.byref(cfg)) == 0
    entries = engine.param_entries(cfg)
    n = lib.skf_model_param_floats(C.byref(cfg))
    assert n > 0 and lib.skf_model_workspace_bytes(C.byref(cfg)) > 0
    # layout is a partition: no overlap, inside the buffer
    import numpy as np
    used = np.zeros(n, dtype=np.int32)
    for e in entries:
        for r in range(e["rows"]):
            used[e["offset"] + r * e["row_stride"]: e["offset"] + r * e["row_stride"] + e["cols"]] += 1
    assert used.max() == 1


def test_layout_names_match_oracle(lib):
    import oracle
    from sketchformer_amd import engine
    ocfg = oracle.Config(num_layers=2, d_model=64, dff=128, num_heads=4, lowerdim=32, vocab_size=52, n_classes=7, seq_len=24)
    cfg = engine.make_config(batch=4, seq_len=24, d_model=64, num_heads=4, dff=128, num_layers=2, vocab_size=52,
                             n_classes=7, lowerdim=32)
    want = {n: s for n, s, _ in oracle.param_specs(ocfg)}
    got = {e["name"]: engine.logical_shape(e) for e in engine.param_entries(cfg)}
    assert got == want


@pytest.mark.parametrize("kw", [dict(do_classification=False), dict(do_reconstruction=False),
                                dict(lowerdim=0, do_classification=False), dict(do_reconstruction=False, attn_version=2, lowerdim=64)])
def test_structural_variant_layouts_match_oracle(lib, kw):
    """do_classification / do_reconstruction off, lowerdim=0 (models/sketchformer.py:76-108): the variable set shrinks."""
    import oracle
    from sketchformer_amd import engine
    base = dict(num_layers=2, d_model=64, dff=128, num_heads=4, lowerdim=32, vocab_size=52, n_classes=7, seq_len=24)
    base.update(kw)
    ocfg = oracle.Config(**base)
    cfg = engine.make_config(batch=4, **base)
    want = [(n, s) for n, s, _ in oracle.param_specs(ocfg)]
    got = [(e["name"], engine.logical_shape(e)) for e in engine.param_entries(cfg)]
    assert dict(got) == dict(want)
    assert [n for n, _ in got if "/mha" not in n] == [n for n, _ in want if "/mha" not in n]


@pytest.mark.parametrize("attn_version,cbuf", [(2, 0), (1, 2), (2, 1)])
def test_variant_layout_names_match_oracle(lib, attn_version, cbuf):
    """SelfAttnV2 (builders/layers/transformer.py:76-131: W (d,d), Dense(lowerdim) -> embedding width = lowerdim, which is
    also the cross-attention K/V input width) and class_buffer_layers (models/sketchformer.py:101-104)."""
    import oracle
    from sketchformer_amd import engine
    ocfg = oracle.Config(num_layers=2, d_model=64, dff=128, num_heads=4, lowerdim=128, vocab_size=52, n_classes=7, seq_len=24,
                         attn_version=attn_version, class_buffer_layers=cbuf)
    cfg = engine.make_config(batch=4, seq_len=24, d_model=64, num_heads=4, dff=128, num_layers=2, vocab_size=52,
                             n_classes=7, lowerdim=128, attn_version=attn_version, class_buffer_layers=cbuf)
    want = {n: s for n, s, _ in oracle.param_specs(ocfg)}
    got = {e["name"]: engine.logical_shape(e) for e in engine.param_entries(cfg)}
    assert got == want
    assert [e["name"] for e in engine.param_entries(cfg) if "/mha" not in e["name"]] == \
           [n for n, _, _ in oracle.param_specs(ocfg) if "/mha" not in n]          # same forward order


def test_unsupported_configs_fail_loudly(lib):
    from sketchformer_amd import engine
    cfg = engine.make_config(batch=4, lowerdim=0)                       # class head without a bottleneck: the reference fails too
    assert lib.skf_config_validate(C.byref(cfg)) == -1
    assert b"lowerdim" in lib.skf_last_error()
    assert lib.skf_config_validate(C.byref(engine.make_config(batch=4, lowerdim=0, do_classification=False))) == 0
    assert lib.skf_config_validate(C.byref(engine.make_config(batch=4, do_reconstruction=False))) == 0
    assert lib.skf_config_validate(C.byref(engine.make_config(batch=4, do_reconstruction=False, do_classification=False))) == -1
    # any d_model % num_heads == 0 like the reference (MFMA kernels for the BASELINE widths, skf_generic.hip for the others), within
    # d_model % 4 == 0, head size % 4 == 0, d_model <= 1024, head size <= 128
    assert lib.skf_config_validate(C.byref(engine.make_config(batch=4, d_model=96))) == 0         # 8 heads of 12
    assert lib.skf_config_validate(C.byref(engine.make_config(batch=4, d_model=80, num_heads=2))) == 0
    assert lib.skf_config_validate(C.byref(engine.make_config(batch=4, d_model=100, num_heads=8))) == -1      # not divisible by the heads
    assert lib.skf_config_validate(C.byref(engine.make_config(batch=4, d_model=72, num_heads=4))) == -2       # head size 18
    assert b"head size" in lib.skf_last_error()
    assert lib.skf_config_validate(C.byref(engine.make_config(batch=4, d_model=2048, num_heads=32))) == -2
    assert lib.skf_config_validate(C.byref(engine.make_config(batch=4, attn_version=3))) == -1
    assert lib.skf_config_validate(C.byref(engine.make_config(batch=4, attn_version=2, lowerdim=100))) == 0
    assert lib.skf_config_validate(C.byref(engine.make_config(batch=4, attn_version=2, lowerdim=102))) == -2
    assert lib.skf_config_validate(C.byref(engine.make_config(batch=4, attn_version=2, class_buffer_layers=2, optimizer="sgd"))) == 0
    with pytest.raises(ValueError):
        engine.make_config(batch=4, optimizer="rmsprop")
    assert lib.skf_config_validate(C.byref(engine.make_config(batch=4, continuous=True, vocab_size=None))) == 0
    with pytest.raises(TypeError):
        engine.make_config(batch=4, lr_scheduler="step-decay")


def test_continuous_layout_names_match_oracle(lib):
    import oracle
    from sketchformer_amd import engine
    ocfg = oracle.Config(num_layers=6, d_model=256, dff=1024, num_heads=8, lowerdim=256, n_classes=345, seq_len=200, continuous=True)
    cfg = engine.make_config(batch=4, num_layers=6, d_model=256, dff=1024, continuous=True, vocab_size=None)
    want = {n: s for n, s, _ in oracle.param_specs(ocfg)}
    got = {e["name"]: engine.logical_shape(e) for e in engine.param_entries(cfg)}
    assert got == want
    assert sum(int(np.prod(s)) for s in want.values()) == 11218670        # P for cfg 3


def test_param_count_matches_survey(lib):
    """P = 2,316,117 for cfg 2 (C=345), 2,271,741 for cfg 1 (C=1)."""
    from sketchformer_amd import engine
    for C_, want in ((345, 2316117), (1, 2271741)):
        cfg = engine.make_config(batch=128, n_classes=C_)
        assert sum(e["rows"] * e["cols"] for e in engine.param_entries(cfg)) == want


def test_no_cpu_fallback():
    import torch
    from sketchformer_amd import engine, _lib
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(_lib.SkfError):
        engine.TrainEngine(engine.make_config(batch=4))
