"""CPU tests of the drop-in boundary against goldens captured from the TF-free parts of the reference
(tests/golden/make_goldens.py -> reference_goldens.json)."""
import hashlib
import json
import os

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
G = json.load(open(os.path.join(HERE, "golden", "reference_goldens.json")))


def test_positional_encoding_matches_reference_bit_for_bit():
    import oracle
    from sketchformer_amd import engine
    for d, want in G["positional_encoding"].items():
        for fn in (lambda p, dm: engine.positional_encoding(p, dm)[None], oracle.positional_encoding):
            t = np.ascontiguousarray(fn(1000, int(d)))
            assert t.dtype == np.float32 and t.shape == (1, 1000, int(d))
            assert hashlib.sha256(t.tobytes()).hexdigest() == want["sha256"]
            for r, vals in zip(want["rows"], want["values"]):
                assert np.array_equal(np.concatenate([t[0, r, :8], t[0, r, -4:]]), np.array(vals, np.float32))
    # KAT quoted in SURVEY.md section 8(a) a3
    t = oracle.positional_encoding(1000, 128)
    assert abs(t[0, 1, 0] - 0.84147096) < 1e-7 and abs(t[0, 1, 1] - 0.54030228) < 1e-7


def _mk():
    from sketchformer_amd.utils.hparams import HParams
    return HParams(num_layers=4, d_model=128, dropout_rate=0.1, do_classification=True, optimizer="Adam", lr=0.01,
                   goal="No description")


@pytest.mark.parametrize("case", G["hparams_parse"], ids=lambda c: c["text"])
def test_hparams_parse_behaviour(case):
    h = _mk()
    if case["ok"]:
        assert h.parse(case["text"]).values() == case["values"]
    else:
        with pytest.raises(ValueError):
            h.parse(case["text"])


def test_hparams_combine_json_and_config_roundtrip(tmp_path):
    from sketchformer_amd.utils import hparams as hp
    assert hp.combine_hparams_into_one(hp.HParams(x=1, y="s"), hp.HParams(y="t", z=2.5)).values() == G["hparams_combine"]
    assert _mk().to_json(indent=2, sort_keys=True) == G["hparams_to_json_sorted"]
    p = str(tmp_path / "config.json")
    h = _mk().parse("num_layers=7")
    hp.save_config(p, h, verbose=False)
    h2 = _mk()
    hp.load_config(h2, p, verbose=False)
    assert h2.values() == h.values()
    with pytest.raises(ValueError):
        h.set_hparam("num_layers", 2.5)


def test_default_hparams_and_plugin_attributes_match_reference():
    from sketchformer_amd import models, dataloaders
    M = models.get_model_by_name("sketch-transformer-tf2")
    assert M.specific_default_hparams().values() == G["model_specific_defaults"]
    assert M.base_default_hparams().values() == G["model_base_defaults"]
    assert {"name": M.name, "quick_metrics": M.quick_metrics, "slow_metrics": M.slow_metrics} == G["model_attrs"]
    L = dataloaders.get_dataloader_by_name("stroke3-distributed")
    assert L.default_hparams().values() == G["loader_defaults"]
    assert M.default_hparams().values() == {**G["model_specific_defaults"], **G["model_base_defaults"]}
    with pytest.raises(KeyError):
        models.get_model_by_name("no-such-model")


def test_grid_tokenizer_matches_reference():
    from sketchformer_amd.utils import GridTokenizer
    tok = GridTokenizer(resolution=100)
    assert {k: getattr(tok, k) for k in ("PAD", "SEP", "SOS", "EOS", "VOCAB_SIZE")} == G["grid_tokenizer_ids"]
    for case in G["grid_tokenizer"]:
        s = np.array(case["stroke3"], dtype=np.float32)
        assert tok.encode(s).tolist() == case["tokens"]
        np.testing.assert_allclose(tok.decode(case["tokens"]), np.array(case["decoded"]), atol=1e-12)
    assert tok.encode(np.array([[.1, .2, 0], [.1, -.1, 1]], np.float32), seq_len=8).tolist()[-3:] == [0, 0, 0]


def test_loader_preprocess_matches_reference():
    from sketchformer_amd import dataloaders
    from sketchformer_amd.utils import GridTokenizer
    L = dataloaders.get_dataloader_by_name("stroke3-distributed")
    want = G["loader_preprocess"]
    raw = [np.array(r, dtype=np.float32) for r in want["raw"]]

    def mk(**over):
        hps = L.default_hparams()
        hps.parse("token_type=grid,max_seq_len=%d" % want["max_seq_len"])
        for k, v in over.items():
            hps.set_hparam(k, v)
        obj = L.__new__(L)
        obj.hps, obj.limit, obj.tokenizer = dict(hps.values()), 1000, GridTokenizer(resolution=100)
        return obj
    grid = mk().preprocess([r.copy() for r in raw], augment=False)
    assert grid.shape == np.array(want["grid_tokens"]).shape and np.array_equal(grid, np.array(want["grid_tokens"]))
    cont = mk(use_continuous_data=True).preprocess([r.copy() for r in raw], augment=False)
    assert np.array_equal(cont, np.array(want["continuous"]))       # bit for bit (float32 division like the reference)
    # use_absolute_strokes (reference :111-112): running positions instead of offsets, both output formats
    wa = G["loader_preprocess_absolute"]
    grid_abs = mk(use_absolute_strokes=True).preprocess([r.copy() for r in raw], augment=False)
    assert np.array_equal(grid_abs, np.array(wa["grid_tokens"]))
    cont_abs = mk(use_continuous_data=True, use_absolute_strokes=True).preprocess([r.copy() for r in raw], augment=False)
    assert np.array_equal(cont_abs, np.array(wa["continuous"]))
    with pytest.raises(NotImplementedError):        # the reference raises too (missing module)
        mk(shuffle_stroke=True).preprocess([r.copy() for r in raw], augment=False)


def _golden_dictionary(tmp_path):
    """The synthetic k-means dictionary of the goldens as a pickle: a sklearn KMeans carrying the stored float32 centres
    (re-running the fit is not needed and would tie the test to sklearn's RNG details)."""
    import pickle
    from sklearn.cluster import KMeans
    c = np.frombuffer(bytes.fromhex(G["dict_tokenizer_centers"]["hex"]), dtype=np.float32).reshape(-1, 2)
    assert hashlib.sha256(c.tobytes()).hexdigest() == G["dict_tokenizer_centers"]["sha256"]
    km = KMeans(n_clusters=len(c))
    km.cluster_centers_, km._n_threads, km.n_features_in_ = c.copy(), 1, 2
    path = str(tmp_path / "token_dict.pkl")
    with open(path, "wb") as f:
        pickle.dump(km, f)
    return path


def test_dictionary_tokenizer_matches_reference(tmp_path):
    """SURVEY 8(c) golden item 3: utils/tokenizer.py:16-101 - special ids, encode (SEP insertion, seq_len padding, the
    max_seq_len pad / truncate branch), decode (single, list, empty) against the reference run on the same dictionary."""
    from sketchformer_amd.utils import Tokenizer
    path = _golden_dictionary(tmp_path)
    tok = Tokenizer(path)
    assert {k: getattr(tok, k) for k in ("PAD", "SEP", "SOS", "EOS", "VOCAB_SIZE")} == G["dict_tokenizer_ids"]
    assert tok.VOCAB_SIZE == 1004                      # the vocabulary of BASELINE cfg 1 / 2
    cap = Tokenizer(path, max_seq_len=16)
    for case in G["dict_tokenizer"]:
        s = np.array(case["stroke3"], dtype=np.float32)
        assert tok.encode(s.copy()).tolist() == case["tokens"]
        assert tok.encode(s.copy(), seq_len=len(s) + 12).tolist() == case["tokens_seq_len"]
        assert cap.encode(s.copy()).tolist() == case["tokens_max16"]
        dec = tok.decode(case["tokens"])
        assert np.array_equal(np.asarray(dec, dtype=np.float64), np.array(case["decoded"]))
    lst = tok.decode([np.array(G["dict_tokenizer"][1]["tokens"]), np.array(G["dict_tokenizer"][2]["tokens"])])
    assert len(lst) == 2 and all(np.array_equal(np.asarray(a, np.float64), np.array(b))
                                 for a, b in zip(lst, G["dict_tokenizer_decode_list"]))
    assert np.array_equal(np.asarray(tok.decode([tok.SOS, tok.EOS])), np.array(G["dict_tokenizer_decode_empty"]))


def test_loader_preprocess_dictionary_and_augmentation_match_reference(tmp_path):
    """dataloaders/distributed_stroke3.py:90-160 run by the reference on the same sketches: dictionary / grid tokens and
    stroke-5 rows bit for bit, incl. the training augmentation (random_scale + utils.sketch.augment_strokes) under the
    same numpy seed; in token mode augmentation is a no-op that draws nothing."""
    from sketchformer_amd import dataloaders
    from sketchformer_amd.utils import GridTokenizer, Tokenizer
    L = dataloaders.get_dataloader_by_name("stroke3-distributed")
    want = G["loader_preprocess2"]
    raw = [np.array(r, dtype=np.float32) for r in want["raw"]]

    def mk(tokenizer, **over):
        hps = L.default_hparams()
        hps.parse("token_type=grid,max_seq_len=%d" % want["max_seq_len"])
        for k, v in over.items():
            hps.set_hparam(k, v)
        obj = L.__new__(L)
        obj.hps, obj.limit, obj.tokenizer = dict(hps.values()), 1000, tokenizer
        return obj
    f64 = lambda h: np.frombuffer(bytes.fromhex(h), dtype=np.float64).reshape(len(raw), want["max_seq_len"], 5)
    ldd = mk(Tokenizer(_golden_dictionary(tmp_path)), token_type="dictionary")
    ldg = mk(GridTokenizer(resolution=100))
    ldc = mk(None, use_continuous_data=True)
    for fn in ("preprocess", "preprocess_per_sketch"):            # the block path and the per-sketch definition
        assert np.array_equal(getattr(ldd, fn)([r.copy() for r in raw], augment=False), np.array(want["dict_tokens"])), fn
        assert np.array_equal(getattr(ldg, fn)([r.copy() for r in raw], augment=False), np.array(want["grid_tokens"])), fn
        got = getattr(ldc, fn)([r.copy() for r in raw], augment=False)
        assert got.dtype == np.float64 and np.array_equal(got, f64(want["continuous_hex"])), fn
        np.random.seed(want["continuous_aug_seed"])
        got = getattr(ldc, fn)([r.copy() for r in raw], augment=True)
        assert np.array_equal(got, f64(want["continuous_aug_hex"])), fn
        np.random.seed(99)
        assert np.array_equal(getattr(ldg, fn)([r.copy() for r in raw], augment=True), np.array(want["grid_tokens"]))
        after = np.random.random()
        np.random.seed(99)
        assert after == np.random.random() and want["grid_aug_equals_plain"]


def test_augment_strokes_matches_reference():
    """utils/sketch.py:127-149 under the reference's own random stream (one draw per point)."""
    from sketchformer_amd.dataloaders.distributed_stroke3 import augment_strokes
    for case in G["augment_strokes"]:
        s = np.frombuffer(bytes.fromhex(case["stroke3_hex"]), dtype=np.float32).reshape(case["n"], 3)
        np.random.seed(case["seed"])
        got = augment_strokes(s.copy(), case["prob"], np.random.random(size=case["n"]))
        want = np.frombuffer(bytes.fromhex(case["result_hex"]), dtype=np.dtype(case["dtype"])).reshape(case["shape"])
        assert list(got.shape) == case["shape"] and np.array_equal(got.astype(want.dtype), want), case["n"]
        assert got.shape[0] < case["n"] or case["prob"] < 0.5 or case["n"] < 8


def test_default_hparams_equal_readme_listing():
    """SURVEY 8(c) golden item 6: the --help-hps listing of the reference's README.md:56-97."""
    from sketchformer_amd import models, dataloaders
    M = models.get_model_by_name("sketch-transformer-tf2")
    L = dataloaders.get_dataloader_by_name("stroke3-distributed")
    R = G["readme_help_hps"]
    assert M.base_default_hparams().values() == R["base"]
    assert M.specific_default_hparams().values() == R["model"]
    assert L.default_hparams().values() == R["loader"]


def test_chunk_loader_end_to_end(tmp_path):
    """stroke3-distributed on a synthetic chunk directory: shapes / dtypes of what the train step receives."""
    from sketchformer_amd import dataloaders
    rng = np.random.RandomState(0)

    def sketches(n):
        out = np.empty(n, dtype=object)
        for i in range(n):
            m = rng.randint(5, 40)
            s = np.zeros((m, 3), np.float32)
            s[:, :2] = rng.randint(-20, 20, (m, 2))
            s[:, 2] = rng.rand(m) < 0.2
            s[-1, 2] = 1
            out[i] = s
        return out
    for name, n in (("train_000", 50), ("train_001", 30), ("valid", 20), ("test", 20)):
        np.savez(str(tmp_path / (name + ".npz")), x=sketches(n), y=rng.randint(0, 3, n))
    np.savez(str(tmp_path / "meta.npz"), n_classes=3, n_samples_train=80, class_names=np.array(["a", "b", "c"]), std=1.0)
    L = dataloaders.get_dataloader_by_name("stroke3-distributed")
    ld = L(L.parse_hparams("token_type=grid,max_seq_len=64"), str(tmp_path))
    assert ld.n_classes == 3 and ld.n_samples == 80 and ld.tokenizer.VOCAB_SIZE == 10004
    it = ld.batch_iterator("train", 16, stop_at_end_of_split=False)
    for _ in range(8):                                     # crosses chunk boundaries
        x, y = next(it)
        assert x.shape == (16, 64) and x.dtype == np.int64 and y.shape == (16, 1)
        assert (x[:, 0] == ld.tokenizer.SOS).all()
    n = sum(len(x) for x, _ in ld.batch_iterator("valid", 8, stop_at_end_of_split=True))
    assert n == 20
    ldc = L(L.parse_hparams("use_continuous_data=true,max_seq_len=64"), str(tmp_path))
    x, y = next(ldc.batch_iterator("test", 4, stop_at_end_of_split=True))
    assert x.shape == (4, 64, 5) and (x[:, -1, 4] == 1).all()


def test_schedules_k1():
    """K1: WarmupDecay(128, 5000): lr(0)=0, lr(1)=2.5e-7, lr(5000)=1.25e-3, lr(20000)=6.25e-4."""
    import oracle
    from sketchformer_amd.builders.schedulers import WarmupDecay, StepDecay
    s = WarmupDecay(128, warmup_steps=5000)
    for step, want in ((0, 0.0), (1, 2.5e-7), (5000, 1.25e-3), (20000, 6.25e-4)):
        assert abs(float(s(step)) - want) <= 1e-6 * max(want, 1e-12)
        assert float(s(step)) == float(oracle.warmup_decay(step, 128, 5000))
    assert StepDecay(0.01, decay_rate=0.5, decay_steps=5000)(12000) == 0.01 * 0.25
    assert StepDecay(0.01, decay_rate=0.5, decay_steps=10)(10 ** 6) == 0.01 * 1e-2


def test_synthetic_batches_shape_contract():
    from sketchformer_amd import synthetic
    x, y = synthetic.token_batch(32, 200, 1004, 345, seed=0)
    assert x.shape == (32, 200) and x.dtype == np.int64 and y.shape == (32, 1) and y.dtype == np.int64
    assert (x[:, 0] == 1002).all() and x.max() <= 1003 and 0.3 < (x == 0).mean() < 0.8
    lens = (x != 0).sum(1)
    assert all(x[b, lens[b] - 1] == 1003 for b in range(32) if lens[b] < 200)      # EOS closes non-truncated rows
    xf, _ = synthetic.token_batch(4, 200, 1004, 345, seed=0, full=True)
    assert (xf != 0).all()
    c, _ = synthetic.continuous_batch(8, 200, 345, seed=1)
    assert c.shape == (8, 200, 5) and c.dtype == np.float32 and (c[:, -1, 4] == 1).all()
    assert np.allclose(c[..., 2:].sum(-1)[c[..., 4] == 0], 1.0)


def _random_sketches(rng, n_s, lo, hi):
    data = []
    for k in range(n_s):
        n = rng.randint(lo, hi)
        s = np.zeros((n, 3), np.float32)
        s[:, :2] = rng.randint(-30, 30, size=(n, 2))
        s[:, 2] = rng.rand(n) < 0.15
        s[-1, 2] = 1
        if k % 3 == 0:
            s[-1, 2] = 0                 # unfinished last stroke: points after the last pen lift are dropped
        if k % 7 == 0:
            s[:, 2] = 0                  # no pen lift at all
        if k % 11 == 0:
            s[rng.randint(n), 0] = 5000  # hits the +-1000 clamp
        data.append(s)
    return data


@pytest.mark.parametrize("mode", ["grid", "continuous", "dictionary"])
def test_block_preprocessing_is_bit_identical_to_per_sketch_pipeline(mode, tmp_path):
    """the loader preprocesses a whole chunk with array operations (the reference loops over
    sketches in Python, ~14k sketches/s/core, below what one GPU consumes); outputs are bit-identical to the per-sketch
    pipeline that the reference goldens pin (test_loader_preprocess_matches_reference), incl. augmentation draws."""
    import pickle
    from sklearn.cluster import KMeans
    from sketchformer_amd import dataloaders
    from sketchformer_amd.utils import GridTokenizer, Tokenizer
    L = dataloaders.get_dataloader_by_name("stroke3-distributed")
    rng = np.random.RandomState(0)
    data = _random_sketches(rng, 700, 1, 260)
    obj = L.__new__(L)
    obj.hps = dict(L.default_hparams().values())
    obj.hps.update(token_type="grid" if mode != "dictionary" else "dictionary", use_continuous_data=mode == "continuous")
    obj.limit = 1000
    if mode == "dictionary":
        km = KMeans(n_clusters=64, n_init=1, max_iter=3, random_state=0).fit(rng.normal(0, 0.1, size=(2000, 2)))
        with open(tmp_path / "dict.pkl", "wb") as f:
            pickle.dump(km, f)
        obj.tokenizer = Tokenizer(str(tmp_path / "dict.pkl"))
        assert obj.tokenizer.VOCAB_SIZE == 68
        # sklearn's predict (what the reference calls) == explicit nearest centre
        p = rng.normal(0, 0.1, size=(500, 2))
        d2 = ((p[:, None, :] - km.cluster_centers_[None]) ** 2).sum(-1)
        assert np.array_equal(obj.tokenizer.nearest_center(p[:, 0], p[:, 1]), d2.argmin(1))
    else:
        obj.tokenizer = GridTokenizer(resolution=100)
    a = obj.preprocess_per_sketch([d.copy() for d in data])
    b = obj.preprocess([d.copy() for d in data])
    assert a.dtype == b.dtype and a.shape == b.shape and np.array_equal(a, b)
    np.random.seed(3)
    a = obj.preprocess_per_sketch([d.copy() for d in data[:300]], augment=True)
    np.random.seed(3)
    b = obj.preprocess([d.copy() for d in data[:300]], augment=True)
    assert np.array_equal(a, b)
