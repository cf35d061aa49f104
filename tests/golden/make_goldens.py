#!/usr/bin/env python
"""Generates tests/golden/reference_goldens.json by IMPORTING the TF-free parts of the reference
(/root/reference) in the build container.  TensorFlow and the rendering libraries are absent, so
they are stubbed in sys.modules; only numpy code paths of the reference are executed:

  * builders.utils.positional_encoding           (builders/utils.py:12-32)
  * utils.hparams.HParams parse behaviours       (utils/hparams.py:232-343, 528-567)
  * default hparams of model / base / loader     (models/sketchformer.py:25-53, core/models.py:19-31,
                                                  dataloaders/distributed_stroke3.py:14-26)
  * utils.tokenizer.GridTokenizer encode/decode  (utils/tokenizer.py:104-198)
  * DistributedStroke3DataLoader.preprocess / _cap_pad_and_convert_sketch on seeded synthetic sketches
                                                 (dataloaders/distributed_stroke3.py:90-153), grid / dictionary tokens
                                                 and stroke-5, with and without augmentation (:155-160)
  * utils.tokenizer.Tokenizer (k-means dictionary) special ids, encode / decode against a synthetic
    MiniBatchKMeans(n_clusters=1000, random_state=0) pickle    (utils/tokenizer.py:16-101)
  * utils.sketch.augment_strokes                 (utils/sketch.py:127-149)
  * the --help-hps listing of README.md:56-97 (parsed from the README text: data, not code)

Only data (inputs and expected outputs) is written - no reference source text.
Run:  python tests/golden/make_goldens.py
"""
import hashlib
import json
import os
import sys
import types

import numpy as np

REF = "/root/reference"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "reference_goldens.json")


def _stub(name, **attrs):
    m = types.ModuleType(name)
    for k, v in attrs.items():
        setattr(m, k, v)
    sys.modules[name] = m
    return m


def install_stubs():
    class _Obj(object):
        def __init__(self, *a, **k):
            pass

    keras_layers = _stub("tensorflow.keras.layers", Layer=_Obj)
    keras_opt_sched = _stub("tensorflow.keras.optimizers.schedules", LearningRateSchedule=_Obj)
    keras_opt = _stub("tensorflow.keras.optimizers", schedules=keras_opt_sched)
    keras_backend = _stub("tensorflow.keras.backend")
    keras = _stub("tensorflow.keras", layers=keras_layers, Model=_Obj, optimizers=keras_opt, backend=keras_backend)
    _stub("tensorflow", keras=keras, cast=lambda x, dtype=None: np.asarray(x, dtype=dtype), float32=np.float32,
          custom_gradient=lambda f: f, function=lambda *a, **k: (lambda f: f))
    for name in ("svgwrite", "svglib", "svglib.svglib", "reportlab", "reportlab.graphics", "rdp", "svgpathtools",
                 "slack", "h5py"):
        _stub(name)
    sys.modules["svglib.svglib"].svg2rlg = None
    sys.modules["reportlab.graphics"].renderPM = None
    sys.modules["rdp"].rdp = None
    for n in ("real", "imag", "svg2paths", "wsvg"):
        setattr(sys.modules["svgpathtools"], n, None)
    for name in ("matplotlib", "matplotlib.pyplot", "matplotlib.cm", "matplotlib.colors", "matplotlib.patches",
                 "matplotlib.gridspec", "matplotlib.ticker", "PIL"):
        try:
            __import__(name)
        except Exception:  # noqa: BLE001
            _stub(name)


def main():
    install_stubs()
    sys.path.insert(0, REF)
    out = {}

    # ---- positional encoding
    from builders.utils import positional_encoding
    pe = {}
    for d in (64, 128, 256, 512):
        t = np.asarray(positional_encoding(1000, d))
        assert t.shape == (1, 1000, d) and t.dtype == np.float32
        rows = [0, 1, 2, 199, 500, 999]
        pe[str(d)] = {"rows": rows, "values": [t[0, r, :8].tolist() + t[0, r, -4:].tolist() for r in rows],
                      "sha256": hashlib.sha256(np.ascontiguousarray(t).tobytes()).hexdigest()}
    out["positional_encoding"] = pe

    # ---- HParams behaviours
    from utils.hparams import HParams, combine_hparams_into_one
    def mk():
        return HParams(num_layers=4, d_model=128, dropout_rate=0.1, do_classification=True, optimizer="Adam",
                       lr=0.01, goal="No description")
    cases = []
    for text in ["num_layers=6", "d_model=256,dropout_rate=0.0", "do_classification=false", "do_classification=False",
                 "do_classification=0", "do_classification=1", "optimizer=sgd,lr=1e-3", "num_layers=6,,d_model=64",
                 "dropout_rate=1", "num_layers=2.5", "unknown_key=3", "num_layers=abc", "lr=", "goal=a b c",
                 "do_classification=yes", "num_layers = 8 , d_model = 32"]:
        h = mk()
        try:
            h.parse(text)
            cases.append({"text": text, "ok": True, "values": h.values()})
        except Exception as e:  # noqa: BLE001
            cases.append({"text": text, "ok": False, "error": type(e).__name__})
    out["hparams_parse"] = cases
    a, b = HParams(x=1, y="s"), HParams(y="t", z=2.5)
    out["hparams_combine"] = combine_hparams_into_one(a, b).values()
    out["hparams_to_json_sorted"] = mk().to_json(indent=2, sort_keys=True)

    # ---- default hparams
    import core.models
    import models.sketchformer as ref_model
    out["model_specific_defaults"] = ref_model.Transformer.specific_default_hparams().values()
    out["model_base_defaults"] = core.models.BaseModel.base_default_hparams().values()
    out["model_attrs"] = {"name": ref_model.Transformer.name, "quick_metrics": ref_model.Transformer.quick_metrics,
                          "slow_metrics": ref_model.Transformer.slow_metrics}
    import dataloaders
    Loader = dataloaders.get_dataloader_by_name("stroke3-distributed")
    out["loader_defaults"] = Loader.default_hparams().values()

    # ---- GridTokenizer
    import utils
    tok = utils.GridTokenizer(resolution=100)
    out["grid_tokenizer_ids"] = {"PAD": tok.PAD, "SEP": tok.SEP, "SOS": tok.SOS, "EOS": tok.EOS, "VOCAB_SIZE": tok.VOCAB_SIZE}
    rng = np.random.RandomState(0)
    samples = [np.array([[.1, .2, 0], [.1, -.1, 1], [.05, .05, 0], [0, .1, 1]], dtype=np.float32)]
    for n in (5, 17, 40):
        s = np.zeros((n, 3), dtype=np.float32)
        s[:, :2] = rng.uniform(-0.04, 0.04, size=(n, 2))
        s[:, 2] = (rng.rand(n) < 0.2)
        s[-1, 2] = 1
        samples.append(s)
    enc = []
    for s in samples:
        e = tok.encode(s)
        dec = tok.decode(e)
        enc.append({"stroke3": s.tolist(), "tokens": [int(v) for v in e], "decoded": np.asarray(dec).tolist()})
    out["grid_tokenizer"] = enc

    # ---- loader preprocessing (grid tokens and continuous), no augmentation, via the reference methods
    def make_loader(**over):
        hps = Loader.default_hparams()
        hps.parse("token_type=grid,max_seq_len=32")
        for k, v in over.items():
            hps.set_hparam(k, v)
        obj = Loader.__new__(Loader)          # skip __init__ (it needs chunk files + threads)
        obj.hps = dict(hps.values())
        obj.limit = 1000
        obj.tokenizer = utils.GridTokenizer(resolution=100)
        return obj

    raw = []
    for n in (6, 20, 60):
        s = np.zeros((n, 3), dtype=np.float32)
        s[:, :2] = rng.randint(-30, 30, size=(n, 2))
        s[:, 2] = (rng.rand(n) < 0.15)
        s[-1, 2] = 1
        raw.append(s)
    raw[1][3, 0] = 5000.0          # exercises the +-1000 clamp
    ld = make_loader()
    grid = ld.preprocess([r.copy() for r in raw], augment=False)
    ldc = make_loader(use_continuous_data=True)
    cont = ldc.preprocess([r.copy() for r in raw], augment=False)
    out["loader_preprocess"] = {"raw": [r.tolist() for r in raw], "max_seq_len": 32,
                                "grid_tokens": np.asarray(grid).tolist(), "grid_dtype": str(np.asarray(grid).dtype),
                                "continuous": np.asarray(cont).tolist(), "continuous_dtype": str(np.asarray(cont).dtype)}
    # use_absolute_strokes (dataloaders/distributed_stroke3.py:111-112 -> utils/sketch.py:169-175), both output formats
    grid_abs = make_loader(use_absolute_strokes=True).preprocess([r.copy() for r in raw], augment=False)
    cont_abs = make_loader(use_continuous_data=True, use_absolute_strokes=True).preprocess([r.copy() for r in raw], augment=False)
    out["loader_preprocess_absolute"] = {"grid_tokens": np.asarray(grid_abs).tolist(), "continuous": np.asarray(cont_abs).tolist()}

    # ---- dictionary Tokenizer (utils/tokenizer.py:16-101) against a synthetic k-means dictionary.  The dictionary is fit
    # on float32 offsets like prep_data/sketch_token/create_token_dict.py:52 makes them; the fixture stores its centres
    # so that the test does not depend on re-running the fit.
    import pickle
    import tempfile
    from sklearn.cluster import MiniBatchKMeans
    drng = np.random.RandomState(0)
    pts = np.concatenate([drng.normal(0, 0.05, size=(60000, 2)), drng.uniform(-0.5, 0.5, size=(20000, 2))]).astype(np.float32)
    km = MiniBatchKMeans(n_clusters=1000, random_state=0, n_init=1, batch_size=4096, max_iter=5).fit(pts)
    assert km.cluster_centers_.dtype == np.float32
    tmpd = tempfile.mkdtemp()
    dict_path = os.path.join(tmpd, "token_dict.pkl")
    with open(dict_path, "wb") as f:
        pickle.dump(km, f)
    dtok = utils.Tokenizer(dict_path)
    out["dict_tokenizer_ids"] = {"PAD": dtok.PAD, "SEP": dtok.SEP, "SOS": dtok.SOS, "EOS": dtok.EOS, "VOCAB_SIZE": dtok.VOCAB_SIZE}
    out["dict_tokenizer_centers"] = {"dtype": str(km.cluster_centers_.dtype), "hex": km.cluster_centers_.tobytes().hex(),
                                     "sha256": hashlib.sha256(km.cluster_centers_.tobytes()).hexdigest()}
    denc = []
    for n in (1, 4, 9, 33, 120):
        s3 = np.zeros((n, 3), dtype=np.float32)
        s3[:, :2] = drng.normal(0, 0.06, size=(n, 2))
        s3[:, 2] = drng.rand(n) < 0.2
        s3[-1, 2] = 1
        e = dtok.encode(s3.copy())
        e_pad = dtok.encode(s3.copy(), seq_len=n + 12)
        dec = dtok.decode(e)
        denc.append({"stroke3": s3.tolist(), "tokens": [int(v) for v in e], "tokens_seq_len": [int(v) for v in e_pad],
                     "decoded": np.asarray(dec, dtype=np.float64).tolist()})
    tok_cap = utils.Tokenizer(dict_path, max_seq_len=16)          # the max_seq_len branch: pad / truncate with SEP, EOS
    for item, n in zip(denc, (1, 4, 9, 33, 120)):
        item["tokens_max16"] = [int(v) for v in tok_cap.encode(np.array(item["stroke3"], dtype=np.float32))]
    out["dict_tokenizer"] = denc
    out["dict_tokenizer_decode_list"] = [np.asarray(a, dtype=np.float64).tolist() for a in
                                         dtok.decode([np.array(denc[1]["tokens"]), np.array(denc[2]["tokens"])])]
    out["dict_tokenizer_decode_empty"] = np.asarray(dtok.decode([dtok.SOS, dtok.EOS])).tolist()

    # ---- loader preprocessing with the dictionary tokenizer, and with augmentation (continuous mode only, :155-160)
    raw2 = []
    for n in (3, 25, 70, 150, 260):
        s3 = np.zeros((n, 3), dtype=np.float32)
        s3[:, :2] = rng.randint(-40, 40, size=(n, 2))
        s3[:, 2] = (rng.rand(n) < 0.12)
        s3[-1, 2] = 1
        raw2.append(s3)
    raw2[2][5, 1] = -7000.0
    ldd = make_loader(token_type="dictionary")
    ldd.tokenizer = dtok
    dict_tokens = ldd.preprocess([r.copy() for r in raw2], augment=False)
    ldg = make_loader()
    grid2 = ldg.preprocess([r.copy() for r in raw2], augment=False)
    lda = make_loader(use_continuous_data=True)
    cont2 = lda.preprocess([r.copy() for r in raw2], augment=False)
    np.random.seed(1234)
    cont_aug = lda.preprocess([r.copy() for r in raw2], augment=True)
    np.random.seed(99)
    grid_aug = ldg.preprocess([r.copy() for r in raw2], augment=True)       # token mode: augment is a no-op, no draws
    after = float(np.random.random())
    np.random.seed(99)
    assert after == float(np.random.random())
    out["loader_preprocess2"] = {"raw": [r.tolist() for r in raw2], "max_seq_len": 32,
                                 "dict_tokens": np.asarray(dict_tokens).tolist(), "grid_tokens": np.asarray(grid2).tolist(),
                                 "continuous_hex": np.asarray(cont2, dtype=np.float64).tobytes().hex(),
                                 "continuous_aug_seed": 1234,
                                 "continuous_aug_hex": np.asarray(cont_aug, dtype=np.float64).tobytes().hex(),
                                 "grid_aug_equals_plain": bool(np.array_equal(grid_aug, grid2))}

    # ---- utils.sketch.augment_strokes on its own
    import utils.sketch as ref_sketch
    aug_cases = []
    for n, prob, seed in ((40, 0.1, 0), (200, 0.5, 1), (90, 0.9, 2), (5, 0.9, 3)):
        s3 = np.zeros((n, 3), dtype=np.float32)
        s3[:, :2] = rng.normal(0, 3, size=(n, 2))
        s3[:, 2] = (rng.rand(n) < 0.08)
        np.random.seed(seed)
        res = ref_sketch.augment_strokes(s3.copy(), prob)
        aug_cases.append({"stroke3_hex": s3.tobytes().hex(), "n": n, "prob": prob, "seed": seed,
                          "dtype": str(np.asarray(res).dtype), "shape": list(np.asarray(res).shape),
                          "result_hex": np.ascontiguousarray(res).tobytes().hex()})
    out["augment_strokes"] = aug_cases

    # ---- README.md:56-97: the three default-parameter listings printed by --help-hps
    import ast
    import re
    readme = open(os.path.join(REF, "README.md")).read()
    blocks = re.findall(r"default parameters:\s*\n(\{.*?\})", readme, flags=re.S)
    assert len(blocks) == 3, len(blocks)
    out["readme_help_hps"] = {"base": ast.literal_eval(blocks[0]), "model": ast.literal_eval(blocks[1]),
                              "loader": ast.literal_eval(blocks[2])}

    def default(o):
        if isinstance(o, (np.integer,)):
            return int(o)
        if isinstance(o, (np.floating,)):
            return float(o)
        if isinstance(o, np.ndarray):
            return o.tolist()
        raise TypeError(type(o))
    with open(OUT, "w") as f:
        json.dump(out, f, indent=1, default=default, sort_keys=True)
    print("wrote", OUT, os.path.getsize(OUT), "bytes")


if __name__ == "__main__":
    main()
