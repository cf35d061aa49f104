"""The reference's plugin surface on the GPU: registry -> Model(hps, dataset, out_dir, id) -> train() /
train_on_batch / checkpoint resume / encoder-side inference, plus the builders front-ends against the oracle."""
import numpy as np
import pytest
import torch

import oracle

pytestmark = pytest.mark.gpu

SMALL = "num_layers=2,d_model=64,dff=128,num_heads=4,lowerdim=32,dropout_rate=0.1"
DATA = "max_seq_len=24,vocab_size=52,n_classes=7,n_samples=64"


def _build(tmp_path, exp_id="t0", base="batch_size=8,num_epochs=1,log_every=4", specific=SMALL):
    from sketchformer_amd import models, dataloaders
    Model = models.get_model_by_name("sketch-transformer-tf2")
    Loader = dataloaders.get_dataloader_by_name("stroke3-synthetic")
    dataset = Loader(Loader.parse_hparams(DATA), None)
    model = Model(Model.parse_hparams(base=base, specific=specific), dataset, str(tmp_path), exp_id)
    return model, dataset


def test_train_loop_and_metric_contract(tmp_path, capsys):
    model, dataset = _build(tmp_path)
    assert sorted(p.name for p in (tmp_path / "sketch-transformer-tf2-t0").iterdir()) == ["plots", "tmp", "weights"]
    assert model.batches_per_epoch == 8
    model.train()
    assert model.current_step == 8 and model.engine.iterations == 8
    out = capsys.readouterr().out
    assert "Epoch 0 Batch 4/8|recon_loss=" in out and "|total_loss=" in out
    # end of epoch: the slow metrics are computed, plotted and appended to the log like in the reference (core/models.py:203-235)
    assert "val-clas-acc=" in out and "sketch-reconstruction=" in out, out
    # status_report returns the log string on every call, also on steps that print nothing
    assert model.status_report().startswith("Epoch 1 Batch 0/8|recon_loss=")
    for name in ("recon_loss", "recon_acc", "class_loss", "class_acc", "total_loss"):
        h = model.quick_metrics[name].history
        assert len(h) == 8 and np.isfinite(h).all()
    # Keras running means: history[k] is the mean of the first k+1 step values
    first = model.quick_metrics["total_loss"].history[0]
    assert 2.0 < first < 9.0          # ~ ln(52)*(non-pad fraction) + ln(7) at random init
    # safety checkpoints every 4 steps (safety_save=.5), fixed every 8
    w = sorted(p.name for p in (tmp_path / "sketch-transformer-tf2-t0" / "weights").iterdir())
    assert w == ["ckpt-1.pt", "ckpt-2.pt", "step7.pt"]


def test_train_on_batch_matches_oracle_running_metrics(tmp_path):
    model, dataset = _build(tmp_path, "t1")
    eng = model.engine
    ocfg = oracle.Config(num_layers=2, d_model=64, dff=128, num_heads=4, dropout_rate=0.0, lowerdim=32, vocab_size=52,
                         n_classes=7, seq_len=24)
    eng.cfg.dropout_rate = 0.0                       # parity run: dropout off (TF's RNG stream cannot be reproduced)
    from sketchformer_amd import engine as E
    import ctypes as C
    model.engine = eng = E.TrainEngine(eng.cfg, init_seed=3)
    st = oracle.TrainState.create({k: v.astype(np.float64) for k, v in eng.state_dict_numpy().items()})
    it = dataset.batch_iterator("train", 8, False)
    for _ in range(3):
        batch = next(it)
        got = model.train_on_batch(batch)
        want, _, _, _ = oracle.train_step(st, ocfg, batch[0], batch[0], batch[1])
        assert set(got) == set(want)
        for k in want:
            assert abs(got[k] - want[k]) < 1e-4 * max(1.0, abs(want[k])), (k, got[k], want[k])


def test_checkpoint_resume_continues_bit_exactly(tmp_path):
    # (dropout off: its key derives from the experiment id, and the two models below are two experiments)
    nodrop = SMALL.replace("dropout_rate=0.1", "dropout_rate=0.0")
    a, dataset = _build(tmp_path, "ra", base="batch_size=8,num_epochs=1,log_every=100", specific=nodrop)
    batches = [next(dataset.batch_iterator("train", 8, False)) for _ in range(1)] * 4
    for b in batches[:2]:
        a.train_on_batch(b)
    a.current_step = 2
    a._save(str(tmp_path / "mid.pt"))
    for b in batches[2:]:
        a.train_on_batch(b)
    b_model, _ = _build(tmp_path, "rb", base="batch_size=8,num_epochs=1,log_every=100", specific=nodrop)
    b_model.restore_checkpoint_if_exists(str(tmp_path / "mid.pt"))
    assert b_model.current_step == 2 and b_model.engine.iterations == 2
    for b in batches[2:]:
        b_model.train_on_batch(b)
    a.engine.synchronize(); b_model.engine.synchronize()
    # bit-exact: every reduction of the step has a fixed order (the embedding gradient too since round 2: stable sort of the
    # positions, split ids combined in chunk order)
    assert torch.equal(a.engine.adam_m, b_model.engine.adam_m) and torch.equal(a.engine.adam_v, b_model.engine.adam_v)
    assert torch.equal(a.engine.params, b_model.engine.params)


def test_encoder_side_inference_api(tmp_path):
    model, dataset = _build(tmp_path, "inf")
    x, _ = next(dataset.batch_iterator("valid", 5, True))
    out = model.predict_class(x)
    assert out["class"].shape == (5,) and out["embedding"].shape == (5, 64) and out["enc_output"].shape == (5, 24, 64)
    P = {k: v.astype(np.float64) for k, v in model.engine.state_dict_numpy().items()}
    ocfg = oracle.Config(num_layers=2, d_model=64, dff=128, num_heads=4, lowerdim=32, vocab_size=52, n_classes=7, seq_len=24)
    pad = np.zeros((8, 24), np.int64); pad[:5] = x
    ref, _ = oracle.forward(P, ocfg, pad, pad[:, :-1], training=False)
    assert np.abs(out["embedding"] - ref["embedding"][:5]).max() < 1e-4
    enc = oracle.encode_from_seq(P, ocfg, x)
    assert np.abs(out["enc_output"] - enc["enc_output"]).max() < 1e-4
    assert np.array_equal(out["class"], enc["class"].argmax(-1))


def test_predict_greedy_reconstruction_matches_oracle(tmp_path):
    """predict / predict_from_embedding (models/sketchformer.py:201-311) through the plugin: token sequences of the
    KV-cached device decode are identical to the naive re-run-the-decoder restatement."""
    model, dataset = _build(tmp_path, "pred")
    x, _ = next(dataset.batch_iterator("valid", 5, True))
    tok = dataset.tokenizer
    P = {k: v.astype(np.float64) for k, v in model.engine.state_dict_numpy().items()}
    ocfg = oracle.Config(num_layers=2, d_model=64, dff=128, num_heads=4, lowerdim=32, vocab_size=52, n_classes=7, seq_len=24)
    want = oracle.predict(P, ocfg, x, tok.SOS, tok.EOS)
    got = model.predict(x)
    assert got["recon"].dtype == np.int32 and got["recon"][:, 0].tolist() == [tok.SOS] * 5
    assert got["recon"].shape == want["recon"].shape, (got["recon"].shape, want["recon"].shape)
    assert np.array_equal(got["recon"], want["recon"])
    assert np.array_equal(got["class"], want["class"])
    assert np.abs(got["embedding"] - want["embedding"]).max() < 1e-4
    # from a given embedding (interpolation experiments): same result as through predict
    again = model.predict_from_embedding(got["embedding"])
    assert np.array_equal(again["recon"], got["recon"]) and again["attn_weights"] is None
    d = model.make_dummy_input(None, 3, 2)
    assert d.shape == (2, 24) and d[:, :3].all() and not d[:, 3:].any()


def test_builders_front_ends_against_oracle():
    from sketchformer_amd import builders
    rng = np.random.RandomState(0)
    B, H, L, dh = 3, 4, 20, 16
    tok = rng.randint(1, 50, size=(B, L)); tok[0, 12:] = 0; tok[1, 5:] = 0
    enc_m, comb_m, dec_m = builders.utils.create_masks(torch.as_tensor(tok).cuda(), torch.as_tensor(tok[:, :-1]).cuda())
    oe, oc, od = oracle.create_masks(tok, tok[:, :-1])
    assert np.array_equal(enc_m.cpu().numpy(), oe) and np.array_equal(comb_m.cpu().numpy(), oc)
    q, k, v = (rng.randn(B, H, L - 1, dh) for _ in range(3))
    want, _, _ = oracle.sdpa_fwd(q, k, v, oc.astype(np.float64))
    got, w = builders.utils.scaled_dot_product_attention(*(torch.as_tensor(t, dtype=torch.float32).cuda() for t in (q, k, v)), comb_m)
    assert w is None and np.abs(got.cpu().numpy() - want).max() < 1e-5
    # any OTHER float mask (builders/utils.py:96-97 adds mask * -1e9 whatever the mask holds): a random 0/1 pattern per query, a
    # fractional mask (a logit lowered, not removed: 1e-9 * -1e9 = -1), a per-head mask, a per-key row broadcast over the queries
    qkv32 = [torch.as_tensor(t, dtype=torch.float32).cuda() for t in (q, k, v)]
    Lq = L - 1
    for mk in (np.round(rng.rand(B, 1, Lq, Lq)), rng.rand(B, 1, Lq, Lq) * 4e-9, np.round(rng.rand(B, H, Lq, Lq) * 0.7),
               rng.rand(1, 1, 1, Lq) * 2e-9, np.round(rng.rand(1, H, 1, Lq))):
        mk = mk.astype(np.float32)
        want_o, want_a, _ = oracle.sdpa_fwd(q, k, v, mk.astype(np.float64))
        got_o, got_a = builders.utils.scaled_dot_product_attention(*qkv32, torch.as_tensor(mk).cuda(), return_weights=True)
        assert got_o.shape == (B, H, Lq, dh) and got_a.shape == (B, H, Lq, Lq)
        assert np.abs(got_o.cpu().numpy() - want_o).max() < 1e-5 and np.abs(got_a.cpu().numpy() - want_a).max() < 1e-6, mk.shape
    # LossManager / MetricManager
    lm = builders.losses.LossManager()
    lm.add_reconstruction_loss("recon", weight=0.5)
    lm.add_sparse_categorical_crossentropy("class")
    logits = rng.randn(B, L - 1, 50)
    want_l, _ = oracle.recon_loss_fwd(tok[:, 1:], logits, 0.5)
    assert abs(float(lm.compute_loss("recon", tok[:, 1:], torch.as_tensor(logits, dtype=torch.float32).cuda())) - want_l) < 1e-5
    with pytest.raises(AssertionError):
        lm.compute_loss("nope")
    # the attention weights the reference returns beside the output (builders/utils.py:105): on request, per call or for the module
    qkv_t = [torch.as_tensor(t, dtype=torch.float32).cuda() for t in (q, k, v)]
    _, want_w, _ = oracle.sdpa_fwd(q, k, v, oc.astype(np.float64))
    got2, w2 = builders.utils.scaled_dot_product_attention(*qkv_t, comb_m, return_weights=True)
    assert torch.equal(got2, got) and w2.shape == (B, H, L - 1, L - 1) and np.abs(w2.cpu().numpy() - want_w).max() < 1e-6
    _, want_wp, _ = oracle.sdpa_fwd(q, k, v, oe[:, :, :, :L - 1].astype(np.float64))
    _, w3 = builders.utils.scaled_dot_product_attention(*qkv_t, enc_m[..., :L - 1], return_weights=True)
    assert np.abs(w3.cpu().numpy() - want_wp).max() < 1e-6
    _, w4 = builders.utils.scaled_dot_product_attention(*qkv_t, None, return_weights=True)
    assert np.abs(w4.cpu().numpy() - oracle.sdpa_fwd(q, k, v, None)[1]).max() < 1e-6
    builders.utils.RETURN_ATTENTION_WEIGHTS = True
    try:
        mha = builders.layers.transformer.MultiHeadAttention(64, 4)
        xx = torch.as_tensor(rng.randn(B, L - 1, 64), dtype=torch.float32).cuda()
        _, wm = mha(xx, xx, xx, comb_m)
        assert wm.shape == (B, 4, L - 1, L - 1) and float((wm.sum(-1) - 1).abs().max()) < 1e-5
        assert float(wm[0, 0, 3, 4:].abs().max()) == 0.0               # look-ahead: no weight on later keys
    finally:
        builders.utils.RETURN_ATTENTION_WEIGHTS = False
    # add_mae_loss / add_mse_loss / add_mean_loss (builders/losses.py:68-75): Keras MAE / MSE reduce the last axis, tf.reduce_mean everything
    a_, b_ = rng.randn(B, L, 7).astype(np.float32), rng.randn(B, L, 7).astype(np.float32)
    lm.add_mae_loss("mae", weight=2.0); lm.add_mse_loss("mse"); lm.add_mean_loss("mean", weight=0.25)
    ta, tb = torch.as_tensor(a_).cuda(), torch.as_tensor(b_).cuda()
    mae, mse = lm.compute_loss("mae", ta, tb), lm.compute_loss("mse", ta, tb)
    assert mae.shape == (B, L) and np.abs(mae.cpu().numpy() - 2.0 * np.abs(b_ - a_).mean(-1)).max() < 1e-6
    assert mse.shape == (B, L) and np.abs(mse.cpu().numpy() - ((b_ - a_) ** 2).mean(-1)).max() < 1e-6
    assert abs(float(lm.compute_loss("mean", ta)) - 0.25 * a_.mean()) < 1e-6
    assert lm.loss_names == ["recon", "class", "mae", "mse", "mean"]
    mm = builders.keras_metrics.MetricManager()
    mm.add_mean_metric("m"); mm.add_sparse_categorical_accuracy("a")
    mm.compute("m", 1.0); mm.compute("m", 3.0); mm.compute("a", tok[:, 1:], torch.as_tensor(logits))
    assert mm.get_results_as_dict()["m"] == 2.0
    assert abs(mm.get_results_as_dict()["a"] - (logits.argmax(-1) == tok[:, 1:]).mean()) < 1e-9
    # layer objects: encoder stack forward == oracle forward with the same weights
    enc = builders.layers.transformer.Encoder(1, 64, 4, 128, 50, rate=0.1)
    P = {"encoder/embedding": enc.embedding.cpu().numpy().astype(np.float64)}
    lay = enc.enc_layers[0]
    for n, dn in (("wq", lay.mha.wq), ("wk", lay.mha.wk), ("wv", lay.mha.wv), ("dense", lay.mha.dense)):
        P["encoder/layer0/mha/%s/kernel" % n] = dn.kernel.cpu().numpy().astype(np.float64)
        P["encoder/layer0/mha/%s/bias" % n] = dn.bias.cpu().numpy().astype(np.float64)
    for n, dn in (("dense1", lay.ffn.d1), ("dense2", lay.ffn.d2)):
        P["encoder/layer0/ffn/%s/kernel" % n] = dn.kernel.cpu().numpy().astype(np.float64)
        P["encoder/layer0/ffn/%s/bias" % n] = dn.bias.cpu().numpy().astype(np.float64)
    for n in ("layernorm1", "layernorm2"):
        P["encoder/layer0/%s/gamma" % n] = np.ones(64); P["encoder/layer0/%s/beta" % n] = np.zeros(64)
    ocfg = oracle.Config(num_layers=1, d_model=64, dff=128, num_heads=4, vocab_size=50, seq_len=L)
    pos = oracle.positional_encoding(1000, 64).astype(np.float64)
    x0, _ = oracle.sketchformer_oracle._embed_fwd(P, "encoder/embedding", tok, ocfg, pos, 0.0, None)
    want_x, _ = oracle.sketchformer_oracle.encoder_layer_fwd(P, "encoder/layer0", x0, oe.astype(np.float64), 4, 0.0, {})
    got_x = enc(tok, False, enc_m)
    assert np.abs(got_x.cpu().numpy() - want_x).max() < 1e-4
    # training=True (round 4): inverted dropout at the reference's three sites of this stack (Encoder.dropout, dropout1, dropout2,
    # builders/layers/transformer.py:212-213,286), masks = the kernels' counter-based ones - the oracle is handed the same masks
    from sketchformer_amd import ops
    got_t = enc(tok, True, enc_m)
    key = ops.read_step_state(enc.drop.state)["drop_key"]
    keep = lambda site: ops.dropout_keep_mask(key, site, 0.1, B * L * 64).reshape(B, L, 64)     # noqa: E731
    x0d, _ = oracle.sketchformer_oracle._embed_fwd(P, "encoder/embedding", tok, ocfg, pos, 0.1, keep(enc.site0))
    drops = {"encoder/layer0/dropout1": keep(lay.site1), "encoder/layer0/dropout2": keep(lay.site2)}
    want_t, _ = oracle.sketchformer_oracle.encoder_layer_fwd(P, "encoder/layer0", x0d, oe.astype(np.float64), 4, 0.1, drops)
    assert np.abs(got_t.cpu().numpy() - want_t).max() < 1e-4
    assert np.abs(got_t.cpu().numpy() - want_x).max() > 1e-2                      # it did drop something
    # a stack's layer called on its own advances the shared state (fresh masks per call, never a stale or missing state), and two
    # stacks draw from different streams (independent Dropout layers of the reference)
    x0t = torch.as_tensor(x0, dtype=torch.float32).cuda()
    l1, l2 = lay(x0t, True, enc_m), lay(x0t, True, enc_m)
    assert float((l1 - l2).abs().max()) > 1e-2 and float((l1 - torch.as_tensor(want_x, dtype=torch.float32).cuda()).abs().max()) > 1e-2
    fresh = builders.layers.transformer.Encoder(1, 64, 4, 128, 50, rate=0.1).enc_layers[0]
    assert float(fresh(x0t, True, enc_m).abs().max()) > 0                          # before its stack ever ran: dropout is applied, state exists
    assert fresh.drop.state is not None and fresh.drop.seed != enc.drop.seed
    again = enc(tok, True, enc_m)                                                 # a new call draws new masks
    assert np.abs(again.cpu().numpy() - got_t.cpu().numpy()).max() > 1e-2
    assert np.abs(enc(tok, False, enc_m).cpu().numpy() - want_x).max() < 1e-4     # and inference is unchanged


def test_builders_layer_variants_against_oracle():
    """The layer front-ends the reference defines beside the defaults (round 3 refused them): SelfAttnV2 (transformer.py:80-137),
    use_continuous_input=True (:267-296: the embedding is Dense(5 -> d)), DenseExpander(feat_dim_out) (:354-376), and a Decoder that
    returns the reference's attention_weights dictionary keys."""
    from sketchformer_amd import builders
    T = builders.layers.transformer
    rng = np.random.RandomState(1)
    B, L, d = 3, 20, 64
    f64 = lambda t: t.detach().cpu().numpy().astype(np.float64)        # noqa: E731
    # SelfAttnV2 with and without the Dense(units) after the pooling
    x = rng.randn(B, L, d)
    for units in (32, None):
        sa = T.SelfAttnV2(units)
        o, a = sa(torch.as_tensor(x, dtype=torch.float32).cuda())
        P = {"bottleneck/W_attn": f64(sa.W), "bottleneck/b_attn": f64(sa.b), "bottleneck/V_attn": f64(sa.V)}
        assert sa.W.shape == (d, d) and sa.V.shape == (d, 1)
        if units:
            P["bottleneck/embeding_layer/kernel"], P["bottleneck/embeding_layer/bias"] = f64(sa.embeding_layer.kernel), f64(sa.embeding_layer.bias)
            want, wa, _ = oracle.sketchformer_oracle.self_attn_v2_fwd(P, x)
        else:
            want, wa, _ = oracle.sketchformer_oracle.self_attn_v1_fwd(P, x)
        assert o.shape == (B, units or d) == sa.compute_output_shape((B, L, d)) and a.shape == (B, L, 1)
        assert np.abs(f64(o) - want).max() < 1e-5 and np.abs(f64(a) - wa).max() < 1e-6
    # continuous input: stroke-5 rows through Dense(5 -> d), * sqrt(d), + pos
    s5 = rng.randn(B, L, 5).astype(np.float32)
    enc = T.Encoder(1, d, 4, 128, None, rate=0.0, use_continuous_input=True)
    assert enc.embedding.kernel.shape == (5, d)
    ocfg = oracle.Config(num_layers=1, d_model=d, dff=128, num_heads=4, seq_len=L, continuous=True)
    pos = oracle.positional_encoding(1000, d).astype(np.float64)
    P = {"encoder/embedding/kernel": f64(enc.embedding.kernel), "encoder/embedding/bias": f64(enc.embedding.bias)}
    lay = enc.enc_layers[0]
    for n, dn in (("mha/wq", lay.mha.wq), ("mha/wk", lay.mha.wk), ("mha/wv", lay.mha.wv), ("mha/dense", lay.mha.dense),
                  ("ffn/dense1", lay.ffn.d1), ("ffn/dense2", lay.ffn.d2)):
        P["encoder/layer0/%s/kernel" % n], P["encoder/layer0/%s/bias" % n] = f64(dn.kernel), f64(dn.bias)
    for n in ("layernorm1", "layernorm2"):
        P["encoder/layer0/%s/gamma" % n] = np.ones(d); P["encoder/layer0/%s/beta" % n] = np.zeros(d)
    x0, _ = oracle.sketchformer_oracle._embed_fwd(P, "encoder/embedding", s5.astype(np.float64), ocfg, pos, 0.0, None)
    want_x, _ = oracle.sketchformer_oracle.encoder_layer_fwd(P, "encoder/layer0", x0, np.zeros((B, 1, 1, L)), 4, 0.0, {})
    got_x = enc(torch.as_tensor(s5).cuda(), False, None)
    assert np.abs(f64(got_x) - want_x).max() < 1e-4
    # DenseExpander with the relu projection in front
    emb = rng.randn(B, d)
    ex = T.DenseExpander(L, feat_dim_out=48)
    pre = ex(torch.as_tensor(emb, dtype=torch.float32).cuda())
    proj = np.maximum(emb @ f64(ex.project_layer.kernel) + f64(ex.project_layer.bias), 0)
    want_pre = proj[:, None, :] * f64(ex.kernel)[0][None, :, None] + f64(ex.bias)[None, :, None]
    assert pre.shape == (B, L, 48) == ex.compute_output_shape((B, d)) and np.abs(f64(pre) - want_pre).max() < 1e-5
    plain = T.DenseExpander(L)
    assert plain(torch.as_tensor(emb, dtype=torch.float32).cuda()).shape == (B, L, d)
    # Decoder: the reference's attention_weights keys (values None: the fused kernel never materialises (B,H,L,L))
    dec = T.Decoder(2, d, 4, 128, 50, rate=0.1)
    tok = rng.randint(1, 50, size=(B, L))
    y, aw = dec(tok, got_x, True, None, None)
    assert y.shape == (B, L, d) and sorted(aw) == ["decoder_layer1_block1", "decoder_layer1_block2", "decoder_layer2_block1", "decoder_layer2_block2"]
    y2, _ = dec(tok, got_x, True, None, None)
    assert float((y - y2).abs().max()) > 1e-3                       # training dropout: new masks per call
    y3, _ = dec(tok, got_x, False, None, None); y4, _ = dec(tok, got_x, False, None, None)
    assert torch.equal(y3, y4)


@pytest.mark.parametrize("extra,keys", [("do_reconstruction=False", ["class_acc", "class_loss", "total_loss"]),
                                        ("do_classification=False", ["recon_acc", "recon_loss", "total_loss"]),
                                        ("lowerdim=0,do_classification=False", ["recon_acc", "recon_loss", "total_loss"])])
def test_plugin_structural_hparams(tmp_path, extra, keys):
    """models/sketchformer.py:76-108: only the heads that exist register losses / metrics; predict returns what exists."""
    from sketchformer_amd import models, dataloaders
    Model = models.get_model_by_name("sketch-transformer-tf2")
    Loader = dataloaders.get_dataloader_by_name("stroke3-synthetic")
    dataset = Loader(Loader.parse_hparams(DATA), None)
    small = SMALL.replace("lowerdim=32,", "") if "lowerdim" in extra else SMALL       # an hparam may be assigned once
    model = Model(Model.parse_hparams(base="batch_size=8,num_epochs=1,log_every=4", specific=small + "," + extra), dataset,
                  str(tmp_path), "v")
    x, y = next(dataset.batch_iterator("train", 8, False))
    res = model.train_on_batch((x, y))
    assert sorted(res) == keys and all(np.isfinite(v) for v in res.values())
    out = model.predict(x[:3])
    assert ("recon" in out) == model.hps["do_reconstruction"]
    if model.hps["do_reconstruction"]:
        assert out["recon"].shape[0] == 3 and out["recon"][:, 0].tolist() == [dataset.tokenizer.SOS] * 3
    if not model.hps["lowerdim"]:
        assert out["embedding"].shape == (3, 24, 64)
    with pytest.raises(ValueError):
        Model(Model.parse_hparams(base="batch_size=8", specific=SMALL.replace("lowerdim=32", "lowerdim=0")), dataset,
              str(tmp_path), "bad")


def test_restore_from_reference_style_tensorflow_checkpoint(tmp_path):
    """--resume <prefix> with a TensorBundle laid out like the reference's tf.train.Checkpoint: weights, Adam slots and
    step counters land in the flat buffers (read without TensorFlow), and the model computes with them."""
    from sketchformer_amd.utils import tf_checkpoint as tfc
    model, dataset = _build(tmp_path, "tf")
    ocfg = oracle.Config(num_layers=2, d_model=64, dff=128, num_heads=4, lowerdim=32, vocab_size=52, n_classes=7, seq_len=24)
    P = oracle.init_params(ocfg, seed=9, dtype=np.float32)
    keymap = tfc.reference_variable_keys(list(P))
    rng = np.random.RandomState(2)
    tensors = {}
    for n, a in P.items():
        tensors[keymap[n] + tfc.SUFFIX] = a
        tensors[keymap[n] + "/.OPTIMIZER_SLOT/optimizer/m" + tfc.SUFFIX] = (1e-3 * rng.randn(*a.shape)).astype(np.float32)
        tensors[keymap[n] + "/.OPTIMIZER_SLOT/optimizer/v" + tfc.SUFFIX] = (1e-6 * rng.rand(*a.shape)).astype(np.float32)
    tensors["optimizer/iter" + tfc.SUFFIX] = np.array(7000, dtype=np.int64)
    tensors["transformer/current_step" + tfc.SUFFIX] = np.array(6999, dtype=np.int64)
    prefix = str(tmp_path / "ckpt-3")
    tfc.write_tensor_bundle(prefix, tensors)
    model.restore_checkpoint_if_exists(prefix)
    assert model.current_step == 6999 and model.engine.iterations == 7000
    got = model.engine.state_dict_numpy()
    assert all(np.array_equal(got[n], P[n]) for n in P)
    m = model.engine.state_dict_numpy("adam_m")
    assert np.array_equal(m["output/kernel"], tensors[keymap["output/kernel"] + "/.OPTIMIZER_SLOT/optimizer/m" + tfc.SUFFIX])
    x, y = next(dataset.batch_iterator("valid", 8, True))
    ref, _ = oracle.forward({k: v.astype(np.float64) for k, v in P.items()}, ocfg, x, x[:, :-1], training=False)
    model.engine.forward(x, training=False)
    torch.cuda.synchronize()
    logits = model.engine.buffer("logits").cpu().numpy().reshape(8, 23, -1)
    assert np.abs(logits - ref["recon"]).max() < 1e-4 * np.abs(ref["recon"]).max()
    res = model.train_on_batch((x, y))                     # and training goes on from step 7000
    assert model.engine.iterations == 7001 and np.isfinite(res["total_loss"])


def test_slow_metrics_and_extract_embeddings(tmp_path):
    """the evaluation plug-ins run end to end on the device model - predictions_on_validation_set
    (greedy reconstructions + class predictions + embeddings), val-clas-acc, t-SNE / PCA projections, the reconstruction
    grid, the evaluation plot, and the extract-embeddings experiment."""
    from sketchformer_amd import experiments, metrics
    model, dataset = _build(tmp_path, "ev")
    assert set(model.slow_metrics) <= set(metrics.metrics_by_name)
    data = model.compute_predictions_on_validation_set()
    x, all_y, pred_x, pred_y, pred_z, tokenizer, plot_fp, tmp_fp, is_cont = data
    n_valid = len(dataset.get_all_data_from("valid")[0])
    assert x.shape == (32, 24) and pred_x.shape == (32, 25) and pred_y.shape == (n_valid,) and pred_z.shape == (n_valid, 64)
    assert all_y.shape == (n_valid,) and not is_cont and tokenizer is dataset.tokenizer
    chosen = {m: metrics.build_metric_by_name(m, model.hps) for m in ("val-clas-acc", "tsne", "tsne-predicted", "pca", "sketch-reconstruction")}
    model.compute_metrics_from(chosen)
    acc = chosen["val-clas-acc"].last_value
    assert acc == pytest.approx(float(np.mean(pred_y == all_y))) and 0.0 <= acc <= 1.0
    for m in ("tsne", "tsne-predicted", "pca"):
        proj = chosen[m].get_data_for_plot()
        assert proj.ndim == 2 and proj.shape[1] == 3 and np.isfinite(proj).all() and len(proj) > 10
    import os
    assert os.path.exists(chosen["sketch-reconstruction"].get_data_for_plot())
    plot = model.plot_and_send_notification_for(chosen)
    assert plot.endswith("evaluation_plots.png") and os.path.getsize(plot) > 1000
    model.clean_up_tmp_dir()
    # a failing metric is reported and swallowed, like in the reference
    class Broken(metrics.metrics_by_name["val-clas-acc"]):
        name = None
        def compute(self, input_data):
            raise RuntimeError("boom")
    b = Broken(model.hps)
    b.computation_worker(data)
    assert b.last_value == 0 and b.history == [0]
    Exp = experiments.get_experiment_by_name("extract-embeddings")
    exp = Exp(Exp.parse_hparams("n_samples_to_reconstruct=6"), "e0", str(tmp_path))
    out = np.load(exp.compute(model), allow_pickle=True)
    assert out["embeddings"].shape == (n_valid, 64) and out["pred_y"].shape == (n_valid,) and out["y"].shape == (n_valid,)
    assert len(out["sketches"]) == 6 and len(out["recon_sketches"]) == 6 and out["sketches"][0].shape[1] == 3


# ------------------------------------------------------------------ data parallelism through the plugin surface
def _dp_plugin_worker(rank, world, port, outdir, backend):
    import os
    local = rank if backend == "nccl" else 0
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(local), HSA_ENABLE_IPC_MODE_LEGACY="0")
    os.environ["SKF_TRAIN_SLOW_METRICS"] = "0"          # (the end-of-epoch slow metrics are covered by the single-rank tests)
    import faulthandler
    faulthandler.dump_traceback_later(150, exit=True)
    import json
    import torch.distributed as dist
    from sketchformer_amd import models, dataloaders, parallel
    torch.cuda.set_device(local)
    _, _, _, pg = parallel.init_from_env(backend=backend)
    Model = models.get_model_by_name("sketch-transformer-tf2")
    Loader = dataloaders.get_dataloader_by_name("stroke3-synthetic")
    dhps = Loader.parse_hparams(DATA)
    dhps.set_hparam("seed", dhps.seed + rank)                       # train.py: every rank draws its own stream
    dataset = Loader(dhps, None)
    # 8 batches per epoch, a safety checkpoint every 2 steps (4 saves -> the max_to_keep=2 rotation runs twice), fixed at 8
    model = Model(Model.parse_hparams(base="batch_size=8,num_epochs=1,log_every=4,safety_save=0.25", specific=SMALL),
                  dataset, outdir, "dp", process_group=pg)
    model.train()
    hist = {k: [float(v) for v in q.history] for k, q in model.quick_metrics.items()}
    info = {"seed": int(model.engine.cfg.seed), "hist": hist, "iters": model.engine.iterations,
            "checksum": float(model.engine.params.double().sum().item())}
    with open(os.path.join(outdir, "rank%d.json" % rank), "w") as f:
        json.dump(info, f)
    # resume: every rank restores the file rank 0 wrote, replicas are checked equal inside load_state_dict
    model2 = Model(Model.parse_hparams(base="batch_size=8,num_epochs=1,log_every=4,safety_save=0.25", specific=SMALL),
                   dataset, outdir, "dp", process_group=pg)
    model2.restore_checkpoint_if_exists("latest")
    assert model2.current_step == 7 and model2.engine.iterations == 7, (model2.current_step, model2.engine.iterations)
    # the file carries the running-metric accumulators of BOTH ranks (folded into rank 0 before the save): rank 0 restores
    # them, the other rank starts from zero, and the reduced running mean equals what train() printed at that step
    counts = model2.engine.metrics[16:21].cpu().numpy()
    assert (counts.max() > 0) == (rank == 0), (rank, counts)
    resumed = model2.engine.resolve_metrics([model2.engine.metrics_snapshot()])[0]
    info["resumed_total_loss"] = resumed["total_loss"]
    with open(os.path.join(outdir, "rank%d.json" % rank), "w") as f:
        json.dump(info, f)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("backend", ["gloo", "nccl"])
def test_data_parallel_plugin_path(tmp_path, backend):
    """train.py under torchrun, in miniature: two ranks build the plugin model over one process group and run train().
    Rank 0 alone writes checkpoints (atomic, rotated without races), ranks draw different dropout masks, replicas stay
    identical, and the printed running metrics are the all-reduced (sum, count) of both ranks."""
    import json
    import socket
    import torch.multiprocessing as mp
    if backend == "nccl" and torch.cuda.device_count() < 2:
        pytest.skip("the RCCL run needs >= 2 visible GPUs (one rank per device)")
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    mp.spawn(_dp_plugin_worker, args=(2, port, str(tmp_path), backend), nprocs=2, join=True)
    r = [json.load(open(tmp_path / ("rank%d.json" % i))) for i in range(2)]
    assert r[0]["seed"] != r[1]["seed"]                              # independent dropout masks per rank
    assert r[0]["checksum"] == r[1]["checksum"] and r[0]["iters"] == r[1]["iters"] == 8
    assert r[0]["hist"] == r[1]["hist"]                              # metrics were reduced over the ranks before use
    assert all(len(h) == 8 and np.isfinite(h).all() for h in r[0]["hist"].values())
    # ckpt-4.pt was written after train step 7 of 8: the restored, reduced running mean is the 7th history entry
    assert abs(r[0]["resumed_total_loss"] - r[0]["hist"]["total_loss"][6]) < 1e-5 * abs(r[0]["hist"]["total_loss"][6]), r[0]
    assert r[0]["resumed_total_loss"] == r[1]["resumed_total_loss"]
    w = sorted(p.name for p in (tmp_path / "sketch-transformer-tf2-dp" / "weights").iterdir())
    assert w == ["ckpt-3.pt", "ckpt-4.pt", "step7.pt"], w
