"""Oracle vs the independent torch-autograd witness (CPU, float64)."""
import numpy as np
import pytest

import oracle
from sketchformer_amd import synthetic
import witness_torch


def _tiny(continuous=False, attn_version=1, blind=True, class_buffer_layers=0, **kw):
    return oracle.Config(**kw, num_layers=2, d_model=16, dff=32, num_heads=4, dropout_rate=0.1, lowerdim=8,
                         attn_version=attn_version, vocab_size=24, n_classes=5, seq_len=12,
                         continuous=continuous, blind_decoder_mask=blind, max_pos=32,
                         class_buffer_layers=class_buffer_layers, class_dropout=0.2)


def _drops(cfg, B, seed=1):
    rng = np.random.RandomState(seed)
    out = {}
    for name, tag in oracle.dropout_sites(cfg):
        if tag == "cls":
            out[name] = rng.rand(B, cfg.lowerdim) >= cfg.class_dropout
            continue
        L = cfg.seq_len if tag == "enc" else cfg.seq_len - 1
        out[name] = rng.rand(B, L, cfg.d_model) >= cfg.dropout_rate
    return out


@pytest.mark.parametrize("continuous,attn_version,blind,cbuf", [(False, 1, True, 0), (False, 2, False, 0), (True, 1, True, 0),
                                                                 (False, 1, True, 2), (False, 2, True, 1)])
def test_oracle_matches_autograd(continuous, attn_version, blind, cbuf):
    cfg = _tiny(continuous, attn_version, blind, cbuf)
    B = 3
    if continuous:
        x, y = synthetic.continuous_batch(B, cfg.seq_len, cfg.n_classes, seed=3)
        x = x.astype(np.float64)
    else:
        x, y = synthetic.token_batch(B, cfg.seq_len, cfg.vocab_size, cfg.n_classes, seed=3)
    x[0, 6:] = 0 if not continuous else x[0, 6:]
    if continuous:
        x[0, 6:, :] = [0, 0, 0, 0, 1]
    P = oracle.init_params(cfg, seed=0)
    # make biases / LN params non-trivial so their gradients are exercised
    rng = np.random.RandomState(5)
    for k in P:
        if k.endswith(("bias", "beta", "b_attn")):
            P[k] = rng.normal(0, 0.1, P[k].shape)
        if k.endswith("gamma"):
            P[k] = 1 + rng.normal(0, 0.1, P[k].shape)
    drops = _drops(cfg, B)
    losses, out, G = oracle.loss_and_grads(P, cfg, x, x, y, drops)
    wl, wo, WG = witness_torch.loss_and_grads(P, cfg, x, x, y, drops)
    for k in ("recon_loss", "class_loss", "total_loss"):
        assert abs(losses[k] - wl[k]) < 1e-12
    np.testing.assert_allclose(out["recon"], wo["recon"], rtol=0, atol=1e-11)


@pytest.mark.parametrize("kw", [dict(do_classification=False), dict(do_reconstruction=False),
                                dict(lowerdim=0, do_classification=False, blind_decoder_mask=False),
                                dict(lowerdim=0, do_classification=False, continuous=True)],
                         ids=["no_class_head", "no_decoder", "no_bottleneck_masked", "no_bottleneck_continuous"])
def test_structural_variants_match_autograd(kw):
    """do_classification / do_reconstruction off and lowerdim=0 (models/sketchformer.py:76-108,149-181): the variable
    set shrinks accordingly and the remaining losses / gradients agree with autograd."""
    kw = dict(kw)
    continuous = kw.pop("continuous", False)
    blind = kw.pop("blind_decoder_mask", True)
    base = dict(num_layers=2, d_model=16, dff=32, num_heads=4, dropout_rate=0.1, lowerdim=8, vocab_size=24, n_classes=5,
                seq_len=12, continuous=continuous, blind_decoder_mask=blind, max_pos=32)
    base.update(kw)
    cfg = oracle.Config(**base)
    B = 3
    if continuous:
        x, y = synthetic.continuous_batch(B, cfg.seq_len, cfg.n_classes, seed=3)
        x = x.astype(np.float64)
        x[0, 6:, :] = [0, 0, 0, 0, 1]
    else:
        x, y = synthetic.token_batch(B, cfg.seq_len, cfg.vocab_size, cfg.n_classes, seed=3)
        x[0, 6:] = 0
    P = oracle.init_params(cfg, seed=0)
    names = set(P)
    assert ("classify/kernel" in names) == cfg.has_classifier
    assert ("output/kernel" in names) == cfg.do_reconstruction
    assert ("bottleneck/W_attn" in names) == cfg.has_bottleneck
    assert ("expand/kernel" in names) == (cfg.has_bottleneck and cfg.do_reconstruction)
    rng = np.random.RandomState(5)
    for k in P:
        if k.endswith(("bias", "beta", "b_attn")):
            P[k] = rng.normal(0, 0.1, P[k].shape)
    drops = _drops(cfg, B)
    losses, out, G = oracle.loss_and_grads(P, cfg, x, x, y, drops)
    wl, wo, WG = witness_torch.loss_and_grads(P, cfg, x, x, y, drops)
    assert ("recon_loss" in losses) == cfg.do_reconstruction and ("class_loss" in losses) == cfg.has_classifier
    for k in losses:
        assert abs(losses[k] - wl[k]) < 1e-12, k
    assert set(G) == set(P)
    for k in P:
        assert WG[k] is not None, k
        np.testing.assert_allclose(G[k], WG[k], rtol=0, atol=1e-11, err_msg=k)
    res, _, _, _ = oracle.train_step(oracle.TrainState.create({k: v.copy() for k, v in P.items()}), cfg, x, x, y, drops)
    assert ("recon_loss" in res) == cfg.do_reconstruction and ("class_acc" in res) == cfg.has_classifier
    assert set(G) == set(P)
    for k in P:
        assert WG[k] is not None, k
        np.testing.assert_allclose(G[k], WG[k], rtol=0, atol=1e-11, err_msg=k)


def test_finite_difference_spot():
    cfg = _tiny()
    cfg.dropout_rate = 0.0
    B = 2
    x, y = synthetic.token_batch(B, cfg.seq_len, cfg.vocab_size, cfg.n_classes, seed=7)
    P = oracle.init_params(cfg, seed=2)
    _, _, G = oracle.loss_and_grads(P, cfg, x, x, y)
    rng = np.random.RandomState(0)
    for name in ["encoder/layer0/mha/wk/kernel", "decoder/layer1/mha2/wv/kernel", "expand/kernel",
                 "bottleneck/V_attn", "decoder/layer0/layernorm2/gamma", "encoder/embedding"]:
        idx = tuple(rng.randint(0, s) for s in P[name].shape)
        if name == "encoder/embedding":
            idx = (int(x[0, 1]), 3)
        eps = 1e-6
        old = P[name][idx]
        P[name][idx] = old + eps
        lp = oracle.loss_and_grads(P, cfg, x, x, y, want_grads=False)[0]["total_loss"]
        P[name][idx] = old - eps
        lm = oracle.loss_and_grads(P, cfg, x, x, y, want_grads=False)[0]["total_loss"]
        P[name][idx] = old
        fd = (lp - lm) / (2 * eps)
        assert abs(fd - G[name][idx]) < 1e-7 * max(1.0, abs(fd)), (name, fd, G[name][idx])


def test_torch_restatement_train_step_matches_numpy_oracle():
    """The PyTorch-CPU restatement bench.py times as the CPU baseline (oracle/torch_restatement.py: autograd backward, Keras-Adam
    with the WarmupDecay schedule on the pre-increment counter) walks the same 3-step trajectory as the numpy oracle."""
    import torch
    from oracle import torch_restatement as tr
    cfg = oracle.Config(seq_len=12, d_model=32, num_heads=4, dff=64, num_layers=2, vocab_size=20, n_classes=5, lowerdim=16,
                        dropout_rate=0.1)
    rng = np.random.RandomState(0)
    P = oracle.init_params(cfg, 3, np.float64)
    st_np = oracle.TrainState.create({k: v.copy() for k, v in P.items()})
    st_t = tr.TorchTrainState(P, dtype=torch.float64)
    st_np.iterations = st_t.iterations = 2000
    for step in range(3):
        x = rng.randint(1, cfg.vocab_size, size=(3, cfg.seq_len))
        x[0, 7:] = 0
        y = rng.randint(0, cfg.n_classes, size=(3, 1))
        drops = {n: rng.rand(3, cfg.seq_len if t == "enc" else cfg.seq_len - 1, cfg.d_model) >= cfg.dropout_rate
                 for n, t in oracle.dropout_sites(cfg)}
        _, losses, _, _ = oracle.train_step(st_np, cfg, x, x, y, drops)
        got = tr.train_step(st_t, cfg, x, x, y, drops)
        # (the two restatements scale kept units by 1/(1-rate) in different association orders: ~1e-8 relative)
        assert abs(got["total_loss"] - losses["total_loss"]) < 1e-7 * abs(losses["total_loss"])
    assert st_t.iterations == st_np.iterations == 2003
    for k in P:
        if k.endswith("wk/bias"):          # analytically zero gradient: Adam amplifies rounding noise there
            continue
        moved = np.abs(st_np.params[k] - P[k]).max()
        assert np.abs(st_t.P[k].detach().numpy() - st_np.params[k]).max() < 1e-4 * max(moved, 1e-12) + 1e-12, k
