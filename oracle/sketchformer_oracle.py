"""NumPy restatement of the reference's sketch-transformer-tf2 train step.

TEST INFRASTRUCTURE (see oracle/__init__.py).  Every function cites the file:line
of /root/reference it restates.  The reference delegates all arithmetic to
TensorFlow 2.1 / Keras (pinned ``tensorflow-gpu == 2.1`` in
dependencies/requirements.txt:4, absent from this image), so the TF-2.1 op
semantics are restated from their published definitions (SURVEY.md section
8(a)): Dense = x.W+b with (in,out) kernels, LayerNormalization(eps=1e-6) with
biased variance, inverted dropout, SparseCategoricalCrossentropy, Keras Adam.

Backward passes are hand derived here (the reference uses tf.GradientTape,
models/sketchformer.py:331,347) and are cross-checked against torch.autograd
in tests/test_oracle_witness.py.

All functions are dtype-polymorphic: they compute in the dtype of the
parameters (float64 for checking, float32 for the timed CPU baseline).
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field, asdict
from typing import Dict, List, Optional, Tuple

import numpy as np

__all__ = [
    "Config", "param_specs", "init_params", "positional_encoding",
    "create_padding_mask", "create_look_ahead_mask", "create_masks",
    "dense_fwd", "dense_bwd", "dropout_fwd", "dropout_bwd", "layernorm_fwd", "layernorm_bwd",
    "sdpa_fwd", "sdpa_bwd", "mha_fwd", "mha_bwd", "self_attn_v1_fwd",
    "self_attn_v1_bwd", "dense_expander_fwd", "dense_expander_bwd",
    "recon_loss_fwd", "recon_loss_bwd", "continuous_recon_loss_fwd",
    "continuous_recon_loss_bwd", "class_loss_fwd", "class_loss_bwd",
    "warmup_decay", "step_decay", "adam_update", "forward", "loss_and_grads",
    "TrainState", "train_step", "dropout_sites", "MetricState",
    "encode_from_seq", "make_dummy_input", "decode", "predict_from_embedding", "predict",
    "classify_from_embedding_fwd", "classify_from_embedding_bwd", "sgd_momentum_update",
    "RELU_MASKS", "RELU_PRE",
]


# --------------------------------------------------------------------------
# configuration / parameters
# --------------------------------------------------------------------------
@dataclass
class Config:
    """Hyper-parameters that shape the graph.

    Defaults = models/sketchformer.py:27-52 (model) plus the data-side values
    the ctor reads at models/sketchformer.py:55-61.
    """
    num_layers: int = 4
    d_model: int = 128
    dff: int = 512
    num_heads: int = 8
    dropout_rate: float = 0.1
    lowerdim: int = 256
    attn_version: int = 1
    class_weight: float = 1.0
    recon_weight: float = 1.0
    blind_decoder_mask: bool = True
    vocab_size: int = 1004          # dataset.tokenizer.VOCAB_SIZE
    n_classes: int = 345            # dataset.n_classes
    seq_len: int = 200              # dataset.hps['max_seq_len']
    continuous: bool = False        # dataset.hps['use_continuous_data']
    max_pos: int = 1000             # builders/layers/transformer.py:268,307
    class_buffer_layers: int = 0    # models/sketchformer.py:44,101-104
    class_dropout: float = 0.1      # models/sketchformer.py:45
    optimizer: str = "adam"         # models/sketchformer.py:120-126 ('adam' | 'sgd')
    do_classification: bool = True  # models/sketchformer.py:42,99 (needs lowerdim > 0)
    do_reconstruction: bool = True  # models/sketchformer.py:46,76

    @property
    def has_bottleneck(self):
        return self.lowerdim > 0

    @property
    def has_classifier(self):
        """the class head only exists inside the `if lowerdim:` block (models/sketchformer.py:96-108)"""
        return self.lowerdim > 0 and self.do_classification

    def as_dict(self):
        return asdict(self)


def param_specs(cfg: Config) -> List[Tuple[str, Tuple[int, ...], str]]:
    """Ordered (name, shape, initializer) of every trainable variable.

    Shapes follow SURVEY.md section 8(a) "Trainable-variable shapes"; the
    initialisers are the Keras defaults used at
    builders/layers/transformer.py:47-58,154-158,196-197,276-278,362-363.
    Order = forward order (the flat device buffer of the product uses it too).
    """
    d, dff, U, L = cfg.d_model, cfg.dff, cfg.lowerdim, cfg.seq_len
    specs: List[Tuple[str, Tuple[int, ...], str]] = []

    def dense(prefix, fan_in, fan_out):
        specs.append((prefix + "/kernel", (fan_in, fan_out), "glorot"))
        specs.append((prefix + "/bias", (fan_out,), "zeros"))

    def ln(prefix):
        specs.append((prefix + "/gamma", (d,), "ones"))
        specs.append((prefix + "/beta", (d,), "zeros"))

    def mha(prefix, kv_in):
        dense(prefix + "/wq", d, d)
        dense(prefix + "/wk", kv_in, d)
        dense(prefix + "/wv", kv_in, d)
        dense(prefix + "/dense", d, d)

    def ffn(prefix):
        dense(prefix + "/dense1", d, dff)
        dense(prefix + "/dense2", dff, d)

    def embed(prefix):
        if cfg.continuous:
            dense(prefix, 5, d)
        else:
            specs.append((prefix, (cfg.vocab_size, d), "uniform05"))

    embed("encoder/embedding")
    for i in range(cfg.num_layers):
        p = "encoder/layer%d" % i
        mha(p + "/mha", d)
        ffn(p + "/ffn")
        ln(p + "/layernorm1")
        ln(p + "/layernorm2")
    emb_dim = d                                            # lowerdim == 0: the decoder attends to the encoder output
    if cfg.has_bottleneck:
        if cfg.attn_version == 1:
            specs.append(("bottleneck/W_attn", (d, U), "normal05"))
            specs.append(("bottleneck/b_attn", (U,), "zeros"))
            specs.append(("bottleneck/V_attn", (U, 1), "uniform05"))
        else:
            specs.append(("bottleneck/W_attn", (d, d), "normal05"))
            specs.append(("bottleneck/b_attn", (d,), "zeros"))
            specs.append(("bottleneck/V_attn", (d, 1), "uniform05"))
            dense("bottleneck/embeding_layer", d, U)
            emb_dim = U
    if cfg.has_classifier:
        for i in range(cfg.class_buffer_layers):           # Dense(lowerdim, relu) buffers, models/sketchformer.py:101
            dense("class_buffer/%d" % i, emb_dim if i == 0 else U, U)
        dense("classify", U if cfg.class_buffer_layers else emb_dim, cfg.n_classes)
    if cfg.do_reconstruction:
        if cfg.has_bottleneck:
            dense("expand", 1, L)                          # only built (Keras: on first call) when the decoder uses it
        embed("decoder/embedding")
        for i in range(cfg.num_layers):
            p = "decoder/layer%d" % i
            mha(p + "/mha1", d)
            mha(p + "/mha2", emb_dim)
            ffn(p + "/ffn")
            ln(p + "/layernorm1")
            ln(p + "/layernorm2")
            ln(p + "/layernorm3")
        dense("output", d, 5 if cfg.continuous else cfg.vocab_size)
    return specs


def init_params(cfg: Config, seed: int = 0, dtype=np.float64) -> Dict[str, np.ndarray]:
    """Keras-default initialisation (distributionally; TF's RNG stream cannot
    be reproduced)."""
    rng = np.random.RandomState(seed)
    out = {}
    for name, shape, kind in param_specs(cfg):
        if kind == "glorot":
            lim = math.sqrt(6.0 / (shape[0] + shape[1]))
            a = rng.uniform(-lim, lim, size=shape)
        elif kind == "uniform05":
            a = rng.uniform(-0.05, 0.05, size=shape)
        elif kind == "normal05":
            a = rng.normal(0.0, 0.05, size=shape)
        elif kind == "ones":
            a = np.ones(shape)
        else:
            a = np.zeros(shape)
        out[name] = a.astype(dtype)
    return out


# --------------------------------------------------------------------------
# builders/utils.py
# --------------------------------------------------------------------------
def positional_encoding(position: int, d_model: int) -> np.ndarray:
    """builders/utils.py:12-32.  float64 numpy, ``np.float32(d_model)``
    divisor, sin on even columns, cos on odd, cast to float32; (1,P,d)."""
    pos = np.arange(position)[:, np.newaxis]
    i = np.arange(d_model)[np.newaxis, :]
    angle_rates = 1 / np.power(10000, (2 * (i // 2)) / np.float32(d_model))
    angle_rads = pos * angle_rates
    angle_rads[:, 0::2] = np.sin(angle_rads[:, 0::2])
    angle_rads[:, 1::2] = np.cos(angle_rads[:, 1::2])
    return angle_rads[np.newaxis, ...].astype(np.float32)


def create_padding_mask(seq: np.ndarray) -> np.ndarray:
    """builders/utils.py:35-43 -> (B,1,1,L) float {0,1}."""
    if seq.ndim < 3:
        m = (seq == 0).astype(np.float32)
    else:
        m = (seq[..., -1] == 1).astype(np.float32)
    return m[:, np.newaxis, np.newaxis, :]


def create_look_ahead_mask(size: int) -> np.ndarray:
    """builders/utils.py:46-49: strict upper triangle of ones."""
    return (1 - np.tril(np.ones((size, size)))).astype(np.float32)


def create_masks(inp, tar):
    """builders/utils.py:52-68."""
    enc_padding_mask = create_padding_mask(inp)
    dec_padding_mask = create_padding_mask(inp)
    look_ahead_mask = create_look_ahead_mask(tar.shape[1])
    dec_target_padding_mask = create_padding_mask(tar)
    combined_mask = np.maximum(dec_target_padding_mask, look_ahead_mask)
    return enc_padding_mask, combined_mask, dec_padding_mask


# --------------------------------------------------------------------------
# primitive layers (Keras semantics)
# --------------------------------------------------------------------------
def dense_fwd(x, W, b, act=None):
    """tf.keras.layers.Dense: x.W + b, kernel (in,out)."""
    y = x @ W + b
    if act == "relu":
        y = np.maximum(y, 0)
    elif act == "tanh":
        y = np.tanh(y)
    return y, (x, W, y, act)


def dense_bwd(dy, cache):
    x, W, y, act = cache
    if act == "relu":
        dy = dy * (y > 0)
    elif act == "tanh":
        dy = dy * (1 - y * y)
    x2 = x.reshape(-1, x.shape[-1])
    dy2 = dy.reshape(-1, dy.shape[-1])
    dW = x2.T @ dy2
    db = dy2.sum(0)
    dx = dy @ W.T
    return dx, dW, db


def layernorm_fwd(x, gamma, beta, eps=1e-6):
    """tf.keras.layers.LayerNormalization(epsilon=1e-6) over the last axis,
    biased variance (builders/layers/transformer.py:209-210,239-241)."""
    mean = x.mean(-1, keepdims=True)
    var = ((x - mean) ** 2).mean(-1, keepdims=True)
    rstd = 1.0 / np.sqrt(var + eps)
    xhat = (x - mean) * rstd
    return xhat * gamma + beta, (xhat, rstd, gamma)


def layernorm_bwd(dy, cache):
    xhat, rstd, gamma = cache
    g = dy * gamma
    dx = rstd * (g - g.mean(-1, keepdims=True) - xhat * (g * xhat).mean(-1, keepdims=True))
    red = tuple(range(dy.ndim - 1))
    return dx, (dy * xhat).sum(red), dy.sum(red)


def dropout_fwd(x, keep, rate):
    """tf.keras.layers.Dropout in training mode: inverted dropout.  ``keep``
    is an externally supplied boolean keep-mask (None -> identity)."""
    if keep is None or rate == 0.0:
        return x
    return x * (1.0 / (1.0 - rate)) * keep


dropout_bwd = dropout_fwd  # linear, same mask


def sdpa_fwd(q, k, v, mask):
    """builders/utils.py:71-105.  Divide AFTER the matmul, additive
    ``mask * -1e9``, softmax over keys."""
    dk = np.asarray(k.shape[-1], dtype=q.dtype)
    logits = (q @ np.swapaxes(k, -1, -2)) / np.sqrt(dk)
    if mask is not None:
        logits = logits + (mask.astype(q.dtype) * q.dtype.type(-1e9))
    m = logits.max(-1, keepdims=True)
    e = np.exp(logits - m)
    a = e / e.sum(-1, keepdims=True)
    out = a @ v
    return out, a, (q, k, v, a)


def sdpa_bwd(dout, cache):
    q, k, v, a = cache
    da = dout @ np.swapaxes(v, -1, -2)
    dv = np.swapaxes(a, -1, -2) @ dout
    ds = a * (da - (da * a).sum(-1, keepdims=True))
    ds = ds / np.sqrt(np.asarray(k.shape[-1], dtype=q.dtype))
    dq = ds @ k
    dk = np.swapaxes(ds, -1, -2) @ q
    return dq, dk, dv


def _split_heads(x, H):
    B, L, d = x.shape
    return x.reshape(B, L, H, d // H).transpose(0, 2, 1, 3)


def _merge_heads(x):
    B, H, L, dh = x.shape
    return x.transpose(0, 2, 1, 3).reshape(B, L, H * dh)


def mha_fwd(P, prefix, v, k, q, mask, H):
    """builders/layers/transformer.py:167-191 (argument order v, k, q)."""
    qp, cq = dense_fwd(q, P[prefix + "/wq/kernel"], P[prefix + "/wq/bias"])
    kp, ck = dense_fwd(k, P[prefix + "/wk/kernel"], P[prefix + "/wk/bias"])
    vp, cv = dense_fwd(v, P[prefix + "/wv/kernel"], P[prefix + "/wv/bias"])
    o, attn, cs = sdpa_fwd(_split_heads(qp, H), _split_heads(kp, H), _split_heads(vp, H), mask)
    concat = _merge_heads(o)
    out, co = dense_fwd(concat, P[prefix + "/dense/kernel"], P[prefix + "/dense/bias"])
    return out, attn, (cq, ck, cv, cs, co, H, prefix)


def mha_bwd(dout, cache, G):
    cq, ck, cv, cs, co, H, prefix = cache
    dconcat, G[prefix + "/dense/kernel"], G[prefix + "/dense/bias"] = dense_bwd(dout, co)
    dq_, dk_, dv_ = sdpa_bwd(_split_heads(dconcat, H), cs)
    dq, G[prefix + "/wq/kernel"], G[prefix + "/wq/bias"] = dense_bwd(_merge_heads(dq_), cq)
    dk, G[prefix + "/wk/kernel"], G[prefix + "/wk/bias"] = dense_bwd(_merge_heads(dk_), ck)
    dv, G[prefix + "/wv/kernel"], G[prefix + "/wv/bias"] = dense_bwd(_merge_heads(dv_), cv)
    return dv, dk, dq


# ReLU branch bookkeeping for parity tests.  relu(z) is not differentiable at z = 0: a pre-activation within rounding
# distance of 0 takes one branch in float64 and possibly the other in the device's fp32, and that single unit then
# changes a whole column of dW1 (observed at the BASELINE sizes: ~1 unit in 10^6).  Like the dropout keep-masks, the
# branch can therefore be SUPPLIED: RELU_MASKS[prefix] = bool array (True = unit active) overrides `pre > 0`, and
# RELU_PRE records every pre-activation so that a test can check that overridden units really sit on the kink.
RELU_MASKS: Dict[str, np.ndarray] = {}
RELU_PRE: Dict[str, np.ndarray] = {}


def ffn_fwd(P, prefix, x):
    """builders/layers/transformer.py:194-198."""
    W1, b1 = P[prefix + "/dense1/kernel"], P[prefix + "/dense1/bias"]
    pre = x @ W1 + b1
    mask = RELU_MASKS.get(prefix)
    if mask is None:
        mask = pre > 0
    RELU_PRE[prefix] = pre
    h = pre * mask                                          # == np.maximum(pre, 0) when mask == (pre > 0)
    y, c2 = dense_fwd(h, P[prefix + "/dense2/kernel"], P[prefix + "/dense2/bias"])
    return y, ((x, W1, mask), c2, prefix)


def ffn_bwd(dy, cache, G):
    (x, W1, mask), c2, prefix = cache
    dh, G[prefix + "/dense2/kernel"], G[prefix + "/dense2/bias"] = dense_bwd(dy, c2)
    dh = dh * mask
    G[prefix + "/dense1/kernel"] = x.reshape(-1, x.shape[-1]).T @ dh.reshape(-1, dh.shape[-1])
    G[prefix + "/dense1/bias"] = dh.reshape(-1, dh.shape[-1]).sum(0)
    return dh @ W1.T


def _ln_fwd(P, prefix, x):
    return layernorm_fwd(x, P[prefix + "/gamma"], P[prefix + "/beta"])


def _ln_bwd(dy, cache, prefix, G):
    dx, G[prefix + "/gamma"], G[prefix + "/beta"] = layernorm_bwd(dy, cache)
    return dx


def encoder_layer_fwd(P, prefix, x, mask, H, rate, drops):
    """builders/layers/transformer.py:215-224 (post-LN)."""
    attn_out, _, cm = mha_fwd(P, prefix + "/mha", x, x, x, mask, H)
    attn_out = dropout_fwd(attn_out, drops.get(prefix + "/dropout1"), rate)
    out1, cl1 = _ln_fwd(P, prefix + "/layernorm1", x + attn_out)
    f, cf = ffn_fwd(P, prefix + "/ffn", out1)
    f = dropout_fwd(f, drops.get(prefix + "/dropout2"), rate)
    out2, cl2 = _ln_fwd(P, prefix + "/layernorm2", out1 + f)
    return out2, (cm, cl1, cf, cl2, prefix, rate, drops)


def encoder_layer_bwd(dout, cache, G):
    cm, cl1, cf, cl2, prefix, rate, drops = cache
    dz2 = _ln_bwd(dout, cl2, prefix + "/layernorm2", G)
    dout1 = dz2 + ffn_bwd(dropout_bwd(dz2, drops.get(prefix + "/dropout2"), rate), cf, G)
    dz1 = _ln_bwd(dout1, cl1, prefix + "/layernorm1", G)
    dv, dk, dq = mha_bwd(dropout_bwd(dz1, drops.get(prefix + "/dropout1"), rate), cm, G)
    return dz1 + dv + dk + dq


def decoder_layer_fwd(P, prefix, x, enc_output, look_ahead_mask, padding_mask, H, rate, drops):
    """builders/layers/transformer.py:245-262."""
    attn1, w1, cm1 = mha_fwd(P, prefix + "/mha1", x, x, x, look_ahead_mask, H)
    attn1 = dropout_fwd(attn1, drops.get(prefix + "/dropout1"), rate)
    out1, cl1 = _ln_fwd(P, prefix + "/layernorm1", attn1 + x)
    attn2, w2, cm2 = mha_fwd(P, prefix + "/mha2", enc_output, enc_output, out1, padding_mask, H)
    attn2 = dropout_fwd(attn2, drops.get(prefix + "/dropout2"), rate)
    out2, cl2 = _ln_fwd(P, prefix + "/layernorm2", attn2 + out1)
    f, cf = ffn_fwd(P, prefix + "/ffn", out2)
    f = dropout_fwd(f, drops.get(prefix + "/dropout3"), rate)
    out3, cl3 = _ln_fwd(P, prefix + "/layernorm3", f + out2)
    return out3, w1, w2, (cm1, cl1, cm2, cl2, cf, cl3, prefix, rate, drops)


def decoder_layer_bwd(dout, cache, G):
    """Returns (dx, denc_output)."""
    cm1, cl1, cm2, cl2, cf, cl3, prefix, rate, drops = cache
    dz3 = _ln_bwd(dout, cl3, prefix + "/layernorm3", G)
    dout2 = dz3 + ffn_bwd(dropout_bwd(dz3, drops.get(prefix + "/dropout3"), rate), cf, G)
    dz2 = _ln_bwd(dout2, cl2, prefix + "/layernorm2", G)
    dv2, dk2, dq2 = mha_bwd(dropout_bwd(dz2, drops.get(prefix + "/dropout2"), rate), cm2, G)
    dout1 = dz2 + dq2
    dz1 = _ln_bwd(dout1, cl1, prefix + "/layernorm1", G)
    dv1, dk1, dq1 = mha_bwd(dropout_bwd(dz1, drops.get(prefix + "/dropout1"), rate), cm1, G)
    return dz1 + dv1 + dk1 + dq1, dv2 + dk2


def _embed_fwd(P, prefix, x, cfg, pos, rate, keep):
    """Embedding stage of Encoder.call / Decoder.call,
    builders/layers/transformer.py:288-296, 325-334: gather (or Dense 5->d),
    ``x *= sqrt(d_model)``, ``x += pos[:, :seq]``, dropout."""
    dt = pos.dtype
    if cfg.continuous:
        e, ce = dense_fwd(x.astype(dt), P[prefix + "/kernel"], P[prefix + "/bias"])
    else:
        e, ce = P[prefix][x], x
    e = e * np.sqrt(np.asarray(cfg.d_model, dtype=dt))
    e = e + pos[:, :x.shape[1], :]
    return dropout_fwd(e, keep, rate), (ce, keep, rate)


def _embed_bwd(dx, cache, prefix, cfg, P, G):
    ce, keep, rate = cache
    dx = dropout_bwd(dx, keep, rate) * np.sqrt(np.asarray(cfg.d_model, dtype=dx.dtype))
    if cfg.continuous:
        _, G[prefix + "/kernel"], G[prefix + "/bias"] = dense_bwd(dx, ce)
    else:
        g = np.zeros_like(P[prefix])
        np.add.at(g, ce.reshape(-1), dx.reshape(-1, dx.shape[-1]))
        G[prefix] = g


def self_attn_v1_fwd(P, x):
    """builders/layers/transformer.py:61-74: u=tanh(xW+b); a=softmax(uV, axis=time)
    (NO padding mask); o=sum_t a*x -> (B,d)."""
    u, cu = dense_fwd(x, P["bottleneck/W_attn"], P["bottleneck/b_attn"], "tanh")
    s = u @ P["bottleneck/V_attn"]                       # (B,T,1)
    m = s.max(1, keepdims=True)
    e = np.exp(s - m)
    a = e / e.sum(1, keepdims=True)
    o = (x * a).sum(1)
    return o, a, (x, u, a, cu)


def self_attn_v1_bwd(do, cache, P, G):
    x, u, a, cu = cache
    do_ = do[:, None, :]
    dx = a * do_
    da = (x * do_).sum(-1, keepdims=True)                # (B,T,1)
    ds = a * (da - (da * a).sum(1, keepdims=True))
    G["bottleneck/V_attn"] = (u * ds).sum((0, 1))[:, None]
    du = ds * P["bottleneck/V_attn"][:, 0]
    dxw, G["bottleneck/W_attn"], G["bottleneck/b_attn"] = dense_bwd(du, cu)
    return dx + dxw


def self_attn_v2_fwd(P, x):
    """builders/layers/transformer.py:116-131."""
    o, a, c = self_attn_v1_fwd(P, x)
    e, ce = dense_fwd(o, P["bottleneck/embeding_layer/kernel"], P["bottleneck/embeding_layer/bias"])
    return e, a, (c, ce)


def self_attn_v2_bwd(de, cache, P, G):
    c, ce = cache
    do, G["bottleneck/embeding_layer/kernel"], G["bottleneck/embeding_layer/bias"] = dense_bwd(de, ce)
    return self_attn_v1_bwd(do, c, P, G)


def dense_expander_fwd(P, emb):
    """builders/layers/transformer.py:370-376:
    pre[b,t,c] = emb[b,c] * w[t] + bias[t]."""
    w = P["expand/kernel"][0]
    b = P["expand/bias"]
    pre = emb[:, None, :] * w[None, :, None] + b[None, :, None]
    return pre, (emb, w)


def dense_expander_bwd(dpre, cache, G):
    emb, w = cache
    G["expand/kernel"] = np.einsum("btc,bc->t", dpre, emb)[None, :]
    G["expand/bias"] = dpre.sum((0, 2))
    return np.einsum("btc,t->bc", dpre, w)


# --------------------------------------------------------------------------
# builders/losses.py / builders/keras_metrics.py
# --------------------------------------------------------------------------
def _log_softmax(z):
    m = z.max(-1, keepdims=True)
    return z - m - np.log(np.exp(z - m).sum(-1, keepdims=True))


def recon_loss_fwd(real, pred, weight=1.0):
    """builders/losses.py:26-41: per-token sparse CE from logits, multiplied
    by (real != 0), reduce_mean over ALL B*L' positions."""
    lsm = _log_softmax(pred)
    B, T = real.shape
    per = -np.take_along_axis(lsm, real[..., None], -1)[..., 0]
    mask = (real != 0).astype(pred.dtype)
    loss = weight * (per * mask).mean()
    return loss, (lsm, real, mask, weight)


def recon_loss_bwd(cache):
    lsm, real, mask, weight = cache
    g = np.exp(lsm)
    B, T = real.shape
    np.put_along_axis(g, real[..., None], np.take_along_axis(g, real[..., None], -1) - 1.0, -1)
    return g * (mask[..., None] * (weight / (B * T)))


def continuous_recon_loss_fwd(real, pred, weight=1.0):
    """builders/losses.py:43-66: per-position MSE on (dx,dy) + a GLOBAL scalar
    mean pen-state CE, masked by real[...,-1] != 1, reduce_mean over all."""
    mask = (real[..., -1] != 1).astype(pred.dtype)
    loc = ((real[..., :2] - pred[..., :2]) ** 2).mean(-1)
    lab = real[..., 2:].argmax(-1)
    lsm = _log_softmax(pred[..., 2:])
    meta = (-np.take_along_axis(lsm, lab[..., None], -1)[..., 0]).mean()
    loss = weight * ((loc + meta) * mask).mean()
    return loss, (real, pred, mask, lab, lsm, weight)


def continuous_recon_loss_bwd(cache):
    real, pred, mask, lab, lsm, weight = cache
    B, T = mask.shape
    n = B * T
    g = np.zeros_like(pred)
    g[..., :2] = (pred[..., :2] - real[..., :2]) * (mask[..., None] * weight / n)   # 2/2 from mean over 2
    dmeta = weight * mask.sum() / n                     # d loss / d meta (scalar)
    p = np.exp(lsm)
    np.put_along_axis(p, lab[..., None], np.take_along_axis(p, lab[..., None], -1) - 1.0, -1)
    g[..., 2:] = p * (dmeta / n)
    return g


def class_loss_fwd(labels, logits, weight=1.0):
    """builders/losses.py:21-24 on the softmax output of classify_layer
    (models/sketchformer.py:99,198).  TF-2.1 graph mode recovers the logits of
    the Softmax producer, so the loss is exact log-softmax CE, mean over B."""
    lsm = _log_softmax(logits)
    lab = labels.reshape(-1)
    per = -lsm[np.arange(lab.shape[0]), lab]
    return weight * per.mean(), (lsm, lab, weight)


def class_loss_bwd(cache):
    lsm, lab, weight = cache
    g = np.exp(lsm)
    g[np.arange(lab.shape[0]), lab] -= 1.0
    return g * (weight / lab.shape[0])


@dataclass
class MetricState:
    """builders/keras_metrics.py:13-42: running Mean / SparseCategoricalAccuracy,
    never reset during train() (core/models.py:183-197)."""
    total: Dict[str, float] = field(default_factory=dict)
    count: Dict[str, float] = field(default_factory=dict)

    def update_mean(self, name, value):
        self.total[name] = self.total.get(name, 0.0) + float(value)
        self.count[name] = self.count.get(name, 0.0) + 1.0

    def update_acc(self, name, labels, pred):
        hit = (pred.argmax(-1).reshape(-1) == labels.reshape(-1))
        self.total[name] = self.total.get(name, 0.0) + float(hit.sum())
        self.count[name] = self.count.get(name, 0.0) + float(hit.size)

    def results(self):
        return {k: self.total[k] / self.count[k] for k in self.total}


# --------------------------------------------------------------------------
# builders/schedulers.py + Keras Adam
# --------------------------------------------------------------------------
def warmup_decay(step, d_model, warmup_steps=5000):
    """builders/schedulers.py:25-29 evaluated in float32, called with
    ``optimizer.iterations`` BEFORE the increment (step 0 -> lr 0).
    models/sketchformer.py:113-114 hard-codes warmup_steps=5000."""
    step = np.float32(step)
    with np.errstate(divide="ignore"):
        arg1 = np.float32(1.0) / np.sqrt(step)
    arg2 = step * np.float32(warmup_steps ** -1.5)
    return np.float32(np.float32(1.0) / np.sqrt(np.float32(d_model)) * np.minimum(arg1, arg2))


def step_decay(step, init_lr, decay_rate=0.1, decay_steps=50000, min_lr_ratio=1e-2):
    """builders/schedulers.py:32-46."""
    return max(init_lr * decay_rate ** math.floor(step / decay_steps), init_lr * min_lr_ratio)


def adam_update(w, g, m, v, iterations, lr, beta1=0.9, beta2=0.98, eps=1e-9):
    """tf.keras.optimizers.Adam (models/sketchformer.py:122-124), non-amsgrad:
    t = iterations+1; alpha = lr*sqrt(1-b2^t)/(1-b1^t);
    m += (g-m)(1-b1); v += (g^2-v)(1-b2); w -= alpha*m/(sqrt(v)+eps)."""
    dt = w.dtype.type
    t = iterations + 1
    alpha = dt(lr) * dt(math.sqrt(1.0 - beta2 ** t) / (1.0 - beta1 ** t))
    m += (g - m) * dt(1.0 - beta1)
    v += (g * g - v) * dt(1.0 - beta2)
    w -= alpha * m / (np.sqrt(v) + dt(eps))


# --------------------------------------------------------------------------
# models/sketchformer.py
# --------------------------------------------------------------------------
def sgd_momentum_update(w, g, vel, lr, momentum=0.9):
    """tf.keras.optimizers.SGD(lr_schedule, momentum=0.9) (models/sketchformer.py:124-126), nesterov=False:
    velocity = momentum * velocity - lr * g ; w += velocity.  In place."""
    vel *= momentum
    vel -= lr * g
    w += vel


def dropout_sites(cfg: Config) -> List[Tuple[str, str]]:
    """All dropout call sites in forward order: (name, 'enc'|'dec') where the
    tag gives the tensor shape (B,L,d) or (B,L-1,d).  The index in this list
    is the ``site`` id of the product's counter-based RNG."""
    sites = [("encoder/dropout", "enc")]
    for i in range(cfg.num_layers):
        sites += [("encoder/layer%d/dropout1" % i, "enc"), ("encoder/layer%d/dropout2" % i, "enc")]
    sites.append(("decoder/dropout", "dec"))               # site ids are fixed; unused ones simply never get drawn
    for i in range(cfg.num_layers):
        sites += [("decoder/layer%d/dropout%d" % (i, j), "dec") for j in (1, 2, 3)]
    sites += [("class_dropout/%d" % i, "cls") for i in range(cfg.class_buffer_layers)]   # (B, lowerdim), rate class_dropout
    return sites


def classify_from_embedding_fwd(P, cfg: Config, emb, drops=None, training=False):
    """models/sketchformer.py:183-199: optional Dense(lowerdim, relu) + Dropout(class_dropout) buffers, then the
    classify layer -> LOGITS (the softmax of Dense(activation='softmax') is applied by the caller)."""
    drops = drops or {}
    rate = cfg.class_dropout if training else 0.0
    fc, caches = emb, []
    for i in range(cfg.class_buffer_layers):
        fc, cd = dense_fwd(fc, P["class_buffer/%d/kernel" % i], P["class_buffer/%d/bias" % i], "relu")
        keep = drops.get("class_dropout/%d" % i)
        fc = dropout_fwd(fc, keep, rate)
        caches.append((cd, keep, rate))
    logits, c = dense_fwd(fc, P["classify/kernel"], P["classify/bias"])
    return logits, (caches, c)


def classify_from_embedding_bwd(dlogits, cache, G):
    caches, c = cache
    d, G["classify/kernel"], G["classify/bias"] = dense_bwd(dlogits, c)
    for i in reversed(range(len(caches))):
        cd, keep, rate = caches[i]
        d = dropout_bwd(d, keep, rate)
        d, G["class_buffer/%d/kernel" % i], G["class_buffer/%d/bias" % i] = dense_bwd(d, cd)
    return d


def forward(P, cfg: Config, inp, tar_inp, drops: Optional[dict] = None, training=True):
    """Transformer.call (models/sketchformer.py:131-147) with the masks of
    model_trainer (:330).  Returns (outputs, cache)."""
    drops = drops or {}
    rate = cfg.dropout_rate if training else 0.0
    dt = next(iter(P.values())).dtype
    pos = positional_encoding(cfg.max_pos, cfg.d_model).astype(dt)
    enc_mask, combined_mask, dec_pad_mask = create_masks(inp, tar_inp)
    H = cfg.num_heads

    # ---- encode (models/sketchformer.py:149-160)
    x, c_eemb = _embed_fwd(P, "encoder/embedding", inp, cfg, pos, rate, drops.get("encoder/dropout"))
    c_enc = []
    for i in range(cfg.num_layers):
        x, c = encoder_layer_fwd(P, "encoder/layer%d" % i, x, enc_mask, H, rate, drops)
        c_enc.append(c)
    enc_output = x
    out = {"enc_output": enc_output}
    c_bott = c_cls = c_exp = c_demb = c_dec = c_out = None
    if cfg.has_bottleneck:
        if cfg.attn_version == 1:
            emb, bott_w, c_bott = self_attn_v1_fwd(P, enc_output)
        else:
            emb, bott_w, c_bott = self_attn_v2_fwd(P, enc_output)
        out["bottleneck_attn"] = bott_w
    else:
        emb = enc_output                                   # models/sketchformer.py:158-159
    out["embedding"] = emb
    if cfg.has_classifier:
        cls_logits, c_cls = classify_from_embedding_fwd(P, cfg, emb, drops, training)
        e = np.exp(cls_logits - cls_logits.max(-1, keepdims=True))
        out["class"] = e / e.sum(-1, keepdims=True)     # Dense(activation='softmax')
        out["class_logits"] = cls_logits

    # ---- decode (models/sketchformer.py:170-181)
    if cfg.do_reconstruction:
        padding_mask = np.zeros_like(dec_pad_mask) if cfg.blind_decoder_mask else dec_pad_mask
        if cfg.has_bottleneck:
            pre, c_exp = dense_expander_fwd(P, emb)
        else:
            pre = emb
        y, c_demb = _embed_fwd(P, "decoder/embedding", tar_inp, cfg, pos, rate, drops.get("decoder/dropout"))
        c_dec = []
        for i in range(cfg.num_layers):
            y, _, _, c = decoder_layer_fwd(P, "decoder/layer%d" % i, y, pre, combined_mask, padding_mask, H, rate, drops)
            c_dec.append(c)
        logits, c_out = dense_fwd(y, P["output/kernel"], P["output/bias"])
        out.update({"recon": logits, "pre_decoder": pre, "dec_output": y})
    cache = (c_eemb, c_enc, c_bott, c_cls, c_exp, c_demb, c_dec, c_out)
    return out, cache


def loss_and_grads(P, cfg: Config, inp, tar, labels, drops=None, want_grads=True):
    """model_trainer (models/sketchformer.py:325-349) minus the optimizer:
    returns (losses dict, outputs, grads dict)."""
    tar_inp, tar_real = tar[:, :-1, ...], tar[:, 1:, ...]
    out, cache = forward(P, cfg, inp, tar_inp, drops, training=True)
    c_eemb, c_enc, c_bott, c_cls, c_exp, c_demb, c_dec, c_out = cache
    losses, total = {}, 0.0
    if cfg.do_reconstruction:
        if cfg.continuous:
            recon, c_rl = continuous_recon_loss_fwd(tar_real.astype(out["recon"].dtype), out["recon"], cfg.recon_weight)
        else:
            recon, c_rl = recon_loss_fwd(tar_real, out["recon"], cfg.recon_weight)
        losses["recon_loss"] = recon
        total = total + recon
    if cfg.has_classifier:
        clas, c_cl = class_loss_fwd(labels, out["class_logits"], cfg.class_weight)
        losses["class_loss"] = clas
        total = total + clas
    losses["total_loss"] = total
    if not want_grads:
        return losses, out, None

    G: Dict[str, np.ndarray] = {}
    demb = 0.0
    if cfg.do_reconstruction:
        dlogits = continuous_recon_loss_bwd(c_rl) if cfg.continuous else recon_loss_bwd(c_rl)
        dy, G["output/kernel"], G["output/bias"] = dense_bwd(dlogits, c_out)
        dpre = 0.0
        for i in reversed(range(cfg.num_layers)):
            dy, dp = decoder_layer_bwd(dy, c_dec[i], G)
            dpre = dpre + dp
        _embed_bwd(dy, c_demb, "decoder/embedding", cfg, P, G)
        demb = dense_expander_bwd(dpre, c_exp, G) if cfg.has_bottleneck else dpre
    if cfg.has_classifier:
        demb = demb + classify_from_embedding_bwd(class_loss_bwd(c_cl), c_cls, G)
    if not cfg.has_bottleneck:
        dx = demb
    elif cfg.attn_version == 1:
        dx = self_attn_v1_bwd(demb, c_bott, P, G)
    else:
        dx = self_attn_v2_bwd(demb, c_bott, P, G)
    for i in reversed(range(cfg.num_layers)):
        dx = encoder_layer_bwd(dx, c_enc[i], G)
    _embed_bwd(dx, c_eemb, "encoder/embedding", cfg, P, G)
    return losses, out, G


@dataclass
class TrainState:
    params: Dict[str, np.ndarray]
    m: Dict[str, np.ndarray]
    v: Dict[str, np.ndarray]
    iterations: int = 0
    metrics: MetricState = field(default_factory=MetricState)

    @classmethod
    def create(cls, params):
        return cls(params, {k: np.zeros_like(a) for k, a in params.items()},
                   {k: np.zeros_like(a) for k, a in params.items()})


def train_step(state: TrainState, cfg: Config, inp, tar, labels, drops=None):
    """One model_trainer call + the metric read-back of train_on_batch
    (models/sketchformer.py:351-359).  Returns the quick-metrics dict."""
    losses, out, G = loss_and_grads(state.params, cfg, inp, tar, labels, drops)
    ms = state.metrics
    if cfg.do_reconstruction:
        ms.update_mean("recon_loss", losses["recon_loss"])
        if not cfg.continuous:
            ms.update_acc("recon_acc", tar[:, 1:], out["recon"])
    if cfg.has_classifier:
        ms.update_mean("class_loss", losses["class_loss"])
        ms.update_acc("class_acc", labels, out["class"])
    ms.update_mean("total_loss", losses["total_loss"])
    lr = warmup_decay(state.iterations, cfg.d_model, 5000)
    for k in state.params:
        if cfg.optimizer == "sgd":
            sgd_momentum_update(state.params[k], G[k], state.m[k], lr)          # m doubles as the velocity slot
        else:
            adam_update(state.params[k], G[k], state.m[k], state.v[k], state.iterations, lr)
    state.iterations += 1
    return ms.results(), losses, out, G

# --------------------------------------------------------------------------
# inference API (models/sketchformer.py:149-168, 201-311) - naive restatement:
# like the reference, the whole decoder is re-run on the growing prefix for
# every emitted token (no KV cache); only small cases are meant to run here.
# --------------------------------------------------------------------------
def encode_from_seq(P, cfg: Config, inp_seq):
    """models/sketchformer.py:162-168 (mask computed inside, training=False).
    Returns dict(enc_output, embedding, class = class probabilities)."""
    inp = np.asarray(inp_seq)
    dt = next(iter(P.values())).dtype
    pos = positional_encoding(cfg.max_pos, cfg.d_model).astype(dt)
    x, _ = _embed_fwd(P, "encoder/embedding", inp, cfg, pos, 0.0, None)
    mask = create_padding_mask(inp)
    for i in range(cfg.num_layers):
        x, _ = encoder_layer_fwd(P, "encoder/layer%d" % i, x, mask, cfg.num_heads, 0.0, {})
    if not cfg.has_bottleneck:
        return {"enc_output": x, "embedding": x, "class": None}
    emb = self_attn_v1_fwd(P, x)[0] if cfg.attn_version == 1 else self_attn_v2_fwd(P, x)[0]
    if not cfg.has_classifier:
        return {"enc_output": x, "embedding": emb, "class": None}
    logits, _ = classify_from_embedding_fwd(P, cfg, emb)
    e = np.exp(logits - logits.max(-1, keepdims=True))
    return {"enc_output": x, "embedding": emb, "class": e / e.sum(-1, keepdims=True)}


def make_dummy_input(cfg: Config, expected_len, nattn, batch_size):
    """models/sketchformer.py:230-253: a fake encoder input whose only use is its padding mask:
    the first nattn positions are 'real', the remaining seq_len - nattn are padding."""
    L = cfg.seq_len
    if cfg.continuous:
        d = np.zeros((batch_size, L, 5), dtype=np.float32)
        d[:, int(nattn):, 4] = 1.0
        return d
    d = np.zeros((batch_size, L), dtype=np.float32)
    if expected_len is None:
        d[:, :int(nattn)] = 1.0
    else:
        for b, n in enumerate(np.asarray(nattn).reshape(-1)):
            d[b, :int(n)] = 1.0
    return d


def decode(P, cfg: Config, embedding, target, dec_padding_mask, look_ahead_mask):
    """models/sketchformer.py:170-181 with training=False -> logits (B, T, V)."""
    dt = next(iter(P.values())).dtype
    pos = positional_encoding(cfg.max_pos, cfg.d_model).astype(dt)
    padding_mask = np.zeros_like(dec_padding_mask) if cfg.blind_decoder_mask else dec_padding_mask
    pre = dense_expander_fwd(P, embedding)[0] if cfg.has_bottleneck else embedding
    y, _ = _embed_fwd(P, "decoder/embedding", target, cfg, pos, 0.0, None)
    for i in range(cfg.num_layers):
        y, _, _, _ = decoder_layer_fwd(P, "decoder/layer%d" % i, y, pre, look_ahead_mask, padding_mask,
                                       cfg.num_heads, 0.0, {})
    return dense_fwd(y, P["output/kernel"], P["output/bias"])[0]


def predict_from_embedding(P, cfg: Config, emb, sos, eos, expected_len=None):
    """Greedy reconstruction, models/sketchformer.py:255-311.
    tokens:     output starts as [SOS]; up to seq_len iterations; every iteration appends argmax of the LAST
                position's logits (first index on ties, like tf.argmax); EOS flags are sticky per sample and the loop
                stops after the iteration in which all samples have emitted an EOS at some point; samples that are
                already finished keep being extended.
    continuous: output starts as [0,0,1,0,0]; the appended row is (xy, softmax(pen logits)); the loop stops after an
                iteration in which argmax(pen) == 2 for ALL samples simultaneously.
    Returns dict(recon = (B, T) int32 tokens incl. the SOS column | (B, T, 5) float rows, class = argmax of the
    class probabilities)."""
    dt = next(iter(P.values())).dtype
    emb = np.asarray(emb, dtype=dt)
    B = emb.shape[0]
    if cfg.continuous:
        output = np.tile(np.array([0., 0., 1., 0., 0.], dtype=dt), (B, 1, 1))
    else:
        output = np.full((B, 1), sos, dtype=np.int64)
        eos_seen = np.zeros(B, dtype=bool)
    for i in range(cfg.seq_len):
        nattn = expected_len if expected_len is not None else i + 1
        dummy = make_dummy_input(cfg, expected_len, nattn, B)
        _, combined_mask, dec_padding_mask = create_masks(dummy, output)
        logits = decode(P, cfg, emb, output, dec_padding_mask, combined_mask)
        last = logits[:, -1:, ...]
        if cfg.continuous:
            pen = last[..., 2:]
            e = np.exp(pen - pen.max(-1, keepdims=True))
            predicted = np.concatenate([last[..., :2], e / e.sum(-1, keepdims=True)], axis=-1)
            output = np.concatenate([output, predicted], axis=1)
            if int(np.sum(np.argmax(predicted[..., 2:], axis=-1) == 2)) == B:
                break
        else:
            predicted = np.argmax(last, axis=-1).astype(np.int64)          # (B, 1)
            output = np.concatenate([output, predicted], axis=1)
            eos_seen |= (predicted[:, 0] == eos)
            if eos_seen.all():
                break
    res = {"recon": output if cfg.continuous else output.astype(np.int32)}
    if cfg.has_classifier:
        cls_logits, _ = classify_from_embedding_fwd(P, cfg, emb)
        res["class"] = np.argmax(cls_logits, axis=-1).astype(np.int32)
    return res


def predict(P, cfg: Config, inp_seq, sos, eos):
    """models/sketchformer.py:201-221: encode, class argmax, greedy reconstruction.  With blind_decoder_mask
    (default) the expected length is not used; otherwise it is the number of non-padding input positions."""
    out = encode_from_seq(P, cfg, inp_seq)
    inp = np.asarray(inp_seq)
    if cfg.blind_decoder_mask:
        tlen = None
    elif cfg.continuous:
        tlen = np.sum(inp[..., -1] != 1, axis=-1)
    else:
        tlen = np.sum(inp > 0, axis=-1)
    res = {"embedding": out["embedding"]}
    if cfg.has_classifier:
        res.update({"class": np.argmax(out["class"], axis=-1).astype(np.int32), "class_probs": out["class"]})
    if cfg.do_reconstruction:
        res["recon"] = predict_from_embedding(P, cfg, out["embedding"], sos, eos, tlen)["recon"]
    return res
