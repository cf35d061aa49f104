"""bf16-storage restatement of the sketch-transformer-tf2 train step: the arithmetic of oracle/sketchformer_oracle.py (the TF2
reference's graph, models/sketchformer.py:131-181, 325-349) with every value ROUNDED TO bf16 WHERE THE MI355X bf16 PATH STORES
IT (BASELINE cfg 5: bf16 activations / weight images / upstream gradients, fp32 accumulation, fp32 master weights, fp32 loss
heads).  TEST INFRASTRUCTURE ONLY (see oracle/__init__.py): tests/test_gpu_bf16_model.py compares the device's gradients with
this restatement at a bar an order of magnitude below the one against the float64 oracle - what is left is fp32-vs-float64
accumulation order and the rare value that sits within that distance of a bf16 rounding boundary.

The storage points follow sketchformer_amd/csrc/skf_model_bf16.inc (launch sequence), skf_bf16_gemm.hip (Dense: bf16 operands,
fp32 accumulate, fp32 bias, result rounded; an accumulating launch rounds the product, adds the old value in fp32 and rounds
again), skf_bf16_rowops.hip (embedding, residual + LayerNorm with moments of the ROUNDED z, cross-entropy gradient in place,
pooling, expander) and skf_bf16_attention.hip (64-key blocks with a running maximum, the unnormalised probabilities of a block
rounded for P.V, the output kept as a bf16 value + bf16 residual, dS and P rounded for the gradient products).  Everything
else (which operations exist, their order, the masks, the losses) is the reference's.

Arithmetic between storage points runs in float64 here and in fp32 on the device.
"""
import numpy as np

from . import sketchformer_oracle as so


# Error model of an fp32 evaluation (tests/test_gpu_bf16_model.py, tools/bf16_rounding_noise.py): (numpy Generator, rel) multiplies every
# value by 1 + rel * N(0, 1) BEFORE it is rounded to bf16 - what a different fp32 accumulation order does to a value that is about to
# be stored.  A value within that distance of a rounding boundary then lands on the other side (a 2^-9 step), and where a later
# difference cancels to 1e-4 of its terms (dS = P o (dP - delta) in the upper encoder layers of a padded batch) that step is amplified.
NOISE = None


def rbf(x):
    """round to nearest-even bf16, returned as float64 (skf_f2bf / v_cvt_pk_bf16_f32)"""
    if NOISE is not None:
        rng, rel = NOISE
        x = np.asarray(x, dtype=np.float64)
        x = x * (1.0 + rel * rng.standard_normal(x.shape))
    f = np.ascontiguousarray(x, dtype=np.float32)
    u = f.view(np.uint32)
    r = ((u >> 16) & 1) + 0x7FFF
    out = ((u + r) & 0xFFFF0000).astype(np.uint32).view(np.float32)
    return out.astype(np.float64)


def _drop(x, keep, rate):
    if keep is None or rate == 0.0:
        return x
    return x * (1.0 / (1.0 - rate)) * keep


# ----------------------------------------------------------------------------------------------- Dense (skf_bf16_gemm.hip)
def dense16(x, W16, b, act=None):
    y = x @ W16 + b
    if act == "relu":
        y = np.maximum(y, 0)
    elif act == "tanh":
        y = np.tanh(y)
    return rbf(y)


def dgrad16(dy, W16, relu_src=None, acc=None):
    """dX = dY . W^T (o relu mask), rounded; accumulate: round(old + round(product))"""
    dx = rbf(dy @ W16.T)
    if relu_src is not None:
        dx = dx * (relu_src > 0)
    return dx if acc is None else rbf(acc + dx)


def wgrad(x, dy):
    x2, dy2 = x.reshape(-1, x.shape[-1]), dy.reshape(-1, dy.shape[-1])
    return x2.T @ dy2, dy2.sum(0)


# ----------------------------------------------------------------------------------------------- LayerNorm (skf_bf16_rowops.hip)
def ln_fwd16(x, y, gamma, beta, keep, rate):
    z = rbf(x + _drop(y, keep, rate))                       # rounded first: the moments are those of what the backward reads
    mean = z.mean(-1, keepdims=True)
    var = ((z - mean) ** 2).mean(-1, keepdims=True)
    rstd = 1.0 / np.sqrt(var + 1e-6)
    out = rbf((z - mean) * rstd * gamma + beta)
    return out, (z, mean, rstd, gamma, keep, rate)


def ln_bwd16(dout, cache):
    """-> dz (rounded), dy = dz o dropout mask (rounded from the unrounded dz), dgamma, dbeta"""
    z, mean, rstd, gamma, keep, rate = cache
    xh = (z - mean) * rstd
    red = tuple(range(dout.ndim - 1))
    dgamma, dbeta = (dout * xh).sum(red), dout.sum(red)
    g = dout * gamma
    dz = rstd * (g - g.mean(-1, keepdims=True) - xh * (g * xh).mean(-1, keepdims=True))
    dy = rbf(_drop(dz, keep, rate)) if (keep is not None and rate > 0.0) else rbf(dz)
    return rbf(dz), dy, dgamma, dbeta


# ----------------------------------------------------------------------------------------------- attention (skf_bf16_attention.hip)
def _heads(x, H):
    B, L, d = x.shape
    return x.reshape(B, L, H, d // H).transpose(0, 2, 1, 3)


def _merge(x):
    B, H, L, dh = x.shape
    return x.transpose(0, 2, 1, 3).reshape(B, L, H * dh)


def _scores2(q, k, mask):
    """base-2 logits: (q.k) log2(e)/sqrt(dh), masked entries SET to -1e9 (skf_attention.hip header: same probabilities)"""
    dh = q.shape[-1]
    s = (q @ np.swapaxes(k, -1, -2)) * (np.log2(np.e) / np.sqrt(dh))
    if mask is not None:
        s = np.where(np.broadcast_to(mask, s.shape) > 0, -1e9, s)
    return s


def attn_fwd16(q, k, v, mask, H, KB=64):
    qh, kh, vh = _heads(q, H), _heads(k, H), _heads(v, H)
    s = _scores2(qh, kh, mask)
    Lk = kh.shape[2]
    m_run = np.full(s.shape[:-1] + (1,), -np.inf)
    l_run = np.zeros_like(m_run)
    acc = np.zeros(qh.shape)
    for k0 in range(0, Lk, KB):                             # 64-key blocks, running maximum (online softmax)
        sb = s[..., k0:k0 + KB]
        m_new = np.maximum(m_run, sb.max(-1, keepdims=True))
        alpha = np.exp2(m_run - m_new)
        p = np.exp2(sb - m_new)
        l_run = l_run * alpha + p.sum(-1, keepdims=True)
        acc = acc * alpha + rbf(p) @ vh[:, :, k0:k0 + KB]    # P goes to the matrix cores in bf16, the row sum does not
        m_run = m_new
    o = acc / l_run
    ohi = rbf(o)
    olo = rbf(o - ohi)
    return _merge(ohi), (qh, kh, vh, mask, m_run, 1.0 / l_run, ohi, olo, H, o)


# Sensitivity experiments of tools/bf16_delta_sensitivity.py (None = the product's arithmetic): a callable
# (delta, dp, do, ohi, olo, o_exact) -> (delta, dp) that replaces / perturbs the two terms whose difference forms dS.
DELTA_HOOK = None


def attn_bwd16(dout, cache):
    qh, kh, vh, mask, m, rinv, ohi, olo, H, o_exact = cache
    do = _heads(dout, H)
    dh = qh.shape[-1]
    delta = (do * (ohi + olo)).sum(-1, keepdims=True)
    s = _scores2(qh, kh, mask)
    p = np.exp2(s - m) * rinv
    dp = do @ np.swapaxes(vh, -1, -2)
    if DELTA_HOOK is not None:
        delta, dp = DELTA_HOOK(delta, dp, do, ohi, olo, o_exact)
    ds = rbf(p * (dp - delta))                              # dS and P are bf16 operands of the gradient products
    scale = 1.0 / np.sqrt(dh)
    dq = rbf((ds @ kh) * scale)
    dk = rbf((np.swapaxes(ds, -1, -2) @ qh) * scale)
    dv = rbf(np.swapaxes(rbf(p), -1, -2) @ do)
    return _merge(dq), _merge(dk), _merge(dv)


# ----------------------------------------------------------------------------------------------- the step
def _w16(P, cfg):
    """bf16 images of every Dense kernel of the bf16 path (the classifier, the embeddings and the expander stay fp32)"""
    W = {}
    for k, v in P.items():
        if k.endswith("/kernel") and not k.startswith(("classify", "expand")) or k == "bottleneck/W_attn":
            W[k] = rbf(v)
    return W


def _qkv(P, W, pre):
    Wq = np.concatenate([W[pre + "/wq/kernel"], W[pre + "/wk/kernel"], W[pre + "/wv/kernel"]], axis=1)
    bq = np.concatenate([P[pre + "/wq/bias"], P[pre + "/wk/bias"], P[pre + "/wv/bias"]])
    return Wq, bq


def _store_qkv_grads(G, pre, dW, db, d):
    for j, n in enumerate(("wq", "wk", "wv")):
        G[pre + "/" + n + "/kernel"] = dW[:, j * d:(j + 1) * d]
        G[pre + "/" + n + "/bias"] = db[j * d:(j + 1) * d]


RELU_KINK = 2.0 ** -7          # relative distance from zero inside which a supplied ReLU branch may replace this restatement's own


def loss_and_grads(P, cfg, inp, tar, labels, drops=None, relu_masks=None, stats=None):
    """model_trainer minus the optimizer on the bf16 path's storage points -> (losses, outputs, gradients); same signature and
    names as oracle.loss_and_grads.  Built for what the bf16 path supports: token mode, attn_version 1, bottleneck +
    classifier + decoder, no class buffers.
    relu_masks: optional {"encoder/layer0/ffn": bool (B, L, dff), ...} = the branch another evaluation (the device) took; it
    replaces this restatement's own `pre > 0` ONLY for units whose pre-activation lies within RELU_KINK * max|pre| of zero
    (a bf16 pre-activation on the kink may round to either side); stats["relu_overrides"] counts them, stats["relu_units"] all."""
    assert not cfg.continuous and cfg.attn_version == 1 and cfg.has_bottleneck and cfg.has_classifier and cfg.do_reconstruction
    drops = drops or {}
    relu_masks = relu_masks or {}
    if stats is not None:
        stats.update(relu_overrides=0, relu_units=0)

    def ffn1(x, Wk, bk, prefix):
        pre = x @ Wk + bk
        act = pre > 0
        m = relu_masks.get(prefix)
        if m is not None:
            near = np.abs(pre) <= RELU_KINK * max(1.0, np.abs(pre).max())
            take = near & (np.asarray(m, bool).reshape(pre.shape) != act)
            if stats is not None:
                stats["relu_overrides"] += int(take.sum())
            act = np.where(take, ~act, act)
        if stats is not None:
            stats["relu_units"] += pre.size
        # an overridden unit takes |pre| (tiny) when switched on: only its sign matters downstream
        return rbf(np.where(act, np.where(pre > 0, pre, np.abs(pre) + 1e-30), 0.0))
    P = {k: np.asarray(v, np.float64) for k, v in P.items()}
    W = _w16(P, cfg)
    d, H, N, rate = cfg.d_model, cfg.num_heads, cfg.num_layers, cfg.dropout_rate
    pos = so.positional_encoding(cfg.max_pos, d).astype(np.float64)
    tar_inp, tar_real = tar[:, :-1], tar[:, 1:]
    enc_mask, comb_mask, dec_pad = so.create_masks(inp, tar_inp)
    cross_mask = None if cfg.blind_decoder_mask else dec_pad
    G = {}

    def embed(prefix, tok, keep):
        e = P[prefix][tok] * np.sqrt(np.float64(d)) + pos[:, :tok.shape[1], :]
        return rbf(_drop(e, keep, rate))

    # ------------------------------------------------------------------ forward
    x = embed("encoder/embedding", inp, drops.get("encoder/dropout"))
    enc_c = []
    for i in range(N):
        p = "encoder/layer%d" % i
        Wq, bq = _qkv(P, W, p + "/mha")
        qkv = dense16(x, Wq, bq)
        o, ca = attn_fwd16(qkv[..., :d], qkv[..., d:2 * d], qkv[..., 2 * d:], enc_mask, H)
        y = dense16(o, W[p + "/mha/dense/kernel"], P[p + "/mha/dense/bias"])
        x1, cl1 = ln_fwd16(x, y, P[p + "/layernorm1/gamma"], P[p + "/layernorm1/beta"], drops.get(p + "/dropout1"), rate)
        h = ffn1(x1, W[p + "/ffn/dense1/kernel"], P[p + "/ffn/dense1/bias"], p + "/ffn")
        y2 = dense16(h, W[p + "/ffn/dense2/kernel"], P[p + "/ffn/dense2/bias"])
        x2, cl2 = ln_fwd16(x1, y2, P[p + "/layernorm2/gamma"], P[p + "/layernorm2/beta"], drops.get(p + "/dropout2"), rate)
        enc_c.append((x, Wq, ca, o, cl1, x1, h, cl2))
        x = x2
    enc_out = x
    u = dense16(enc_out, W["bottleneck/W_attn"], P["bottleneck/b_attn"], "tanh")
    sc = u @ P["bottleneck/V_attn"]                                        # (B, T, 1), fp32 on the device
    e = np.exp(sc - sc.max(1, keepdims=True))
    a = e / e.sum(1, keepdims=True)
    emb = (enc_out * a).sum(1)                                             # fp32 embedding
    cls_logits = emb @ P["classify/kernel"] + P["classify/bias"]           # fp32 head on the master weights
    clas, c_cl = so.class_loss_fwd(labels, cls_logits, cfg.class_weight)
    w_exp, b_exp = P["expand/kernel"][0], P["expand/bias"]
    pre = rbf(emb[:, None, :] * w_exp[None, :, None] + b_exp[None, :, None])
    y = embed("decoder/embedding", tar_inp, drops.get("decoder/dropout"))
    dec_c = []
    for i in range(N):
        p = "decoder/layer%d" % i
        Wq, bq = _qkv(P, W, p + "/mha1")
        qkv = dense16(y, Wq, bq)
        o1, ca1 = attn_fwd16(qkv[..., :d], qkv[..., d:2 * d], qkv[..., 2 * d:], comb_mask, H)
        z = dense16(o1, W[p + "/mha1/dense/kernel"], P[p + "/mha1/dense/bias"])
        out1, cl1 = ln_fwd16(y, z, P[p + "/layernorm1/gamma"], P[p + "/layernorm1/beta"], drops.get(p + "/dropout1"), rate)
        q2 = dense16(out1, W[p + "/mha2/wq/kernel"], P[p + "/mha2/wq/bias"])
        Wkv = np.concatenate([W[p + "/mha2/wk/kernel"], W[p + "/mha2/wv/kernel"]], axis=1)
        bkv = np.concatenate([P[p + "/mha2/wk/bias"], P[p + "/mha2/wv/bias"]])
        kv2 = dense16(pre, Wkv, bkv)
        o2, ca2 = attn_fwd16(q2, kv2[..., :d], kv2[..., d:], cross_mask, H)
        z = dense16(o2, W[p + "/mha2/dense/kernel"], P[p + "/mha2/dense/bias"])
        out2, cl2 = ln_fwd16(out1, z, P[p + "/layernorm2/gamma"], P[p + "/layernorm2/beta"], drops.get(p + "/dropout2"), rate)
        h = ffn1(out2, W[p + "/ffn/dense1/kernel"], P[p + "/ffn/dense1/bias"], p + "/ffn")
        z = dense16(h, W[p + "/ffn/dense2/kernel"], P[p + "/ffn/dense2/bias"])
        out3, cl3 = ln_fwd16(out2, z, P[p + "/layernorm3/gamma"], P[p + "/layernorm3/beta"], drops.get(p + "/dropout3"), rate)
        dec_c.append((y, Wq, ca1, o1, cl1, out1, Wkv, ca2, o2, cl2, out2, h, cl3))
        y = out3
    logits = dense16(y, W["output/kernel"], P["output/bias"])             # bf16 logits
    recon, c_rl = so.recon_loss_fwd(tar_real, logits, cfg.recon_weight)
    losses = {"recon_loss": recon, "class_loss": clas, "total_loss": recon + clas}
    outputs = {"recon": logits, "class_logits": cls_logits, "embedding": emb}

    # ------------------------------------------------------------------ backward
    dlog = rbf(so.recon_loss_bwd(c_rl))                                    # cross-entropy gradient, written in place in bf16
    G["output/kernel"], G["output/bias"] = wgrad(y, dlog)
    g = dgrad16(dlog, W["output/kernel"])
    dpre = None
    for i in reversed(range(N)):
        p = "decoder/layer%d" % i
        y_in, Wq, ca1, o1, cl1, out1, Wkv, ca2, o2, cl2, out2, h, cl3 = dec_c[i]
        g2, dy, G[p + "/layernorm3/gamma"], G[p + "/layernorm3/beta"] = ln_bwd16(g, cl3)
        G[p + "/ffn/dense2/kernel"], G[p + "/ffn/dense2/bias"] = wgrad(h, dy)
        dh = dgrad16(dy, W[p + "/ffn/dense2/kernel"], relu_src=h)
        G[p + "/ffn/dense1/kernel"], G[p + "/ffn/dense1/bias"] = wgrad(out2, dh)
        g2 = dgrad16(dh, W[p + "/ffn/dense1/kernel"], acc=g2)
        g, dy, G[p + "/layernorm2/gamma"], G[p + "/layernorm2/beta"] = ln_bwd16(g2, cl2)
        G[p + "/mha2/dense/kernel"], G[p + "/mha2/dense/bias"] = wgrad(o2, dy)
        dq2, dk2, dv2 = attn_bwd16(dgrad16(dy, W[p + "/mha2/dense/kernel"]), ca2)
        G[p + "/mha2/wq/kernel"], G[p + "/mha2/wq/bias"] = wgrad(out1, dq2)
        g = dgrad16(dq2, W[p + "/mha2/wq/kernel"], acc=g)
        dkv2 = np.concatenate([dk2, dv2], axis=-1)
        dWkv, dbkv = wgrad(pre, dkv2)
        G[p + "/mha2/wk/kernel"], G[p + "/mha2/wv/kernel"] = dWkv[:, :d], dWkv[:, d:]
        G[p + "/mha2/wk/bias"], G[p + "/mha2/wv/bias"] = dbkv[:d], dbkv[d:]
        dpre = dgrad16(dkv2, Wkv, acc=dpre)
        g2, dy, G[p + "/layernorm1/gamma"], G[p + "/layernorm1/beta"] = ln_bwd16(g, cl1)
        G[p + "/mha1/dense/kernel"], G[p + "/mha1/dense/bias"] = wgrad(o1, dy)
        dq, dk, dv = attn_bwd16(dgrad16(dy, W[p + "/mha1/dense/kernel"]), ca1)
        dqkv = np.concatenate([dq, dk, dv], axis=-1)
        dW, db = wgrad(y_in, dqkv)
        _store_qkv_grads(G, p + "/mha1", dW, db, d)
        g = dgrad16(dqkv, Wq, acc=g2)

    def embed_bwd(prefix, tok, dx, keep):
        dx = _drop(dx, keep, rate) * np.sqrt(np.float64(d))
        t = np.zeros_like(P[prefix])
        np.add.at(t, tok.reshape(-1), dx.reshape(-1, d))
        G[prefix] = t

    embed_bwd("decoder/embedding", tar_inp, g, drops.get("decoder/dropout"))
    # expander, classifier (fp32), pooling
    G["expand/kernel"] = np.einsum("btc,bc->t", dpre, emb)[None, :]
    G["expand/bias"] = dpre.sum((0, 2))
    demb = np.einsum("btc,t->bc", dpre, w_exp)
    dcls = so.class_loss_bwd(c_cl)
    G["classify/kernel"], G["classify/bias"] = emb.T @ dcls, dcls.sum(0)
    demb = demb + dcls @ P["classify/kernel"].T
    de = demb[:, None, :]
    g = rbf(a * de)                                                        # d(enc_out) through the weighted sum, stored bf16
    da = (enc_out * de).sum(-1, keepdims=True)
    dsc = a * (da - (da * a).sum(1, keepdims=True))
    G["bottleneck/V_attn"] = (u * dsc).sum((0, 1))[:, None]
    du = rbf(dsc * P["bottleneck/V_attn"][:, 0] * (1.0 - u * u))          # written over u in bf16
    G["bottleneck/W_attn"], G["bottleneck/b_attn"] = wgrad(enc_out, du)
    g = dgrad16(du, W["bottleneck/W_attn"], acc=g)
    for i in reversed(range(N)):
        p = "encoder/layer%d" % i
        x_in, Wq, ca, o, cl1, x1, h, cl2 = enc_c[i]
        g2, dy, G[p + "/layernorm2/gamma"], G[p + "/layernorm2/beta"] = ln_bwd16(g, cl2)
        G[p + "/ffn/dense2/kernel"], G[p + "/ffn/dense2/bias"] = wgrad(h, dy)
        dh = dgrad16(dy, W[p + "/ffn/dense2/kernel"], relu_src=h)
        G[p + "/ffn/dense1/kernel"], G[p + "/ffn/dense1/bias"] = wgrad(x1, dh)
        g2 = dgrad16(dh, W[p + "/ffn/dense1/kernel"], acc=g2)
        g, dy, G[p + "/layernorm1/gamma"], G[p + "/layernorm1/beta"] = ln_bwd16(g2, cl1)
        G[p + "/mha/dense/kernel"], G[p + "/mha/dense/bias"] = wgrad(o, dy)
        dq, dk, dv = attn_bwd16(dgrad16(dy, W[p + "/mha/dense/kernel"]), ca)
        dqkv = np.concatenate([dq, dk, dv], axis=-1)
        dW, db = wgrad(x_in, dqkv)
        _store_qkv_grads(G, p + "/mha", dW, db, d)
        g = dgrad16(dqkv, Wq, acc=g)
    embed_bwd("encoder/embedding", inp, g, drops.get("encoder/dropout"))
    return losses, outputs, G
