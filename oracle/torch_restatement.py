"""PyTorch-CPU restatement of the sketch-transformer-tf2 train step (SURVEY.md section 8(c) "second witness", 8(d) CPU baseline).

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py): imported by tests/ (as the independent autograd witness of the numpy
oracle, float64) and by bench.py's cpu_baseline leg (float32, all host cores).  The forward is composed from
torch.nn.functional primitives (F.layer_norm(eps=1e-6), F.scaled_dot_product_attention with an additive float mask,
F.cross_entropy), the gradients come from torch.autograd, the optimizer is the Keras-Adam / WarmupDecay formula written out.
It shares no arithmetic with oracle/sketchformer_oracle.py (only Config and the parameter names).

Follows: models/sketchformer.py:131-181 (call), :313-349 (model_trainer), builders/layers/transformer.py:13-376,
builders/utils.py:17-105, builders/losses.py:21-66, builders/schedulers.py:13-31.
"""

import math

import torch
import torch.nn.functional as F


def _pos(max_pos, d, dtype):
    import numpy as np
    pos = np.arange(max_pos)[:, None]
    i = np.arange(d)[None, :]
    ang = pos * (1 / np.power(10000, (2 * (i // 2)) / np.float32(d)))
    ang[:, 0::2] = np.sin(ang[:, 0::2])
    ang[:, 1::2] = np.cos(ang[:, 1::2])
    return torch.tensor(ang.astype(np.float32)).to(dtype)[None]


def _drop(x, keep, rate):
    if keep is None or rate == 0.0:
        return x
    return x * (1.0 / (1.0 - rate)) * torch.as_tensor(keep, dtype=x.dtype)


def _mha(P, pre, v, k, q, add_mask, H):
    def proj(x, n):
        return x @ P[pre + "/" + n + "/kernel"] + P[pre + "/" + n + "/bias"]

    def split(x):
        B, L, d = x.shape
        return x.view(B, L, H, d // H).permute(0, 2, 1, 3)
    qh, kh, vh = split(proj(q, "wq")), split(proj(k, "wk")), split(proj(v, "wv"))
    dh = qh.shape[-1]
    # reference divides by sqrt(dk) after the matmul; F.sdpa scales by 1/sqrt(dk) as well
    o = F.scaled_dot_product_attention(qh, kh, vh, attn_mask=add_mask, scale=1.0 / math.sqrt(dh))
    B, _, L, _ = o.shape
    o = o.permute(0, 2, 1, 3).reshape(B, L, H * dh)
    return proj(o, "dense")


def _ln(P, pre, x):
    return F.layer_norm(x, (x.shape[-1],), P[pre + "/gamma"], P[pre + "/beta"], eps=1e-6)


def _ffn(P, pre, x):
    h = F.relu(x @ P[pre + "/dense1/kernel"] + P[pre + "/dense1/bias"])
    return h @ P[pre + "/dense2/kernel"] + P[pre + "/dense2/bias"]


def _embed(P, pre, x, cfg, pos, keep, rate):
    if cfg.continuous:
        e = torch.as_tensor(x, dtype=pos.dtype) @ P[pre + "/kernel"] + P[pre + "/bias"]
    else:
        e = F.embedding(torch.as_tensor(x), P[pre])
    e = e * math.sqrt(cfg.d_model) + pos[:, :e.shape[1]]
    return _drop(e, keep, rate)


def loss_and_grads(params_np, cfg, inp, tar, labels, drops=None):
    """float64 witness: numpy parameters in, (losses, outputs, numpy gradients) out."""
    P = {k: torch.tensor(v, dtype=torch.float64, requires_grad=True) for k, v in params_np.items()}
    losses, out = forward_losses(P, cfg, inp, tar, labels, drops, torch.float64)
    losses["total_loss"].backward()
    grads = {k: (v.grad.numpy() if v.grad is not None else None) for k, v in P.items()}
    return ({k: v.item() for k, v in losses.items()},
            {k: (None if v is None else v.detach().numpy()) for k, v in out.items()}, grads)


def forward_losses(P, cfg, inp, tar, labels, drops=None, dtype=torch.float64):
    """Transformer.call + the losses of model_trainer on torch tensors P (requires_grad as the caller set it)."""
    drops = drops or {}
    rate, H = cfg.dropout_rate, cfg.num_heads
    pos = _pos(cfg.max_pos, cfg.d_model, dtype)
    tar_inp, tar_real = tar[:, :-1], tar[:, 1:]
    if cfg.continuous:
        enc_pad = torch.tensor(inp[..., -1] == 1)
        tar_pad = torch.tensor(tar_inp[..., -1] == 1)
    else:
        enc_pad = torch.tensor(inp == 0)
        tar_pad = torch.tensor(tar_inp == 0)
    Lp = tar_inp.shape[1]
    enc_add = enc_pad[:, None, None, :].to(dtype) * -1e9
    la = torch.triu(torch.ones(Lp, Lp, dtype=dtype), diagonal=1)
    comb = torch.maximum(tar_pad[:, None, None, :].to(dtype), la[None, None]) * -1e9
    cross = None if cfg.blind_decoder_mask else enc_add

    x = _embed(P, "encoder/embedding", inp, cfg, pos, drops.get("encoder/dropout"), rate)
    for i in range(cfg.num_layers):
        p = "encoder/layer%d" % i
        a = _drop(_mha(P, p + "/mha", x, x, x, enc_add, H), drops.get(p + "/dropout1"), rate)
        o1 = _ln(P, p + "/layernorm1", x + a)
        f = _drop(_ffn(P, p + "/ffn", o1), drops.get(p + "/dropout2"), rate)
        x = _ln(P, p + "/layernorm2", o1 + f)
    has_bott = cfg.lowerdim > 0
    has_cls = has_bott and getattr(cfg, "do_classification", True)
    do_recon = getattr(cfg, "do_reconstruction", True)
    if has_bott:
        u = torch.tanh(x @ P["bottleneck/W_attn"] + P["bottleneck/b_attn"])
        a = torch.softmax(u @ P["bottleneck/V_attn"], dim=1)
        emb = (x * a).sum(1)
        if cfg.attn_version != 1:
            emb = emb @ P["bottleneck/embeding_layer/kernel"] + P["bottleneck/embeding_layer/bias"]
    else:
        emb = x
    recon = clas = torch.zeros((), dtype=dtype)
    logits = cls_logits = None
    if has_cls:
        fc = emb
        for i in range(getattr(cfg, "class_buffer_layers", 0)):
            fc = torch.relu(fc @ P["class_buffer/%d/kernel" % i] + P["class_buffer/%d/bias" % i])
            fc = _drop(fc, drops.get("class_dropout/%d" % i), cfg.class_dropout if rate > 0 else 0.0)
        cls_logits = fc @ P["classify/kernel"] + P["classify/bias"]
        clas = cfg.class_weight * F.cross_entropy(cls_logits, torch.tensor(labels).reshape(-1))
    if do_recon:
        pre = emb[:, None, :] * P["expand/kernel"][0][None, :, None] + P["expand/bias"][None, :, None] if has_bott else emb
        y = _embed(P, "decoder/embedding", tar_inp, cfg, pos, drops.get("decoder/dropout"), rate)
        for i in range(cfg.num_layers):
            p = "decoder/layer%d" % i
            a1 = _drop(_mha(P, p + "/mha1", y, y, y, comb, H), drops.get(p + "/dropout1"), rate)
            o1 = _ln(P, p + "/layernorm1", a1 + y)
            a2 = _drop(_mha(P, p + "/mha2", pre, pre, o1, cross, H), drops.get(p + "/dropout2"), rate)
            o2 = _ln(P, p + "/layernorm2", a2 + o1)
            f = _drop(_ffn(P, p + "/ffn", o2), drops.get(p + "/dropout3"), rate)
            y = _ln(P, p + "/layernorm3", f + o2)
        logits = y @ P["output/kernel"] + P["output/bias"]
        if cfg.continuous:
            real = torch.tensor(tar_real, dtype=dtype)
            mask = (real[..., -1] != 1).to(dtype)
            loc = ((real[..., :2] - logits[..., :2]) ** 2).mean(-1)
            meta = F.cross_entropy(logits[..., 2:].reshape(-1, 3), real[..., 2:].argmax(-1).reshape(-1))
            recon = cfg.recon_weight * ((loc + meta) * mask).mean()
        else:
            real = torch.tensor(tar_real)
            per = F.cross_entropy(logits.reshape(-1, logits.shape[-1]), real.reshape(-1), reduction="none")
            recon = cfg.recon_weight * (per * (real.reshape(-1) != 0).to(dtype)).mean()
    total = recon + clas
    return ({"recon_loss": recon, "class_loss": clas, "total_loss": total},
            {"recon": logits, "class_logits": cls_logits, "embedding": emb})


class TorchTrainState:
    """fp32 parameters + Keras-Adam slots as torch tensors (the CPU-baseline leg of bench.py)."""

    def __init__(self, params_np, dtype=torch.float32):
        self.dtype = dtype
        self.P = {k: torch.tensor(v, dtype=dtype, requires_grad=True) for k, v in params_np.items()}
        self.m = {k: torch.zeros_like(v) for k, v in self.P.items()}
        self.v = {k: torch.zeros_like(v) for k, v in self.P.items()}
        self.iterations = 0


def train_step(state, cfg, inp, tar, labels, drops=None, beta1=0.9, beta2=0.98, eps=1e-9, warmup_steps=5000):
    """One model_trainer call (models/sketchformer.py:325-349): forward, tape gradient, Keras Adam with the WarmupDecay
    schedule evaluated on the pre-increment step counter (builders/schedulers.py:13-31; tf.keras Adam: epsilon outside the
    square root, bias correction folded into the step size)."""
    losses, _ = forward_losses(state.P, cfg, inp, tar, labels, drops, state.dtype)
    grads = torch.autograd.grad(losses["total_loss"], list(state.P.values()), allow_unused=True)
    step = float(state.iterations)
    lr = (cfg.d_model ** -0.5) * min(step ** -0.5 if step > 0 else float("inf"), step * warmup_steps ** -1.5)
    t = state.iterations + 1
    alpha = lr * math.sqrt(1.0 - beta2 ** t) / (1.0 - beta1 ** t)
    with torch.no_grad():
        for (k, w), g in zip(state.P.items(), grads):
            if g is None:
                continue
            state.m[k].mul_(beta1).add_(g, alpha=1.0 - beta1)
            state.v[k].mul_(beta2).addcmul_(g, g, value=1.0 - beta2)
            w.addcdiv_(state.m[k], state.v[k].sqrt().add_(eps), value=-alpha)
    state.iterations += 1
    return {k: float(v.detach()) for k, v in losses.items()}
