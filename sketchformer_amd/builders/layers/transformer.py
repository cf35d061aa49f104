"""Inference-mode layer objects with the constructor arguments of builders/layers/transformer.py (:13-376).
They own Keras-initialised torch parameters and run the same HIP kernels as the train step; the training
path itself is the fused C-ABI step (TrainEngine), so ``training=True`` with dropout is refused here.
"""
import math

import torch

from ... import ops
from ..utils import positional_encoding, scaled_dot_product_attention


def _glorot(fan_in, fan_out, device):
    lim = math.sqrt(6.0 / (fan_in + fan_out))
    return (torch.rand(fan_in, fan_out, device=device) * 2 - 1) * lim


class Dense(object):
    def __init__(self, fan_in, units, activation=None, device="cuda"):
        self.kernel, self.bias = _glorot(fan_in, units, device), torch.zeros(units, device=device)
        self.act = {None: 0, "relu": 1, "tanh": 2}[activation]

    def __call__(self, x):
        y = ops.gemm(x.reshape(-1, x.shape[-1]).contiguous(), self.kernel, bias=self.bias, act=self.act)
        return y.view(*x.shape[:-1], -1)


class LayerNorm(object):
    def __init__(self, d, device="cuda"):
        self.gamma, self.beta = torch.ones(d, device=device), torch.zeros(d, device=device)

    def residual(self, x, y):
        out, _, _ = ops.layernorm_residual_fwd(x.contiguous(), y.contiguous(), self.gamma, self.beta)
        return out


def _no_training(training, rate):
    if training and rate > 0:
        raise NotImplementedError("training-mode dropout runs only inside the fused train step (TrainEngine)")


class MultiHeadAttention(object):
    def __init__(self, d_model, num_heads, device="cuda"):
        assert d_model % num_heads == 0
        self.num_heads, self.d_model, self.depth = num_heads, d_model, d_model // num_heads
        self.wq, self.wk, self.wv, self.dense = (Dense(d_model, d_model, device=device) for _ in range(4))

    def split_heads(self, x, batch_size):
        return x.view(batch_size, -1, self.num_heads, self.depth).permute(0, 2, 1, 3)

    def call(self, v, k, q, mask):
        B = q.shape[0]
        q, k, v = self.split_heads(self.wq(q), B), self.split_heads(self.wk(k), B), self.split_heads(self.wv(v), B)
        o, w = scaled_dot_product_attention(q, k, v, mask)
        return self.dense(o.permute(0, 2, 1, 3).reshape(B, -1, self.d_model)), w

    __call__ = call


class _FFN(object):
    def __init__(self, d_model, dff, device):
        self.d1, self.d2 = Dense(d_model, dff, "relu", device), Dense(dff, d_model, device=device)

    def __call__(self, x):
        return self.d2(self.d1(x))


def point_wise_feed_forward_network(d_model, dff, device="cuda"):
    return _FFN(d_model, dff, device)


class EncoderLayer(object):
    def __init__(self, d_model, num_heads, dff, rate=0.1, device="cuda"):
        self.mha, self.ffn, self.rate = MultiHeadAttention(d_model, num_heads, device), _FFN(d_model, dff, device), rate
        self.layernorm1, self.layernorm2 = LayerNorm(d_model, device), LayerNorm(d_model, device)

    def call(self, x, training, mask):
        _no_training(training, self.rate)
        attn_output, _ = self.mha(x, x, x, mask)
        out1 = self.layernorm1.residual(x, attn_output)
        return self.layernorm2.residual(out1, self.ffn(out1))

    __call__ = call


class DecoderLayer(object):
    def __init__(self, d_model, num_heads, dff, rate=0.1, device="cuda"):
        self.mha1, self.mha2 = MultiHeadAttention(d_model, num_heads, device), MultiHeadAttention(d_model, num_heads, device)
        self.ffn, self.rate = _FFN(d_model, dff, device), rate
        self.layernorm1, self.layernorm2, self.layernorm3 = (LayerNorm(d_model, device) for _ in range(3))

    def call(self, x, enc_output, training, look_ahead_mask, padding_mask):
        _no_training(training, self.rate)
        attn1, w1 = self.mha1(x, x, x, look_ahead_mask)
        out1 = self.layernorm1.residual(x, attn1)
        attn2, w2 = self.mha2(enc_output, enc_output, out1, padding_mask)
        out2 = self.layernorm2.residual(out1, attn2)
        return self.layernorm3.residual(out2, self.ffn(out2)), w1, w2

    __call__ = call


class _Stack(object):
    def __init__(self, num_layers, d_model, vocab_size, maximum_position_encoding, rate, use_continuous_input, device):
        if use_continuous_input:
            raise NotImplementedError("use_continuous_input=True is not implemented on the HIP path yet")
        self.d_model, self.num_layers, self.rate = d_model, num_layers, rate
        self.embedding = (torch.rand(vocab_size, d_model, device=device) - 0.5) * 0.1      # uniform(-0.05, 0.05)
        self.pos_encoding = positional_encoding(maximum_position_encoding, d_model).to(device)

    def _embed(self, x):
        return ops.embed_fwd(torch.as_tensor(x).to(torch.int64).to(self.embedding.device).contiguous(),
                             self.embedding, self.pos_encoding[0])


class Encoder(_Stack):
    def __init__(self, num_layers, d_model, num_heads, dff, input_vocab_size, maximum_position_encoding=1000, rate=0.1,
                 use_continuous_input=False, device="cuda"):
        super().__init__(num_layers, d_model, input_vocab_size, maximum_position_encoding, rate, use_continuous_input, device)
        self.enc_layers = [EncoderLayer(d_model, num_heads, dff, rate, device) for _ in range(num_layers)]

    def call(self, x, training, mask):
        _no_training(training, self.rate)
        x = self._embed(x)
        for layer in self.enc_layers:
            x = layer(x, training, mask)
        return x

    __call__ = call


class Decoder(_Stack):
    def __init__(self, num_layers, d_model, num_heads, dff, target_vocab_size, maximum_position_encoding=1000, rate=0.1,
                 use_continuous_input=False, device="cuda"):
        super().__init__(num_layers, d_model, target_vocab_size, maximum_position_encoding, rate, use_continuous_input, device)
        self.dec_layers = [DecoderLayer(d_model, num_heads, dff, rate, device) for _ in range(num_layers)]

    def call(self, x, enc_output, training, look_ahead_mask, padding_mask):
        _no_training(training, self.rate)
        x = self._embed(x)
        for layer in self.dec_layers:
            x, _, _ = layer(x, enc_output, training, look_ahead_mask, padding_mask)
        return x, {}        # attention weights are not materialised

    __call__ = call


class SelfAttnV1(object):
    """u = tanh(xW+b); a = softmax(uV, axis=time) (no padding mask); o = sum_t a*x."""

    def __init__(self, units=None, device="cuda"):
        self.units, self.device, self.W = units, device, None

    def build(self, fdim):
        self.units = self.units or fdim
        self.W = torch.randn(fdim, self.units, device=self.device) * 0.05
        self.b = torch.zeros(self.units, device=self.device)
        self.V = (torch.rand(self.units, 1, device=self.device) - 0.5) * 0.1

    def call(self, x):
        if self.W is None:
            self.build(x.shape[-1])
        B, L, d = x.shape
        u = ops.gemm(x.reshape(-1, d).contiguous(), self.W, bias=self.b, act=2).view(B, L, -1)
        a, o = ops.pool_fwd(u, self.V[:, 0].contiguous(), x.contiguous())
        return o, a[..., None]

    __call__ = call


class DenseExpander(object):
    """(B, feat) -> (B, seq_len, feat): pre[b,t,c] = x[b,c] * w[t] + bias[t]."""

    def __init__(self, seq_len, feat_dim_out=0, device="cuda"):
        if feat_dim_out:
            raise NotImplementedError("feat_dim_out projection is not used by sketch-transformer-tf2")
        self.seq_len = seq_len
        lim = math.sqrt(6.0 / (1 + seq_len))
        self.kernel = (torch.rand(1, seq_len, device=device) * 2 - 1) * lim
        self.bias = torch.zeros(seq_len, device=device)

    def call(self, x):
        return ops.expander_fwd(x.contiguous(), self.kernel[0].contiguous(), self.bias)

    __call__ = call
