"""Layer objects with the constructor arguments and call signatures of builders/layers/transformer.py (:13-376).
They own Keras-initialised torch parameters and run the HIP kernels of libskf.so through ``ops`` (forward only: the train step with
its hand-written backward is the fused C-ABI step, TrainEngine).  ``training=True`` applies inverted dropout with the kernels'
counter-based masks (skf_dropout / the dropout inputs of the embedding and LayerNorm kernels): every layer stack owns a
``DropoutState`` whose step counter advances once per training call, every Dropout layer of the reference is one numbered site.
"""
import itertools
import math

import torch

from ... import ops
from ..utils import positional_encoding, scaled_dot_product_attention


def _glorot(fan_in, fan_out, device):
    lim = math.sqrt(6.0 / (fan_in + fan_out))
    return (torch.rand(fan_in, fan_out, device=device) * 2 - 1) * lim


class DropoutState(object):
    """Device-side step scalars (skf_step_prologue) for the training-mode dropout of a layer stack: ``advance()`` once per
    training call draws fresh masks (tf.keras.layers.Dropout draws new ones per call), ``site()`` numbers the Dropout layers."""

    _seeds = itertools.count()          # every state object draws from its own stream: two stacks (or two stand-alone layers) must
                                        # not repeat each other's keep masks (tf.keras.layers.Dropout instances are independent)

    def __init__(self, device="cuda", seed=None):
        self.device, self.calls = device, 0
        self.seed = next(DropoutState._seeds) if seed is None else seed
        self.state = None
        self.in_stack_call = False          # True while the owning stack's call() runs: its layers then share the advanced state
        self._sites = itertools.count()

    def site(self):
        return next(self._sites)

    def advance(self):
        if self.state is None:
            self.state = ops.new_step_state(self.device)
        self.state[0] = self.calls
        ops.step_prologue(self.state, seed=self.seed)
        self.calls += 1
        return self.state


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

    def residual(self, x, y, rate=0.0, site=0, state=None):
        """LayerNormalization(1e-6)(x + Dropout(rate)(y)) in one launch (builders/layers/transformer.py:217-222)"""
        out, _, _ = ops.layernorm_residual_fwd(x.contiguous(), y.contiguous(), self.gamma, self.beta, rate=rate if state is not None else 0.0,
                                               site=site, state=state)
        return out


class _Droppable(object):
    """shared by the layer classes: the stack's DropoutState (or a private one when a layer is used on its own)"""
    rate = 0.0
    drop = None

    def _drop_state(self, training):
        """step state for this call, or None when nothing is dropped; a layer used stand-alone advances its own state, a layer
        inside a stack uses the state its stack advanced"""
        if not training or self.rate <= 0.0:
            return None
        if self.drop is None:
            self.drop = DropoutState(self._device)
            self._own_drop = True
        if getattr(self, "_own_drop", False) or not self.drop.in_stack_call:
            return self.drop.advance()      # stand-alone layer, or a stack's layer called on its own: fresh masks per call
        return self.drop.state


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


class EncoderLayer(_Droppable):
    def __init__(self, d_model, num_heads, dff, rate=0.1, device="cuda", drop=None):
        self.mha, self.ffn, self.rate = MultiHeadAttention(d_model, num_heads, device), _FFN(d_model, dff, device), rate
        self.layernorm1, self.layernorm2 = LayerNorm(d_model, device), LayerNorm(d_model, device)
        self._device, self.drop = device, drop
        sites = drop if drop is not None else DropoutState(device)
        self.site1, self.site2 = sites.site(), sites.site()          # dropout1, dropout2 (transformer.py:212-213)
        if drop is None:
            self.drop, self._own_drop = sites, True

    def call(self, x, training, mask):
        st = self._drop_state(training)
        attn_output, _ = self.mha(x, x, x, mask)
        out1 = self.layernorm1.residual(x, attn_output, self.rate, self.site1, st)
        return self.layernorm2.residual(out1, self.ffn(out1), self.rate, self.site2, st)

    __call__ = call


class DecoderLayer(_Droppable):
    def __init__(self, d_model, num_heads, dff, rate=0.1, device="cuda", drop=None):
        self.mha1, self.mha2 = MultiHeadAttention(d_model, num_heads, device), MultiHeadAttention(d_model, num_heads, device)
        self.ffn, self.rate = _FFN(d_model, dff, device), rate
        self.layernorm1, self.layernorm2, self.layernorm3 = (LayerNorm(d_model, device) for _ in range(3))
        self._device, self.drop = device, drop
        sites = drop if drop is not None else DropoutState(device)
        self.site1, self.site2, self.site3 = sites.site(), sites.site(), sites.site()
        if drop is None:
            self.drop, self._own_drop = sites, True

    def call(self, x, enc_output, training, look_ahead_mask, padding_mask):
        st = self._drop_state(training)
        attn1, w1 = self.mha1(x, x, x, look_ahead_mask)
        out1 = self.layernorm1.residual(x, attn1, self.rate, self.site1, st)
        attn2, w2 = self.mha2(enc_output, enc_output, out1, padding_mask)
        out2 = self.layernorm2.residual(out1, attn2, self.rate, self.site2, st)
        return self.layernorm3.residual(out2, self.ffn(out2), self.rate, self.site3, st), w1, w2

    __call__ = call


class _Stack(_Droppable):
    def __init__(self, num_layers, d_model, vocab_size, maximum_position_encoding, rate, use_continuous_input, device):
        self.d_model, self.num_layers, self.rate = d_model, num_layers, rate
        self.use_continuous_input = use_continuous_input
        self._device = device
        if use_continuous_input:
            # tf.keras.layers.Dense(d_model) on the stroke-5 rows (transformer.py:275-276, 314-315)
            self.embedding = Dense(5, d_model, device=device)
        else:
            self.embedding = (torch.rand(vocab_size, d_model, device=device) - 0.5) * 0.1      # Embedding: uniform(-0.05, 0.05)
        self.pos_encoding = positional_encoding(maximum_position_encoding, d_model).to(device)
        self.drop = DropoutState(device)
        self._own_drop = True
        self.site0 = self.drop.site()                                   # self.dropout (transformer.py:286, 323)

    def _embed(self, x, st):
        rate = self.rate if st is not None else 0.0
        if self.use_continuous_input:
            x = torch.as_tensor(x).to(self.pos_encoding.device, dtype=torch.float32).contiguous()
            return ops.embed_continuous_fwd(x, self.embedding.kernel, self.embedding.bias, self.pos_encoding[0], rate=rate, site=self.site0, state=st)
        return ops.embed_fwd(torch.as_tensor(x).to(torch.int64).to(self.embedding.device).contiguous(),
                             self.embedding, self.pos_encoding[0], rate=rate, site=self.site0, state=st)


class Encoder(_Stack):
    def __init__(self, num_layers, d_model, num_heads, dff, input_vocab_size, maximum_position_encoding=1000, rate=0.1,
                 use_continuous_input=False, device="cuda"):
        super().__init__(num_layers, d_model, input_vocab_size, maximum_position_encoding, rate, use_continuous_input, device)
        self.enc_layers = [EncoderLayer(d_model, num_heads, dff, rate, device, drop=self.drop) for _ in range(num_layers)]

    def call(self, x, training, mask):
        st = self._drop_state(training)
        self.drop.in_stack_call = True
        try:
            x = self._embed(x, st)
            for layer in self.enc_layers:
                x = layer(x, training, mask)
        finally:
            self.drop.in_stack_call = False
        return x

    __call__ = call


class Decoder(_Stack):
    def __init__(self, num_layers, d_model, num_heads, dff, target_vocab_size, maximum_position_encoding=1000, rate=0.1,
                 use_continuous_input=False, device="cuda"):
        super().__init__(num_layers, d_model, target_vocab_size, maximum_position_encoding, rate, use_continuous_input, device)
        self.dec_layers = [DecoderLayer(d_model, num_heads, dff, rate, device, drop=self.drop) for _ in range(num_layers)]

    def call(self, x, enc_output, training, look_ahead_mask, padding_mask):
        st = self._drop_state(training)
        attention_weights = {}
        self.drop.in_stack_call = True
        try:
            x = self._embed(x, st)
            for i, layer in enumerate(self.dec_layers):
                x, block1, block2 = layer(x, enc_output, training, look_ahead_mask, padding_mask)
                attention_weights['decoder_layer{}_block1'.format(i + 1)] = block1
                attention_weights['decoder_layer{}_block2'.format(i + 1)] = block2
        finally:
            self.drop.in_stack_call = False
        return x, attention_weights

    __call__ = call


class SelfAttnV1(object):
    """u = tanh(xW+b); a = softmax(uV, axis=time) (no padding mask); o = sum_t a*x  (transformer.py:13-77)."""

    def __init__(self, units=None, device="cuda"):
        self.units, self.device, self.W = units, device, None

    def build(self, fdim):
        self.units = self.units or fdim
        self.W = torch.randn(fdim, self.units, device=self.device) * 0.05
        self.b = torch.zeros(self.units, device=self.device)
        self.V = (torch.rand(self.units, 1, device=self.device) - 0.5) * 0.1

    def _pool(self, x):
        B, L, d = x.shape
        u = ops.gemm(x.reshape(-1, d).contiguous(), self.W, bias=self.b, act=2).view(B, L, -1)
        a, o = ops.pool_fwd(u, self.V[:, 0].contiguous(), x.contiguous())
        return o, a[..., None]

    def call(self, x):
        if self.W is None:
            self.build(x.shape[-1])
        return self._pool(x)

    __call__ = call

    def compute_output_shape(self, input_shape):
        return input_shape[0], self.units


class SelfAttnV2(SelfAttnV1):
    """Version 2 (transformer.py:80-137): the scorer is W (fdim, fdim), V (fdim, 1); with ``units`` a Dense(units) follows the
    pooling, so the output is (B, units) instead of (B, fdim)."""

    def build(self, fdim):
        self.embeding_layer = Dense(fdim, self.units, device=self.device) if self.units else None      # (the reference's spelling)
        self.W = torch.randn(fdim, fdim, device=self.device) * 0.05
        self.b = torch.zeros(fdim, device=self.device)
        self.V = (torch.rand(fdim, 1, device=self.device) - 0.5) * 0.1

    def call(self, x):
        if self.W is None:
            self.build(x.shape[-1])
        o, a = self._pool(x)
        if self.units:
            o = self.embeding_layer(o)
        return o, a

    __call__ = call

    def compute_output_shape(self, input_shape):
        return (input_shape[0], self.units) if self.units else (input_shape[0], input_shape[-1])


class DenseExpander(object):
    """(B, feat_in) -> (B, seq_len, feat_out): pre[b,t,c] = x[b,c] * w[t] + bias[t]; with ``feat_dim_out`` a Dense(feat_dim_out, relu)
    projects x first (transformer.py:347-376)."""

    def __init__(self, seq_len, feat_dim_out=0, device="cuda"):
        self.seq_len, self.feat_dim_out, self.device = seq_len, feat_dim_out, device
        self.project_layer = None
        lim = math.sqrt(6.0 / (1 + seq_len))
        self.kernel = (torch.rand(1, seq_len, device=device) * 2 - 1) * lim
        self.bias = torch.zeros(seq_len, device=device)

    def compute_output_shape(self, input_shape):
        return input_shape[0], self.seq_len, self.feat_dim_out if self.feat_dim_out else input_shape[-1]

    def call(self, x):
        assert x.dim() == 2, 'Error! input tensor must be 2D'
        if self.feat_dim_out:
            if self.project_layer is None:
                self.project_layer = Dense(x.shape[-1], self.feat_dim_out, "relu", self.device)
            x = self.project_layer(x)
        return ops.expander_fwd(x.contiguous(), self.kernel[0].contiguous(), self.bias)

    __call__ = call
