"""builders/utils.py of the reference (:12-105): positional encoding, masks, scaled dot-product attention."""
import numpy as np
import torch

from .. import ops
from ..engine import positional_encoding as _pe_table


def positional_encoding(position, d_model):
    """(1, position, d_model) float32 table computed on the host in float64 exactly like the reference."""
    return torch.from_numpy(_pe_table(position, d_model)[np.newaxis, ...])


def create_padding_mask(seq):
    """(B,1,1,L) float mask: 1 where token == 0 (tokens) or pad bit == 1 (continuous stroke-5)."""
    seq = torch.as_tensor(seq)
    if seq.dim() < 3:
        if seq.is_cuda and seq.dtype == torch.int64:
            m = ops.padding_mask(seq.contiguous()).to(torch.float32)
        else:
            m = (seq == 0).to(torch.float32)
    else:
        m = (seq[..., -1] == 1).to(torch.float32)
    return m[:, None, None, :]


def create_look_ahead_mask(size, device=None):
    return torch.triu(torch.ones(size, size, dtype=torch.float32, device=device), diagonal=1)


def create_masks(inp, tar):
    enc_padding_mask = create_padding_mask(inp)
    dec_padding_mask = create_padding_mask(inp)
    look_ahead_mask = create_look_ahead_mask(tar.shape[1], device=enc_padding_mask.device)
    combined_mask = torch.maximum(create_padding_mask(tar), look_ahead_mask)
    return enc_padding_mask, combined_mask, dec_padding_mask


def _decode_mask(mask, B, Lq, Lk):
    """A reference-style float mask -> (key padding bytes or None, causal flag) understood by the fused kernel, or None when the
    mask is neither a padding mask nor padding + look-ahead (fractional values, per-head or per-query patterns): those take the
    float-mask kernel, which ADDS mask * -1e9 like builders/utils.py:96-97."""
    if mask is None:
        return None, False
    mask = torch.as_tensor(mask)
    if mask.dim() == 4 and mask.shape[1] != 1:
        return None
    if not bool(((mask == 0) | (mask == 1)).all()):
        return None
    m = torch.broadcast_to(mask, (B, 1, Lq, Lk))[:, 0] != 0                       # (B,Lq,Lk) bool
    key = m[:, -1, :]                                                             # last query row: no look-ahead term
    la = torch.triu(torch.ones(Lq, Lk, dtype=torch.bool, device=m.device), diagonal=1) if Lq == Lk else None
    if torch.equal(m, key[:, None, :].expand(B, Lq, Lk)):
        causal = False
    elif la is not None and torch.equal(m, key[:, None, :] | la[None]):
        causal = True
    else:
        return None
    km = key.to(torch.uint8).contiguous() if bool(key.any()) else None
    return km, causal


# The reference returns (output, attention_weights) from every call (builders/utils.py:105) and its train step drops the weights
# (models/sketchformer.py:140-145).  The fused kernel never forms them; a caller that wants them asks - per call
# (``return_weights=True``) or for every call of the front-end (``RETURN_ATTENTION_WEIGHTS = True``) - and gets the (B,H,Lq,Lk)
# tensor from a second launch (skf_attention_weights).
RETURN_ATTENTION_WEIGHTS = False


def scaled_dot_product_attention(q, k, v, mask, return_weights=None):
    """q,k,v: (B, H, L, depth) as in the reference.  Returns (output (B,H,Lq,depth), attention_weights (B,H,Lq,Lk) or None):
    the weights are materialised on request only (see RETURN_ATTENTION_WEIGHTS)."""
    B, H, Lq, dh = q.shape
    Lk = k.shape[2]
    dec = _decode_mask(mask, B, Lq, Lk)

    def flat(x):
        return x.permute(0, 2, 1, 3).reshape(B, x.shape[2], H * dh).contiguous()
    qf, kf = flat(q), flat(k)
    want = RETURN_ATTENTION_WEIGHTS if return_weights is None else return_weights
    if dec is None:         # any other float mask (builders/utils.py:96-97 adds mask * -1e9 whatever it holds): the generic kernel
        o, w = ops.attention_fwd_float_mask(qf, kf, flat(v), H, mask=mask, return_weights=want)
        return o.view(B, Lq, H, dh).permute(0, 2, 1, 3), w
    km, causal = dec
    o, _ = ops.attention_fwd(qf, kf, flat(v), H, key_mask=km, causal=causal)
    w = ops.attention_weights(qf, kf, H, key_mask=km, causal=causal) if want else None
    return o.view(B, Lq, H, dh).permute(0, 2, 1, 3), w
