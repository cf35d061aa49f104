"""Thin torch-tensor front-ends over the C ABI (one function per kernel family).

PyTorch only owns the device memory and the stream; every function checks
dtype / device / contiguity, passes raw pointers to libskf.so and raises
``SkfError`` on failure.  No CPU path exists.
"""
import ctypes as C

import torch

from . import _lib


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _p(t):
    if t is None:
        return None
    if not t.is_cuda:
        raise _lib.SkfError("sketchformer_amd ops need CUDA(HIP) tensors; got a CPU tensor (no CPU fallback)")
    return C.c_void_p(t.data_ptr())


def _prec(precision):
    return _lib.default_precision() if precision is None else int(precision)


def _f32(t, name):
    if t.dtype != torch.float32:
        raise TypeError("%s must be float32" % name)
    if t.stride(-1) != 1:
        raise ValueError("%s must have a unit innermost stride" % name)
    return t


def _ws(nbytes, device):
    return torch.empty(max(int(nbytes), 256), dtype=torch.uint8, device=device)


def new_step_state(device, iterations=0):
    """Device-resident per-step scalars {int64 iterations; f32 lr, alpha; u32 drop_key, pad}."""
    n = _lib.load().skf_step_state_bytes()
    st = torch.zeros(n // 8 + 1, dtype=torch.int64, device=device)
    st[0] = iterations
    return st


def step_prologue(state, schedule=0, p0=128.0, p1=5000 ** -1.5, p2=0.0, p3=0.0, beta1=0.9, beta2=0.98, seed=0):
    _lib.call("skf_step_prologue", _p(state), schedule, p0, p1, p2, p3, beta1, beta2, seed, _stream())


def step_epilogue(state):
    _lib.call("skf_step_epilogue", _p(state), _stream())


def read_step_state(state):
    """-> dict(iterations, lr, alpha, drop_key) (host sync)."""
    raw = state.cpu().numpy().tobytes()
    import struct
    it, lr, alpha, key, _ = struct.unpack("<qffII", raw[:24])
    return {"iterations": it, "lr": lr, "alpha": alpha, "drop_key": key}


def target_live_len(tar, Ld):
    """(B, >= Ld + 1) int64 target tokens -> int32 (B,): 1 + last decoder row t < Ld trained on a non-PAD token tar[b, t+1]."""
    assert tar.dtype == torch.int64 and tar.dim() == 2 and tar.shape[1] > Ld
    out = torch.empty(tar.shape[0], dtype=torch.int32, device=tar.device)
    _lib.call("skf_target_live_len", _p(tar), tar.stride(0), tar.shape[0], Ld, _p(out), _stream())
    return out


def row_blocks(live_len, rows_per_sample, granule):
    """{n_live, n_blocks, live block ids, dead block ids} (int32) over `granule`-row blocks of the (B * rows_per_sample) rows."""
    B = live_len.shape[0]
    n = _lib.load().skf_row_blocks_bytes(B * rows_per_sample, granule) // 4
    out = torch.empty(n, dtype=torch.int32, device=live_len.device)
    _lib.call("skf_row_blocks_build", _p(live_len), B, rows_per_sample, granule, _p(out), _stream())
    return out


def relu_bits(M, N, K, device, precision=None):
    """Zeroed sign-bit buffer for gemm(..., relu_bits_out= / relu_bits_in=), or None when the shape has no such path."""
    n = _lib.load().skf_gemm_relu_bits_bytes(M, N, K, _prec(precision))
    return torch.zeros(n // 8, dtype=torch.int64, device=device) if n else None


def gemm(a, b, a_kcontig=True, b_kcontig=False, bias=None, act=0, relu_src=None, out=None, accumulate=False,
         splits=1, bias_grad=None, precision=None, row_blocks=None, row_block_rows=0, relu_bits_out=None, relu_bits_in=None):
    """C[M,N] (+)= opA(a) . opB(b).  a: [M,K] (a_kcontig) or [K,M]; b: [K,N] or [N,K] (b_kcontig).
    precision: SKF_PREC_* (0 fp32 MFMA, 6 bf16x6, 3 bf16x3); None = _lib.default_precision().
    row_blocks: list from ``row_blocks()`` - dgrad form with 16-row blocks (dead rows of a are zero), or the weight
    gradient (splits > 1 / bias_grad) with 32-row blocks over the contraction rows."""
    _f32(a, "a"); _f32(b, "b")
    M, K = (a.shape[0], a.shape[1]) if a_kcontig else (a.shape[1], a.shape[0])
    N = b.shape[0] if b_kcontig else b.shape[1]
    assert (b.shape[1] if b_kcontig else b.shape[0]) == K, "inner dimensions differ"
    if out is None:
        out = torch.empty(M, N, dtype=torch.float32, device=a.device)
    ws = None
    wsb = 0
    if splits > 1 or bias_grad is not None:
        wsb = _lib.load().skf_gemm_workspace_bytes(M, N, K, splits, 1)
        ws = _ws(wsb, a.device)
    if row_blocks is not None and (splits > 1 or bias_grad is not None):
        # the weight gradient in its two phases: partial tiles over the live contraction blocks, then the slab reduction
        assert not a_kcontig and not b_kcontig
        used = C.c_int(0)
        _lib.call("skf_gemm_wgrad_partial_rows", M, N, K, _p(a), a.stride(0), _p(b), b.stride(0), splits,
                  int(bias_grad is not None), _p(ws), wsb, C.byref(used), _prec(precision), _p(row_blocks), row_block_rows, _stream())
        _lib.call("skf_splitk_reduce", _p(ws), used.value, M, N, _p(out), out.stride(0), int(accumulate), _p(bias_grad), 0, _stream())
        return out
    if relu_bits_out is not None or relu_bits_in is not None:
        _lib.call("skf_gemm_f32_bits", int(a_kcontig), int(b_kcontig), M, N, K, _p(a), a.stride(0), _p(b), b.stride(0),
                  _p(out), out.stride(0), _p(bias), act, _p(relu_src), relu_src.stride(0) if relu_src is not None else 0,
                  int(accumulate), splits, _p(bias_grad), 0, _p(ws), wsb, _prec(precision), _p(row_blocks), row_block_rows,
                  _p(relu_bits_out), _p(relu_bits_in), _stream())
        return out
    if row_blocks is not None:
        _lib.call("skf_gemm_f32_rows", int(a_kcontig), int(b_kcontig), M, N, K, _p(a), a.stride(0), _p(b), b.stride(0),
                  _p(out), out.stride(0), _p(bias), act, _p(relu_src), relu_src.stride(0) if relu_src is not None else 0,
                  int(accumulate), splits, _p(bias_grad), 0, _p(ws), wsb, _prec(precision), _p(row_blocks), row_block_rows, _stream())
        return out
    _lib.call("skf_gemm_f32", int(a_kcontig), int(b_kcontig), M, N, K, _p(a), a.stride(0), _p(b), b.stride(0),
              _p(out), out.stride(0), _p(bias), act, _p(relu_src), relu_src.stride(0) if relu_src is not None else 0,
              int(accumulate), splits, _p(bias_grad), 0, _p(ws), wsb, _prec(precision), _stream())
    return out


def sample_order(mask_a=None, mask_b=None):
    """uint8 padding masks (B, La) / (B, Lb), 1 = padded -> int32 (B,): the samples by unmasked positions, most first (skf_sample_order)."""
    m = mask_a if mask_a is not None else mask_b
    out = torch.empty(m.shape[0], dtype=torch.int32, device=m.device)
    _lib.call("skf_sample_order", _p(mask_a), mask_a.stride(0) if mask_a is not None else 0, mask_a.shape[1] if mask_a is not None else 0,
              _p(mask_b), mask_b.stride(0) if mask_b is not None else 0, mask_b.shape[1] if mask_b is not None else 0, m.shape[0], _p(out),
              _stream())
    return out


def attention_fwd(q, k, v, num_heads, key_mask=None, causal=False, precision=None, sample_order=None):
    """q (B,Lq,d) k,v (B,Lk,d) views with unit inner stride -> o (B,Lq,d), stats (B,H,Lq,2).
    sample_order: optional int32 (B,) from ``sample_order`` - workgroup numbering only, never a result bit."""
    B, Lq, d = q.shape
    Lk = k.shape[1]
    o = torch.empty(B, Lq, d, dtype=torch.float32, device=q.device)
    stats = torch.empty(B, num_heads, Lq, 2, dtype=torch.float32, device=q.device)
    _lib.call("skf_attention_fwd_ordered", _p(q), q.stride(1), _p(k), k.stride(1), _p(v), v.stride(1), _p(key_mask),
              key_mask.stride(0) if key_mask is not None else 0, int(causal), B, num_heads, Lq, Lk, d // num_heads,
              _p(o), o.stride(1), _p(stats), _prec(precision), _p(sample_order), _stream())
    return o, stats


def attention_weights(q, k, num_heads, key_mask=None, causal=False):
    """softmax(q.k/sqrt(dh) + mask*-1e9) as the (B,H,Lq,Lk) tensor builders/utils.py:105 returns; q (B,Lq,d), k (B,Lk,d) views."""
    B, Lq, d = q.shape
    Lk = k.shape[1]
    w = torch.empty(B, num_heads, Lq, Lk, dtype=torch.float32, device=q.device)
    _lib.call("skf_attention_weights", _p(q), q.stride(1), _p(k), k.stride(1), _p(key_mask),
              key_mask.stride(0) if key_mask is not None else 0, int(causal), B, num_heads, Lq, Lk, d // num_heads, _p(w), _stream())
    return w


def attention_fwd_float_mask(q, k, v, num_heads, mask=None, return_weights=False):
    """softmax(q.k/sqrt(dh) + mask * -1e9) . v for ANY float mask broadcastable to (B,H,Lq,Lk) (builders/utils.py:90-105);
    q (B,Lq,d), k/v (B,Lk,d) views -> (o (B,Lq,d), weights (B,H,Lq,Lk) or None).  skf_attention_fwd_float_mask."""
    B, Lq, d = q.shape
    Lk = k.shape[1]
    o = torch.empty(B, Lq, d, dtype=torch.float32, device=q.device)
    w = torch.empty(B, num_heads, Lq, Lk, dtype=torch.float32, device=q.device) if return_weights else None
    sb = sh = sq = 0
    if mask is not None:
        mask = torch.as_tensor(mask, device=q.device).detach().to(torch.float32)
        while mask.dim() < 4:
            mask = mask[None]
        if mask.stride(-1) != 1 and mask.shape[-1] != 1:
            mask = mask.contiguous()
        mask = torch.broadcast_to(mask, (B, num_heads, Lq, Lk))
        if mask.stride(-1) != 1:                 # a broadcast key axis: materialise (the kernel wants unit stride there)
            mask = mask.contiguous()
        sb, sh, sq = mask.stride(0), mask.stride(1), mask.stride(2)
    _lib.call("skf_attention_fwd_float_mask", _p(q), q.stride(1), _p(k), k.stride(1), _p(v), v.stride(1), _p(mask), sb, sh, sq,
              B, num_heads, Lq, Lk, d // num_heads, _p(o), o.stride(1), _p(w), _stream())
    return o, w


def row_mean(a, b=None, mode=0):
    """mean over the last axis of a (mode 0), |a - b| (1) or (a - b)^2 (2) -> a.shape[:-1]  (skf_row_mean)"""
    a = torch.as_tensor(a).detach().to(device="cuda", dtype=torch.float32).contiguous()
    if b is not None:
        b = torch.as_tensor(b, device=a.device).detach().to(torch.float32).expand_as(a).contiguous()
    cols = a.shape[-1]
    out = torch.empty(a.shape[:-1], dtype=torch.float32, device=a.device)
    _lib.call("skf_row_mean", _p(a), _p(b), a.numel() // cols, cols, int(mode), _p(out), _stream())
    return out


def attention_decode(q, k, v, num_heads, n_keys=None, key_mask=None, key_limit=None, key_limit_all=0, step=None,
                     k_new=None, v_new=None, limit_from_step=False):
    """One query row per (sample, head): q (B,d), k/v (B,Lcap,d) cache views with unit inner stride, the first
    n_keys rows of each sample are used -> o (B,d)."""
    B, d = q.shape
    Lk = k.shape[1] if n_keys is None else int(n_keys)
    assert k.stride(0) == v.stride(0) and k.stride(1) == v.stride(1)
    o = torch.empty(B, d, dtype=torch.float32, device=q.device)
    _lib.call("skf_attention_decode", _p(q), q.stride(0), _p(k), _p(v), k.stride(1), k.stride(0), _p(key_mask),
              key_mask.stride(0) if key_mask is not None else 0, _p(key_limit), int(key_limit_all), B, num_heads, Lk,
              d // num_heads, _p(o), o.stride(0), _p(step), _p(k_new), _p(v_new), k_new.stride(0) if k_new is not None else 0,
              int(limit_from_step), _stream())
    return o


def attention_bwd(q, k, v, o, do, stats, num_heads, key_mask=None, causal=False, precision=None, q_live_len=None, two_pass=False,
                  sample_order=None):
    """q_live_len: optional int32 (B,) - query rows at or behind it have do == 0 exactly (``target_live_len``).
    two_pass: OR SKF_ATTN_TWO_PASS into the precision argument (the two-pass kernel of skf_attention_bwd2.hip also where the
    dispatch would take the one-pass kernel: head size 16)."""
    B, Lq, d = q.shape
    Lk = k.shape[1]
    dq = torch.full((B, Lq, d), float("nan"), dtype=torch.float32, device=q.device)
    dk = torch.empty(B, Lk, d, dtype=torch.float32, device=q.device)
    dv = torch.empty(B, Lk, d, dtype=torch.float32, device=q.device)
    _lib.call("skf_attention_bwd_ordered", _p(q), q.stride(1), _p(k), k.stride(1), _p(v), v.stride(1), _p(o), o.stride(1),
              _p(do), do.stride(1), _p(stats), _p(key_mask), key_mask.stride(0) if key_mask is not None else 0,
              int(causal), B, num_heads, Lq, Lk, d // num_heads, _p(dq), dq.stride(1), _p(dk), dk.stride(1),
              _p(dv), dv.stride(1), _prec(precision) | (_lib.ATTN_TWO_PASS if two_pass else 0), _p(q_live_len), _p(sample_order), _stream())
    return dq, dk, dv


def padding_mask(tokens, L=None):
    B, ld = tokens.shape
    L = L or ld
    out = torch.empty(B, L, dtype=torch.uint8, device=tokens.device)
    _lib.call("skf_padding_mask", _p(tokens), ld, B, L, _p(out), _stream())
    return out


def embed_fwd(tokens, table, pos, L=None, rate=0.0, site=0, state=None):
    B, ld = tokens.shape
    L = L or ld
    V, d = table.shape
    out = torch.empty(B, L, d, dtype=torch.float32, device=table.device)
    _lib.call("skf_embed_fwd", _p(tokens), ld, B, L, _p(table), V, d, _p(pos), _p(out), rate, site, _p(state), _stream())
    return out


def embed_bwd(tokens, dx, vocab, L=None, rate=0.0, site=0, state=None):
    B, ld = tokens.shape
    L = L or ld
    d = dx.shape[-1]
    dtable = torch.zeros(vocab, d, dtype=torch.float32, device=dx.device)
    _lib.call("skf_embed_bwd", _p(tokens), ld, B, L, _p(dx), vocab, d, _p(dtable), rate, site, _p(state), _stream())
    return dtable


def embed_bwd_sorted(tokens, dx, vocab, L=None, rate=0.0, site=0, state=None, prefill=None):
    """The two-launch form (skf_embed_sort + skf_embed_bwd_sorted); dtable starts as `prefill` (garbage is fine: every row is
    written) to show that no pre-zeroed table is needed."""
    B, ld = tokens.shape
    L = L or ld
    d = dx.shape[-1]
    dtable = torch.full((vocab, d), float("nan") if prefill is None else prefill, dtype=torch.float32, device=dx.device)
    nbytes = _lib.load().skf_embed_sort_workspace_bytes(B, L, vocab)
    ws = torch.empty((nbytes + 3) // 4, dtype=torch.int32, device=dx.device)
    _lib.call("skf_embed_sort", _p(tokens), ld, B, L, vocab, _p(dtable), d, _p(ws), nbytes, _stream())
    _lib.call("skf_embed_bwd_sorted", _p(ws), B, L, _p(dx), vocab, d, _p(dtable), rate, site, _p(state), _stream())
    return dtable


def layernorm_residual_fwd(x, y, gamma, beta, rate=0.0, site=0, state=None):
    """-> out, z (= x + drop(y), written over a copy of y), stats."""
    d = x.shape[-1]
    rows = x.numel() // d
    z = y.clone()
    out = torch.empty_like(x)
    stats = torch.empty(rows, 2, dtype=torch.float32, device=x.device)
    _lib.call("skf_layernorm_residual_fwd", _p(x), _p(z), _p(gamma), _p(beta), _p(out), _p(stats), rows, d, rate, site,
              _p(state), _stream())
    return out, z, stats


def gemm_ln_residual(a, w, bias, x, gamma, beta, rate=0.0, site=0, state=None, precision=None):
    """One launch: z = x + dropout(a . w + bias), out = LayerNorm(z) -> out, z, stats (skf_gemm_ln_residual_f32;
    K = N = 128 in a split-arithmetic mode only)."""
    _f32(a, "a"); _f32(w, "w"); _f32(x, "x")
    M, K = a.shape
    N = w.shape[1]
    z = torch.empty(M, N, dtype=torch.float32, device=a.device)
    out = torch.empty_like(z)
    stats = torch.empty(M, 2, dtype=torch.float32, device=a.device)
    _lib.call("skf_gemm_ln_residual_f32", M, N, K, _p(a), a.stride(0), _p(w), w.stride(0), _p(bias), _p(x), _p(gamma), _p(beta),
              _p(z), _p(out), _p(stats), rate, site, _p(state), _prec(precision), _stream())
    return out, z, stats


def ffn_fused_supported(M, d, dff, precision=None):
    return bool(_lib.load().skf_ffn_fused_supported(M, d, dff, _prec(precision)))


def ffn_weight_images(pairs, transpose, precision=None):
    """pairs: [(W1 [d, dff], W2 [dff, d]), ...] -> one uint8 tensor per pair holding the pre-split operand image
    (transpose=False: the forward's, True: the input gradient's); all pairs in ONE launch (skf_ffn_weight_images)."""
    n = len(pairs)
    d, dff = pairs[0][0].shape
    nbytes = _lib.load().skf_ffn_image_bytes(d, dff, _prec(precision))
    if not nbytes:
        raise _lib.SkfError("no fused feed-forward path for d=%d dff=%d" % (d, dff))
    imgs = [torch.empty(nbytes, dtype=torch.uint8, device=pairs[0][0].device) for _ in range(n)]
    PA = C.c_void_p * n
    IA = C.c_int * n
    for w1, w2 in pairs:
        _f32(w1, "W1"); _f32(w2, "W2")
    _lib.call("skf_ffn_weight_images", n, PA(*[w1.data_ptr() for w1, _ in pairs]), IA(*[w1.stride(0) for w1, _ in pairs]),
              PA(*[w2.data_ptr() for _, w2 in pairs]), IA(*[w2.stride(0) for _, w2 in pairs]), IA(*([int(bool(transpose))] * n)),
              PA(*[im.data_ptr() for im in imgs]), d, dff, _prec(precision), _stream())
    return imgs


def ffn_fused_fwd(x, image, b1, b2, gamma, beta, dff, rate=0.0, site=0, state=None, precision=None, proj=None):
    """One launch: h = relu(x.W1 + b1), z = x + dropout(h.W2 + b2), out = LayerNorm(z) -> out, z, stats, h, sign bits.
    proj = (image of Wp [d, n], bias [n]): also out . Wp + bias in the same launch -> (..., proj_out)."""
    _f32(x, "x")
    M, d = x.shape
    h = torch.empty(M, dff, dtype=torch.float32, device=x.device)
    nb = _lib.load().skf_ffn_relu_bits_bytes(M, d, dff, _prec(precision))
    bits = torch.zeros(max(nb // 8, 1), dtype=torch.int64, device=x.device)
    z = torch.empty_like(x)
    out = torch.empty_like(x)
    stats = torch.empty(M, 2, dtype=torch.float32, device=x.device)
    if proj is not None:
        pimg, pbias = proj
        po = torch.empty(M, pbias.numel(), dtype=torch.float32, device=x.device)
        _lib.call("skf_ffn_fused_fwd_proj_f32", M, d, dff, _p(x), _p(image), _p(b1), _p(b2), _p(h), _p(bits), _p(gamma), _p(beta),
                  _p(z), _p(out), _p(stats), rate, site, _p(state), _p(pimg), _p(pbias), pbias.numel(), _p(po), _prec(precision), _stream())
        return out, z, stats, h, bits, po
    _lib.call("skf_ffn_fused_fwd_f32", M, d, dff, _p(x), _p(image), _p(b1), _p(b2), _p(h), _p(bits), _p(gamma), _p(beta),
              _p(z), _p(out), _p(stats), rate, site, _p(state), _prec(precision), _stream())
    return out, z, stats, h, bits


def ffn_block_fwd(a, residual, pre, image, b1, b2, gamma, beta, dff, rate=0.0, pre_site=0, site=0, state=None, precision=None, proj=None):
    """The forward launch in its general form (skf_ffn_block_fwd_f32): pre = (image of Wo, bias, gamma1, beta1): the launch starts at the
    attention output `a` and forms z1 = residual + dropout(a.Wo + bo), x1 = LayerNorm(z1) in front of the feed-forward block
    -> dict(z1, x1, stats1, h, bits, z, out, stats[, proj_out])."""
    _f32(a, "a")
    M, d = a.shape
    dev = a.device
    e = lambda *sh: torch.empty(*sh, dtype=torch.float32, device=dev)  # noqa: E731
    r = {"h": e(M, dff), "z": e(M, d), "out": e(M, d), "stats": e(M, 2), "z1": e(M, d), "x1": e(M, d), "stats1": e(M, 2)}
    nb = _lib.load().skf_ffn_relu_bits_bytes(M, d, dff, _prec(precision))
    r["bits"] = torch.zeros(max(nb // 8, 1), dtype=torch.int64, device=dev)
    pimg, pbias, pg, pb = pre
    P = lambda t: t.data_ptr() if t is not None else None  # noqa: E731
    blk = _lib.SkfFfnBlockFwd(struct_size=C.sizeof(_lib.SkfFfnBlockFwd), M=M, d=d, dff=dff, precision=_prec(precision), x=P(a), image=P(image),
                              b1=P(b1), b2=P(b2), h=P(r["h"]), relu_bits_out=P(r["bits"]), gamma=P(gamma), beta=P(beta), z=P(r["z"]),
                              out=P(r["out"]), stats=P(r["stats"]), rate=rate, site=site, step_state=P(state), pre_image=P(pimg),
                              pre_bias=P(pbias), pre_residual=P(residual), pre_gamma=P(pg), pre_beta=P(pb), pre_z=P(r["z1"]), pre_out=P(r["x1"]),
                              pre_stats=P(r["stats1"]), pre_site=pre_site)
    if proj is not None:
        r["proj_out"] = e(M, proj[1].numel())
        blk.proj_image, blk.proj_bias, blk.proj_out, blk.proj_n = P(proj[0]), P(proj[1]), P(r["proj_out"]), proj[1].numel()
    _lib.call("skf_ffn_block_fwd_f32", C.byref(blk), _stream())
    return r


def attn_tail_proj(a, residual, pre, proj, rate=0.0, pre_site=0, state=None, precision=None):
    """skf_ffn_block_fwd_f32 without a feed-forward image: z1 = residual + dropout(a.Wo + bo), x1 = LayerNorm(z1), proj_out = x1.Wp + bp in
    one launch (pre = (image of Wo, bo, gamma, beta), proj = (image of Wp, bp)) -> dict(z1, x1, stats1, proj_out)."""
    _f32(a, "a")
    M, d = a.shape
    e = lambda *sh: torch.empty(*sh, dtype=torch.float32, device=a.device)  # noqa: E731
    r = {"z1": e(M, d), "x1": e(M, d), "stats1": e(M, 2), "proj_out": e(M, proj[1].numel())}
    pimg, pbias, pg, pb = pre
    P = lambda t: t.data_ptr() if t is not None else None  # noqa: E731
    blk = _lib.SkfFfnBlockFwd(struct_size=C.sizeof(_lib.SkfFfnBlockFwd), M=M, d=d, dff=4 * d, precision=_prec(precision), x=P(a), rate=rate,
                              step_state=P(state), pre_image=P(pimg), pre_bias=P(pbias), pre_residual=P(residual), pre_gamma=P(pg), pre_beta=P(pb),
                              pre_z=P(r["z1"]), pre_out=P(r["x1"]), pre_stats=P(r["stats1"]), pre_site=pre_site, proj_image=P(proj[0]),
                              proj_bias=P(proj[1]), proj_out=P(r["proj_out"]), proj_n=proj[1].numel())
    _lib.call("skf_ffn_block_fwd_f32", C.byref(blk), _stream())
    return r


def ffn_fused_bwd(dy, image_t, bits, dff, dx=None, row_blocks=None, precision=None):
    """One launch: dh = (dy.W2^T) o relu'(h), dx (+)= dh.W1^T -> dh, dx (accumulated into `dx` when given)."""
    _f32(dy, "dy")
    M, d = dy.shape
    dh = torch.empty(M, dff, dtype=torch.float32, device=dy.device)
    acc = dx is not None
    if dx is None:
        dx = torch.empty_like(dy)
    _lib.call("skf_ffn_fused_bwd_f32", M, d, dff, _p(dy), _p(image_t), _p(bits), _p(dh), _p(dx), int(acc), _p(row_blocks),
              16 if row_blocks is not None else 0, _prec(precision), _stream())
    return dh, dx


def ffn_fused_bwd_ln(dout, z, stats, gamma, image_t, bits, dff, rate=0.0, site=0, state=None, row_blocks=None, precision=None):
    """One launch from the gradient of the closing LayerNorm's output: -> dy, dh, dx (= dz + dh.W1^T), dgamma, dbeta."""
    _f32(dout, "dout")
    M, d = dout.shape
    lib = _lib.load()
    n = lib.skf_ffn_fused_ln_partials(M)
    part = torch.empty(n, 2, d, dtype=torch.float32, device=dout.device)
    dy, dx = torch.empty_like(dout), torch.empty_like(dout)
    dh = torch.empty(M, dff, dtype=torch.float32, device=dout.device)
    _lib.call("skf_ffn_fused_bwd_ln_f32", M, d, dff, _p(dout), _p(z), _p(stats), _p(gamma), rate, site, _p(state), _p(image_t), _p(bits),
              _p(dy), _p(dh), _p(dx), _p(part), part.numel() * 4, _p(row_blocks), 16 if row_blocks is not None else 0, _prec(precision),
              _stream())
    g = part.sum(0)
    return dy, dh, dx, g[0], g[1]


def dense_weight_image(w, transpose, precision=None):
    """Pre-split MFMA operand image of B = w [K, N] (transpose=False) or w^T (w is [N, K]) - skf_dense_weight_images."""
    _f32(w, "w")
    K, N = (w.shape[1], w.shape[0]) if transpose else (w.shape[0], w.shape[1])
    nbytes = _lib.load().skf_dense_image_bytes(K, N, _prec(precision))
    if not nbytes:
        raise _lib.SkfError("no operand image for K=%d N=%d" % (K, N))
    img = torch.empty(nbytes, dtype=torch.uint8, device=w.device)
    PA, IA = C.c_void_p * 1, C.c_int * 1
    _lib.call("skf_dense_weight_images", 1, PA(w.data_ptr()), IA(w.stride(0)), IA(int(bool(transpose))), IA(K), IA(N), PA(img.data_ptr()),
              _prec(precision), _stream())
    return img


def layernorm_bwd_dgrad(dout, z, stats, gamma, image_t, rate=0.0, site=0, state=None, row_blocks=None, precision=None, lead=None):
    """One launch: dz = LayerNorm'(dout), dy = dropout'(dz), da = dy . W^T -> dz, dy, da, dgamma, dbeta (skf_layernorm_bwd_dgrad_f32).
    lead = (rows a [M, d], transposed image of Wl): the LayerNorm output's gradient is dout + a . Wl^T (skf_layernorm_bwd_dgrad_lead_f32)."""
    _f32(dout, "dout")
    M, d = dout.shape
    n = _lib.load().skf_layernorm_bwd_dgrad_partials(M)
    part = torch.empty(n, 2, d, dtype=torch.float32, device=dout.device)
    dz, dy, da = torch.empty_like(dout), torch.empty_like(dout), torch.empty_like(dout)
    la, li = lead if lead is not None else (None, None)
    _lib.call("skf_layernorm_bwd_dgrad_lead_f32", M, d, _p(dout), _p(la), _p(li), _p(z), _p(stats), _p(gamma), rate, site, _p(state), _p(image_t),
              _p(dz), _p(dy), _p(da), _p(part), part.numel() * 4, _p(row_blocks), 16 if row_blocks is not None else 0, _prec(precision), _stream())
    g = part.sum(0)
    return dz, dy, da, g[0], g[1]


def layernorm_residual_bwd(dout, z, stats, gamma, rate=0.0, site=0, state=None):
    d = dout.shape[-1]
    rows = dout.numel() // d
    dz = torch.empty_like(dout)
    dy = torch.empty_like(dout) if rate > 0 else None
    dg = torch.empty(d, dtype=torch.float32, device=dout.device)
    db = torch.empty(d, dtype=torch.float32, device=dout.device)
    wsb = _lib.load().skf_layernorm_bwd_workspace_bytes(rows, d)
    ws = _ws(wsb, dout.device)
    _lib.call("skf_layernorm_residual_bwd", _p(dout), _p(z), _p(stats), _p(gamma), _p(dz), _p(dy), _p(dg), _p(db), rows,
              d, rate, site, _p(state), _p(ws), wsb, _stream())
    return dz, (dy if dy is not None else dz), dg, db


def softmax_ce(logits, target, tgt_cols, tgt_off=0, mask_pad=False, scale=1.0, want_probs=False, write_grad=True):
    """In place: logits (rows, ncls) become the gradient.  target: int64 (B, tgt_ld)."""
    rows, ncls = logits.shape
    row_loss = torch.empty(rows, dtype=torch.float32, device=logits.device)
    row_hit = torch.empty(rows, dtype=torch.float32, device=logits.device)
    probs = torch.empty(rows, ncls, dtype=torch.float32, device=logits.device) if want_probs else None
    _lib.call("skf_softmax_ce", _p(logits), logits.stride(0), rows, ncls, _p(target), target.stride(0), tgt_cols, tgt_off,
              int(mask_pad), scale, _p(row_loss), _p(row_hit), _p(probs), int(write_grad), _stream())
    return row_loss, row_hit, probs


def pool_fwd(u, Vw, x):
    B, L, U = u.shape
    d = x.shape[-1]
    a = torch.empty(B, L, dtype=torch.float32, device=x.device)
    emb = torch.empty(B, d, dtype=torch.float32, device=x.device)
    _lib.call("skf_pool_fwd", _p(u), _p(Vw), _p(x), B, L, U, d, _p(a), _p(emb), _stream())
    return a, emb


def pool_bwd(u, Vw, x, a, demb):
    """-> dpre (overwrites a copy of u), dx_direct, dV."""
    B, L, U = u.shape
    d = x.shape[-1]
    dpre = u.clone()
    dx = torch.empty_like(x)
    dV = torch.empty(U, dtype=torch.float32, device=x.device)
    ws = _ws(B * U * 4, x.device)
    _lib.call("skf_pool_bwd", _p(dpre), _p(Vw), _p(x), _p(a), _p(demb), B, L, U, d, _p(dx), _p(dV), _p(ws), B * U * 4,
              _stream())
    return dpre, dx, dV


def expander_fwd(emb, w, bias):
    B, d = emb.shape
    L = w.numel()
    pre = torch.empty(B, L, d, dtype=torch.float32, device=emb.device)
    _lib.call("skf_expander_fwd", _p(emb), _p(w), _p(bias), B, L, d, _p(pre), _stream())
    return pre


def expander_bwd(dpre, emb, w):
    B, L, d = dpre.shape
    demb = torch.empty(B, d, dtype=torch.float32, device=emb.device)
    dw = torch.empty(L, dtype=torch.float32, device=emb.device)
    db = torch.empty(L, dtype=torch.float32, device=emb.device)
    ws = _ws(2 * B * L * 4, emb.device)
    _lib.call("skf_expander_bwd", _p(dpre), _p(emb), _p(w), B, L, d, _p(demb), 0, _p(dw), _p(db), _p(ws), 2 * B * L * 4,
              _stream())
    return demb, dw, db


def adam_step(w, g, m, v, state, grad_scale=1.0, beta1=0.9, beta2=0.98, eps=1e-9):
    _lib.call("skf_adam_step", _p(w), _p(g), _p(m), _p(v), w.numel(), _p(state), grad_scale, beta1, beta2, eps, _stream())


def dropout(x, rate, site=0, state=None):
    """Inverted dropout (tf.keras.layers.Dropout in training mode) with the kernels' counter-based mask of (step key, site, element):
    skf_dropout.  rate 0 or no state = the input itself."""
    if rate <= 0.0 or state is None:
        return x
    _f32(x, "x")
    y = torch.empty_like(x)
    _lib.call("skf_dropout", _p(x), _p(y), x.numel(), float(rate), int(site), _p(state), _stream())
    return y


def embed_continuous_fwd(x, W, bias, pos, L=None, rate=0.0, site=0, state=None):
    """Encoder / Decoder embed stage in continuous mode (builders/layers/transformer.py:276,288-296): Dense(5 -> d) of the stroke-5
    rows, * sqrt(d), + pos[:L], dropout.  x (B, ld, 5) float32."""
    _f32(x, "x")
    B, ld = x.shape[0], x.shape[1]
    L = L or ld
    d = W.shape[1]
    out = torch.empty(B, L, d, dtype=torch.float32, device=x.device)
    _lib.call("skf_embed_continuous_fwd", _p(x), ld, B, L, _p(W), _p(bias), d, _p(pos), _p(out), float(rate), int(site), _p(state), _stream())
    return out


def dropout_keep_mask(drop_key, site, rate, n):
    """Host replica of the kernels' counter-based keep mask (for parity tests)."""
    import numpy as np
    out = np.empty(n, dtype=np.uint8)
    _lib.call("skf_dropout_keep_mask", drop_key, site, rate, n, out.ctypes.data_as(C.c_void_p))
    return out.astype(bool)
