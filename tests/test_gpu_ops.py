"""Kernel-level parity: every HIP kernel family, called through the C ABI, against
the CPU oracle (float64) on the same seeded inputs.  Tolerances are fp32
accumulation-order tolerances (north star: logits within 1e-3 rel)."""
import numpy as np
import pytest
import torch

import oracle

pytestmark = pytest.mark.gpu

RTOL = 2e-5   # relative to the max |reference| of the tensor


def _dev(a, dtype=torch.float32):
    return torch.as_tensor(np.ascontiguousarray(a)).to(dtype).cuda()


def _close(got, want, rtol=RTOL, name="", floor=1e-30):
    got = got.detach().cpu().numpy().astype(np.float64) if torch.is_tensor(got) else np.asarray(got, np.float64)
    want = np.asarray(want, np.float64)
    assert got.shape == want.shape, (name, got.shape, want.shape)
    assert np.isfinite(got).all(), name + ": non-finite output"
    scale = max(np.abs(want).max(), floor)
    err = np.abs(got - want).max() / scale
    assert err <= rtol, "%s: max err %.3e of scale %.3e (rel %.3e > %.1e)" % (name, np.abs(got - want).max(), scale, err, rtol)


@pytest.fixture(scope="module")
def ops():
    from sketchformer_amd import ops
    return ops


# ------------------------------------------------------------------ GEMM
@pytest.mark.parametrize("M,N,K", [(256, 128, 128), (200, 384, 128), (331, 1004, 128), (128, 345, 128), (77, 52, 36),
                                   (1000, 128, 512), (25472, 128, 128)])
def test_gemm_forward_bias_act(ops, M, N, K):
    rng = np.random.RandomState(M + N + K)
    x, w, b = rng.randn(M, K), rng.randn(K, N) / np.sqrt(K), rng.randn(N)
    for act, fn in ((0, lambda v: v), (1, lambda v: np.maximum(v, 0)), (2, np.tanh)):
        y = ops.gemm(_dev(x), _dev(w), bias=_dev(b), act=act)
        _close(y, fn(x @ w + b), name="fwd act=%d" % act)


@pytest.mark.parametrize("M,N,K", [(256, 128, 512), (199, 128, 1004), (300, 128, 384), (128, 128, 345), (64, 36, 52)])
def test_gemm_dgrad_relu_mask_accumulate(ops, M, N, K):
    rng = np.random.RandomState(M * 3 + N + K)
    dy, w, h, c0 = rng.randn(M, K), rng.randn(N, K) / np.sqrt(K), rng.randn(M, N), rng.randn(M, N)
    got = ops.gemm(_dev(dy), _dev(w), a_kcontig=True, b_kcontig=True, relu_src=_dev(h))
    _close(got, (dy @ w.T) * (h > 0), name="dgrad+relu mask")
    out = _dev(c0)
    ops.gemm(_dev(dy), _dev(w), a_kcontig=True, b_kcontig=True, out=out, accumulate=True)
    _close(out, c0 + dy @ w.T, name="dgrad accumulate")


@pytest.mark.parametrize("M,N,K", [(25472, 128, 1004), (3000, 128, 1004), (2048, 256, 516), (1500, 128, 2044)])
def test_gemm_dgrad_masked_last_slice(ops, M, N, K):
    """Input gradient with a contraction that is no multiple of 128 (the logits layer: K = V = 1004): 512-deep slices on the
    split-arithmetic weight-stationary kernel, the last one with its surplus weight columns counted as zeros and the A rows read
    past their end (the next row's values times those zeros; nothing behind the last row).  Plain, accumulating, and over a
    live-row-block list; a NaN-free neighbour is all the over-read needs."""
    rng = np.random.RandomState(M + N + K)
    dy, w, c0 = rng.randn(M, K), rng.randn(N, K) / np.sqrt(K), rng.randn(M, N)
    got = ops.gemm(_dev(dy), _dev(w), a_kcontig=True, b_kcontig=True)
    _close(got, dy @ w.T, name="dgrad K=%d" % K)
    out = _dev(c0)
    ops.gemm(_dev(dy), _dev(w), a_kcontig=True, b_kcontig=True, out=out, accumulate=True)
    _close(out, c0 + dy @ w.T, name="dgrad K=%d accumulate" % K)
    rps = 199
    if M % rps == 0:
        live = torch.from_numpy(rng.randint(8, rps, size=M // rps).astype(np.int32)).cuda()
        blocks = ops.row_blocks(live, rps, 16)
        dyr = dy.copy().reshape(M // rps, rps, K)
        for b, n in enumerate(live.cpu().numpy()):
            dyr[b, n:] = 0
        dyr = dyr.reshape(M, K)
        got = ops.gemm(_dev(dyr), _dev(w), a_kcontig=True, b_kcontig=True, row_blocks=blocks, row_block_rows=16)
        _close(got, dyr @ w.T, name="dgrad K=%d over live row blocks" % K)


@pytest.mark.parametrize("M,N,K", [(25600, 512, 128), (5000, 1024, 256), (3001, 132, 128), (4096, 512, 512)])
def test_gemm_relu_sign_bits(ops, M, N, K):
    """ffn backward without re-reading the hidden tensor: the relu forward launch leaves one sign bit per output element in the
    layout of its own tiles (skf_gemm_f32_bits), the input-gradient launch of the same (M, N, K) multiplies by them - bit for
    bit what the relu_src form computes, also over a live-row-block list and when accumulating."""
    from sketchformer_amd import _lib
    rng = np.random.RandomState(M + N + K)
    x, w1, b1 = rng.randn(M, K), rng.randn(K, N) / np.sqrt(K), rng.randn(N)
    dy, c0 = rng.randn(M, K), rng.randn(M, N)
    bits = ops.relu_bits(M, N, K, "cuda")
    assert bits is not None and bits.numel() * 8 == _lib.load().skf_gemm_relu_bits_bytes(M, N, K, _lib.default_precision())
    h = ops.gemm(_dev(x), _dev(w1), bias=_dev(b1), act=1, relu_bits_out=bits)
    _close(h, np.maximum(x @ w1 + b1, 0), name="relu forward with sign bits")
    assert torch.equal(h, ops.gemm(_dev(x), _dev(w1), bias=_dev(b1), act=1))
    w2t = _dev(rng.randn(N, K) / np.sqrt(K))              # dgrad form: B is [N][K]
    want = ops.gemm(_dev(dy), w2t, a_kcontig=True, b_kcontig=True, relu_src=h)
    got = ops.gemm(_dev(dy), w2t, a_kcontig=True, b_kcontig=True, relu_bits_in=bits)
    assert torch.equal(got, want)
    acc_w, acc_g = _dev(c0), _dev(c0)
    ops.gemm(_dev(dy), w2t, a_kcontig=True, b_kcontig=True, relu_src=h, out=acc_w, accumulate=True)
    ops.gemm(_dev(dy), w2t, a_kcontig=True, b_kcontig=True, relu_bits_in=bits, out=acc_g, accumulate=True)
    assert torch.equal(acc_g, acc_w)
    # live 16-row blocks: dead rows of dy are zero, their output rows come out as zeros
    live = torch.full((M // 200 + 1,), 120, dtype=torch.int32, device="cuda")
    rps = 200
    Mr = (M // rps) * rps
    if Mr >= 1024 and _lib.load().skf_gemm_relu_bits_bytes(Mr, N, K, _lib.default_precision()):
        live = live[:Mr // rps].contiguous()
        blocks = ops.row_blocks(live, rps, 16)
        dyr = dy[:Mr].copy()
        dyr.reshape(Mr // rps, rps, K)[:, 120:] = 0
        bits_r = ops.relu_bits(Mr, N, K, "cuda")
        hr = ops.gemm(_dev(x[:Mr]), _dev(w1), bias=_dev(b1), act=1, relu_bits_out=bits_r)
        want = ops.gemm(_dev(dyr), w2t, a_kcontig=True, b_kcontig=True, relu_src=hr)
        got = ops.gemm(_dev(dyr), w2t, a_kcontig=True, b_kcontig=True, relu_bits_in=bits_r, row_blocks=blocks, row_block_rows=16)
        assert torch.equal(got, want)
    # shapes without the path say so: size 0, and bits on such a launch are refused
    assert _lib.load().skf_gemm_relu_bits_bytes(128, 64, 52, _lib.default_precision()) == 0
    assert _lib.load().skf_gemm_relu_bits_bytes(M, N, K, 0) == 0


@pytest.mark.parametrize("rows,inf,outf", [(25600, 128, 128), (25472, 128, 1004), (5000, 512, 128), (3000, 128, 384),
                                           (128, 128, 345), (999, 36, 52)])
def test_gemm_wgrad_splitk_bias_grad(ops, rows, inf, outf):
    from sketchformer_amd import _lib
    rng = np.random.RandomState(rows + inf + outf)
    x, dy = rng.randn(rows, inf), rng.randn(rows, outf)
    splits = _lib.load().skf_gemm_default_splits(inf, outf, rows)
    bg = torch.empty(outf, dtype=torch.float32, device="cuda")
    dw = ops.gemm(_dev(x), _dev(dy), a_kcontig=False, b_kcontig=False, splits=splits, bias_grad=bg)
    _close(dw, x.T @ dy, rtol=5e-5, name="wgrad splits=%d" % splits)
    _close(bg, dy.sum(0), rtol=5e-5, name="bias grad")


@pytest.mark.parametrize("M,N,K", [(2000, 128, 128), (1100, 384, 128), (1500, 128, 512), (1030, 256, 256),
                                   (1025, 1004, 128), (3001, 128, 384), (25600, 512, 128)])
def test_gemm_weight_stationary_path(ops, M, N, K):
    """K in {128,256,384,512}, M >= 1024: the persistent weight-in-registers kernel (both B layouts, every epilogue)."""
    rng = np.random.RandomState(M + 7 * N + K)
    x, w, b = rng.randn(M, K), rng.randn(K, N) / np.sqrt(K), rng.randn(N)
    h, c0 = rng.randn(M, N), rng.randn(M, N)
    _close(ops.gemm(_dev(x), _dev(w), bias=_dev(b), act=1), np.maximum(x @ w + b, 0), name="ws fwd relu")
    _close(ops.gemm(_dev(x), _dev(w), bias=_dev(b), act=2), np.tanh(x @ w + b), name="ws fwd tanh")
    wt = np.ascontiguousarray(w.T)                      # dgrad layout: B stored [N][K]
    _close(ops.gemm(_dev(x), _dev(wt), b_kcontig=True, relu_src=_dev(h)), (x @ w) * (h > 0), name="ws dgrad relu mask")
    out = _dev(c0)
    ops.gemm(_dev(x), _dev(wt), b_kcontig=True, out=out, accumulate=True)
    _close(out, c0 + x @ w, name="ws dgrad accumulate")


@pytest.mark.parametrize("M,N,K,bkc", [(2000, 128, 128, False), (1100, 384, 128, False), (1500, 128, 512, True),
                                       (1030, 256, 256, False), (1025, 1004, 128, False), (3001, 128, 384, True)])
def test_gemm_arithmetic_modes(ops, M, N, K, bkc):
    """The `precision` argument: 0 = fp32 MFMA, 6 = fp32 operands split exactly into three bf16 pieces (six products on
    the bf16 matrix cores), 3 = two pieces.  Against float64: bf16x6 is at least as accurate as the fp32-MFMA kernel,
    bf16x3 stays inside 1e-4 of sum|a||b| (north_star tolerance: 1e-3 relative)."""
    from sketchformer_amd import _lib
    rng = np.random.RandomState(M + N + K)
    x = rng.randn(M, K) * np.exp(rng.randn(M, 1))          # rows of very different magnitude
    w = rng.randn(N, K) if bkc else rng.randn(K, N)
    b = rng.randn(N)
    xf, wf, bf = x.astype(np.float32), w.astype(np.float32), b.astype(np.float32)
    wm = wf.T.astype(np.float64) if bkc else wf.astype(np.float64)
    want = xf.astype(np.float64) @ wm + bf
    scale = np.abs(xf).astype(np.float64) @ np.abs(wm)
    err = {}
    for mode in (0, 6, 3):
        y = ops.gemm(_dev(xf), _dev(wf), b_kcontig=bkc, bias=_dev(bf), precision=mode).cpu().numpy().astype(np.float64)
        err[mode] = np.abs(y - want) / scale
    with pytest.raises(_lib.SkfError):                       # not a SKF_PREC_* value
        ops.gemm(_dev(xf), _dev(wf), b_kcontig=bkc, bias=_dev(bf), precision=5)
    assert err[0].max() < 2e-6, err[0].max()
    assert err[6].max() < 2e-6 and err[6].mean() <= 1.25 * err[0].mean(), (err[6].max(), err[6].mean(), err[0].mean())
    assert err[3].max() < 1e-4, err[3].max()


@pytest.mark.parametrize("M,N,K,bkc", [(1500, 256, 1024, False), (1200, 256, 768, True), (1100, 128, 2048, False), (1030, 256, 640, True)])
def test_gemm_long_k_chain(ops, M, N, K, bkc):
    """K = a multiple of 128 in (512, 2048]: chained <= 512-deep weight-stationary launches (bias in the first, every
    later one accumulating), with and without an initial accumulate."""
    rng = np.random.RandomState(M + N + K)
    x, w, b, c0 = rng.randn(M, K), rng.randn(K, N) / np.sqrt(K), rng.randn(N), rng.randn(M, N)
    wd = _dev(np.ascontiguousarray(w.T)) if bkc else _dev(w)
    _close(ops.gemm(_dev(x), wd, b_kcontig=bkc, bias=_dev(b)), x @ w + b, name="chain")
    out = _dev(c0)
    ops.gemm(_dev(x), wd, b_kcontig=bkc, out=out, accumulate=True)
    _close(out, c0 + x @ w, name="chain accumulate")
    _close(ops.gemm(_dev(x), wd, b_kcontig=bkc, bias=_dev(b), act=1), np.maximum(x @ w + b, 0), name="relu: generic kernel")


@pytest.mark.parametrize("rows,inf,outf", [(25600, 128, 128), (6000, 128, 512), (5000, 512, 128)])
def test_wgrad_arithmetic_modes(ops, rows, inf, outf):
    """Weight gradient dW = X^T dY in the fp32-MFMA and the bf16x6 arithmetic against float64: the split kernel is at
    least as accurate (exact piece products, small-to-large accumulation)."""
    from sketchformer_amd import _lib
    lib = _lib.load()
    rng = np.random.RandomState(rows + inf + outf)
    x = (rng.randn(rows, inf) * np.exp(rng.randn(rows, 1))).astype(np.float32)
    dy = (rng.randn(rows, outf) * np.exp(rng.randn(1, outf))).astype(np.float32)
    want = x.astype(np.float64).T @ dy.astype(np.float64)
    scale = np.abs(x).astype(np.float64).T @ np.abs(dy).astype(np.float64)
    splits = lib.skf_gemm_default_splits(inf, outf, rows)
    err = {}
    for mode in (0, 6):
        bg = torch.empty(outf, dtype=torch.float32, device="cuda")
        dw = ops.gemm(_dev(x), _dev(dy), a_kcontig=False, b_kcontig=False, splits=splits, bias_grad=bg, precision=mode)
        err[mode] = np.abs(dw.cpu().numpy().astype(np.float64) - want) / scale
        np.testing.assert_allclose(bg.cpu().numpy(), dy.astype(np.float64).sum(0), rtol=1e-4, atol=1e-3 * np.abs(dy).sum(0).max() / rows ** 0.5)
    assert err[0].max() < 2e-6 and err[6].max() < 2e-6, (err[0].max(), err[6].max())
    assert err[6].mean() <= 1.25 * err[0].mean(), (err[6].mean(), err[0].mean())


@pytest.mark.parametrize("kind", ["fwd", "dgrad", "wgrad"])
def test_gemm_modes_extreme_operands(ops, kind):
    """Robustness of the split arithmetic (mode 6) beside the fp32 MFMA (mode 0) on operands the randn tests never see:
    rows / columns scaled over 1e+-30 (products up to 1e+-34, pieces far apart in exponent), fp32 subnormals, +-inf, NaN.
    Defined behaviour (include/skf.h): finite results within one fp32 rounding of sum|a||b| plus the flush of pieces below
    the bf16 normal range; an output is non-finite in mode 6 exactly where it is in mode 0 (and in float64 arithmetic on
    the fp32 operands); the non-finite value itself may be NaN where mode 0 gives +-inf."""
    rng = np.random.RandomState({"fwd": 1, "dgrad": 2, "wgrad": 3}[kind])
    M, K, N = (1100, 128, 384) if kind != "wgrad" else (128, 4096, 384)        # wgrad: K = rows of the batch
    a = rng.randn(M, K).astype(np.float32)
    b = rng.randn(K, N).astype(np.float32)

    def run(a, b, mode):
        if kind == "fwd":
            return ops.gemm(_dev(a), _dev(b), precision=mode).cpu().numpy()
        if kind == "dgrad":
            return ops.gemm(_dev(a), _dev(np.ascontiguousarray(b.T)), b_kcontig=True, precision=mode).cpu().numpy()
        splits = 4
        return ops.gemm(_dev(np.ascontiguousarray(a.T)), _dev(b), a_kcontig=False, b_kcontig=False, splits=splits,
                        precision=mode).cpu().numpy()

    # (1) huge dynamic range: rows of A scaled by 10^U(-30, 4), columns of B by 10^U(-4, 30) / 10^U(-30,-4) halves
    ra = (10.0 ** rng.uniform(-30, 4, size=(M, 1))).astype(np.float32)
    cb = (10.0 ** np.where(rng.rand(1, N) < 0.5, rng.uniform(-30, -4, size=(1, N)), rng.uniform(-4, 30, size=(1, N)))).astype(np.float32)
    a1, b1 = a * ra, b * cb
    want = a1.astype(np.float64) @ b1.astype(np.float64)
    scale = np.abs(a1).astype(np.float64) @ np.abs(b1).astype(np.float64)
    tiny = 2.0 ** -120 * K                                  # flushed sub-bf16-normal pieces / fp32 underflow of the products
    for mode in (0, 6):
        y = run(a1, b1, mode).astype(np.float64)
        fin = np.isfinite(want.astype(np.float32))          # fp32 overflow of the exact result is legitimate
        assert np.isfinite(y[fin & (np.abs(want) < 1e37)]).all(), mode
        ok = fin & (np.abs(want) < 1e37)
        err = np.abs(y - want)[ok] / (scale[ok] + tiny)
        assert err.max() < 4e-6, (kind, mode, err.max())
    # (2) fp32 subnormal operands beside normal ones: result error bounded by the flush (absolute), never NaN
    a2 = a.copy(); a2[:, ::7] = (rng.randn(M, len(range(0, K, 7))) * 1e-41).astype(np.float32)
    b2 = b.copy(); b2[::5, :] = (rng.randn(len(range(0, K, 5)), N) * 1e-42).astype(np.float32)
    want = a2.astype(np.float64) @ b2.astype(np.float64)
    scale = np.abs(a2).astype(np.float64) @ np.abs(b2).astype(np.float64)
    for mode in (0, 6):
        y = run(a2, b2, mode).astype(np.float64)
        assert np.isfinite(y).all()
        assert (np.abs(y - want) / (scale + 1e-30)).max() < 4e-6, (kind, mode)
    # (3) non-finite operands: the same output positions are non-finite in both modes, finite ones agree
    a3, b3 = a.copy(), b.copy()
    a3[3, 5] = np.inf; a3[17, 100] = -np.inf; a3[40, 0] = np.nan
    b3[9, 11] = np.inf; b3[77, 200] = np.nan
    with np.errstate(invalid="ignore", over="ignore"):
        want = a3.astype(np.float64) @ b3.astype(np.float64)
    y0, y6 = run(a3, b3, 0), run(a3, b3, 6)
    nf = ~np.isfinite(want)
    assert nf.any() and (~nf).any()
    assert np.array_equal(~np.isfinite(y0), nf), "mode 0: non-finite outputs differ from IEEE arithmetic"
    assert np.array_equal(~np.isfinite(y6), nf), "mode 6: non-finite outputs differ from mode 0"
    assert np.array_equal(np.isnan(y0), np.isnan(want.astype(np.float32)))     # the fp32 MFMA keeps inf vs NaN like IEEE
    scale = np.abs(np.nan_to_num(a3, posinf=0, neginf=0)).astype(np.float64) @ np.abs(np.nan_to_num(b3, posinf=0, neginf=0)).astype(np.float64)
    assert (np.abs(y6.astype(np.float64) - want)[~nf] / scale[~nf]).max() < 2e-6


def _padded_case(rng, B, Ld, width):
    """targets with PAD tails (one empty, one full sample) -> live_len, and a (B*Ld, width) gradient that is zero on dead rows"""
    lens = rng.randint(1, Ld + 1, size=B); lens[0] = 0; lens[1] = Ld
    tar = np.zeros((B, Ld + 1), np.int64)
    for b in range(B):
        tar[b, :lens[b] + 1] = rng.randint(1, 1000, size=lens[b] + 1)
        if lens[b] > 3:
            tar[b, 2] = 0                                  # a PAD inside the live range does not end it
    tar[0, 0] = 5                                           # SOS only: no trained position
    g = rng.randn(B * Ld, width)
    for b in range(B):
        g[b * Ld + lens[b]:(b + 1) * Ld] = 0.0
    return tar, lens, g


def test_row_blocks_lists(ops):
    rng = np.random.RandomState(3)
    B, Ld = 37, 199
    tar, lens, _ = _padded_case(rng, B, Ld, 4)
    ll = ops.target_live_len(_dev(tar, torch.int64), Ld)
    assert np.array_equal(ll.cpu().numpy(), lens)
    for g in (1, 16, 32, 64, 256):
        got = ops.row_blocks(ll, Ld, g).cpu().numpy()
        nb = -(-B * Ld // g)
        live_row = np.concatenate([np.arange(Ld) < n for n in lens])
        live = np.array([live_row[k * g:(k + 1) * g].any() for k in range(nb)])
        assert got[0] == live.sum() and got[1] == nb
        assert np.array_equal(got[2:2 + got[0]], np.nonzero(live)[0]) and np.array_equal(got[2 + got[0]:2 + nb], np.nonzero(~live)[0])
        assert np.array_equal(got[2 + nb:], live.astype(np.int32))


@pytest.mark.parametrize("N,K,relu,acc", [(128, 128, False, False), (128, 384, False, True), (512, 128, True, False), (128, 512, False, False),
                                          (128, 256, False, False), (1004, 128, False, False), (128, 1004, False, False)])   # (the last: generic tiled kernel)
def test_gemm_dgrad_over_live_row_blocks_is_exact(ops, N, K, relu, acc):
    """dX = dY W^T visiting only the 16-row tiles that hold live rows: bit-identical to the dense call (dead rows of dY are
    zero, so the dense kernel computes exact zeros there), for every weight-stationary shape family of the step."""
    rng = np.random.RandomState(N + K)
    B, Ld = 40, 199
    tar, lens, dy = _padded_case(rng, B, Ld, K)
    w = rng.randn(N, K) / np.sqrt(K)
    h = rng.randn(B * Ld, N) if relu else None
    c0 = rng.randn(B * Ld, N)
    for b in range(B):
        c0[b * Ld + lens[b]:(b + 1) * Ld] = 0.0
    blocks = ops.row_blocks(ops.target_live_len(_dev(tar, torch.int64), Ld), Ld, 16)
    kw = dict(a_kcontig=True, b_kcontig=True, relu_src=_dev(h) if relu else None, accumulate=acc)
    dense = ops.gemm(_dev(dy), _dev(w), out=_dev(c0), **kw)
    rows = ops.gemm(_dev(dy), _dev(w), out=_dev(c0) if acc else torch.full((B * Ld, N), 7.0, device="cuda"), row_blocks=blocks,
                    row_block_rows=16, **kw)
    assert torch.equal(dense, rows)
    want = dy @ w.T
    if relu:
        want = want * (h > 0)
    _close(rows, want + (c0 if acc else 0.0), name="dgrad over live blocks")


@pytest.mark.parametrize("inf,outf", [(128, 384), (128, 128), (512, 128), (128, 1004)])
def test_gemm_wgrad_over_live_row_blocks(ops, inf, outf):
    """dW = X^T dY contracting over the live 32-row blocks only; X is dense (forward activations exist at padded positions)."""
    from sketchformer_amd import _lib
    rng = np.random.RandomState(inf + outf)
    B, Ld = 64, 199
    tar, lens, dy = _padded_case(rng, B, Ld, outf)
    x = rng.randn(B * Ld, inf)
    blocks = ops.row_blocks(ops.target_live_len(_dev(tar, torch.int64), Ld), Ld, 32)
    splits = _lib.load().skf_gemm_default_splits(inf, outf, B * Ld)
    bg = torch.zeros(outf, device="cuda"); bg2 = torch.zeros(outf, device="cuda")
    dense = ops.gemm(_dev(x), _dev(dy), a_kcontig=False, b_kcontig=False, splits=splits, bias_grad=bg)
    rows = ops.gemm(_dev(x), _dev(dy), a_kcontig=False, b_kcontig=False, splits=splits, bias_grad=bg2, row_blocks=blocks, row_block_rows=32)
    _close(rows, x.T @ dy, rtol=5e-5, name="wgrad over live blocks")
    _close(bg2, dy.sum(0), rtol=5e-5, name="bias grad over live blocks")
    _close(rows, dense.cpu().numpy(), rtol=2e-6, name="vs dense")


def test_gemm_strided_views(ops):
    """fused QKV layout: W stored [d][3d]; outputs written into a (rows, 3d) buffer at a column offset."""
    rng = np.random.RandomState(5)
    rows, d = 300, 128
    x, w = rng.randn(rows, d), rng.randn(d, 3 * d) / np.sqrt(d)
    W = _dev(w)
    out = torch.zeros(rows, 3 * d, dtype=torch.float32, device="cuda")
    ops.gemm(_dev(x), W[:, d:2 * d], out=out[:, d:2 * d])
    want = np.zeros((rows, 3 * d)); want[:, d:2 * d] = x @ w[:, d:2 * d]
    _close(out, want, name="strided")


# ------------------------------------------------------------------ attention
def _attn_case(B, H, Lq, Lk, dh, causal, with_mask, seed, all_pad_row=False):
    rng = np.random.RandomState(seed)
    d = H * dh
    q, k, v = rng.randn(B, Lq, d), rng.randn(B, Lk, d), rng.randn(B, Lk, d)
    do = rng.randn(B, Lq, d)
    km = None
    if with_mask:
        lens = rng.randint(1, Lk + 1, size=B)
        lens[0] = Lk
        km = (np.arange(Lk)[None, :] >= lens[:, None])
        if all_pad_row:
            km[-1, :] = True
    mask = np.zeros((B, 1, Lq, Lk), np.float32)
    if km is not None:
        mask = np.maximum(mask, km[:, None, None, :].astype(np.float32))
    if causal:
        mask = np.maximum(mask, oracle.create_look_ahead_mask(Lq)[None, None])
    return q, k, v, do, km, mask


def _split(x, H):
    B, L, d = x.shape
    return x.reshape(B, L, H, d // H).transpose(0, 2, 1, 3)


def _merge(x):
    B, H, L, dh = x.shape
    return x.transpose(0, 2, 1, 3).reshape(B, L, H * dh)


@pytest.mark.parametrize("B,H,Lq,Lk,dh,causal,with_mask", [
    (3, 8, 200, 200, 16, False, True),     # encoder self-attention, padding mask
    (3, 8, 199, 199, 16, True, True),      # decoder self-attention, combined mask
    (3, 8, 199, 200, 16, False, False),    # blind cross-attention
    (2, 4, 37, 53, 16, False, True),       # ragged tiles
    (2, 8, 200, 200, 32, False, True),     # cfg 3 head dim
    (2, 4, 199, 199, 32, True, True),      # cfg 3 decoder self-attention
    (2, 4, 199, 200, 32, False, False),    # cfg 3 cross-attention
    (2, 4, 64, 64, 64, True, False),       # head dim 64
    (2, 2, 16, 16, 16, True, True),
    (1, 1, 1, 1, 16, False, False),        # degenerate
])
def test_attention_fwd_bwd(ops, B, H, Lq, Lk, dh, causal, with_mask):
    q, k, v, do, km, mask = _attn_case(B, H, Lq, Lk, dh, causal, with_mask, seed=B * 1000 + Lq + dh)
    want_o, _, cache = oracle.sdpa_fwd(_split(q, H), _split(k, H), _split(v, H), mask.astype(np.float64))
    kmd = _dev(km, torch.uint8) if km is not None else None
    o, stats = ops.attention_fwd(_dev(q), _dev(k), _dev(v), H, key_mask=kmd, causal=causal)
    _close(o, _merge(want_o), name="attn fwd")
    dq, dk, dv = oracle.sdpa_bwd(_split(do, H), cache)
    gq, gk, gv = ops.attention_bwd(_dev(q), _dev(k), _dev(v), o, _dev(do), stats, H, key_mask=kmd, causal=causal)
    # (floor: with a single key dQ = dK = 0 exactly; dP - delta is then the rounding noise of two fp32 evaluations, ~1e-6)
    _close(gq, _merge(dq), rtol=5e-5, name="attn dQ", floor=0.1)
    _close(gk, _merge(dk), rtol=5e-5, name="attn dK", floor=0.1)
    _close(gv, _merge(dv), rtol=5e-5, name="attn dV")


def test_attention_fully_padded_sequence_matches_fp32_reference_semantics(ops):
    """A row whose keys are ALL masked: x + (-1e9) rounds to -1e9 in fp32, so the
    reference yields a uniform distribution over every key (incl. look-ahead ones)."""
    B, H, L, dh = 2, 2, 40, 16
    q, k, v, do, km, mask = _attn_case(B, H, L, L, dh, True, True, seed=11, all_pad_row=True)
    f32 = np.float32
    want_o, _, _ = oracle.sdpa_fwd(_split(q, H).astype(f32), _split(k, H).astype(f32), _split(v, H).astype(f32), mask)
    o, _ = ops.attention_fwd(_dev(q), _dev(k), _dev(v), H, key_mask=_dev(km, torch.uint8), causal=True)
    _close(o, _merge(want_o), rtol=1e-4, name="all-pad row")


@pytest.mark.parametrize("causal", [False, True])
def test_attention_padding_tile_skipping_is_exact(ops, causal):
    """Short sequences (several trailing all-padding key tiles), holes inside the valid range, one fully padded sample."""
    B, H, L, dh = 6, 4, 200, 16
    rng = np.random.RandomState(17)
    d = H * dh
    q, k, v, do = (rng.randn(B, L, d) for _ in range(4))
    lens = np.array([200, 80, 17, 16, 1, 50])
    km = np.arange(L)[None, :] >= lens[:, None]
    km[1, 30:40] = True                      # padded keys inside the valid range
    km[5, :] = True                          # a fully padded sample: uniform attention over every key
    mask = km[:, None, None, :].astype(np.float32)
    if causal:
        mask = np.maximum(mask, oracle.create_look_ahead_mask(L)[None, None])
    f32 = np.float32
    want, _, cache = oracle.sdpa_fwd(_split(q, H).astype(f32), _split(k, H).astype(f32), _split(v, H).astype(f32), mask)
    kmd = _dev(km, torch.uint8)
    o, stats = ops.attention_fwd(_dev(q), _dev(k), _dev(v), H, key_mask=kmd, causal=causal)
    _close(o, _merge(want), rtol=1e-4, name="fwd")
    dq, dk, dv = oracle.sdpa_bwd(_split(do, H).astype(f32), cache)
    gq, gk, gv = ops.attention_bwd(_dev(q), _dev(k), _dev(v), o, _dev(do), stats, H, key_mask=kmd, causal=causal)
    _close(gq, _merge(dq), rtol=2e-4, name="dQ"); _close(gk, _merge(dk), rtol=2e-4, name="dK"); _close(gv, _merge(dv), rtol=2e-4, name="dV")
    assert float(gk[2, 32:].abs().max()) == 0.0 and float(gv[2, 32:].abs().max()) == 0.0     # skipped tiles: exact zeros


@pytest.mark.parametrize("seed", range(24))
def test_attention_backward_random_shapes_masks_and_live_lengths(ops, seed):
    """Round 5 (skf_attention_bwd3.hip serves head size 16 up to 208 positions): random batch / head counts, query and key lengths off the
    tile grid (1 ... 208), key masks with holes and fully padded samples, look-ahead masks, zero dO rows behind random live lengths with
    and without the length list - every case against the float64 oracle, and the two calls against each other bit for bit."""
    rng = np.random.RandomState(1000 + seed)
    B, H, dh = int(rng.randint(1, 5)), int(rng.choice([1, 2, 4, 8])), 16
    causal = bool(rng.rand() < 0.4)
    Lq = int(rng.choice([1, 7, 16, 17, 63, 100, 199, 200, 208, int(rng.randint(1, 209))]))
    Lk = Lq if causal else int(rng.choice([1, 15, 16, 33, 128, 200, 208, int(rng.randint(1, 209))]))
    d = H * dh
    q, k, v, do = rng.randn(B, Lq, d), rng.randn(B, Lk, d), rng.randn(B, Lk, d), rng.randn(B, Lq, d)
    km = None
    if rng.rand() < 0.7:
        lens = rng.randint(0 if rng.rand() < 0.3 else 1, Lk + 1, size=B)           # length 0: a fully padded sample (uniform weights)
        km = np.arange(Lk)[None, :] >= lens[:, None]
        if Lk > 8 and rng.rand() < 0.5:
            a = int(rng.randint(0, Lk - 4))
            km[0, a:a + int(rng.randint(1, 5))] = True                               # padded keys inside the valid range
    live = rng.randint(0, Lq + 1, size=B).astype(np.int32) if rng.rand() < 0.6 else None
    if live is not None:
        do = do * (np.arange(Lq)[None, :, None] < live[:, None, None])
    mask = np.zeros((B, 1, Lq, Lk), np.float32)
    if km is not None:
        mask = np.maximum(mask, km[:, None, None, :].astype(np.float32))
    if causal:
        mask = np.maximum(mask, oracle.create_look_ahead_mask(Lq)[None, None])
    f32 = np.float32      # (the additive -1e9 only rounds away in fp32: the fully-padded-sample semantics are fp32 semantics)
    want_o, _, cache = oracle.sdpa_fwd(_split(q, H).astype(f32), _split(k, H).astype(f32), _split(v, H).astype(f32), mask)
    dq, dk, dv = oracle.sdpa_bwd(_split(do, H).astype(f32), cache)
    kmd = _dev(km, torch.uint8) if km is not None else None
    o, stats = ops.attention_fwd(_dev(q), _dev(k), _dev(v), H, key_mask=kmd, causal=causal)
    _close(o, _merge(want_o), rtol=1e-4, name="fwd")
    a = ops.attention_bwd(_dev(q), _dev(k), _dev(v), o, _dev(do), stats, H, key_mask=kmd, causal=causal)
    for g, w, n in zip(a, (dq, dk, dv), ("dQ", "dK", "dV")):
        _close(g, _merge(w), rtol=3e-4, name=n, floor=0.1)
    if live is not None:
        b = ops.attention_bwd(_dev(q), _dev(k), _dev(v), o, _dev(do), stats, H, key_mask=kmd, causal=causal,
                              q_live_len=torch.as_tensor(live).cuda())
        for x, y in zip(a, b):
            assert torch.equal(x, y)


@pytest.mark.parametrize("seed", range(6))
def test_sample_order_and_dealt_attention_workgroups(ops, seed):
    """skf_sample_order = a stable descending sort of the samples by unmasked positions (one or two masks); the attention launches that
    deal their (sample, head) workgroups over the shader engines from such a list (B * H a multiple of 32; any permutation will do) return
    the bits of the plain numbering, forward and backward, dh = 16 (forward + skf_attention_bwd3) and a size the list is ignored for."""
    rng = np.random.RandomState(2000 + seed)
    B, H, dh = int(rng.choice([4, 8, 16, 33])), int(rng.choice([8, 4])), 16
    causal = bool(seed & 1)
    Lq = int(rng.choice([17, 100, 200]))
    Lk = Lq if causal else int(rng.choice([33, 128, 200]))
    d = H * dh
    la, lb = rng.randint(0, Lk + 1, size=B), rng.randint(0, Lq + 1, size=B)
    la[rng.randint(B)] = la[rng.randint(B)]                      # ties: the sort is stable
    ma = np.arange(Lk)[None, :] >= la[:, None]
    mb = np.arange(Lq)[None, :] >= lb[:, None]
    ma_d, mb_d = _dev(ma, torch.uint8), _dev(mb, torch.uint8)
    for a_, b_, key in ((ma_d, mb_d, la + lb), (ma_d, None, la), (None, mb_d, lb)):
        got = ops.sample_order(a_, b_).cpu().numpy()
        want = np.argsort(-key, kind="stable")
        assert np.array_equal(got, want), (got, want)
    wide = _dev(np.concatenate([ma, np.zeros((B, 3), bool)], axis=1), torch.uint8)          # rows with a pitch: the byte path
    assert np.array_equal(ops.sample_order(wide[:, :Lk], None).cpu().numpy(), np.argsort(-la, kind="stable"))
    if seed == 0:       # more samples than threads per pass of the rank phase, rows that are not a multiple of four bytes
        big = rng.randint(0, 2, size=(3000, 37)).astype(bool)
        assert np.array_equal(ops.sample_order(_dev(big, torch.uint8), None).cpu().numpy(), np.argsort(-(~big).sum(1), kind="stable"))
    order = ops.sample_order(ma_d, mb_d)
    perm = torch.as_tensor(rng.permutation(B).astype(np.int32)).cuda()
    q, k, v, do = (_dev(rng.randn(B, n, d)) for n in (Lq, Lk, Lk, Lq))
    live = torch.as_tensor(lb.astype(np.int32)).cuda()
    do = do * (torch.arange(Lq, device="cuda")[None, :, None] < live[:, None, None])
    o, st = ops.attention_fwd(q, k, v, H, key_mask=ma_d, causal=causal)
    g = ops.attention_bwd(q, k, v, o, do, st, H, key_mask=ma_d, causal=causal, q_live_len=live)
    for od in (order, perm):
        o2, st2 = ops.attention_fwd(q, k, v, H, key_mask=ma_d, causal=causal, sample_order=od)
        assert torch.equal(o2, o) and torch.equal(st2, st)
        g2 = ops.attention_bwd(q, k, v, o, do, st, H, key_mask=ma_d, causal=causal, q_live_len=live, sample_order=od)
        for x, y in zip(g, g2):
            assert torch.equal(x, y)


@pytest.fixture
def ops_two_pass(ops, monkeypatch):
    """`ops` with attention_bwd asking for the two-pass kernel (SKF_ATTN_TWO_PASS in the precision argument)."""
    import functools
    monkeypatch.setattr(ops, "attention_bwd", functools.partial(ops.attention_bwd, two_pass=True))
    return ops


def test_attention_two_pass_backward_head_size_16(ops_two_pass):
    """Head size 16 takes the one-pass backward by default since round 3; the two-pass kernel (`attn_bwd2<1, *>`, still the head-size-32
    default) stays reachable for that size through SKF_ATTN_TWO_PASS: the same oracle tests, dh = 16 cases."""
    for (B, H, Lq, Lk, dh, causal, with_mask) in [(2, 8, 200, 200, 16, False, True), (3, 8, 199, 199, 16, True, True),
                                                  (2, 8, 199, 200, 16, False, False), (1, 4, 33, 47, 16, False, True)]:
        test_attention_fwd_bwd(ops_two_pass, B, H, Lq, Lk, dh, causal, with_mask)
    for causal in (False, True):
        test_attention_padding_tile_skipping_is_exact(ops_two_pass, causal)
    for (H, causal, mode) in [(8, False, None), (8, True, None), (4, False, None)]:
        test_attention_bwd_live_query_counts_is_exact(ops_two_pass, H, causal, mode)


def test_attention_strided_qkv(ops):
    """q/k/v as column slices of one (B,L,3d) projection buffer, as the train step uses them."""
    B, H, L, dh = 2, 8, 50, 16
    d = H * dh
    rng = np.random.RandomState(3)
    qkv = rng.randn(B, L, 3 * d)
    t = _dev(qkv)
    o, _ = ops.attention_fwd(t[..., :d], t[..., d:2 * d], t[..., 2 * d:], H)
    want, _, _ = oracle.sdpa_fwd(_split(qkv[..., :d], H), _split(qkv[..., d:2 * d], H), _split(qkv[..., 2 * d:], H), None)
    _close(o, _merge(want), name="strided qkv")


# ------------------------------------------------------------------ embedding / layernorm / dropout
@pytest.mark.parametrize("H,causal,mode", [(8, False, None), (8, True, None), (8, False, 0), (4, False, None)])
def test_attention_bwd_live_query_counts_is_exact(ops, H, causal, mode):
    """dO is zero behind each sample's live length (decoder rows of a padded batch): with the counts the one-pass kernel leaves
    the dead query tiles out (not staged, not visited, dQ stored as zeros) - every output equals the call without them
    (head size 16 non-causal and the fp32-MFMA mode take the one-pass kernel, the causal case the two-pass one)."""
    B, Lq, Lk, d = 7, 199, 200, 128
    rng = np.random.RandomState(3 + H)
    q, k, v = (_dev(rng.randn(B, L, d)) for L in (Lq, Lk if not causal else Lq, Lk if not causal else Lq))
    lens = np.array([0, 199, 1, 15, 16, 17, 120], np.int32)
    do = _dev(rng.randn(B, Lq, d) * (np.arange(Lq)[None, :, None] < lens[:, None, None]))
    o, st = ops.attention_fwd(q, k, v, H, causal=causal, precision=mode)
    ll = torch.as_tensor(lens).cuda()
    a = ops.attention_bwd(q, k, v, o, do, st, H, causal=causal, precision=mode)
    b = ops.attention_bwd(q, k, v, o, do, st, H, causal=causal, precision=mode, q_live_len=ll)
    for x, y in zip(a, b):
        assert torch.equal(x, y)
    assert float(b[0][0].abs().max()) == 0.0


def test_embed_fwd_bwd_with_dropout(ops):
    from sketchformer_amd import engine
    B, L, V, d, rate, site = 5, 33, 52, 128, 0.1, 3
    rng = np.random.RandomState(0)
    tok = rng.randint(0, V, size=(B, L + 1)); tok[:, 20:] = 0
    table = rng.uniform(-0.05, 0.05, (V, d))
    pos = engine.positional_encoding(64, d)
    st = ops.new_step_state("cuda", iterations=7)
    ops.step_prologue(st, seed=123)
    key = ops.read_step_state(st)["drop_key"]
    keep = ops.dropout_keep_mask(key, site, rate, B * L * d).reshape(B, L, d)
    assert 0.85 < keep.mean() < 0.95
    x = table[tok[:, :L]] * np.sqrt(np.float64(d)) + pos[None, :L]
    out = ops.embed_fwd(_dev(tok, torch.int64), _dev(table), _dev(pos), L=L, rate=rate, site=site, state=st)
    _close(out, x * keep / (1 - rate), name="embed fwd")
    dx = rng.randn(B, L, d)
    want = np.zeros((V, d))
    np.add.at(want, tok[:, :L].reshape(-1), (dx * keep / (1 - rate) * np.sqrt(np.float64(d))).reshape(-1, d))
    got = ops.embed_bwd(_dev(tok, torch.int64), _dev(dx), V, L=L, rate=rate, site=site, state=st)
    _close(got, want, rtol=5e-5, name="embed bwd")


@pytest.mark.parametrize("B,L,V,d,rate", [(5, 33, 52, 128, 0.1), (128, 200, 1004, 128, 0.1), (16, 199, 10004, 64, 0.0), (3, 7, 5, 256, 0.0)])
def test_embed_bwd_sorted(ops, B, L, V, d, rate):
    """Sorted two-launch embedding gradient: equals the scatter-add definition, writes every table row (the table starts as
    NaN), handles ids split over several 64-position chunks (PAD), out-of-range ids, unused ids, d != 128."""
    site = 2
    rng = np.random.RandomState(B + L + V)
    tok = rng.randint(0, V, size=(B, L + 1))
    tok[:, (2 * L) // 3:] = 0                         # PAD tail: one id with hundreds of positions
    tok[0, 0] = V + 7                                  # out of range: ignored
    st = ops.new_step_state("cuda", iterations=3)
    ops.step_prologue(st, seed=5)
    key = ops.read_step_state(st)["drop_key"]
    keep = ops.dropout_keep_mask(key, site, rate, B * L * d).reshape(B, L, d) if rate > 0 else np.ones((B, L, d))
    dx = rng.randn(B, L, d)
    want = np.zeros((V, d))
    tk = tok[:, :L].reshape(-1)
    ok = tk < V
    np.add.at(want, tk[ok], (dx * keep / (1 - rate) * np.sqrt(np.float64(d))).reshape(-1, d)[ok])
    got = ops.embed_bwd_sorted(_dev(tok, torch.int64), _dev(dx), V, L=L, rate=rate, site=site, state=st)
    assert torch.isfinite(got).all()
    _close(got, want, rtol=5e-5, name="embed bwd sorted")
    ref = ops.embed_bwd(_dev(np.where(tok < V, tok, 0) * (tok < V), torch.int64), _dev(dx * (tok[:, :L] < V)[..., None]), V, L=L,
                        rate=rate, site=site, state=st)
    _close(got, ref.cpu().numpy(), rtol=5e-5, name="embed bwd sorted vs atomic kernel")


@pytest.mark.parametrize("V", [1004, 10004])
def test_embed_bwd_sorted_is_run_to_run_deterministic(ops, V):
    """cfg-2 size with a PAD id of ~17k positions (68 chunks) and a few frequent ids: the stable counting sort and the in-order
    combination of a split id's chunks make the gradient bit-identical from run to run (round 1: LDS-atomic scatter + float
    atomics, the PAD row changed in the last bits).  V = 10004 (grid tokens, utils/tokenizer.py:104-198): the one-table sort
    with a single scattering wave (round 2 fell back to an unordered scatter above 2032 ids)."""
    B, L, d = 128, 200, 128
    rng = np.random.RandomState(11)
    tok = rng.randint(1, V, size=(B, L))
    tok[rng.rand(B, L) < 0.15] = 7                    # a frequent id: several chunks
    lens = rng.randint(20, L, size=B)
    tok[np.arange(L)[None, :] >= lens[:, None]] = 0    # PAD tails
    dx = _dev(rng.randn(B, L, d) * np.exp(rng.randn(B, L, 1)))
    tokd = _dev(tok, torch.int64)
    st = ops.new_step_state("cuda", iterations=3)
    ops.step_prologue(st, seed=5)
    first = ops.embed_bwd_sorted(tokd, dx, V, L=L, rate=0.1, site=2, state=st)
    # one sort serves any number of gradient passes (the split ids' tickets are reset by the chunk that used them up)
    import ctypes as C
    from sketchformer_amd import _lib
    nbytes = _lib.load().skf_embed_sort_workspace_bytes(B, L, V)
    ws = torch.empty((nbytes + 3) // 4, dtype=torch.int32, device="cuda")
    stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    _lib.call("skf_embed_sort", C.c_void_p(tokd.data_ptr()), L, B, L, V, None, d, C.c_void_p(ws.data_ptr()), nbytes, stream)
    for _ in range(2):
        t = torch.full((V, d), float("nan"), device="cuda")
        _lib.call("skf_embed_bwd_sorted", C.c_void_p(ws.data_ptr()), B, L, C.c_void_p(dx.data_ptr()), V, d, C.c_void_p(t.data_ptr()), 0.1, 2,
                  C.c_void_p(st.data_ptr()), stream)
        assert torch.equal(t, first)
    for _ in range(8):
        busy = torch.randn(4096, 4096, device="cuda") @ torch.randn(4096, 4096, device="cuda")      # perturb the scheduling
        again = ops.embed_bwd_sorted(tokd, dx, V, L=L, rate=0.1, site=2, state=st)
        assert torch.equal(first, again)
        del busy


def test_embed_bwd_sorted_more_than_65535_positions_of_one_id(ops):
    """B * L > 65536 rows with a PAD id of > 65536 positions: the ordered scatter's per-(wave, id) cursor is an offset inside the
    id's whole segment (round 2 kept it in 16 bits: it overflowed into the group count and positions landed in wrong slots)."""
    B, L, V, d = 320, 256, 1004, 64
    rng = np.random.RandomState(3)
    tok = rng.randint(1, V, size=(B, L))
    lens = rng.randint(8, 48, size=B)
    tok[np.arange(L)[None, :] >= lens[:, None]] = 0
    assert (tok == 0).sum() > 65536
    dx = rng.randn(B, L, d)
    want = np.zeros((V, d))
    np.add.at(want, tok.reshape(-1), (dx * np.sqrt(np.float64(d))).reshape(-1, d))
    tokd, dxd = _dev(tok, torch.int64), _dev(dx)
    got = ops.embed_bwd_sorted(tokd, dxd, V, L=L, rate=0.0, site=2)
    assert torch.isfinite(got).all()
    _close(got, want, rtol=5e-5, name="embed bwd sorted, 70k PAD positions")
    for _ in range(3):
        assert torch.equal(got, ops.embed_bwd_sorted(tokd, dxd, V, L=L, rate=0.0, site=2))


@pytest.mark.parametrize("d,rate", [(128, 0.0), (128, 0.1), (256, 0.1), (64, 0.0), (512, 0.0)])
def test_layernorm_residual_fwd_bwd(ops, d, rate):
    rows = 1031
    rng = np.random.RandomState(d)
    x, y, dout = rng.randn(rows, d), rng.randn(rows, d), rng.randn(rows, d)
    gamma, beta = 1 + 0.1 * rng.randn(d), 0.1 * rng.randn(d)
    st = ops.new_step_state("cuda", iterations=3)
    ops.step_prologue(st, seed=5)
    keep = np.ones((rows, d), bool)
    if rate > 0:
        keep = ops.dropout_keep_mask(ops.read_step_state(st)["drop_key"], 9, rate, rows * d).reshape(rows, d)
    z = x + oracle.dropout_fwd(y, keep, rate)
    want, cache = oracle.layernorm_fwd(z, gamma, beta)
    out, zz, stats = ops.layernorm_residual_fwd(_dev(x), _dev(y), _dev(gamma), _dev(beta), rate=rate, site=9, state=st)
    _close(out, want, name="ln fwd")
    _close(zz, z, name="ln z")
    dz, dg, db = oracle.layernorm_bwd(dout, cache)
    gz, gy, gg, gb = ops.layernorm_residual_bwd(_dev(dout), zz, stats, _dev(gamma), rate=rate, site=9, state=st)
    _close(gz, dz, rtol=5e-5, name="ln dz")
    _close(gy, oracle.dropout_fwd(dz, keep, rate), rtol=5e-5, name="ln dy")
    _close(gg, dg, rtol=5e-5, name="ln dgamma")
    _close(gb, db, rtol=5e-5, name="ln dbeta")


@pytest.mark.parametrize("N,K,acc", [(256, 768, False), (256, 256, True), (256, 1024, False), (256, 512, True), (128, 256, True), (128, 128, True),
                                     (128, 384, True), (768, 256, True), (1024, 256, True), (512, 128, True)])
def test_gemm_dgrad_accumulate_and_chains_at_full_rows_are_deterministic(ops, N, K, acc):
    """The accumulate / chained input-gradient launches at the benchmark's row count (M = 25600: every template variant of the
    weight-stationary kernel the cfg 2 / cfg 3 steps dispatch for them), against float64 and run against run.  Added after an
    'accumulate only' variant of one instantiation produced run-to-run different results that no small-M test could see."""
    M = 25600
    g = torch.Generator(device="cuda").manual_seed(N * 7919 + K)
    x = torch.randn(M, K, device="cuda", generator=g)
    wt = torch.randn(N, K, device="cuda", generator=g) / K ** 0.5
    c0 = torch.randn(M, N, device="cuda", generator=g)
    ref = x.double() @ wt.double().t() + (c0.double() if acc else 0)
    outs = []
    for _ in range(3):
        out = c0.clone()
        ops.gemm(x, wt, b_kcontig=True, out=out, accumulate=acc, precision=6)
        outs.append(out)
    assert torch.equal(outs[0], outs[1]) and torch.equal(outs[0], outs[2]), "run-to-run different results"
    err = (outs[0].double() - ref).abs().max().item() / ref.abs().max().item()
    assert err < 2e-6, err


# every instantiation of gemm_wsx_kernel<K, NB, P, B_KC, EXTRA, KMASK, LNF, KS> the cfg 2 / cfg2grid / cfg 3 steps dispatch, as
# (N, K, form, epilogue): form "fwd" = weights [K][N], "dgrad" = weights [N][K]; epilogue in {"", "relu_bits_out", "bits", "bits_acc",
# "relu_src", "acc", "act_relu", "act_tanh", "ln", "list", "list_acc"}
_WSX_CASES = [
    # cfg 2 forward (K = 128: NB = 2; K = 512: KS = 2)
    (128, 128, "fwd", ""), (384, 128, "fwd", ""), (256, 128, "fwd", ""), (512, 128, "fwd", "relu_bits_out"), (1004, 128, "fwd", ""),
    (10004, 128, "fwd", ""), (256, 128, "fwd", "act_tanh"), (128, 512, "fwd", ""), (128, 128, "fwd", "ln"), (512, 128, "fwd", "act_relu"),
    # cfg 2 input gradients
    (128, 128, "dgrad", ""), (128, 128, "dgrad", "acc"), (128, 384, "dgrad", ""), (128, 384, "dgrad", "acc"), (128, 256, "dgrad", "acc"),
    (512, 128, "dgrad", "bits"), (512, 128, "dgrad", "relu_src"), (128, 512, "dgrad", ""), (128, 512, "dgrad", "acc"),
    (128, 1004, "dgrad", ""), (128, 1004, "dgrad", "list"), (128, 128, "dgrad", "list"), (128, 384, "dgrad", "list_acc"),
    (128, 10004, "dgrad", ""), (128, 10004, "dgrad", "list"),        # cfg2grid logits input gradient: twenty chained 512-deep launches (round 6)
    (512, 128, "dgrad", "bits_list"), (128, 512, "dgrad", "list_acc"),
    # cfg 3 (d = 256, dff = 1024): K = 256 one column per lane (KS = 1 for N > 128), chained 512-deep launches for K = 768 / 1024
    (256, 256, "fwd", ""), (768, 256, "fwd", ""), (512, 256, "fwd", ""), (1024, 256, "fwd", "relu_bits_out"), (256, 1024, "fwd", ""),
    (256, 256, "dgrad", ""), (256, 256, "dgrad", "acc"), (256, 768, "dgrad", ""), (256, 768, "dgrad", "acc"), (256, 512, "dgrad", "acc"),
    (1024, 256, "dgrad", "bits"), (1024, 256, "dgrad", "relu_src"), (256, 1024, "dgrad", "acc"), (768, 256, "dgrad", "acc"),
    (1024, 256, "dgrad", "acc"), (256, 256, "dgrad", "list_acc"),
]


@pytest.mark.parametrize("N,K,form,epi", _WSX_CASES, ids=["%s-N%d-K%d-%s" % (f, n, k, e or "plain") for n, k, f, e in _WSX_CASES])
def test_gemm_wsx_every_dispatched_instantiation_is_deterministic_over_50_runs(ops, N, K, form, epi):
    """Run-to-run bit equality of every template variant of the split-arithmetic weight-stationary kernel that the cfg 2 / cfg2grid /
    cfg 3 steps launch, at the benchmark's row count, 50 launches each, plus float64 accuracy.  Round 3 parked an 'accumulate only'
    epilogue kind whose <K = 256, one column per lane> instantiation was run-to-run different; round 4 bisected it to a compiler-formed
    v_pk_add_f32 with crossed operand selects and removed the kind (skf_gemm_wsx.hip: launch_wsx) - this test is the net under the rest."""
    M, reps = 25600, 50
    g = torch.Generator(device="cuda").manual_seed(N * 7919 + K * 31 + len(epi))
    x = torch.randn(M, K, device="cuda", generator=g)
    w = torch.randn((N, K) if form == "dgrad" else (K, N), device="cuda", generator=g) / K ** 0.5
    c0 = torch.randn(M, N, device="cuda", generator=g)
    kw = dict(b_kcontig=form == "dgrad", precision=6)
    wd = w.double().t() if form == "dgrad" else w.double()
    ref = x.double() @ wd
    acc = epi in ("acc", "list_acc")
    extra_outs = []
    if epi == "ln":
        bias = torch.randn(N, device="cuda", generator=g); res = torch.randn(M, N, device="cuda", generator=g)
        gam = torch.randn(N, device="cuda", generator=g); bet = torch.randn(N, device="cuda", generator=g)
        st = ops.new_step_state("cuda", iterations=3); ops.step_prologue(st, seed=5)
        run = lambda: ops.gemm_ln_residual(x, w, bias, res, gam, bet, rate=0.1, site=2, state=st, precision=6)   # noqa: E731
        first = run()
        for _ in range(reps - 1):
            again = run()
            assert all(torch.equal(a, b) for a, b in zip(first, again)), "run-to-run different results"
        return
    if epi in ("act_relu", "act_tanh"):
        kw["act"] = 1 if epi == "act_relu" else 2
        kw["bias"] = torch.randn(N, device="cuda", generator=g)
        pre = ref + kw["bias"].double()
        ref = pre.clamp_min(0) if epi == "act_relu" else pre.tanh()
    lens = None
    if "list" in epi:
        rows_per = 200
        lens = torch.randint(0, rows_per + 1, (M // rows_per,), generator=torch.Generator().manual_seed(1)).to(torch.int32).cuda()
        live = (torch.arange(rows_per, device="cuda")[None, :] < lens[:, None]).reshape(-1)
        x = x * live[:, None]                                   # dead rows of A are exact zeros (what the list promises)
        kw["row_blocks"], kw["row_block_rows"] = ops.row_blocks(lens, rows_per, 16), 16
        ref = x.double() @ wd
    if epi in ("bits", "bits_list", "relu_src"):
        # the relu forward of the same (M, N, K) - hidden (M, N) from K inputs - leaves the sign bits this input gradient multiplies by
        xf = torch.randn(M, K, device="cuda", generator=g); wf = torch.randn(K, N, device="cuda", generator=g) / K ** 0.5
        bits = ops.relu_bits(M, N, K, "cuda", precision=6)
        assert bits is not None
        hid = ops.gemm(xf, wf, act=1, relu_bits_out=bits, precision=6)
        ref = ref * (hid > 0)
        if epi == "relu_src":
            kw["relu_src"] = hid
        else:
            kw["relu_bits_in"] = bits
    if epi == "relu_bits_out":
        kw["act"] = 1
        kw["relu_bits_out"] = ops.relu_bits(M, N, K, "cuda", precision=6)
        ref = ref.clamp_min(0)
    if acc:
        ref = ref + c0.double()
    first = None
    for _ in range(reps):
        out = c0.clone() if acc else torch.full((M, N), float("nan"), device="cuda")
        ops.gemm(x, w, out=out, accumulate=acc, **kw)
        if first is None:
            first = out
            if epi == "relu_bits_out":
                extra_outs.append(kw["relu_bits_out"].clone())
        else:
            assert torch.equal(first, out), "run-to-run different results"
            if epi == "relu_bits_out":
                assert torch.equal(extra_outs[0], kw["relu_bits_out"])
    err = (first.double() - ref).abs().max().item() / max(ref.abs().max().item(), 1e-30)
    assert err < 3e-6, err


@pytest.mark.parametrize("rows,rate", [(25600, 0.1), (1031, 0.1), (25472, 0.0), (7, 0.1)])
def test_gemm_ln_residual_one_launch(ops, rows, rate):
    """Dense + residual + dropout + LayerNorm in one launch (the attention output projection, K = N = 128) against the oracle,
    and against the two launches it replaces: z bit-identical, out / stats to rounding."""
    d = 128
    rng = np.random.RandomState(rows)
    a, x = rng.randn(rows, d), rng.randn(rows, d)
    x[rows // 2] = 100.0 + 1e-3 * rng.randn(d)               # a row whose mean dwarfs its spread: the centred sums must hold
    w, b = rng.randn(d, d) / np.sqrt(d), 0.1 * rng.randn(d)
    gamma, beta = 1 + 0.1 * rng.randn(d), 0.1 * rng.randn(d)
    st = ops.new_step_state("cuda", iterations=3)
    ops.step_prologue(st, seed=5)
    keep = np.ones((rows, d), bool)
    if rate > 0:
        keep = ops.dropout_keep_mask(ops.read_step_state(st)["drop_key"], 4, rate, rows * d).reshape(rows, d)
    z = x + oracle.dropout_fwd(a @ w + b, keep, rate)
    want, cache = oracle.layernorm_fwd(z, gamma, beta)
    A, X, W, Bv, G, Be = _dev(a), _dev(x), _dev(w), _dev(b), _dev(gamma), _dev(beta)
    out, zz, stats = ops.gemm_ln_residual(A, W, Bv, X, G, Be, rate=rate, site=4, state=st, precision=6)
    _close(zz, z, name="fused z")
    _close(out, want, rtol=2e-5, name="fused ln out")
    mean, var = z.mean(-1), z.var(-1)
    _close(stats[:, 0], mean, rtol=2e-5, name="fused mean")
    _close(stats[:, 1], 1.0 / np.sqrt(var + 1e-6), rtol=2e-5, name="fused rstd")
    if rows > 2048:                                            # below that the plain call takes the small fp32 kernel
        y = ops.gemm(A, W, bias=Bv, precision=6)
        out2, z2, stats2 = ops.layernorm_residual_fwd(X, y, G, Be, rate=rate, site=4, state=st)
        assert torch.equal(zz, z2), "z differs from the Dense + LayerNorm launch pair"
        assert (out - out2).abs().max().item() <= 2e-5 * max(1.0, out2.abs().max().item())
        assert (stats - stats2).abs().max().item() <= 2e-5 * max(1.0, stats2.abs().max().item())
    # the backward consumes (z, stats) exactly as the LayerNorm launch leaves them
    dout = rng.randn(rows, d)
    dz, dg, db = oracle.layernorm_bwd(dout, cache)
    gz, gy, gg, gb = ops.layernorm_residual_bwd(_dev(dout), zz, stats, G, rate=rate, site=4, state=st)
    _close(gz, dz, rtol=5e-5, name="ln dz from fused stats")
    _close(gg, dg, rtol=5e-5, name="ln dgamma from fused stats")


def test_gemm_ln_residual_refuses_other_shapes(ops):
    lib = ops._lib.load()
    assert lib.skf_gemm_ln_residual_supported(25600, 128, 128, 6) == 1
    assert lib.skf_gemm_ln_residual_supported(25600, 128, 128, 0) == 0      # fp32-MFMA mode: two launches
    assert lib.skf_gemm_ln_residual_supported(25600, 128, 512, 6) == 0
    assert lib.skf_gemm_ln_residual_supported(25600, 256, 256, 6) == 0
    a = torch.zeros(64, 256, device="cuda"); w = torch.zeros(256, 256, device="cuda"); v = torch.zeros(256, device="cuda")
    with pytest.raises(RuntimeError):
        ops.gemm_ln_residual(a, w, v, a, v, v, precision=6)


# ------------------------------------------------------------------ feed-forward block in one launch per direction
@pytest.mark.parametrize("rows,rate,prec", [(25600, 0.1, 6), (1031, 0.1, 6), (25472, 0.0, 6), (7, 0.1, 6), (4000, 0.1, 3)])
def test_ffn_fused_forward(ops, rows, rate, prec):
    """h = relu(x.W1 + b1), z = x + dropout(h.W2 + b2), out = LayerNorm(z) in one launch (builders/layers/transformer.py:194-198,
    221-224) against the oracle, and against the three launches it replaces."""
    d, dff = 128, 512
    rng = np.random.RandomState(rows)
    x = rng.randn(rows, d)
    x[rows // 2] = 100.0 + 1e-3 * rng.randn(d)
    w1, b1 = rng.randn(d, dff) / np.sqrt(d), 0.1 * rng.randn(dff)
    w2, b2 = rng.randn(dff, d) / np.sqrt(dff), 0.1 * rng.randn(d)
    gamma, beta = 1 + 0.1 * rng.randn(d), 0.1 * rng.randn(d)
    st = ops.new_step_state("cuda", iterations=3)
    ops.step_prologue(st, seed=5)
    keep = np.ones((rows, d), bool)
    if rate > 0:
        keep = ops.dropout_keep_mask(ops.read_step_state(st)["drop_key"], 9, rate, rows * d).reshape(rows, d)
    h = np.maximum(x @ w1 + b1, 0)
    z = x + oracle.dropout_fwd(h @ w2 + b2, keep, rate)
    want, _ = oracle.layernorm_fwd(z, gamma, beta)
    X, W1, B1, W2, B2, G, Be = _dev(x), _dev(w1), _dev(b1), _dev(w2), _dev(b2), _dev(gamma), _dev(beta)
    img, = ops.ffn_weight_images([(W1, W2)], transpose=False, precision=prec)
    out, zz, stats, hh, bits = ops.ffn_fused_fwd(X, img, B1, B2, G, Be, dff, rate=rate, site=9, state=st, precision=prec)
    tol = RTOL if prec == 6 else 2e-4
    _close(hh, h, rtol=tol, name="fused h")
    _close(zz, z, rtol=tol, name="fused z")
    _close(out, want, rtol=max(tol, 2e-5), name="fused ln out")
    _close(stats[:, 0], z.mean(-1), rtol=max(tol, 2e-5), name="fused mean")
    _close(stats[:, 1], 1.0 / np.sqrt(z.var(-1) + 1e-6), rtol=max(tol, 2e-5), name="fused rstd")
    # sign bits: word [tile][block][wave][r], bit 16g + i <-> h[16 tile + i][128 block + 16 wave + 4g + r] > 0
    nt = (rows + 15) // 16
    hp = np.zeros((nt * 16, dff)); hp[:rows] = hh.cpu().numpy(); hp[rows:] = np.maximum(b1, 0)   # rows behind M: x reads as 0
    lanes = (hp.reshape(nt, 16, 4, 8, 4, 4) > 0)                      # [tile][i][block][wave][g][r]
    lanes = lanes.transpose(0, 2, 3, 5, 4, 1).reshape(nt, 4, 8, 4, 64)  # [tile][block][wave][r][lane = 16g + i]
    words = (lanes.astype(np.uint64) << np.arange(64, dtype=np.uint64)).sum(-1, dtype=np.uint64)
    got_words = bits.cpu().numpy().view(np.uint64)[:words.size].reshape(words.shape)
    assert np.array_equal(got_words, words), "sign bits differ from the stored hidden tensor"
    if rows > 2048 and prec == 6:
        h2 = ops.gemm(X, W1, bias=B1, act=1, precision=prec)
        y2 = ops.gemm(h2, W2, bias=B2, precision=prec)
        out2, z2, stats2 = ops.layernorm_residual_fwd(X, y2, G, Be, rate=rate, site=9, state=st)
        assert (hh - h2).abs().max().item() <= 2e-6 * max(1.0, h2.abs().max().item())
        assert (zz - z2).abs().max().item() <= 2e-6 * max(1.0, z2.abs().max().item())
        assert (out - out2).abs().max().item() <= 2e-5 * max(1.0, out2.abs().max().item())


@pytest.mark.parametrize("rows,listed,acc", [(25472, True, True), (25600, False, True), (1031, False, False), (3184, True, False), (7, False, True)])
def test_ffn_fused_backward(ops, rows, listed, acc):
    """dh = (dy.W2^T) o relu'(h), dx (+)= dh.W1^T in one launch against float64, with and without the live-row-block list of the
    decoder-side backward (dead rows: dy == 0 -> dh rows stored as zeros, dx rows left alone / zeroed)."""
    d, dff = 128, 512
    rng = np.random.RandomState(rows + 1)
    x = rng.randn(rows, d)
    w1, b1 = rng.randn(d, dff) / np.sqrt(d), 0.1 * rng.randn(dff)
    w2, b2 = rng.randn(dff, d) / np.sqrt(dff), 0.1 * rng.randn(d)
    dy = rng.randn(rows, d)
    dx0 = rng.randn(rows, d)
    blocks = None
    if listed:
        Ld = 199
        B = rows // Ld
        assert B * Ld == rows
        live = rng.randint(0, Ld + 1, size=B).astype(np.int32)
        live[0] = 0; live[-1] = Ld
        for b in range(B):
            dy[b * Ld + live[b]:(b + 1) * Ld] = 0.0
        blocks = ops.row_blocks(_dev(live, torch.int32), Ld, 16)
    X, W1, B1, W2, B2 = _dev(x), _dev(w1), _dev(b1), _dev(w2), _dev(b2)
    g = _dev(np.ones(d)); be = _dev(np.zeros(d))
    img, = ops.ffn_weight_images([(W1, W2)], transpose=False)
    _, _, _, hh, bits = ops.ffn_fused_fwd(X, img, B1, B2, g, be, dff)
    imgt, = ops.ffn_weight_images([(W1, W2)], transpose=True)
    hmask = hh.cpu().numpy().astype(np.float64) > 0
    dh = (dy @ w2.T) * hmask
    dx = dh @ w1.T + (dx0 if acc else 0.0)
    DX = _dev(dx0) if acc else None
    gdh, gdx = ops.ffn_fused_bwd(_dev(dy), imgt, bits, dff, dx=DX, row_blocks=blocks)
    _close(gdh, dh, name="fused dh")
    _close(gdx, dx, name="fused dx")
    if rows > 2048:
        dh2 = ops.gemm(_dev(dy), W2, a_kcontig=True, b_kcontig=True, relu_src=hh)
        assert (gdh - dh2).abs().max().item() <= 2e-6 * max(1.0, dh2.abs().max().item())


@pytest.mark.parametrize("rows,listed,rate", [(25472, True, 0.1), (25600, False, 0.1), (1031, False, 0.0), (3184, True, 0.0), (7, False, 0.1)])
def test_ffn_fused_backward_from_layernorm_gradient(ops, rows, listed, rate):
    """The backward launch with the LayerNorm-backward prologue: dz = LN'(dout), dy = dropout'(dz), dh, dx = dz + dh.W1^T, dgamma, dbeta
    against the oracle / float64, and against the LayerNorm launch + plain fused launch it replaces."""
    d, dff = 128, 512
    rng = np.random.RandomState(rows + 2)
    x = rng.randn(rows, d)
    w1, b1 = rng.randn(d, dff) / np.sqrt(d), 0.1 * rng.randn(dff)
    w2, b2 = rng.randn(dff, d) / np.sqrt(dff), 0.1 * rng.randn(d)
    gamma, beta = 1 + 0.1 * rng.randn(d), 0.1 * rng.randn(d)
    dout = rng.randn(rows, d)
    blocks = None
    if listed:
        Ld = 199
        B = rows // Ld
        live = rng.randint(0, Ld + 1, size=B).astype(np.int32)
        live[0] = 0; live[-1] = Ld
        for b in range(B):
            dout[b * Ld + live[b]:(b + 1) * Ld] = 0.0
        blocks = ops.row_blocks(_dev(live, torch.int32), Ld, 16)
    st = ops.new_step_state("cuda", iterations=3)
    ops.step_prologue(st, seed=5)
    X, W1, B1, W2, B2, G, Be = _dev(x), _dev(w1), _dev(b1), _dev(w2), _dev(b2), _dev(gamma), _dev(beta)
    img, = ops.ffn_weight_images([(W1, W2)], transpose=False)
    imgt, = ops.ffn_weight_images([(W1, W2)], transpose=True)
    out, zz, stats, hh, bits = ops.ffn_fused_fwd(X, img, B1, B2, G, Be, dff, rate=rate, site=9, state=st)
    keep = np.ones((rows, d), bool)
    if rate > 0:
        keep = ops.dropout_keep_mask(ops.read_step_state(st)["drop_key"], 9, rate, rows * d).reshape(rows, d)
    z64 = zz.cpu().numpy().astype(np.float64)
    _, cache = oracle.layernorm_fwd(z64, gamma, beta)
    dz, dg, db = oracle.layernorm_bwd(dout, cache)
    dy = dz * keep / (1.0 - rate)
    dh = (dy @ w2.T) * (hh.cpu().numpy() > 0)
    dx = dz + dh @ w1.T
    gdy, gdh, gdx, gdg, gdb = ops.ffn_fused_bwd_ln(_dev(dout), zz, stats, G, imgt, bits, dff, rate=rate, site=9, state=st, row_blocks=blocks)
    _close(gdy, dy, rtol=5e-5, name="dy")
    _close(gdh, dh, rtol=5e-5, name="dh")
    _close(gdx, dx, rtol=5e-5, name="dx")
    _close(gdg, dg, rtol=5e-5, name="dgamma")
    _close(gdb, db, rtol=5e-5, name="dbeta")
    if rows > 2048 and not listed:
        rdz, rdy, _, _ = ops.layernorm_residual_bwd(_dev(dout), zz, stats, G, rate=rate, site=9, state=st)
        assert (gdy - rdy).abs().max().item() <= 2e-6 * max(1.0, rdy.abs().max().item())


@pytest.mark.parametrize("rows,listed,rate", [(25472, True, 0.1), (25600, False, 0.1), (1031, False, 0.0), (3184, True, 0.0), (7, False, 0.1)])
def test_layernorm_bwd_dgrad_one_launch(ops, rows, listed, rate):
    """dz = LayerNorm'(dout), dy = dropout'(dz), da = dy . W^T in one launch (the attention output projection's backward,
    builders/layers/transformer.py:186, 221-224) against the oracle, and against the two launches it replaces."""
    d = 128
    rng = np.random.RandomState(rows + 3)
    z = rng.randn(rows, d) * 1.5 + 0.3
    w = rng.randn(d, d) / np.sqrt(d)
    gamma, beta = 1 + 0.1 * rng.randn(d), 0.1 * rng.randn(d)
    dout = rng.randn(rows, d)
    blocks = None
    if listed:
        Ld = 199
        B = rows // Ld
        live = rng.randint(0, Ld + 1, size=B).astype(np.int32)
        live[0] = 0; live[-1] = Ld
        for b in range(B):
            dout[b * Ld + live[b]:(b + 1) * Ld] = 0.0
        blocks = ops.row_blocks(_dev(live, torch.int32), Ld, 16)
    st = ops.new_step_state("cuda", iterations=3)
    ops.step_prologue(st, seed=5)
    keep = np.ones((rows, d), bool)
    if rate > 0:
        keep = ops.dropout_keep_mask(ops.read_step_state(st)["drop_key"], 6, rate, rows * d).reshape(rows, d)
    Z, W, G = _dev(z), _dev(w), _dev(gamma)
    z32 = Z.cpu().numpy().astype(np.float64)
    _, cache = oracle.layernorm_fwd(z32, gamma, beta)
    stats = _dev(np.stack([z32.mean(-1), 1.0 / np.sqrt(z32.var(-1) + 1e-6)], -1))
    dz, dg, db = oracle.layernorm_bwd(dout, cache)
    dy = dz * keep / (1.0 - rate)
    da = dy @ w.T
    img = ops.dense_weight_image(W, transpose=True)
    gdz, gdy, gda, gdg, gdb = ops.layernorm_bwd_dgrad(_dev(dout), Z, stats, G, img, rate=rate, site=6, state=st, row_blocks=blocks)
    _close(gdz, dz, rtol=5e-5, name="dz")
    _close(gdy, dy, rtol=5e-5, name="dy")
    _close(gda, da, rtol=5e-5, name="da")
    _close(gdg, dg, rtol=5e-5, name="dgamma")
    _close(gdb, db, rtol=5e-5, name="dbeta")
    if rows > 2048 and not listed:
        rdz, rdy, _, _ = ops.layernorm_residual_bwd(_dev(dout), Z, stats, G, rate=rate, site=6, state=st)
        rda = ops.gemm(rdy, W, a_kcontig=True, b_kcontig=True)
        assert (gdy - rdy).abs().max().item() <= 2e-6 * max(1.0, rdy.abs().max().item())
        assert (gda - rda).abs().max().item() <= 4e-6 * max(1.0, rda.abs().max().item())


@pytest.mark.parametrize("rows,listed,rate", [(25472, True, 0.1), (25600, False, 0.1), (1031, False, 0.0), (7, False, 0.1)])
def test_layernorm_bwd_dgrad_with_a_leading_product(ops, rows, listed, rate):
    """skf_layernorm_bwd_dgrad_lead_f32 (round 5): the gradient of the LayerNorm output is dout + a . Wl^T - the decoder's cross-attention
    query projection's input gradient formed inside the self-attention sublayer's LayerNorm-backward launch
    (builders/layers/transformer.py:258-262) - against the oracle and against the accumulating GEMM launch followed by the plain launch."""
    d = 128
    rng = np.random.RandomState(rows + 17)
    z = rng.randn(rows, d) * 1.5 + 0.3
    w, wl = rng.randn(d, d) / np.sqrt(d), rng.randn(d, d) / np.sqrt(d)
    gamma, beta = 1 + 0.1 * rng.randn(d), 0.1 * rng.randn(d)
    dout, a = rng.randn(rows, d), rng.randn(rows, d)
    blocks = None
    if listed:
        Ld = 199
        B = rows // Ld
        live = rng.randint(0, Ld + 1, size=B).astype(np.int32)
        live[0] = 0; live[-1] = Ld
        for b in range(B):
            dout[b * Ld + live[b]:(b + 1) * Ld] = 0.0
            a[b * Ld + live[b]:(b + 1) * Ld] = 0.0
        blocks = ops.row_blocks(_dev(live, torch.int32), Ld, 16)
    st = ops.new_step_state("cuda", iterations=3)
    ops.step_prologue(st, seed=5)
    keep = np.ones((rows, d), bool)
    if rate > 0:
        keep = ops.dropout_keep_mask(ops.read_step_state(st)["drop_key"], 6, rate, rows * d).reshape(rows, d)
    Z, W, WL, G, A, DOUT = _dev(z), _dev(w), _dev(wl), _dev(gamma), _dev(a), _dev(dout)
    z32 = Z.cpu().numpy().astype(np.float64)
    _, cache = oracle.layernorm_fwd(z32, gamma, beta)
    stats = _dev(np.stack([z32.mean(-1), 1.0 / np.sqrt(z32.var(-1) + 1e-6)], -1))
    dz, dg, db = oracle.layernorm_bwd(dout + a @ wl.T, cache)
    dy = dz * keep / (1.0 - rate)
    da = dy @ w.T
    img, limg = ops.dense_weight_image(W, transpose=True), ops.dense_weight_image(WL, transpose=True)
    got = ops.layernorm_bwd_dgrad(DOUT, Z, stats, G, img, rate=rate, site=6, state=st, row_blocks=blocks, lead=(A, limg))
    for g_, w_, n in zip(got, (dz, dy, da, dg, db), ("dz", "dy", "da", "dgamma", "dbeta")):
        _close(g_, w_, rtol=6e-5, name=n)
    # the two launches it replaces: dout += a . Wl^T (accumulating input-gradient GEMM), then the plain launch
    acc = ops.gemm(A, WL, a_kcontig=True, b_kcontig=True, out=DOUT.clone(), accumulate=True)
    ref = ops.layernorm_bwd_dgrad(acc, Z, stats, G, img, rate=rate, site=6, state=st, row_blocks=blocks)
    for g_, r_, n in zip(got[:3], ref[:3], ("dz", "dy", "da")):
        assert (g_ - r_).abs().max().item() <= 8e-6 * max(1.0, r_.abs().max().item()), n


@pytest.mark.parametrize("rows,n2", [(25600, 384), (25472, 128), (1031, 384), (7, 128)])
def test_ffn_fused_forward_with_chained_projection(ops, rows, n2):
    """The forward launch going on to the next layer's q|k|v (N = 384) / query (N = 128) projection of its LayerNorm output."""
    d, dff = 128, 512
    rng = np.random.RandomState(rows + n2)
    x = rng.randn(rows, d)
    w1, b1 = rng.randn(d, dff) / np.sqrt(d), 0.1 * rng.randn(dff)
    w2, b2 = rng.randn(dff, d) / np.sqrt(dff), 0.1 * rng.randn(d)
    wp, bp = rng.randn(d, n2) / np.sqrt(d), 0.1 * rng.randn(n2)
    gamma, beta = 1 + 0.1 * rng.randn(d), 0.1 * rng.randn(d)
    st = ops.new_step_state("cuda", iterations=3)
    ops.step_prologue(st, seed=5)
    keep = ops.dropout_keep_mask(ops.read_step_state(st)["drop_key"], 9, 0.1, rows * d).reshape(rows, d)
    h = np.maximum(x @ w1 + b1, 0)
    z = x + oracle.dropout_fwd(h @ w2 + b2, keep, 0.1)
    want, _ = oracle.layernorm_fwd(z, gamma, beta)
    X, W1, B1, W2, B2, G, Be, WP, BP = _dev(x), _dev(w1), _dev(b1), _dev(w2), _dev(b2), _dev(gamma), _dev(beta), _dev(wp), _dev(bp)
    img, = ops.ffn_weight_images([(W1, W2)], transpose=False)
    pimg = ops.dense_weight_image(WP, transpose=False)
    out, zz, stats, hh, bits, po = ops.ffn_fused_fwd(X, img, B1, B2, G, Be, dff, rate=0.1, site=9, state=st, proj=(pimg, BP))
    _close(zz, z, name="z")
    _close(out, want, name="ln out")
    _close(po, want @ wp + bp, rtol=3e-5, name="chained projection")
    out1, z1, _, _, _ = ops.ffn_fused_fwd(X, img, B1, B2, G, Be, dff, rate=0.1, site=9, state=st)
    assert torch.equal(out, out1) and torch.equal(zz, z1), "the chained launch changes the block's own outputs"
    if rows > 2048:
        ref = ops.gemm(out, WP, bias=BP)
        assert (po - ref).abs().max().item() <= 4e-6 * max(1.0, ref.abs().max().item())


@pytest.mark.parametrize("rows,n2,rate", [(25600, 384, 0.1), (25472, 0, 0.1), (1031, 384, 0.0), (7, 0, 0.1)])
def test_ffn_block_forward_from_attention_output(ops, rows, n2, rate):
    """The forward launch starting at the attention output: x1 = LayerNorm(x + dropout(a.Wo + bo)) (builders/layers/transformer.py:186,
    216-224), the feed-forward block on x1, optionally the next q|k|v projection - against the oracle and the launches it replaces."""
    d, dff = 128, 512
    rng = np.random.RandomState(rows + n2 + 5)
    a, x = rng.randn(rows, d), rng.randn(rows, d)
    wo, bo = rng.randn(d, d) / np.sqrt(d), 0.1 * rng.randn(d)
    w1, b1 = rng.randn(d, dff) / np.sqrt(d), 0.1 * rng.randn(dff)
    w2, b2 = rng.randn(dff, d) / np.sqrt(dff), 0.1 * rng.randn(d)
    g1, be1, g2, be2 = 1 + 0.1 * rng.randn(d), 0.1 * rng.randn(d), 1 + 0.1 * rng.randn(d), 0.1 * rng.randn(d)
    st = ops.new_step_state("cuda", iterations=3)
    ops.step_prologue(st, seed=5)
    key = ops.read_step_state(st)["drop_key"]
    k1 = k2 = np.ones((rows, d), bool)
    if rate > 0:
        k1 = ops.dropout_keep_mask(key, 4, rate, rows * d).reshape(rows, d)
        k2 = ops.dropout_keep_mask(key, 9, rate, rows * d).reshape(rows, d)
    z1 = x + oracle.dropout_fwd(a @ wo + bo, k1, rate)
    x1, _ = oracle.layernorm_fwd(z1, g1, be1)
    h = np.maximum(x1 @ w1 + b1, 0)
    z2 = x1 + oracle.dropout_fwd(h @ w2 + b2, k2, rate)
    x2, _ = oracle.layernorm_fwd(z2, g2, be2)
    A, X, WO, BO, W1, B1, W2, B2 = (_dev(v) for v in (a, x, wo, bo, w1, b1, w2, b2))
    G1, BE1, G2, BE2 = (_dev(v) for v in (g1, be1, g2, be2))
    img, = ops.ffn_weight_images([(W1, W2)], transpose=False)
    pimg = ops.dense_weight_image(WO, transpose=False)
    proj = None
    if n2:
        wp, bp = rng.randn(d, n2) / np.sqrt(d), 0.1 * rng.randn(n2)
        proj = (ops.dense_weight_image(_dev(wp), transpose=False), _dev(bp))
    r = ops.ffn_block_fwd(A, X, (pimg, BO, G1, BE1), img, B1, B2, G2, BE2, dff, rate=rate, pre_site=4, site=9, state=st, proj=proj)
    _close(r["z1"], z1, name="z1")
    _close(r["x1"], x1, name="x1")
    _close(r["stats1"][:, 0], z1.mean(-1), name="mean1")
    _close(r["h"], h, rtol=4e-5, name="h")
    _close(r["z"], z2, rtol=4e-5, name="z2")
    _close(r["out"], x2, rtol=4e-5, name="x2")
    if n2:
        _close(r["proj_out"], x2 @ wp + bp, rtol=6e-5, name="chained projection")
    if rows > 2048:
        o1, zz1, _ = ops.gemm_ln_residual(A, WO, BO, X, G1, BE1, rate=rate, site=4, state=st, precision=6)
        assert (r["z1"] - zz1).abs().max().item() <= 2e-6 * max(1.0, zz1.abs().max().item())
        assert (r["x1"] - o1).abs().max().item() <= 2e-5 * max(1.0, o1.abs().max().item())


@pytest.mark.parametrize("rows,n2,rate", [(25472, 128, 0.1), (1031, 128, 0.0), (7, 128, 0.1), (4000, 384, 0.1)])
def test_attention_tail_with_the_next_projection(ops, rows, n2, rate):
    """skf_ffn_block_fwd_f32 without a feed-forward image (round 5): the decoder's self-attention tail x1 = LayerNorm(x + dropout(a.Wo + bo))
    and the cross-attention query projection q = x1.Wq + bq (builders/layers/transformer.py:258-262) in one row-owner launch - against
    the oracle, against the two launches it replaces, and with the leading stage's outputs bit-equal to the full block's."""
    d = 128
    rng = np.random.RandomState(rows + n2 + 11)
    a, x = rng.randn(rows, d), rng.randn(rows, d)
    wo, bo = rng.randn(d, d) / np.sqrt(d), 0.1 * rng.randn(d)
    wp, bp = rng.randn(d, n2) / np.sqrt(d), 0.1 * rng.randn(n2)
    g1, be1 = 1 + 0.1 * rng.randn(d), 0.1 * rng.randn(d)
    st = ops.new_step_state("cuda", iterations=3)
    ops.step_prologue(st, seed=5)
    key = ops.read_step_state(st)["drop_key"]
    k1 = ops.dropout_keep_mask(key, 4, rate, rows * d).reshape(rows, d) if rate > 0 else np.ones((rows, d), bool)
    z1 = x + oracle.dropout_fwd(a @ wo + bo, k1, rate)
    x1, _ = oracle.layernorm_fwd(z1, g1, be1)
    A, X, WO, BO, WP, BP, G1, BE1 = (_dev(v) for v in (a, x, wo, bo, wp, bp, g1, be1))
    pre = (ops.dense_weight_image(WO, transpose=False), BO, G1, BE1)
    r = ops.attn_tail_proj(A, X, pre, (ops.dense_weight_image(WP, transpose=False), BP), rate=rate, pre_site=4, state=st)
    _close(r["z1"], z1, name="z1")
    _close(r["x1"], x1, name="x1")
    _close(r["stats1"][:, 0], z1.mean(-1), name="mean1")
    _close(r["proj_out"], x1 @ wp + bp, rtol=3e-5, name="projection")
    # the two launches it replaces
    o1, zz1, _ = ops.gemm_ln_residual(A, WO, BO, X, G1, BE1, rate=rate, site=4, state=st, precision=6)
    ref = ops.gemm(r["x1"], WP, bias=BP)
    assert (r["z1"] - zz1).abs().max().item() <= 2e-6 * max(1.0, zz1.abs().max().item())
    assert (r["x1"] - o1).abs().max().item() <= 2e-5 * max(1.0, o1.abs().max().item())
    assert (r["proj_out"] - ref).abs().max().item() <= 4e-6 * max(1.0, ref.abs().max().item())
    # the same leading stage inside the full block
    w1, w2 = _dev(rng.randn(d, 512) / np.sqrt(d)), _dev(rng.randn(512, d) / np.sqrt(512))
    img, = ops.ffn_weight_images([(w1, w2)], transpose=False)
    full = ops.ffn_block_fwd(A, X, pre, img, _dev(np.zeros(512)), _dev(np.zeros(d)), G1, BE1, 512, rate=rate, pre_site=4, site=9, state=st)
    assert torch.equal(full["z1"], r["z1"]) and torch.equal(full["x1"], r["x1"]) and torch.equal(full["stats1"], r["stats1"])


def test_ffn_fused_refuses_other_shapes(ops):
    lib = ops._lib.load()
    assert lib.skf_ffn_fused_supported(25600, 128, 512, 6) == 1
    assert lib.skf_ffn_fused_supported(25600, 128, 512, 0) == 0
    assert lib.skf_ffn_fused_supported(25600, 256, 1024, 6) == 0
    assert lib.skf_ffn_image_bytes(128, 512, 6) == 786432 and lib.skf_ffn_image_bytes(256, 1024, 6) == 0


# ------------------------------------------------------------------ loss heads
@pytest.mark.parametrize("V", [1004, 52, 10004])
def test_recon_softmax_ce(ops, V):
    B, L = 4, 31
    rng = np.random.RandomState(V)
    logits = rng.randn(B, L - 1, V) * 3
    tar = rng.randint(1, V, size=(B, L)); tar[:, 20:] = 0
    logits[0, 0, :] = 0.0                      # all-equal row: argmax must be index 0
    loss, cache = oracle.recon_loss_fwd(tar[:, 1:], logits, weight=1.0)
    g = oracle.recon_loss_bwd(cache)
    lg = _dev(logits.reshape(-1, V))
    rl, rh, _ = ops.softmax_ce(lg, _dev(tar, torch.int64), tgt_cols=L - 1, tgt_off=1, mask_pad=True,
                               scale=1.0 / (B * (L - 1)))
    assert abs(rl.sum().item() / (B * (L - 1)) - loss) < 1e-5 * max(1, abs(loss))
    _close(lg, g.reshape(-1, V), rtol=5e-5, name="dlogits")
    hits = (logits.argmax(-1) == tar[:, 1:]).reshape(-1)
    assert np.array_equal(rh.cpu().numpy() > 0.5, hits)


def test_class_softmax_ce(ops):
    B, C_ = 128, 345
    rng = np.random.RandomState(1)
    logits = rng.randn(B, C_)
    lab = rng.randint(0, C_, size=(B, 1))
    loss, cache = oracle.class_loss_fwd(lab, logits)
    lg = _dev(logits)
    rl, rh, probs = ops.softmax_ce(lg, _dev(lab, torch.int64), tgt_cols=1, scale=1.0 / B, want_probs=True)
    assert abs(rl.mean().item() - loss) < 1e-5
    _close(lg, oracle.class_loss_bwd(cache), rtol=5e-5, name="dclass")
    e = np.exp(logits - logits.max(-1, keepdims=True))
    _close(probs, e / e.sum(-1, keepdims=True), name="probs")


# ------------------------------------------------------------------ bottleneck / expander / adam
def test_pool_and_expander(ops):
    B, L, U, d = 6, 200, 256, 128
    rng = np.random.RandomState(2)
    x = rng.randn(B, L, d)
    P = {"bottleneck/W_attn": rng.randn(d, U) * 0.05, "bottleneck/b_attn": rng.randn(U) * 0.1,
         "bottleneck/V_attn": rng.uniform(-0.5, 0.5, (U, 1)), "expand/kernel": rng.randn(1, L), "expand/bias": rng.randn(L)}
    emb, a, cache = oracle.self_attn_v1_fwd(P, x)
    u = np.tanh(x @ P["bottleneck/W_attn"] + P["bottleneck/b_attn"])
    ga, gemb = ops.pool_fwd(_dev(u), _dev(P["bottleneck/V_attn"][:, 0]), _dev(x))
    _close(gemb, emb, name="pool emb")
    _close(ga, a[..., 0], name="pool a")
    demb = rng.randn(B, d)
    G = {}
    dx_total = oracle.self_attn_v1_bwd(demb, cache, P, G)
    dpre, dx, dV = ops.pool_bwd(_dev(u), _dev(P["bottleneck/V_attn"][:, 0]), _dev(x), ga, _dev(demb))
    _close(dV, G["bottleneck/V_attn"][:, 0], rtol=5e-5, name="dV_attn")
    dpre_np = dpre.cpu().numpy().astype(np.float64)
    _close(x.reshape(-1, d).T @ dpre_np.reshape(-1, U), G["bottleneck/W_attn"], rtol=5e-5, name="dW_attn via dpre")
    _close(dx.cpu().numpy().astype(np.float64) + dpre_np @ P["bottleneck/W_attn"].T, dx_total, rtol=5e-5, name="pool dx")
    # expander
    pre, c2 = oracle.dense_expander_fwd(P, emb)
    _close(ops.expander_fwd(_dev(emb), _dev(P["expand/kernel"][0]), _dev(P["expand/bias"])), pre, name="expander fwd")
    dpre2 = rng.randn(B, L, d)
    G2 = {}
    de = oracle.dense_expander_bwd(dpre2, c2, G2)
    gde, gdw, gdb = ops.expander_bwd(_dev(dpre2), _dev(emb), _dev(P["expand/kernel"][0]))
    _close(gde, de, rtol=5e-5, name="expander demb")
    _close(gdw, G2["expand/kernel"][0], rtol=5e-5, name="expander dw")
    _close(gdb, G2["expand/bias"], rtol=5e-5, name="expander dbias")


def test_warmup_decay_and_adam(ops):
    n = 10007
    rng = np.random.RandomState(4)
    for it in (0, 1, 4999, 5000, 20000):
        st = ops.new_step_state("cuda", iterations=it)
        ops.step_prologue(st, schedule=0, p0=128.0, p1=float(5000 ** -1.5))
        s = ops.read_step_state(st)
        want_lr = float(oracle.warmup_decay(it, 128, 5000))
        assert abs(s["lr"] - want_lr) <= 1e-6 * max(want_lr, 1e-12), (it, s["lr"], want_lr)
        w, g, m, v = rng.randn(n), rng.randn(n), rng.randn(n) * 0.1, np.abs(rng.randn(n)) * 0.01
        W, M_, V_ = w.copy(), m.copy(), v.copy()
        oracle.adam_update(W, g * 0.5, M_, V_, it, want_lr)
        tw, tm, tv = _dev(w), _dev(m), _dev(v)
        ops.adam_step(tw, _dev(g), tm, tv, st, grad_scale=0.5)
        _close(tm, M_, name="adam m"); _close(tv, V_, name="adam v")
        assert np.abs(tw.cpu().numpy() - W).max() <= 1e-6 * max(1.0, want_lr * 1e3), it
        ops.step_epilogue(st)
        assert ops.read_step_state(st)["iterations"] == it + 1
    # KATs K1: lr(1)=2.5e-7, lr(5000)=1.25e-3, lr(20000)=6.25e-4, lr(0)=0
    for it, want in ((0, 0.0), (1, 2.5e-7), (5000, 1.25e-3), (20000, 6.25e-4)):
        st = ops.new_step_state("cuda", iterations=it)
        ops.step_prologue(st)
        assert abs(ops.read_step_state(st)["lr"] - want) <= 2e-6 * max(want, 1e-9)


@pytest.mark.parametrize("dh,Lk,cap", [(16, 1, 8), (16, 37, 64), (32, 200, 200), (64, 300, 320)])
def test_attention_decode_single_query(ops, dh, Lk, cap):
    """skf_attention_decode == scaled_dot_product_attention for one query row per (sample, head), reading the first Lk
    rows of a (B, cap, 2d) K|V cache, with a byte key mask and a per-sample key limit."""
    B, H = 5, 4
    d = H * dh
    rng = np.random.RandomState(dh + Lk)
    q = rng.randn(B, d).astype(np.float32)
    cache = rng.randn(B, cap, 2 * d).astype(np.float32)
    mask = (rng.rand(B, cap) < 0.3)
    mask[0, :] = True                       # fully masked sample: uniform weights over its Lk keys
    limit = rng.randint(1, Lk + 1, size=B).astype(np.int32)
    ct = torch.from_numpy(cache).cuda()
    for use_mask, use_limit in ((False, False), (True, False), (True, True)):
        o = ops.attention_decode(torch.from_numpy(q).cuda(), ct[:, :, :d], ct[:, :, d:], H, n_keys=Lk,
                                 key_mask=torch.from_numpy(mask.astype(np.uint8)).cuda() if use_mask else None,
                                 key_limit=torch.from_numpy(limit).cuda() if use_limit else None)
        m = np.zeros((B, 1, 1, Lk), np.float32)
        if use_mask:
            m = np.maximum(m, mask[:, None, None, :Lk].astype(np.float32))
        if use_limit:
            m = np.maximum(m, (np.arange(Lk)[None, :] >= limit[:, None])[:, None, None, :].astype(np.float32))
        # float32 oracle: for the fully masked sample the reference's fp32 `logits + (-1e9)` rounds to exactly -1e9
        # (uniform weights); a float64 oracle would keep the score differences
        sp = lambda t, n: t.reshape(B, n, H, dh).transpose(0, 2, 1, 3).astype(np.float32)  # noqa: E731
        want, _, _ = oracle.sdpa_fwd(sp(q[:, None, :], 1), sp(cache[:, :Lk, :d], Lk), sp(cache[:, :Lk, d:], Lk), m)
        want = want.transpose(0, 2, 1, 3).reshape(B, d)
        _close(o, want, 3e-5, "attention_decode mask=%s limit=%s" % (use_mask, use_limit))


def test_attention_decode_appends_new_row_and_reads_step_from_device(ops):
    """Graph-replayable form: the number of keys is *step + 1, the newest K/V row comes from the projection output and is
    appended to the cache at row *step; limit_from_step masks keys >= step + 1 of a full-length cross cache."""
    B, H, dh, cap = 3, 4, 16, 24
    d = H * dh
    rng = np.random.RandomState(7)
    cache = rng.randn(B, cap, 2 * d).astype(np.float32)
    ct = torch.from_numpy(cache.copy()).cuda()
    step = torch.zeros(1, dtype=torch.int32, device="cuda")
    sp = lambda t, n: t.reshape(B, n, H, dh).transpose(0, 2, 1, 3).astype(np.float32)  # noqa: E731
    for i in (0, 5, 23):
        step.fill_(i)
        q = rng.randn(B, d).astype(np.float32)
        kvn = rng.randn(B, 2 * d).astype(np.float32)
        kvt = torch.from_numpy(kvn).cuda()
        o = ops.attention_decode(torch.from_numpy(q).cuda(), ct[:, :, :d], ct[:, :, d:], H, n_keys=cap, step=step,
                                 k_new=kvt[:, :d], v_new=kvt[:, d:])
        cache[:, i] = kvn                                        # what the kernel must have appended
        want, _, _ = oracle.sdpa_fwd(sp(q[:, None, :], 1), sp(cache[:, :i + 1, :d], i + 1), sp(cache[:, :i + 1, d:], i + 1), None)
        _close(o, want.transpose(0, 2, 1, 3).reshape(B, d), 3e-5, "append step %d" % i)
        assert np.array_equal(ct[:, i].cpu().numpy(), kvn)
        # cross attention over the whole cache with keys >= step + 1 masked
        o2 = ops.attention_decode(torch.from_numpy(q).cuda(), ct[:, :, :d], ct[:, :, d:], H, n_keys=cap, step=step,
                                  limit_from_step=True)
        m = (np.arange(cap)[None, None, None, :] > i).astype(np.float32)
        cur = ct.cpu().numpy()
        want2, _, _ = oracle.sdpa_fwd(sp(q[:, None, :], 1), sp(cur[:, :, :d], cap), sp(cur[:, :, d:], cap), m)
        _close(o2, want2.transpose(0, 2, 1, 3).reshape(B, d), 3e-5, "limit_from_step %d" % i)


def test_gemm_wgrad_partial_group_equals_single_launches(ops):
    """Four weight gradients of a cfg-2 layer (one with a live-row-block list) as ONE grouped launch of the partial-tile kernel:
    the slabs are bit-identical to those of four separate launches."""
    import ctypes as C
    from sketchformer_amd import _lib
    lib = _lib.load()
    rng = np.random.RandomState(17)
    B, Ld = 64, 199
    R = B * Ld
    tar, lens, _ = _padded_case(rng, B, Ld, 4)
    blocks = ops.row_blocks(ops.target_live_len(_dev(tar, torch.int64), Ld), Ld, 32)

    class Prob(C.Structure):
        _fields_ = [("A", C.c_void_p), ("B", C.c_void_p), ("slab", C.c_void_p), ("slab_bytes", C.c_size_t), ("row_blocks", C.c_void_p),
                    ("M", C.c_int32), ("N", C.c_int32), ("K", C.c_int32), ("lda", C.c_int32), ("ldb", C.c_int32), ("splits", C.c_int32),
                    ("with_bias_grad", C.c_int32), ("row_block_rows", C.c_int32), ("splits_used", C.c_int32), ("pad", C.c_int32)]

    shapes = [(128, 384, False), (128, 128, True), (128, 512, False), (512, 128, False)]
    keep, probs, singles = [], (Prob * len(shapes))(), []
    stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    for i, (inf, outf, rows_list) in enumerate(shapes):
        x = _dev(rng.randn(R, inf)); dy = _dev(rng.randn(R, outf))
        sp = lib.skf_gemm_default_splits(inf, outf, R)
        nb = lib.skf_gemm_workspace_bytes(inf, outf, R, sp, 1)
        slab_g = torch.full((nb // 4,), 7.0, device="cuda"); slab_s = torch.full((nb // 4,), 9.0, device="cuda")
        keep += [x, dy, slab_g, slab_s]
        probs[i] = Prob(x.data_ptr(), dy.data_ptr(), slab_g.data_ptr(), nb, blocks.data_ptr() if rows_list else None, inf, outf, R, inf, outf, sp, 1,
                        32, 0, 0)
        used = C.c_int(0)
        _lib.call("skf_gemm_wgrad_partial_rows", inf, outf, R, C.c_void_p(x.data_ptr()), inf, C.c_void_p(dy.data_ptr()), outf, sp, 1,
                  C.c_void_p(slab_s.data_ptr()), nb, C.byref(used), 6, C.c_void_p(blocks.data_ptr()) if rows_list else None, 32, stream)
        singles.append((slab_s, slab_g, used.value, inf, outf))
    _lib.call("skf_gemm_wgrad_partial_group", C.byref(probs), len(shapes), 6, stream)
    torch.cuda.synchronize()
    for i, (slab_s, slab_g, used, inf, outf) in enumerate(singles):
        assert probs[i].splits_used == used
        n = used * (inf * outf + outf)
        assert torch.equal(slab_s[:n], slab_g[:n])
