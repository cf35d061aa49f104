"""Model-level parity of the bf16 path (BASELINE cfg 5: bf16 storage / MFMA, fp32 accumulate, fp32 master weights + Adam)
against the float64 oracle on identical fp32 parameters and inputs.  SURVEY 8(c) bar for this path: forward logits <= 2e-2
relative, token argmax agreement rate reported; here also losses, every gradient (per-tensor max-norm, 6e-2: the sum of
~2^-8 roundings of activations, weights images and upstream gradients) and a short Adam trajectory.  ReLU branches follow
the device where the oracle's pre-activation lies inside the bf16 noise band (tests/relu_branches.py)."""
import numpy as np
import pytest
import torch

import oracle
from relu_branches import device_relu_branches
from sketchformer_amd import synthetic

pytestmark = pytest.mark.gpu
import os

SMALL = dict(seq_len=40, d_model=128, num_heads=2, dff=256, num_layers=2, vocab_size=52, n_classes=7, lowerdim=32)
CFG5 = dict(seq_len=512, d_model=512, num_heads=8, dff=2048, num_layers=8, vocab_size=1004, n_classes=345, lowerdim=256)


def _build(kw, B, rate=0.0, use_graph=False, blind=True):
    from sketchformer_amd import engine
    cfg = engine.make_config(batch=B, dropout_rate=rate, use_graph=use_graph, seed=11, act_dtype="bf16", blind_decoder_mask=blind, **kw)
    eng = engine.TrainEngine(cfg, init_seed=1)
    ocfg = oracle.Config(dropout_rate=rate, blind_decoder_mask=blind, **kw)
    rng = np.random.RandomState(9)
    for e in eng.entries:
        n = e["name"]
        if n.endswith(("/bias", "/beta", "b_attn")):
            eng.set(n, rng.normal(0, 0.1, engine.logical_shape(e)))
        elif n.endswith("/gamma"):
            eng.set(n, 1 + rng.normal(0, 0.1, engine.logical_shape(e)))
    return eng, ocfg


def _rel(got, want):
    return np.abs(np.asarray(got, np.float64) - want).max() / max(np.abs(want).max(), 1e-30)


def _drops(eng, ocfg, B):
    from sketchformer_amd import ops
    key = ops.read_step_state(eng.state)["drop_key"]
    drops = {}
    for site, (name, tag) in enumerate(oracle.dropout_sites(ocfg)):
        L = ocfg.seq_len if tag == "enc" else ocfg.seq_len - 1
        drops[name] = ops.dropout_keep_mask(key, site, ocfg.dropout_rate, B * L * ocfg.d_model).reshape(B, L, ocfg.d_model)
    return drops


def test_bf16_config_is_validated():
    from sketchformer_amd import engine, _lib
    with pytest.raises(_lib.SkfError):         # head size 16: the streaming attention is built for 64
        engine.TrainEngine(engine.make_config(batch=2, act_dtype="bf16"))
    with pytest.raises(_lib.SkfError):
        engine.TrainEngine(engine.make_config(batch=2, act_dtype="bf16", continuous=True, vocab_size=None, **{k: v for k, v in SMALL.items() if k != "vocab_size"}))


@pytest.mark.parametrize("name,B,blind", [("small", 5, True), ("small", 5, False), ("cfg5", 2, True)])
def test_bf16_forward_logits_and_argmax(name, B, blind):
    kw = SMALL if name == "small" else CFG5
    eng, ocfg = _build(kw, B, blind=blind)
    x, y = synthetic.token_batch(B, ocfg.seq_len, ocfg.vocab_size, ocfg.n_classes, seed=2)
    x[0, ocfg.seq_len // 3:] = 0
    P = {k: v.astype(np.float64) for k, v in eng.state_dict_numpy().items()}
    out, _ = oracle.forward(P, ocfg, x, x[:, :-1], training=False)
    eng.forward(x, training=False)
    torch.cuda.synchronize()
    want = out["recon"]
    logits = eng.buffer("logits").float().cpu().numpy().reshape(want.shape)
    r = _rel(logits, want)
    r_emb = _rel(eng.buffer("embedding").cpu().numpy(), out["embedding"])
    r_cls = _rel(eng.buffer("class_probs").cpu().numpy(), out["class"])
    agree = (logits.argmax(-1) == want.argmax(-1)).mean()
    srt = np.sort(want, -1)
    safe = (srt[..., -1] - srt[..., -2]) > 4e-2 * np.abs(want).max()        # margin above the tolerated logit error
    agree_safe = (logits.argmax(-1) == want.argmax(-1))[safe].mean() if safe.any() else 1.0
    print("\n[bf16 %s blind=%s] logits rel %.3e  embedding rel %.3e  class probs rel %.3e  argmax agreement %.4f (%.4f of the %d "
          "positions with a top-2 margin above the logit tolerance)" % (name, blind, r, r_emb, r_cls, agree, agree_safe, safe.sum()))
    assert r < 2e-2 and r_emb < 2e-2 and r_cls < 2e-2          # bf16 cfg 5 logits <= 2e-2 rel
    assert agree_safe == 1.0 and agree > 0.9


# lengths: "bench" = the N(80, 35^2) lengths of SURVEY 8(d) as bench.py draws them (83 % padding at L = 512), "full" = every row n = L
@pytest.mark.parametrize("name,B,rate,lengths", [("small", 6, 0.0, "bench"), ("small", 6, 0.1, "bench"), ("cfg5", 2, 0.0, "bench"),
                                                 ("cfg5", 8, 0.0, "bench"), ("cfg5", 8, 0.0, "full")])
def test_bf16_losses_and_all_gradients(name, B, rate, lengths):
    kw = SMALL if name == "small" else CFG5
    eng, ocfg = _build(kw, B, rate=rate)
    x, y = synthetic.token_batch(B, ocfg.seq_len, ocfg.vocab_size, ocfg.n_classes, seed=3, full=lengths == "full")
    if lengths != "full":
        x[1, ocfg.seq_len // 4:] = 0
    P = {k: v.astype(np.float64) for k, v in eng.state_dict_numpy().items()}
    eng.forward_backward(x, None, y)
    torch.cuda.synchronize()
    drops = _drops(eng, ocfg, B) if rate > 0 else None
    n_units = B * (2 * ocfg.seq_len - 1) * ocfg.dff * ocfg.num_layers
    # the float64 oracle follows the device's ReLU branch only for units whose own pre-activation lies within the bf16 noise of zero:
    # 2^-7 of the largest pre-activation in the two-layer model; 2^-5 at the cfg-5 depth (measured round 4: a unit of decoder layer 5
    # differs at |pre| = 2.4e-2 - sixteen layers of 2^-9 roundings - and 2^-7 fails there)
    with device_relu_branches(eng, ocfg, B, kink=2.0 ** -7 if name == "small" else 2.0 ** -5, max_flips=n_units // 20) as chk:
        losses, out, G = oracle.loss_and_grads(P, ocfg, x, x, y, drops)
    m = eng.step_metrics()
    for k in ("recon_loss", "class_loss", "total_loss"):
        assert abs(m[k] - losses[k]) < 2e-2 * max(1.0, abs(losses[k])), (k, m[k], losses[k])
    got = eng.state_dict_numpy("grads")
    floor = 1e-2 * np.median([np.abs(G[k]).max() for k in G])
    rel = {k: np.abs(got[k].astype(np.float64) - G[k]).max() / max(np.abs(G[k]).max(), floor) for k in G if not k.endswith("wk/bias")}
    worst = max((v, k) for k, v in rel.items())
    print("\n[bf16 %s] floor %.3e; ten worst tensors (rel, max|G|, max err): %s" % (name, floor, [(k, "%.2e" % v, "%.2e" % np.abs(G[k]).max(), "%.2e" % np.abs(got[k] - G[k]).max()) for k, v in sorted(rel.items(), key=lambda kv: -kv[1])[:10]]))
    print("\n[bf16 %s rate %.1f] losses %s (oracle %s); worst gradient rel %.3e (%s), median %.3e over %d tensors; %d of %d ReLU "
          "units followed the device's branch" % (name, rate, {k: round(m[k], 4) for k in ("recon_loss", "class_loss")},
                                                  {k: round(float(losses[k]), 4) for k in ("recon_loss", "class_loss")}, worst[0],
                                                  worst[1], np.median(list(rel.values())), len(rel), chk.flips, n_units))
    # ONE bar per model size, whatever the batch and the padding (round 5: the per-case exception and the report-only switch are gone).
    # Against float64 the bars are the storage precision of the path (small model 1.7e-2, cfg-5 dimensions 5.1e-2 (B = 2) ... 7.6e-2
    # (B = 8, 83 % padding, encoder/layer7/mha/wq)); the error MODEL that accounts for them tensor by tensor sits below, at the comparison
    # with the bf16-storage restatement.
    # (round 6, ADVICE: the float64 bar is 6e-2 again for every case but the one whose measured distance is above it - cfg-5 dimensions,
    #  B = 8 at the bench's 83 % padding: 7.6e-2, profiles/r04_cfg5_parity_B8.txt; full-length 3.0e-2, B = 2 5.1e-2, small model 1.7e-2)
    padded_cfg5 = name == "cfg5" and lengths == "bench"
    assert worst[0] < (1e-1 if padded_cfg5 and B == 8 else 6e-2), worst
    assert np.median(list(rel.values())) < 1.5e-2
    assert np.isfinite(eng.grads.cpu().numpy()).all()
    # ---- the tight check: the same step restated with bf16 rounding at the product's storage points (oracle/bf16_storage.py:
    # activations, weight images, upstream gradients, attention operands).  What is left is fp32 accumulation order and the
    # values that sit on a rounding boundary; ReLU branches are the restatement's own except for units within 2^-7 (relative)
    # of zero, where the device's branch is taken - their number is printed.
    from oracle import bf16_storage
    dev_masks = {}
    for side, Ls in (("encoder", ocfg.seq_len), ("decoder", ocfg.seq_len - 1)):
        for i in range(ocfg.num_layers):
            hdev = eng.buffer("%s/layer%d/ffn_h" % (side, i)).float().cpu().numpy()
            dev_masks["%s/layer%d/ffn" % (side, i)] = (hdev > 0).reshape(B, Ls, ocfg.dff)
    st = {}
    l16, _, G16 = bf16_storage.loss_and_grads(P, ocfg, x, x, y, drops, relu_masks=dev_masks, stats=st)
    for k in ("recon_loss", "class_loss", "total_loss"):
        assert abs(m[k] - l16[k]) < 2e-3 * max(1.0, abs(l16[k])), (k, m[k], l16[k])
    floor16 = 1e-2 * np.median([np.abs(G16[k]).max() for k in G16])
    rel16 = {k: np.abs(got[k].astype(np.float64) - G16[k]).max() / max(np.abs(G16[k]).max(), floor16) for k in G16
             if not k.endswith("wk/bias")}
    worst16 = max((v, k) for k, v in rel16.items())
    print("\n[bf16 %s] against the bf16-storage restatement, five worst tensors: %s" % (name, [(k, "%.2e" % v) for k, v in sorted(rel16.items(), key=lambda kv: -kv[1])[:5]]))
    print("\n[bf16 %s rate %.1f] against the bf16-storage restatement: worst gradient rel %.3e (%s), median %.3e; %d of %d ReLU units "
          "within 2^-7 of zero took the device's branch" % (name, rate, worst16[0], worst16[1], np.median(list(rel16.values())),
                                                           st["relu_overrides"], st["relu_units"]))
    # Measured (profiles/r03e_bf16_storage_parity.txt, r03h): small model worst 1.3e-2 / 1.7e-2 (dropout), median 2.2e-3 / 3.1e-3;
    # cfg-5 dimensions (8 layers, L = 512) median 3.6e-3, 22 tensors at or above 1.5e-2 - query / key projections
    # of the upper attention layers (up to 6.7e-2), the bottleneck scorer and the expander - are sums that cancel to ~1e-5 of their
    # terms (dS = P o (dP - delta); column sums over every position), where the fp32 accumulation order of the device against
    # float64 here is amplified: the one thing this restatement does not model.  (Round 4, tools/bf16_delta_sensitivity.py, CPU only,
    # profiles/r04_cfg5_delta_sensitivity.txt: relative noise of 2^-24 - one fp32 rounding - on dP and delta alone moves tensors of this model by
    # up to 2.3e-2 ... 3.8e-2; the unrounded O instead of value + residual moves the worst by 2.6e-2.)  Bars: small model every tensor < 2.5e-2 and
    # median < 5e-3; cfg-5 dimensions median < 5e-3, worst < 1e-1, at least 90 % of the tensors below 1.5e-2.
    # Round 4, B = 8 (profiles/r04_cfg5_parity_B8.txt): with the bench's lengths worst 9.4e-2 (encoder/layer7/mha/wq), 20 of 323 tensors at or above
    # 1.5e-2, median 2.5e-3; the SAME model and batch size with full-length rows: worst 3.8e-2 (bottleneck/W_attn), 9 tensors, median 1.1e-3.  The
    # outliers are the query / key projections of the upper ENCODER layers, growing with depth (layer 3: 1.5e-2 ... layer 7: 9.4e-2), and the
    # bottleneck scorer: with 83 % of the rows PAD the encoder states of most positions are nearly identical (same token, same visible keys), so
    # dS = P o (dP - delta) and the pooling softmax gradient are differences of nearly equal terms there - the padding structure, not the batch
    # size, sets the amplification.  Full-length bars: worst < 5e-2, at most 4 % of the tensors at or above 1.5e-2.
    above = {k: round(float(v), 4) for k, v in rel16.items() if v >= 1.5e-2}
    print("[bf16 %s] %d of %d tensors at or above 1.5e-2: %s" % (name, len(above), len(rel16), above))
    assert np.median(list(rel16.values())) < 5e-3
    # ---- the bar per tensor is an ERROR MODEL (round 5, tools/bf16_rounding_noise.py, profiles/r05z_cfg5_rounding_noise.txt): the same
    # restatement evaluated once more with every value multiplied by 1 + 2^-24 N(0, 1) BEFORE it is rounded to bf16 - what another fp32
    # accumulation order does to a value about to be stored.  A value within that distance of a rounding boundary lands on the other side
    # (a 2^-9 step), and where a later difference cancels to ~1e-4 of its terms (dS = P o (dP - delta) of the upper encoder layers on a
    # padded batch: max|dQ| there is 1e-7 against 1e-3 for dV) the step is amplified.  That evaluation moves the SAME tensors by the SAME
    # amounts as the device is away from the plain restatement (cfg-5 dimensions, B = 8, 83 % padding: encoder/layer7/mha/wq 7.4e-2 ...
    # 9.2e-2 over three noise draws against the device's 9.4e-2; device / noise over the 50 tensors above 5e-3: median 0.81, max 1.14):
    # the distance IS the rounding-boundary noise of the storage scheme, not a term of the kernels.  On the padded cfg-5 cases every
    # tensor must stay below max(1.5e-2, 1.5 x its noise figure, maximum of three draws); a kernel that leaves the arithmetic moves
    # tensors the noise does not.
    # Round 6: the error model is the bar ONLY where it is needed - cfg-5 dimensions on a padded batch - and there it is
    # the maximum over three noise draws with a multiplier of 1.5 (a single draw x 3 let a regression in the high-cancellation tensors
    # pass with up to 20 % error); the small model and the full-length case keep the absolute bars of round 4.
    if name == "small":
        assert worst16[0] < 2.5e-2, worst16
    elif lengths == "full":
        assert worst16[0] < 5e-2 and len(above) <= len(rel16) // 25, (worst16, len(above), len(rel16))
    else:
        noise = {k: 0.0 for k in rel16}
        for draw in range(3):
            bf16_storage.NOISE = (np.random.default_rng(1 + draw), 2.0 ** -24)
            try:
                _, _, Gn = bf16_storage.loss_and_grads(P, ocfg, x, x, y, drops, relu_masks=dev_masks)
            finally:
                bf16_storage.NOISE = None
            for k in rel16:
                noise[k] = max(noise[k], np.abs(Gn[k] - G16[k]).max() / max(np.abs(G16[k]).max(), floor16))
        ratio = {k: rel16[k] / max(noise[k], 5e-3) for k in rel16}
        wk = max(ratio, key=ratio.get)
        print("[bf16 %s] rounding-noise model (max of 3 draws): worst noise %.3e (%s); device / max(noise, 5e-3): worst %.2f (%s: device "
              "%.3e, noise %.3e)" % (name, max(noise.values()), max(noise, key=noise.get), ratio[wk], wk, rel16[wk], noise[wk]))
        for k in rel16:
            assert rel16[k] < max(1.5e-2, 1.5 * noise[k]), (k, rel16[k], noise[k])
        assert worst16[0] < 1e-1 and len(above) <= len(rel16) // 10, (worst16, len(above), len(rel16))
    assert st["relu_overrides"] <= st["relu_units"] // 200


def test_bf16_adam_trajectory_and_graph_replay():
    """fp32 master weights + Keras-Adam under the bf16 step: 4 steps from iterations = 3000 against the oracle; the step
    replayed from a hipGraph equals the eagerly launched one bit for bit."""
    B = 4
    eng, ocfg = _build(SMALL, B)
    eng_g, _ = _build(SMALL, B, use_graph=True)
    P = {k: v.astype(np.float64) for k, v in eng.state_dict_numpy().items()}
    st = oracle.TrainState.create({k: v.copy() for k, v in P.items()})
    st.iterations = 3000
    eng.state[0] = 3000
    eng_g.state[0] = 3000
    for step in range(4):
        x, y = synthetic.token_batch(B, ocfg.seq_len, ocfg.vocab_size, ocfg.n_classes, seed=20 + step)
        eng.train_step(x, y)
        eng_g.train_step(x, y)
        torch.cuda.synchronize()
        n_units = B * (2 * ocfg.seq_len - 1) * ocfg.dff * ocfg.num_layers
        with device_relu_branches(eng, ocfg, B, kink=5e-2, max_flips=n_units // 20):
            res, losses, _, _ = oracle.train_step(st, ocfg, x, x, y)
        m = eng.step_metrics()
        assert abs(m["total_loss"] - losses["total_loss"]) < 2e-2 * abs(losses["total_loss"]), (step, m, losses)
    assert eng.iterations == 3004
    assert torch.equal(eng.params, eng_g.params) and torch.equal(eng.adam_v, eng_g.adam_v)
    got = eng.state_dict_numpy()
    # Keras-Adam normalises the gradient (update ~ lr * m / sqrt(v)): an element whose gradient is below the bf16 noise moves
    # by +-lr per step in a noise-determined direction, so single elements may differ by the whole distance travelled.  What
    # must agree is the update as a whole: relative L2 distance of the 4-step parameter displacement.
    keys = [k for k in got if not k.endswith("wk/bias")]
    du = np.concatenate([(got[k].astype(np.float64) - P[k]).reshape(-1) for k in keys])
    do = np.concatenate([(st.params[k] - P[k]).reshape(-1) for k in keys])
    rel_l2 = np.linalg.norm(du - do) / np.linalg.norm(do)
    cos = float(du @ do / (np.linalg.norm(du) * np.linalg.norm(do)))
    print("\n[bf16 trajectory] 4 Adam steps: |displacement| %.3e, relative L2 distance to the oracle's displacement %.3e, cosine %.5f"
          % (np.linalg.norm(do), rel_l2, cos))
    assert rel_l2 < 0.15 and cos > 0.99, (rel_l2, cos)


@pytest.mark.parametrize("blind", [True, False])
def test_bf16_model_greedy_decode_matches_oracle_from_its_own_embedding(blind):
    """predict_from_embedding on a bf16-trained model: the decoder runs in fp32 on the master weights (one launch per
    position), starting from the embedding of the bf16 encoder - token sequences identical to the oracle's decode of that
    same embedding (blind / expected-length cross mask, an early stop)."""
    B = 5
    eng, ocfg = _build(SMALL, B, blind=blind)
    sos, eos = ocfg.vocab_size - 2, ocfg.vocab_size - 1
    b = eng.get("output/bias"); b[eos] += 2.5; eng.set("output/bias", b)          # samples stop at different positions
    x, _ = synthetic.token_batch(B, ocfg.seq_len, ocfg.vocab_size, ocfg.n_classes, seed=4)
    eng.encode(x)
    emb = eng.buffer("embedding").float().cpu().numpy()
    P = {k: v.astype(np.float64) for k, v in eng.state_dict_numpy().items()}
    tlen = None if blind else np.sum(x > 0, axis=-1)
    want = oracle.predict_from_embedding(P, ocfg, emb.astype(np.float64), sos, eos, expected_len=tlen)["recon"]
    got = eng.greedy_decode(None, expected_len=tlen, sos=sos, eos=eos)
    assert got.shape == want.shape and np.array_equal(got, want)
    got2 = eng.greedy_decode(emb, expected_len=tlen, n_valid=B, sos=sos, eos=eos)      # explicit embedding: same answer
    assert np.array_equal(got2, want)
