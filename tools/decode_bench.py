#!/usr/bin/env python
"""Greedy reconstruction (predict_from_embedding) at the cfg-2 dimensions: B = 128 samples x 200 positions.
    python tools/decode_bench.py [out.json]        # SKF_DECODE_FUSED=0 python ... = the layer-by-layer path of round 1"""
import json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sketchformer_amd import engine
from sketchformer_amd import synthetic

B, L, V = 128, 200, 1004
cfg = engine.make_config(batch=B, seq_len=L, d_model=128, num_heads=8, dff=512, num_layers=4, vocab_size=V, n_classes=345,
                         lowerdim=128, dropout_rate=0.0, use_graph=False, seed=1)
eng = engine.TrainEngine(cfg, init_seed=2)
x, _ = synthetic.token_batch(B, L, V, 345, seed=5)
eng.encode(x)
sos, eos = V - 2, V - 1
got = eng.greedy_decode(None, sos=sos, eos=eos)         # warm-up (graph capture / attribute calls)
ts = []
for _ in range(5):
    t0 = time.perf_counter()
    got = eng.greedy_decode(None, sos=sos, eos=eos)
    ts.append(time.perf_counter() - t0)
t = float(np.median(ts))
npos = got.shape[1] - 1
rec = {"what": "greedy reconstruction, cfg-2 dimensions (4L/8H/d128/dff512, V=1004), B=128, random weights (no EOS: all %d positions)" % npos,
       "path": "layer-by-layer (51 launches / position)" if os.environ.get("SKF_DECODE_FUSED") == "0" else "one launch / position",
       "positions": npos, "seconds_per_call": t, "ms_per_position": 1e3 * t / npos, "tokens_per_second": B * npos / t,
       "includes": "K/V projection of pre_decoder for all layers, host read-back of the result"}
print(json.dumps(rec))
if len(sys.argv) > 1:
    json.dump(rec, open(sys.argv[1], "w"), indent=1)
np.save("/tmp/decode_tokens_%s.npy" % ("unfused" if os.environ.get("SKF_DECODE_FUSED") == "0" else "fused"), got)
