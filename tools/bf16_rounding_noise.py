#!/usr/bin/env python
"""Does fp32-sized noise in front of the bf16 storage roundings explain the distance between the device and the bf16-storage restatement
(cfg-5 dimensions, B = 8, 83 % padding: encoder/layer7/mha/wq 9.4e-2)?  The restatement is evaluated once as it is and twice with every value
multiplied by 1 + rel * N(0, 1) before it is rounded (oracle/bf16_storage.NOISE); per tensor: |noisy - plain|_max / max(|plain|_max, floor),
next to the device's distance from the plain restatement."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from oracle import bf16_storage  # noqa: E402
from sketchformer_amd import synthetic  # noqa: E402
import test_gpu_bf16_model as T  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
full = len(sys.argv) > 2 and sys.argv[2] == "full"
eng, ocfg = T._build(T.CFG5, B)
x, y = synthetic.token_batch(B, ocfg.seq_len, ocfg.vocab_size, ocfg.n_classes, seed=3, full=full)
if not full:
    x[1, ocfg.seq_len // 4:] = 0
P = {k: v.astype(np.float64) for k, v in eng.state_dict_numpy().items()}
eng.forward_backward(x, None, y)
torch.cuda.synchronize()
got = eng.state_dict_numpy("grads")
dev_masks = {}
for side, Ls in (("encoder", ocfg.seq_len), ("decoder", ocfg.seq_len - 1)):
    for i in range(ocfg.num_layers):
        dev_masks["%s/layer%d/ffn" % (side, i)] = (eng.buffer("%s/layer%d/ffn_h" % (side, i)).float().cpu().numpy() > 0).reshape(B, Ls, ocfg.dff)
_, _, G0 = bf16_storage.loss_and_grads(P, ocfg, x, x, y, None, relu_masks=dev_masks)
floor = 1e-2 * np.median([np.abs(G0[k]).max() for k in G0])
keys = [k for k in G0 if not k.endswith("wk/bias")]
rel = lambda A, k: np.abs(np.asarray(A[k], np.float64) - G0[k]).max() / max(np.abs(G0[k]).max(), floor)  # noqa: E731
dev = {k: rel(got, k) for k in keys}
rows = {}
for tag, rl, seed in (("2^-24 a", 2.0 ** -24, 1), ("2^-24 b", 2.0 ** -24, 2), ("2^-22", 2.0 ** -22, 3)):
    bf16_storage.NOISE = (np.random.default_rng(seed), rl)
    _, _, Gn = bf16_storage.loss_and_grads(P, ocfg, x, x, y, None, relu_masks=dev_masks)
    bf16_storage.NOISE = None
    rows[tag] = {k: rel(Gn, k) for k in keys}
    print("noise %-8s worst %.2e (%s)  median %.2e  tensors >= 1.5e-2: %d of %d" % (tag, max(rows[tag].values()), max(rows[tag], key=rows[tag].get),
                                                                                  np.median(list(rows[tag].values())), sum(v >= 1.5e-2 for v in rows[tag].values()), len(keys)), flush=True)
print("device           worst %.2e (%s)  median %.2e  tensors >= 1.5e-2: %d of %d" % (max(dev.values()), max(dev, key=dev.get), np.median(list(dev.values())),
                                                                                   sum(v >= 1.5e-2 for v in dev.values()), len(keys)))
print("the fifteen tensors the device is furthest from the restatement on: device | noise 2^-24 a | b | 2^-22 | max|G|/floor")
for k in sorted(keys, key=lambda k: -dev[k])[:15]:
    print("  %-44s %.2e | %.2e | %.2e | %.2e | %.1f" % (k, dev[k], rows["2^-24 a"][k], rows["2^-24 b"][k], rows["2^-22"][k], np.abs(G0[k]).max() / floor))
ratio = [dev[k] / max(rows["2^-24 a"][k], rows["2^-24 b"][k], 1e-12) for k in keys if dev[k] >= 5e-3]
print("device / max(noise a, b) over the %d tensors with device >= 5e-3: median %.2f  max %.2f" % (len(ratio), np.median(ratio), max(ratio)))
