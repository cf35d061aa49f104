import numpy as np, torch, sys
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
import test_gpu_model as T
from sketchformer_amd import synthetic
B = 4
for trial in range(4):
    res = []
    for bucketed in (False, True):
        eng, ocfg = T._mk(B, rate=0.1)
        eng.state[0] = 3000
        for step in range(3):
            x, y = synthetic.token_batch(B, ocfg.seq_len, ocfg.vocab_size, ocfg.n_classes, seed=50 + step)
            eng.forward_backward(x, None, y)
            eng.apply_gradients(bucketed=bucketed)
        torch.cuda.synchronize()
        res.append((eng.params.clone(), eng.adam_m.clone(), eng.adam_v.clone(), eng))
    for nm, a, b in zip(("params", "m", "v"), res[0][:3], res[1][:3]):
        diff = (a - b).abs()
        i = int(diff.argmax())
        ent = [e["name"] for e in res[0][3].entries if e["offset"] <= i < e["offset"] + e.get("size", 10**9)]
        name = None
        for e in res[0][3].entries:
            if e["offset"] <= i: name = e["name"]
        print(trial, nm, float(diff.max()), "at", i, name, float(a[i]), float(b[i]))
