"""Debug helper: cfg-2 dims at B=4, one forward_backward, per-tensor gradient error against the float64 oracle and the
location of the largest error.  python tools/dbg_fullsize.py [mode] [name]"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import oracle
import test_gpu_fullsize as T
mode = int(sys.argv[1]) if len(sys.argv) > 1 else 6
name = sys.argv[2] if len(sys.argv) > 2 else "cfg2"
eng, ocfg = T._build(name, 4, mode)
x, y, xo = T._batch(name, 4, ocfg, seed=3)
P = {k: v.astype(np.float64) for k, v in eng.state_dict_numpy().items()}
losses, out, G = oracle.loss_and_grads(P, ocfg, xo, xo, y)
for rep in range(3):
    eng.forward_backward(x, None, y)
    torch.cuda.synchronize()
    got = eng.state_dict_numpy("grads")
    floor = 1e-3 * np.median([np.abs(G[k]).max() for k in G])
    rel = {k: np.abs(got[k].astype(np.float64) - G[k]).max() / max(np.abs(G[k]).max(), floor) for k in G if not k.endswith("wk/bias")}
    bad = sorted(rel.items(), key=lambda kv: -kv[1])[:4]
    print("rep", rep, [(k, "%.2e" % v) for k, v in bad])
    k = bad[0][0]
    d = np.abs(got[k].astype(np.float64) - G[k])
    if d.ndim == 2:
        print("   ", k, "shape", d.shape, "argmax", np.unravel_index(d.argmax(), d.shape), "rows with err > 10% of max:", int((d.max(1) > 0.1 * d.max()).sum()),
              "cols:", int((d.max(0) > 0.1 * d.max()).sum()), "max|G|", np.abs(G[k]).max(), "max err", d.max())
        r, c = np.unravel_index(d.argmax(), d.shape)
        print("    got", got[k][r, c], "want", G[k][r, c], " col profile:", np.round(d[:, c][:8] / d.max(), 3), " row profile:", np.round(d[r, :][:8] / d.max(), 3))
