"""Debug helper: which rows / columns of the fused feed-forward launch differ from float64."""
import sys
import numpy as np
import torch
from sketchformer_amd import ops

rows = int(sys.argv[1]) if len(sys.argv) > 1 else 25600
d, dff = 128, 512
rng = np.random.RandomState(0)
x = rng.randn(rows, d); w1 = rng.randn(d, dff) / np.sqrt(d); b1 = 0.1 * rng.randn(dff)
w2 = rng.randn(dff, d) / np.sqrt(dff); b2 = 0.1 * rng.randn(d)
dev = lambda a: torch.as_tensor(a).float().cuda()
X, W1, B1, W2, B2 = dev(x), dev(w1), dev(b1), dev(w2), dev(b2)
g, be = dev(np.ones(d)), dev(np.zeros(d))
img, = ops.ffn_weight_images([(W1, W2)], transpose=False)
out, z, stats, h, bits = ops.ffn_fused_fwd(X, img, B1, B2, g, be, dff)
torch.cuda.synchronize()
hw = np.maximum(x @ w1 + b1, 0)
zw = x + hw @ w2 + b2
for name, got, want in (("h", h, hw), ("z", z, zw)):
    e = np.abs(got.cpu().numpy() - want) / np.abs(want).max()
    bad = e > 1e-4
    br = np.where(bad.any(1))[0]; bc = np.where(bad.any(0))[0]
    print(name, "max err", e.max(), "bad rows", len(br), br[:20], br[-5:] if len(br) else "", "bad cols", len(bc), bc[:20])
    if len(br):
        t = br // 16
        print("  bad tiles", len(np.unique(t)), np.unique(t)[:30], "rows-in-tile", np.unique(br % 16))
zz = z.cpu().numpy().astype(np.float64)
mean = zz.mean(-1); rstd = 1 / np.sqrt(zz.var(-1) + 1e-6)
st = stats.cpu().numpy()
for name, got, want in (("mean", st[:, 0], mean), ("rstd", st[:, 1], rstd), ("out", out.cpu().numpy(), (zz - mean[:, None]) * rstd[:, None])):
    e = np.abs(got - want) / np.abs(want).max()
    bad = e > 1e-4
    br = np.where(bad.reshape(rows, -1).any(1))[0]
    print(name, "max err", e.max(), "bad rows", len(br), br[:24], "rows-in-tile", np.unique(br % 16) if len(br) else "")
    if name == "mean" and len(br):
        r0 = br[0]
        print("   row", r0, "got", st[r0, 0], "want", mean[r0], "ratio", st[r0, 0] / mean[r0], "partial sums of 16-col groups / 128:", [zz[r0, 16 * k:16 * k + 16].sum() / 128 for k in range(8)])
