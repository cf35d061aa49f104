"""Debug helper: rows of dy / dx of the LayerNorm-prologue backward that differ from float64 (listed / unlisted)."""
import sys
import numpy as np
import torch
sys.path.insert(0, ".")
import oracle
from sketchformer_amd import ops

rows, listed, rate = int(sys.argv[1]), int(sys.argv[2]), float(sys.argv[3])
d, dff = 128, 512
rng = np.random.RandomState(rows + 2)
x = rng.randn(rows, d); w1 = rng.randn(d, dff) / np.sqrt(d); b1 = 0.1 * rng.randn(dff)
w2 = rng.randn(dff, d) / np.sqrt(dff); b2 = 0.1 * rng.randn(d)
gamma, beta = 1 + 0.1 * rng.randn(d), 0.1 * rng.randn(d)
dout = rng.randn(rows, d)
dev = lambda a, t=torch.float32: torch.as_tensor(np.ascontiguousarray(a)).to(t).cuda()
blocks = None
if listed:
    Ld = 199; B = rows // Ld
    live = rng.randint(0, Ld + 1, size=B).astype(np.int32); live[0] = 0; live[-1] = Ld
    for b in range(B):
        dout[b * Ld + live[b]:(b + 1) * Ld] = 0.0
    blocks = ops.row_blocks(dev(live, torch.int32), Ld, 16)
st = ops.new_step_state("cuda", iterations=3); ops.step_prologue(st, seed=5)
X, W1, B1, W2, B2, G, Be = dev(x), dev(w1), dev(b1), dev(w2), dev(b2), dev(gamma), dev(beta)
img, = ops.ffn_weight_images([(W1, W2)], transpose=False); imgt, = ops.ffn_weight_images([(W1, W2)], transpose=True)
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
gdy, gdh, gdx, gdg, gdb = ops.ffn_fused_bwd_ln(dev(dout), zz, stats, G, imgt, bits, dff, rate=rate, site=9, state=st, row_blocks=blocks)
torch.cuda.synchronize()
for name, got, want in (("dy", gdy, dy), ("dh", gdh, dh), ("dx", gdx, dx)):
    e = np.abs(got.cpu().numpy() - want) / np.abs(want).max()
    br = np.where((e > 1e-4).any(1))[0]
    print(name, "max err", e.max(), "bad rows", len(br), br[:16], "tiles", np.unique(br // 16)[:12], "rows-in-tile", np.unique(br % 16))
    if len(br):
        r0 = br[0]; print("   row", r0, "got", got[r0, :4].cpu().numpy(), "want", want[r0, :4], "dout", dout[r0, :3])
print("dg err", np.abs(gdg.cpu().numpy() - dg).max() / np.abs(dg).max(), "db err", np.abs(gdb.cpu().numpy() - db).max() / np.abs(db).max())
