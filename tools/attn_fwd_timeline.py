"""wall_clock64 stamps (10-ns ticks) of a few workgroups of attn_fwd_kernel (-DSKF_MEASURE=1 build, SKF_ATTN_DBG): per wave
start | own share of K / V staged | staging complete | one stamp per query tile ..."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
dbg = torch.zeros(8 * 4 * 16, dtype=torch.int64, device="cuda")
os.environ["SKF_ATTN_DBG"] = str(dbg.data_ptr())
from sketchformer_amd import ops, synthetic
B, L, d, H = 128, 200, 128, 8
mode = sys.argv[1] if len(sys.argv) > 1 else "full"
qkv = torch.randn(B, L, 3 * d, device="cuda")
q, k, v = qkv[..., :d], qkv[..., d:2 * d], qkv[..., 2 * d:]
km = None
if mode == "bench":
    xs, _ = synthetic.token_batch(B, L, 1004, 345, seed=0)
    km = torch.from_numpy((xs == 0).astype("uint8")).cuda()
for _ in range(3):
    dbg.zero_()
    ops.attention_fwd(q, k, v, H, key_mask=km)
torch.cuda.synchronize()
d_ = dbg.view(8, 4, 16).cpu().numpy()
for wg in range(8):
    t0 = d_[wg, :, 0].min()
    for w in (0, 3):
        st_ = [int(x - t0) for x in d_[wg, w] if x != 0]
        print("wg%d wave%d start %5d | " % (wg, w, st_[0]) + " ".join("%5d" % (b_ - a_) for a_, b_ in zip(st_[:-1], st_[1:])) + " | end %d (x10 ns)" % st_[-1])
