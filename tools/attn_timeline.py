import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
dbg = torch.zeros(8 * 4 * 32, dtype=torch.int64, device="cuda")
os.environ["SKF_ATTN_DBG"] = str(dbg.data_ptr())
from sketchformer_amd import ops
B, L, d, H = 128, 200, 128, 8
qkv = torch.randn(B, L, 3 * d, device="cuda")
q, k, v = qkv[..., :d], qkv[..., d:2 * d], qkv[..., 2 * d:]
LEN = int(sys.argv[1]) if len(sys.argv) > 1 else 200
km = (torch.arange(L, device='cuda')[None, :] >= LEN).expand(B, L).to(torch.uint8).contiguous()
o, st = ops.attention_fwd(q, k, v, H, key_mask=km)
do = torch.randn(B, L, d, device="cuda")
for _ in range(3):
    ops.attention_bwd(q, k, v, o, do, st, H, key_mask=km)
torch.cuda.synchronize()
d_ = dbg.view(8, 4, 32).cpu().numpy()
for wg in range(3):
    t0 = d_[wg, :, 0].min()
    for w in range(4):
        r = d_[wg, w]
        st_ = [int(x - t0) for x in r if x != 0]
        print("wg%d wave%d start %5d | " % (wg, w, st_[0]) + " ".join("%5d" % (b_ - a_) for a_, b_ in zip(st_[:-1], st_[1:])) + " | end %d" % st_[-1])
