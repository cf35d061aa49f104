import sys, torch, numpy as np
sys.path.insert(0, "/root/repo")
from sketchformer_amd import ops
B, L, d, H = 128, 200, 128, 8
qkv = torch.randn(B, L, 3 * d, device="cuda")
q, k, v = qkv[..., :d], qkv[..., d:2 * d], qkv[..., 2 * d:]
def t(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
for ln in (200, 100, 40, 16):
    km = (torch.arange(L, device="cuda")[None, :] >= ln).expand(B, L).to(torch.uint8).contiguous()
    o, st = ops.attention_fwd(q, k, v, H, key_mask=km)
    do = torch.randn(B, L, d, device="cuda")
    print("len", ln, "fwd %.1f us" % t(lambda: ops.attention_fwd(q, k, v, H, key_mask=km)),
          "bwd %.1f us" % t(lambda: ops.attention_bwd(q, k, v, o, do, st, H, key_mask=km)))
km = torch.ones(2, 40, dtype=torch.uint8, device="cuda")
x = torch.randn(2, 40, 3 * 32, device="cuda")
o, st = ops.attention_fwd(x[..., :32], x[..., 32:64], x[..., 64:], 2, key_mask=km)
print("fwd finite", torch.isfinite(o).all().item(), st[0, 0, :3])
dq, dk, dv = ops.attention_bwd(x[..., :32], x[..., 32:64], x[..., 64:], o, torch.randn(2, 40, 32, device="cuda"), st, 2, key_mask=km)
print("bwd finite", torch.isfinite(dq).all().item(), torch.isfinite(dk).all().item(), torch.isfinite(dv).all().item())
print(dq[0, :3, :4])
