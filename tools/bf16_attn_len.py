import ctypes as C, json, os, sys
import numpy as np, torch
sys.path.insert(0, "/root/repo")
from sketchformer_amd import _lib
lib = _lib.load()
BF = torch.bfloat16
B, H, L, dh = 128, 8, 512, 64
d = H * dh
p = lambda t: None if t is None else C.c_void_p(t.data_ptr())
s = lambda: C.c_void_p(torch.cuda.current_stream().cuda_stream)
Q = torch.randn(B, L, d, device="cuda").to(BF); K = torch.randn(B, L, d, device="cuda").to(BF); V = torch.randn(B, L, d, device="cuda").to(BF)
dO = torch.randn(B, L, d, device="cuda").to(BF)
O = torch.empty_like(Q); Olo = torch.empty_like(Q); stats = torch.empty(B, H, L, 2, device="cuda")
ws = torch.empty(B * H * L, device="cuda"); dQ = torch.empty_like(Q); dK = torch.empty_like(Q); dV = torch.empty_like(Q)
for n in (512, 256, 128, 64, 8):
    mask = (torch.arange(L, device="cuda")[None, :] >= n).expand(B, L).to(torch.uint8).contiguous()
    def fwd():
        _lib.call("skf_attention_bf16_fwd", p(Q), d, p(K), d, p(V), d, p(mask), L, 0, B, H, L, L, dh, p(O), d, p(Olo), p(stats), s())
    def bwd():
        _lib.call("skf_attention_bf16_bwd", p(Q), d, p(K), d, p(V), d, p(O), d, p(Olo), p(dO), d, p(stats), p(mask), L, 0,
                  B, H, L, L, dh, p(dQ), d, p(dK), d, p(dV), d, p(ws), ws.numel() * 4, s())
    for _ in range(3): fwd(); bwd()
    torch.cuda.synchronize(); lib.skf_profiler_enable(1)
    for _ in range(20): fwd(); bwd()
    torch.cuda.synchronize()
    buf = C.create_string_buffer(1 << 16); lib.skf_profiler_report(buf, len(buf)); lib.skf_profiler_enable(0)
    print("keys %3d of 512:" % n, "  ".join("%s %.1f us" % (r["tag"].replace("attn_bf16_", "").replace("<dh64>", ""), r["ms"] / r["count"] * 1e3) for r in json.loads(buf.value.decode())), flush=True)
