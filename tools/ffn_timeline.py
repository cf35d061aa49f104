#!/usr/bin/env python
"""Phase stamps of the fused feed-forward launch (library built with -DSKF_FFN_STAMPS=1: tools/ffn_variants.sh stamps).
Prints, per stamped wave, the cycles between consecutive stamps: start, [rows staged, barrier, 4 x (stage 1, barrier, stage 2), epilogue] per sub-group."""
import ctypes as C
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sketchformer_amd import ops, _lib  # noqa: E402

M, d, F = int(sys.argv[1]) if len(sys.argv) > 1 else 25600, 128, 512
mode = sys.argv[2] if len(sys.argv) > 2 else "fwd"
dev = "cuda"
r = lambda *s: torch.randn(*s, device=dev)  # noqa: E731
x, w1, b1, w2, b2, g, be = r(M, d), r(d, F) / 11, r(F), r(F, d) / 22, r(d), r(d), r(d)
img, = ops.ffn_weight_images([(w1, w2)], transpose=False)
imgt, = ops.ffn_weight_images([(w1, w2)], transpose=True)
proj = None
if mode == "proj":
    wp, bp = r(d, 384) / 11, r(384)
    proj = (ops.dense_weight_image(wp, transpose=False), bp)
for _ in range(3):
    out, z, stats, h, bits = ops.ffn_fused_fwd(x, img, b1, b2, g, be, F, proj=proj)[:5]
    if mode == "bwd":
        ops.ffn_fused_bwd(r(M, d), imgt, bits, F, dx=r(M, d))
torch.cuda.synchronize()
lib = C.CDLL(_lib.LIB_PATH)
buf = (C.c_longlong * (16 * 64))()
assert lib.skf_ffn_debug_stamps(buf) == 0
a = np.frombuffer(buf, dtype=np.int64).reshape(16, 64)
names = ["staged", "barrier"] + sum([["s1.%d" % b, "bar.%d" % b, "s2.%d" % b] for b in range(4)], []) + (["rowepi", "bar", "post0", "post1", "post2", "end"] if mode == "proj" else ["epilogue"])
for w in range(16):
    t = a[w]
    n = int((t != 0).sum())
    if n < 2:
        continue
    dlt = np.diff(t[:n])
    print("wg %3d wave %d: total %6d cycles" % ((w // 2) * 32, 7 * (w % 2), t[n - 1] - t[0]))
    per = len(names)
    for sg in range((n - 1) // per):
        seg = dlt[sg * per:(sg + 1) * per]
        print("   sub-group %d: " % sg + " ".join("%s %d" % (nm, v) for nm, v in zip(names, seg)))
