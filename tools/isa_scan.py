#!/usr/bin/env python
"""Scan the gfx950 ISA of every HIP source for the pattern that cost this code base three kernels' worth of time in round 3:
a global / buffer load whose result the compiler needs at once - closed with `s_waitcnt vmcnt(0)` (and usually `v_readfirstlane`) -
inside a loop.  Such a wait also drains every load that was issued ahead on purpose (tile prefetch, DMA).

usage: python tools/isa_scan.py [file.hip ...]      (default: every sketchformer_amd/csrc/*.hip; needs hipcc, no GPU)
Prints, per kernel, the loops that contain `load ; ... ; s_waitcnt vmcnt(0)` within a few instructions, and the uniform-load form
(`global_load` -> `vmcnt(0)` -> `v_readfirstlane`) anywhere."""
import glob, os, re, subprocess, sys, tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-munsafe-fp-atomics", "-mllvm", "-amdgpu-mfma-vgpr-form=1", "-S", "--cuda-device-only", "-w"]


def scan(path):
    with tempfile.TemporaryDirectory() as tmp:
        out = os.path.join(tmp, "k.s")
        hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
        if subprocess.run([hipcc] + FLAGS + [path, "-o", out], capture_output=True).returncode != 0:
            print("%s: did not compile" % path)
            return
        text = open(out).read()
    for m in re.finditer(r"^(_Z\w+):[^\n]*\n(.*?)^\.Lfunc_end", text, re.M | re.S):
        name, body = m.group(1), [l.strip() for l in m.group(2).split("\n")]
        loops, stack = [], []
        in_loop = [False] * len(body)
        # a loop = from a "Loop Header" label to the last backward branch to it
        labels = {l[:-1]: k for k, l in enumerate(body) if l.endswith(":") and l.startswith(".LBB")}
        for k, l in enumerate(body):
            b = re.match(r"s_cbranch_\w+ (\.LBB\w+)|s_branch (\.LBB\w+)", l)
            if b:
                t = labels.get(b.group(1) or b.group(2))
                if t is not None and t < k:
                    for q in range(t, k + 1):
                        in_loop[q] = True
        hits = []
        for k, l in enumerate(body):
            if not re.match(r"(global|buffer)_load", l):
                continue
            window = body[k + 1:k + 7]
            w0 = next((d for d, n in enumerate(window) if "s_waitcnt vmcnt(0)" in n), None)
            if w0 is None:
                continue
            uniform = any("v_readfirstlane" in n for n in body[k + 1:k + 9])
            if in_loop[k] or uniform:
                hits.append((k, "loop" if in_loop[k] else "once", "uniform" if uniform else "vector", l[:70]))
        if hits:
            short = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip() or name
            print("%s :: %s" % (os.path.basename(path), short[:150]))
            for h in hits:
                print("    line %5d  %-4s %-7s %s" % h)


if __name__ == "__main__":
    files = sys.argv[1:] or sorted(glob.glob(os.path.join(ROOT, "sketchformer_amd", "csrc", "*.hip")))
    for f in files:
        scan(f)
