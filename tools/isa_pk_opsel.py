#!/usr/bin/env python
"""List packed-fp32 VALU instructions with CROSSED operand selects (op_sel / op_sel_hi other than the default [0,0] / [1,1]) in the
gfx950 ISA of every HIP source.  Round 4 bisected the run-to-run wrong results of one gemm_wsx instantiation (accumulate-only epilogue,
K = 256, one column per lane) to the compiler's `v_pk_add_f32 ... op_sel:[0,1] op_sel_hi:[1,0]` in the kernel's exit block (DESIGN section 6,
round 4): this scan shows where else the compiler forms such instructions.

usage: python tools/isa_pk_opsel.py [extra hipcc flags ...]     (needs hipcc, no GPU)"""
import glob, os, re, subprocess, sys, tempfile
from concurrent.futures import ThreadPoolExecutor

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-munsafe-fp-atomics", "-mllvm", "-amdgpu-mfma-vgpr-form=1", "-S", "--cuda-device-only", "-w"]


def scan(path, extra):
    with tempfile.TemporaryDirectory() as tmp:
        out = os.path.join(tmp, "k.s")
        r = subprocess.run([os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")] + FLAGS + extra + [path, "-o", out], capture_output=True, text=True)
        if r.returncode != 0:
            return path, None
        text = open(out).read()
    hits = {}
    for m in re.finditer(r"^(_Z\w+):[^\n]*\n(.*?)^\.Lfunc_end", text, re.M | re.S):
        n = 0
        for l in m.group(2).split("\n"):
            s = re.search(r"(v_pk_\w+_f32) .*op_sel:\[(\d),(\d)(?:,\d)?\](?: op_sel_hi:\[(\d),(\d)(?:,\d)?\])?", l)
            if s and (s.group(2), s.group(3)) != ("0", "0"):
                n += 1
        if n:
            hits[m.group(1)] = n
    return path, hits


if __name__ == "__main__":
    extra = sys.argv[1:]
    files = sorted(glob.glob(os.path.join(ROOT, "sketchformer_amd", "csrc", "*.hip")))
    total = 0
    with ThreadPoolExecutor(6) as ex:
        for path, hits in ex.map(lambda f: scan(f, extra), files):
            if hits is None:
                print("%s: did not compile" % os.path.basename(path)); continue
            n = sum(hits.values()); total += n
            print("%-28s %4d crossed-select packed-fp32 instructions in %d kernels" % (os.path.basename(path), n, len(hits)))
            for k, v in sorted(hits.items(), key=lambda kv: -kv[1])[:4]:
                short = subprocess.run(["c++filt", k], capture_output=True, text=True).stdout.strip()[:110]
                print("      %3d  %s" % (v, short))
    print("total", total)
