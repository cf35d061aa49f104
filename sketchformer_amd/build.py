"""Builds libskf.so (the C-ABI HIP library) in-tree with hipcc for gfx950."""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libskf.so")
SOURCES = ["skf_model.hip", "skf_gemm.hip", "skf_gemm_ws.hip", "skf_gemm_small.hip", "skf_gemm_wsx.hip", "skf_gemm_wgrad.hip", "skf_ffn_fused.hip", "skf_attention.hip", "skf_attention_bwd2.hip", "skf_attention_bwd3.hip", "skf_rowops.hip", "skf_continuous.hip", "skf_optimizer.hip", "skf_decode.hip", "skf_decode_fused.hip", "skf_row_blocks.hip", "skf_generic.hip", "skf_bf16_gemm.hip", "skf_bf16_attention.hip", "skf_bf16_rowops.hip"]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-munsafe-fp-atomics",
         # f32-input MFMA shares the VALU pipe on gfx950 (tools/micro/mfma_valu_overlap.hip): keep accumulators in VGPRs so
         # the epilogues need no v_accvgpr_read/write moves (they were ~30 % of the VALU instructions of the attention loops)
         "-mllvm", "-amdgpu-mfma-vgpr-form=1", "-Wall", "-Wno-unused-function", "-Wno-unused-value"]
FLAGS += os.environ.get("SKF_EXTRA_HIPCC_FLAGS", "").split()   # e.g. -DSKF_WS_STAMPS=1 for tools/ws_timeline.py


def _hipcc():
    for cand in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if cand and (os.path.isabs(cand) and os.path.exists(cand) or not os.path.isabs(cand)):
            return cand
    return "hipcc"


STAMP = LIB + ".stamp"


def _source_digest():
    """sha256 over every file the library is built from + the compiler flags (file times do not survive a snapshot copy)."""
    import hashlib
    h = hashlib.sha256(" ".join(FLAGS).encode())
    files = sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC)) + [os.path.join(HERE, "..", "include", "skf.h")]
    for path in files:
        h.update(os.path.basename(path).encode())
        with open(path, "rb") as f:
            h.update(f.read())
    return h.hexdigest()


def needs_build():
    if not os.path.exists(LIB) or not os.path.exists(STAMP):
        return True
    with open(STAMP) as f:
        return f.read().strip() != _source_digest()


def build_library(force=False, verbose=True):
    """Compile every HIP source for gfx950 and link libskf.so next to this file."""
    if not force and not needs_build():
        return LIB
    objs = []
    procs = []
    os.makedirs(os.path.join(HERE, "build"), exist_ok=True)
    for src in SOURCES:
        obj = os.path.join(HERE, "build", src.replace(".hip", ".o"))
        cmd = [_hipcc()] + FLAGS + ["-c", os.path.join(CSRC, src), "-o", obj]
        if verbose:
            print(" ".join(cmd), flush=True)
        procs.append((src, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)))
        objs.append(obj)
    failed = False
    for src, p in procs:
        out, _ = p.communicate()
        if out and verbose:
            sys.stdout.write(out.decode(errors="replace"))
        if p.returncode != 0:
            failed = True
            sys.stderr.write("hipcc failed on %s\n%s\n" % (src, out.decode(errors="replace")))
    if failed:
        raise RuntimeError("libskf.so build failed")
    cmd = [_hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    with open(STAMP, "w") as f:
        f.write(_source_digest())
    return LIB


if __name__ == "__main__":
    build_library(force="--force" in sys.argv)
    print(LIB)
