#!/usr/bin/env python
"""Headline benchmark: stroke-tokens/sec of the sketch-transformer-tf2 train step
(forward + backward + Keras-Adam/WarmupDecay) on synthetic QuickDraw-shaped batches.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Workload (BASELINE.json configs[1] / cfg 2): 4 layers, 8 heads, d_model 128, dff 512,
L=200, per-GPU B=128, V=1004, 345 classes, dropout 0.1, fp32.  Weak scaling: every rank
runs B=128 (global batch 128*N); one RCCL all-reduce of the flat fp32 gradient buffer
per step.  Inputs are resident in HBM before the timed region.  Prints ONE JSON line.

The line carries, beside the contract keys: `roofline` (dominant kernel by time, HIP events on the launch stream, HBM
traffic from two rocprofv3 PMC passes of this same script), `cpu_baseline`, and - single GPU only, none of them part of
`value` - `full_length` (no padding), `fp32_mfma_mode`, `plugin_path` (the step through train_on_batch), `kernels`
(serialised per-kernel timing) next to `kernels_concurrent` (rocprofv3 kernel trace of the real two-stream step), `cfg3` and
`cfg5` sub-records (BASELINE configs[2] and configs[4] per GPU).  `--workload cfg3|cfg5` makes one of those the main line.
"""
import argparse
import csv
import ctypes as C
import glob
import json
import os
import re
import shutil
import subprocess
import sys
import tempfile
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_F32_MFMA_TFLOPS = 157.3   # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32 / 16x16x4_f32 dense peak
PEAK_BF16_MFMA_TFLOPS = 2500.0  # MI355X_MICROARCH.md: dense bf16 MFMA peak (v_mfma_f32_16x16x32_bf16 / 32x32x16)
PEAK_HBM_GBS = 8000.0

WORKLOADS = {
    # name: (L, d, dff, N, V, continuous, act dtype, description)
    "cfg2": dict(L=200, d=128, dff=512, N=4, V=1004, cont=False, act="f32",
                 text="cfg2: sketch-transformer-tf2 4L/8H/d128/dff512 L=200 V=1004 C=345 dropout=0.1, fwd+bwd+Adam(WarmupDecay)"),
    "cfg2grid": dict(L=200, d=128, dff=512, N=4, V=10004, cont=False, act="f32",
                     text="cfg2grid: cfg2 with the grid tokenizer's vocabulary (V=10004, utils/tokenizer.py:104-198), F_step = 0.576 TFLOP"),
    "cfg3": dict(L=200, d=256, dff=1024, N=6, V=5, cont=True, act="f32",
                 text="cfg3: sketch-transformer-tf2 6L/8H/d256/dff1024 L=200 continuous stroke-5 C=345 dropout=0.1, fwd+bwd+Adam(WarmupDecay)"),
    "cfg5": dict(L=512, d=512, dff=2048, N=8, V=1004, cont=False, act="bf16",
                 text="cfg5: sketch-transformer-tf2 8L/8H/d512/dff2048 L=512 V=1004 C=345 dropout=0.1, bf16 storage/MFMA, fp32 master "
                      "weights + Adam(WarmupDecay), per-GPU B=128 (BASELINE does not state B)"),
}
U, CN = 256, 345


_T0 = time.perf_counter()


def _progress(msg):
    """phase log on stderr (the one JSON line goes to stdout): where a slow run spends its time"""
    print("[bench %7.1f s] %s" % (time.perf_counter() - _T0, msg), file=sys.stderr, flush=True)


def _elapsed():
    return time.perf_counter() - _T0


def step_flops(B, L, d, dff, N, V, U, Cn, cont=False):
    """Algorithmic FLOPs of one train step, SURVEY.md section 8(d): F_step = 3 * F_fwd."""
    Le, Ld, Lk = L, L - 1, L
    f_enc = N * (8 * B * Le * d * d + 4 * B * Le * Le * d + 4 * B * Le * d * dff)
    f_dec = N * (8 * B * Ld * d * d + 4 * B * Ld * Ld * d + 4 * B * Ld * d * d + 4 * B * Lk * d * d
                 + 4 * B * Ld * Lk * d + 4 * B * Ld * d * dff)
    f_out = 2 * B * Ld * d * V
    f_emb = 2 * B * (Le + Ld) * 5 * d if cont else 0
    f_bott = 2 * B * Le * d * U + 2 * B * Le * U + 2 * B * Le * d
    f_cls = 2 * B * d * Cn + 2 * B * d * L
    return 3 * (f_enc + f_dec + f_out + f_emb + f_bott + f_cls)


def step_bytes(B, L, d, dff, N, V, P, act_bytes):
    """Algorithmic HBM bytes of one train step, SURVEY.md section 8(d): every saved activation written once forward and read
    once backward (flash-style attention: no L x L tensors), weights read twice, optimizer 28 B / parameter."""
    Le, Ld, Lk = L, L - 1, L
    A = B * Le * N * (7 * d + dff) + B * Ld * N * (11 * d + dff) + B * Lk * N * 2 * d + B * Ld * V
    return 2 * A * act_bytes + 2 * P * act_bytes + 28 * P


def cpu_baseline(seconds_budget=20.0):
    """The CPU restatement of the TF2 reference timed on this host at SURVEY 8(d)'s definition (TensorFlow itself cannot run here):
    the PyTorch-CPU eager restatement of the identical graph (oracle/torch_restatement.py: torch.nn.functional forward, autograd
    backward, Keras-Adam/WarmupDecay) in fp32, cfg 1 (4L/8H/d128/dff512, L=200, V=1004, C=1), the full B=128 batch with dropout
    0.1, at the fastest thread count of a short ladder (see below), >= 3 timed steps (bounded to ~20 s).  The numpy oracle (the
    parity checker) is timed beside it as `numpy_port` (one warm-up step, >= 1 timed step)."""
    import oracle
    from oracle import torch_restatement as tr
    from sketchformer_amd import synthetic
    B, L = 128, 200
    cfg = oracle.Config(n_classes=1)
    x, y = synthetic.token_batch(B, L, cfg.vocab_size, 1, seed=0)
    ncpu = os.cpu_count() or 1
    old_threads = torch.get_num_threads()
    try:
        state = tr.TorchTrainState(oracle.init_params(cfg, 0, np.float32))
        gen = torch.Generator().manual_seed(0)
        drops = {n: torch.rand(B, L if t == "enc" else L - 1, cfg.d_model, generator=gen) >= cfg.dropout_rate
                 for n, t in oracle.dropout_sites(cfg)}
        # Thread count: SURVEY 8(d) asks for torch.set_num_threads(os.cpu_count()), but eager PyTorch with one thread per logical
        # core of a 256-thread host is pathological (measured on the MI355X box, profiles/r03b_bench.json: 129 s per step at 256
        # threads - slower than the single-process numpy oracle).  The baseline is the BEST of a short ladder of thread counts up
        # to 64 (one untimed + one timed step each, ascending, abandoned as soon as a count is clearly slower than the best so
        # far); the ladder and the choice are reported.
        ladder, trials = [n for n in (8, 16, 32, 64) if n <= ncpu] or [ncpu], []
        best = None
        for n in ladder:
            torch.set_num_threads(n)
            tr.train_step(state, cfg, x, x, y, drops)
            t0 = time.perf_counter()
            tr.train_step(state, cfg, x, x, y, drops)
            dt = time.perf_counter() - t0
            trials.append((n, round(dt, 3)))
            if best is None or dt < best[1]:
                best = (n, dt)
            elif dt > 1.25 * best[1]:
                break
        torch.set_num_threads(best[0])
        _progress("cpu baseline: thread ladder %s -> %d threads" % (trials, best[0]))
        t_start = time.perf_counter()
        times = []
        while len(times) < 3 or (len(times) < 10 and time.perf_counter() - t_start < seconds_budget):
            t0 = time.perf_counter()
            tr.train_step(state, cfg, x, x, y, drops)
            times.append(time.perf_counter() - t0)
        med = float(np.median(times))
        threads = torch.get_num_threads()
        _progress("cpu baseline: torch %.2f s/step on %d threads" % (med, threads))
    finally:
        torch.set_num_threads(old_threads)
    out = {"value": B * L / med, "unit": "stroke-tokens/sec", "cores": int(threads), "kind": "port",
           "sample": "PyTorch-CPU fp32 restatement of the TF2 reference (eager forward, autograd backward, Keras-Adam/WarmupDecay), cfg1 "
                     "(4L/8H/d128/dff512, L=200, V=1004, C=1), B=128, dropout 0.1, median of %d full train steps after the warm-up ladder "
                     "(%.2f s each), torch.set_num_threads(%d) = the fastest of the ladder %s (threads, s/step) on %d logical cores "
                     "(one thread per logical core is pathological for eager PyTorch on this host)" % (len(times), med, int(threads), trials, ncpu)}
    out["sample_short"] = ("torch-CPU fp32 port of the TF2 step, cfg1 B=128 L=200 dropout 0.1, median of %d steps (%.2f s each), %d threads "
                           "(best of ladder) on %d logical cores" % (len(times), med, int(threads), ncpu))
    # the numpy oracle (what the parity tests check against), for continuity with rounds 1-2
    nstate = oracle.TrainState.create(oracle.init_params(cfg, 0, np.float32))
    rng = np.random.RandomState(0)
    ndrops = {n: rng.rand(B, L if t == "enc" else L - 1, cfg.d_model) >= cfg.dropout_rate for n, t in oracle.dropout_sites(cfg)}
    t0 = time.perf_counter()
    oracle.train_step(nstate, cfg, x, x, y, ndrops)
    warm = time.perf_counter() - t0
    ntimes = []
    while warm < 10.0 and (len(ntimes) < 1 or (len(ntimes) < 3 and (len(ntimes) + 2) * warm < 12.0)):
        t0 = time.perf_counter()
        oracle.train_step(nstate, cfg, x, x, y, ndrops)
        ntimes.append(time.perf_counter() - t0)
    first_only = not ntimes
    if first_only:          # a slow host: the first (cold) step is the sample
        ntimes = [warm]
    try:
        from threadpoolctl import threadpool_info
        ncores = max([p.get("num_threads", 1) for p in threadpool_info()] or [1])
    except Exception:
        ncores = ncpu
    out["numpy_port"] = {"value": B * L / float(np.median(ntimes)), "unit": "stroke-tokens/sec", "cores": int(ncores),
                         "sample": "numpy float32 oracle, same workload, " + ("the first step (no warm-up: > 10 s per step on this host)" if first_only
                                                                                     else "median of %d steps after 1 warm-up" % len(ntimes))}
    return out


def _kernel_prefix(tag):
    """profiler tag -> prefix of the demangled kernel name in rocprofv3 output"""
    m = re.match(r"gemm_ws(x?)<K(\d+),CW(\d+)", tag)
    if m:   # kernel template is <K, columns per lane = CW/16, ...>
        return "gemm_ws%s_kernel<%s, %d," % (m.group(1), m.group(2), int(m.group(3)) // 16)
    m = re.match(r"ffn_fused<d128,dff512,bf16x(\d)>", tag)
    if m:   # ffn_fused_kernel<P, MODE, LNB, POST>: the family = every variant of one piece count
        return "ffn_fused_kernel<%d," % (3 if m.group(1) == "6" else 2)
    m = re.match(r"ln_bwd_dgrad<d128,bf16x(\d)>", tag)
    if m:
        return "ln_bwd_dgrad_kernel<%d>" % (3 if m.group(1) == "6" else 2)
    table = {"wgrad<64x64>": "wgrad_kernel", "wgrad<64x64,bf16x6>": "wgrad_x_kernel<3", "wgrad<64x64,bf16x3>": "wgrad_x_kernel<2",
             "attn_bwd<dh16>": "attn_bwd_kernel<16", "attn_fwd<dh16>": "attn_fwd_kernel<16", "attn_bwd<dh32>": "attn_bwd_kernel<32",
             "attn_fwd<dh32>": "attn_fwd_kernel<32", "ln_fwd": "ln_fwd", "ln_bwd": "ln_bwd", "gemm_bf16_nt": "gemm_bf16_nt_kernel",
             "gemm_bf16_tn(wgrad)": "gemm_bf16_tn_kernel", "attn_bf16_fwd<dh64>": "attn_bf16_q_kernel<0",
             "attn_bf16_bwd_dq<dh64>": "attn_bf16_q_kernel<1", "attn_bf16_bwd_dkv<dh64>": "attn_bf16_kv_kernel"}
    return table.get(tag)


def _clean(name):
    return name.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]


def _run_self_under_rocprof(extra_rocprof, bench_args, outdir, timeout=240):
    """rocprofv3 <extra> -- python bench.py <bench_args> with the profiler's files under outdir; returns the CSV paths."""
    exe = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(exe):
        return None
    env = dict(os.environ, TMPDIR="/tmp")
    cmd = [exe] + extra_rocprof + ["--output-format", "csv", "-d", outdir, "--", sys.executable, os.path.join(ROOT, "bench.py")] + bench_args
    try:
        subprocess.run(cmd, cwd="/tmp", env=env, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, timeout=timeout, check=False)
    except Exception:
        return None
    return glob.glob(os.path.join(outdir, "**", "*.csv"), recursive=True)


def rocprof_views(workload, steps=15, want_trace=True):
    """Two facts only a hardware profiler of the REAL (two-stream, un-instrumented) step can give, collected by running this
    script as a child of rocprofv3 (separate runs: a kernel trace, then one PMC pass per counter, as MI355X_MICROARCH.md
    prescribes): per-kernel average durations under concurrency, and HBM bytes per launch (FETCH_SIZE doubled for the gfx950
    wide-load under-count, x1024)."""
    out = {"kernels_concurrent": None, "traffic": None, "wall_per_step_us": None, "traffic_per_step": None}
    launches = {}
    base = ["--workload", workload, "--no-profile", "--no-cpu-baseline", "--no-extras"]
    tmp = tempfile.mkdtemp(prefix="skf_rocprof_")
    try:
        files = _run_self_under_rocprof(["--kernel-trace"], base + ["--steps", str(steps), "--warmup", "5"], os.path.join(tmp, "kt")) if want_trace else None
        trace = [f for f in (files or []) if f.endswith("kernel_trace.csv")]
        if trace:
            rows = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), _clean(r["Kernel_Name"])) for r in csv.DictReader(open(trace[0]))]
            rows.sort()
            ends = [i for i, r in enumerate(rows) if "adam" in r[2] or "sgd" in r[2]][-(steps + 1):]
            if len(ends) >= 2:
                seg = rows[ends[0] + 1: ends[-1] + 1]
                n = len(ends) - 1
                agg = {}
                for s, e, k in seg:
                    a = agg.setdefault(k, [0, 0.0])
                    a[0] += 1; a[1] += (e - s) / 1e3
                out["kernels_concurrent"] = sorted(({"kernel": k[:90], "launches_per_step": round(v[0] / n, 2), "avg_us": round(v[1] / v[0], 2),
                                                     "per_step_ms": round(v[1] / n / 1e3, 4)} for k, v in agg.items()),
                                                   key=lambda r: -r["per_step_ms"])[:16]
                launches = {k: v[0] / n for k, v in agg.items()}
                out["wall_per_step_us"] = (seg[-1][1] - seg[0][0]) / 1e3 / n
                out["kernels_per_step"] = len(seg) / n
        traffic = {}
        for counter, scale in (("FETCH_SIZE", 2.0), ("WRITE_SIZE", 1.0)):
            files = _run_self_under_rocprof(["--pmc", counter], base + ["--steps", "3", "--warmup", "1"], os.path.join(tmp, counter))
            cc = [f for f in (files or []) if f.endswith("counter_collection.csv")]
            if not cc:
                traffic = None
                break
            for r in csv.DictReader(open(cc[0])):
                if r["Counter_Name"] == counter:
                    t = traffic.setdefault(_clean(r["Kernel_Name"]), {"FETCH_SIZE": [0.0, 0], "WRITE_SIZE": [0.0, 0]})
                    t[counter][0] += float(r["Counter_Value"]) * scale * 1024.0
                    t[counter][1] += 1
        if traffic:
            out["traffic"] = {k: (v["FETCH_SIZE"][0] / max(v["FETCH_SIZE"][1], 1)) + (v["WRITE_SIZE"][0] / max(v["WRITE_SIZE"][1], 1))
                              for k, v in traffic.items()}
            # step-level HBM bytes: every kernel of the traced step x its PMC bytes per launch (the figure to hold against the
            # algorithmic bytes of SURVEY 8(d): step_bytes())
            hit = [k for k in launches if k in out["traffic"]]
            if hit:
                out["traffic_per_step"] = {"bytes": sum(out["traffic"][k] * launches[k] for k in hit),
                                           "launches_covered": sum(launches[k] for k in hit) / max(sum(launches.values()), 1e-9)}
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    return out


def committed_traffic(tag):
    """fallback when rocprofv3 cannot run here: the latest committed PMC summary under profiles/"""
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "*pmc_traffic.json")))
    prefix = _kernel_prefix(tag)
    if not files or not prefix:
        return None
    table = json.load(open(files[-1]))
    rows = [v for k, v in table.items() if k.startswith(prefix)]
    n = sum(r["launches"] for r in rows)
    return sum(r["hbm_bytes_per_launch"] * r["launches"] for r in rows) / n if n else None


def kernel_profile(engine_mod, cfg_kwargs, x, y, steps=3):
    """Per-kernel launch timing with HIP events (libskf's launch profiler) on an un-captured replica."""
    from sketchformer_amd import _lib
    lib = _lib.load()
    eng = engine_mod.TrainEngine(engine_mod.make_config(use_graph=False, **cfg_kwargs), init_seed=0)
    for _ in range(2):
        eng.train_step(x, y)
    eng.synchronize()
    lib.skf_profiler_enable(1)
    for _ in range(steps):
        eng.train_step(x, y)
    eng.synchronize()
    buf = C.create_string_buffer(1 << 16)
    _lib.check(lib.skf_profiler_report(buf, len(buf)), "skf_profiler_report")
    lib.skf_profiler_enable(0)
    rows = json.loads(buf.value.decode())
    for r in rows:
        r["avg_us"] = 1e3 * r["ms"] / r["count"]
        r["per_step_ms"] = r["ms"] / steps
        r["launches_per_step"] = r["count"] / steps
    n_floats = eng.n_floats
    del eng
    torch.cuda.empty_cache()
    return rows, n_floats


def make_batch(synthetic, w, B, seed, full):
    if w["cont"]:
        xs, ys = synthetic.continuous_batch(B, w["L"], CN, seed=seed, full=full)
    else:
        xs, ys = synthetic.token_batch(B, w["L"], w["V"], CN, seed=seed, full=full)
    return xs, ys


def timed(eng, x, y, steps, warmup):
    for _ in range(warmup):
        eng.train_step(x, y)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        eng.train_step(x, y)
    torch.cuda.synchronize()
    return time.perf_counter() - t0


def roofline_of(rows, act, steps_profiled=3):
    """`roofline` object for the dominant kernel of a per-kernel profile (rows sorted by time).  `achieved` / `frac` count the
    work the launches actually did (live row tiles of list-driven launches, visited attention tiles: `flops_done` /
    `bytes_done` of the launch profiler); `frac_dense_counted` is the same time against the dense 2MNK figure."""
    top = rows[0]
    is_mfma = top["flops"] > 0
    bf16_pipe = act == "bf16"
    sec = top["ms"] * 1e-3
    if is_mfma:
        ach = top["flops_done"] / sec / 1e12
        peak = PEAK_BF16_MFMA_TFLOPS if bf16_pipe else PEAK_F32_MFMA_TFLOPS
        roof = {"bound": "mfma", "achieved": ach, "peak": peak, "unit": "TFLOP/s", "frac": ach / peak, "traffic": None,
                "frac_dense_counted": top["flops"] / sec / 1e12 / peak, "work_done_fraction": top["flops_done"] / top["flops"]}
        if "bf16x" in top["tag"]:
            # split-operand kernel: fp32-equivalent FLOPs priced against the fp32 MFMA peak (the dtype of the path); the bf16
            # matrix cores execute n_prod times as many - frac_of_executing_pipe prices THAT against the pipe that runs it
            n_prod = int(top["tag"].split("bf16x")[1].rstrip(">"))
            roof["issued_bf16_tflops"] = ach * n_prod
            roof["frac_of_executing_pipe"] = ach * n_prod / PEAK_BF16_MFMA_TFLOPS
            roof["executing_pipe"] = "bf16 MFMA, %d products per fp32 product, peak %.0f TFLOP/s" % (n_prod, PEAK_BF16_MFMA_TFLOPS)
        else:
            roof["frac_of_executing_pipe"] = roof["frac"]
            roof["executing_pipe"] = "bf16 MFMA" if bf16_pipe else "fp32 MFMA"
    else:
        ach = top["bytes_done"] / sec / 1e9
        roof = {"bound": "hbm", "achieved": ach, "peak": PEAK_HBM_GBS, "unit": "GB/s", "frac": ach / PEAK_HBM_GBS, "traffic": None,
                "frac_dense_counted": top["bytes"] / sec / 1e9 / PEAK_HBM_GBS, "work_done_fraction": top["bytes_done"] / max(top["bytes"], 1.0)}
    roof["algorithmic_per_launch"] = (top["flops_done"] if is_mfma else top["bytes_done"]) / top["count"]
    roof["dense_per_launch"] = (top["flops"] if is_mfma else top["bytes"]) / top["count"]
    roof.update({"kernel": top["tag"], "launches_per_step": top["launches_per_step"], "avg_launch_us": top["avg_us"],
                 "per_step_ms": top["per_step_ms"], "timing": "HIP events on the launch stream (in-library launch profiler)"})
    return roof


def apply_concurrent(roof, views):
    """Re-price `roof` with the rocprofv3 kernel-trace average of the same kernel in the real two-stream step (the HIP-event
    figures stay beside it as *_hip_events) and attach the PMC traffic."""
    prefix = _kernel_prefix(roof["kernel"])
    if not views or not prefix:
        return
    if views.get("traffic"):
        vals = [v for k, v in views["traffic"].items() if k.startswith(prefix)]
        if vals:
            roof["traffic"] = float(np.mean(vals))
            roof["traffic_source"] = "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, two child runs of this script (FETCH doubled, x1024)"
    conc = [r for r in (views.get("kernels_concurrent") or []) if r["kernel"].startswith(prefix)]
    if conc:
        # `frac` must follow from the fields printed beside it: frac = algorithmic_per_launch / avg_launch_us / peak, with
        # avg_launch_us = this kernel's rocprofv3 kernel-trace average in the real two-stream step; the in-library HIP-event
        # figures (which serialise the two streams) stay beside them as *_hip_events.
        n = sum(r["launches_per_step"] for r in conc)
        avg = sum(r["avg_us"] * r["launches_per_step"] for r in conc) / n
        scale = roof["avg_launch_us"] / avg
        for k in ("achieved", "frac", "frac_dense_counted", "issued_bf16_tflops", "frac_of_executing_pipe"):
            if k in roof:
                roof[k + "_hip_events"] = roof[k]
                roof[k] = roof[k] * scale
        roof["avg_launch_us_hip_events"] = roof["avg_launch_us"]
        roof["avg_launch_us"] = avg
        roof["timing"] = ("rocprofv3 --kernel-trace average of this kernel in the un-instrumented two-stream step (child run of this script); "
                          "*_hip_events = the in-library HIP-event figures, which serialise the two streams")


def roofline_identity_error(roof):
    """|frac - algorithmic_per_launch / avg_launch_us / peak| / frac from the fields of a `roofline` object alone (0 when the
    object is self-consistent; tests/test_bench_line_cpu.py asserts it on canned and on freshly built records)."""
    unit = 1e12 if roof["unit"] == "TFLOP/s" else 1e9
    derived = roof["algorithmic_per_launch"] / (roof["avg_launch_us"] * 1e-6) / unit / roof["peak"]
    return abs(derived - roof["frac"]) / max(abs(roof["frac"]), 1e-30)


def kernel_table(rows):
    return [{"tag": r["tag"], "launches_per_step": round(r["launches_per_step"], 2), "avg_us": round(r["avg_us"], 2),
             "per_step_ms": round(r["per_step_ms"], 4),
             "tflops": round(r["flops_done"] / (r["ms"] * 1e-3) / 1e12, 2) if r["flops"] else None,
             "tflops_dense_counted": round(r["flops"] / (r["ms"] * 1e-3) / 1e12, 2) if r["flops"] else None,
             "gbs": round(r["bytes_done"] / (r["ms"] * 1e-3) / 1e9, 1) if r["bytes"] else None} for r in rows]


def sub_record(engine, synthetic, name, B, steps, warmup, seed=0, traffic=True):
    """ms/step, tokens/s, dominant kernel + roofline fraction of another BASELINE config on this GPU (not part of `value`)."""
    w = WORKLOADS[name]
    kw = dict(batch=B, seq_len=w["L"], d_model=w["d"], num_heads=8, dff=w["dff"], num_layers=w["N"], vocab_size=None if w["cont"] else w["V"],
              n_classes=CN, lowerdim=U, dropout_rate=0.1, seed=1234, continuous=w["cont"], act_dtype=w["act"])
    rec = {"workload": w["text"], "per_gpu_batch": B, "dtype": w["act"]}
    eng = engine.TrainEngine(engine.make_config(use_graph=False, **kw), init_seed=0)
    for full in (False, True):
        xs, ys = make_batch(synthetic, w, B, seed, full)
        x, y = torch.from_numpy(xs).cuda(), torch.from_numpy(ys).cuda()
        e = timed(eng, x, y, steps, warmup)
        key = "full_length" if full else "padded"
        rec[key] = {"ms_per_step": 1e3 * e / steps, "value": B * w["L"] * steps / e,
                    "pad_fraction": float((xs[..., 4] == 1).mean() if w["cont"] else (xs == 0).mean())}
    if not w["cont"] and w["L"] != 200:
        # SURVEY 8(d) draws the lengths from N(80, 35^2) whatever the sequence length: 83 % padding at L = 512.  The same distribution
        # stretched with the sequence length (58 % padding, like the L = 200 batches) beside it.
        xs, ys = synthetic.token_batch(B, w["L"], w["V"], CN, seed=seed, length_scale=w["L"] / 200.0)
        e = timed(eng, torch.from_numpy(xs).cuda(), torch.from_numpy(ys).cuda(), steps, warmup)
        rec["scaled_lengths"] = {"ms_per_step": 1e3 * e / steps, "value": B * w["L"] * steps / e, "pad_fraction": float((xs == 0).mean()),
                                 "what": "lengths ~ N(80, 35^2) x L / 200"}
    assert np.isfinite(eng.step_metrics()["total_loss"])
    P = eng.n_floats
    del eng
    torch.cuda.empty_cache()
    xs, ys = make_batch(synthetic, w, B, seed, False)
    x, y = torch.from_numpy(xs).cuda(), torch.from_numpy(ys).cuda()
    rows, _ = kernel_profile(engine, kw, x, y)
    rows.sort(key=lambda r: -r["ms"])
    f_step = step_flops(B, w["L"], w["d"], w["dff"], w["N"], w["V"], U, CN, w["cont"])
    peak = PEAK_BF16_MFMA_TFLOPS if w["act"] == "bf16" else PEAK_F32_MFMA_TFLOPS
    t = rec["padded"]["ms_per_step"] * 1e-3
    roof = roofline_of(rows, w["act"])
    if traffic:
        apply_concurrent(roof, rocprof_views(name, steps=3, want_trace=False))
    rec.update({"ms_per_step": rec["padded"]["ms_per_step"], "value": rec["padded"]["value"], "unit": "stroke-tokens/sec",
                "step_tflops": f_step / t / 1e12, "step_mfma_frac": f_step / t / (peak * 1e12), "step_mfma_peak_tflops": peak,
                "achieved_hbm": step_bytes(B, w["L"], w["d"], w["dff"], w["N"], w["V"], P, 2 if w["act"] == "bf16" else 4) / t / (PEAK_HBM_GBS * 1e9),
                "roofline": roof, "kernels": kernel_table(rows)[:10]})
    return rec


def plugin_path_record(steps, warmup, B):
    """The same cfg-2 step driven through the reference's plugin surface (Model.train_on_batch, models/sketchformer.py:351-359):
    the metrics come back as a lazily read mapping, so the loop is not host-synchronised per step."""
    from sketchformer_amd import models, dataloaders
    Model = models.get_model_by_name("sketch-transformer-tf2")
    Loader = dataloaders.get_dataloader_by_name("stroke3-synthetic")
    dataset = Loader(Loader.parse_hparams("max_seq_len=200,vocab_size=1004,n_classes=345,n_samples=%d" % (B * 64)), None)
    tmp = tempfile.mkdtemp(prefix="skf_bench_")
    try:
        model = Model(Model.parse_hparams(base="batch_size=%d,log_every=1000000" % B, specific=None), dataset, tmp, "bench")
        it = dataset.batch_iterator("train", B, False)
        xs, ys = next(it)
        batch = (torch.from_numpy(xs).cuda(), torch.from_numpy(ys).cuda())
        pend = []
        for _ in range(warmup):
            pend.append(model.train_on_batch(batch))
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            pend.append(model.train_on_batch(batch))
        torch.cuda.synchronize()
        e = time.perf_counter() - t0
        last = dict(pend[-1])
        assert np.isfinite(last["total_loss"])
        return {"ms_per_step": 1e3 * e / steps, "value": B * 200 * steps / e,
                "what": "Model.train_on_batch on a device-resident batch, metrics deferred (one read-back at the end)"}
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def _r(v, sig=5):
    """floats to `sig` significant digits (the compact line is read by people and by the driver, not used for arithmetic)"""
    if isinstance(v, float):
        return float("%.*g" % (sig, v)) if np.isfinite(v) else None
    return v


def _pick(d, keys, sig=5):
    return {k: _r(d[k], sig) for k in keys if isinstance(d, dict) and k in d and d[k] is not None}


def detail_paths():
    """where the full record (kernel tables, sub-records, every roofline side figure) goes: next to the script, and under
    gpurun_out/ when that exists so that it travels back from a GPU box"""
    paths = [os.path.join(ROOT, "bench_detail.json")]
    if os.path.isdir(os.path.join(ROOT, "gpurun_out")):
        paths.append(os.path.join(ROOT, "gpurun_out", "bench_detail.json"))
    return paths


COMPACT_LIMIT = 4000


def compact_line(out):
    """The ONE stdout line the driver parses: the contract keys + `roofline` + `cpu_baseline` and a one-level summary of the
    optional legs, kept under COMPACT_LIMIT bytes (round 3's 20.9 KB line could not be parsed: BENCH_r03.parsed = null).
    Kernel tables, sub-record detail and the long `sample` / `timing` strings live in bench_detail.json."""
    line = {k: _r(out[k], 8) for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                                    "vs_baseline", "dtype", "data") if k in out}
    cfg = dict(out.get("config") or {})
    for k in ("dense_gemm_arithmetic",):
        if isinstance(cfg.get(k), str) and len(cfg[k]) > 60:
            cfg[k] = cfg[k][:57] + "..."
    if isinstance(cfg.get("workload"), str) and len(cfg["workload"]) > 140:
        cfg["workload"] = cfg["workload"][:137] + "..."
    line["config"] = {k: _r(v) for k, v in cfg.items()}
    roof = out.get("roofline")
    if roof:
        line["roofline"] = _pick(roof, ("bound", "achieved", "peak", "unit", "frac", "traffic", "kernel", "avg_launch_us", "launches_per_step",
                                        "per_step_ms", "frac_dense_counted", "frac_of_executing_pipe", "algorithmic_per_launch",
                                        "avg_launch_us_hip_events", "frac_hip_events", "traffic_per_step", "algorithmic_bytes_per_step"))
        line["roofline"].setdefault("traffic", None)
        if roof.get("timing"):
            line["roofline"]["timing"] = "rocprofv3 kernel trace" if roof["timing"].startswith("rocprofv3") else "HIP events"
    cpu = out.get("cpu_baseline")
    if cpu:
        line["cpu_baseline"] = _pick(cpu, ("value", "unit", "cores", "kind"))
        sample = cpu.get("sample_short") or cpu.get("sample") or ""
        line["cpu_baseline"]["sample"] = sample if len(sample) <= 200 else sample[:197] + "..."
    line.update(_pick(out, ("step_mfma_frac", "step_tflops", "achieved_hbm", "kernels_per_step", "rocprof_wall_per_step_us", "rccl_ranks",
                            "final_total_loss")))
    for leg in ("full_length", "fp32_mfma_mode", "plugin_path", "hip_graph_mode"):
        if isinstance(out.get(leg), dict):
            line[leg] = _pick(out[leg], ("ms_per_step", "value", "step_mfma_frac", "skipped", "error"), 4)
    for leg in ("cfg3", "cfg5", "cfg2grid"):
        rec = out.get(leg)
        if isinstance(rec, dict):
            s = _pick(rec, ("ms_per_step", "value", "step_mfma_frac", "skipped"), 4)
            if isinstance(rec.get("full_length"), dict):
                s["full_length_ms"] = _r(rec["full_length"].get("ms_per_step"), 4)
            if isinstance(rec.get("padded"), dict):
                s["pad_fraction"] = _r(rec["padded"].get("pad_fraction"), 3)
            if isinstance(rec.get("scaled_lengths"), dict):
                s["scaled_lengths_ms"] = _r(rec["scaled_lengths"].get("ms_per_step"), 4)
                s["scaled_pad_fraction"] = _r(rec["scaled_lengths"].get("pad_fraction"), 3)
            if "error" in rec:
                s["error"] = str(rec["error"])[:80]
            line[leg] = s
    line["detail"] = "bench_detail.json"
    text = json.dumps(line, separators=(",", ":"))
    # belt and braces: drop optional legs (least important first) rather than ever exceed the limit
    for leg in ("cfg2grid", "cfg3", "cfg5", "plugin_path", "fp32_mfma_mode", "hip_graph_mode", "full_length", "final_total_loss",
                "rocprof_wall_per_step_us", "kernels_per_step"):
        if len(text) <= COMPACT_LIMIT:
            break
        line.pop(leg, None)
        text = json.dumps(line, separators=(",", ":"))
    assert len(text) <= COMPACT_LIMIT and "\n" not in text, len(text)
    return text


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--batch", type=int, default=128, help="per-GPU batch")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-profile", action="store_true", help="skip the per-kernel profile (and with it `roofline`)")
    ap.add_argument("--no-extras", action="store_true", help="skip the sub-records, the rocprofv3 child runs and the plugin-path leg")
    ap.add_argument("--graph", type=int, nargs="?", const=1, default=0, choices=(0, 1, 2),
                    help="replay the step from hipGraphs instead of eager launches: 1 = captured on one stream, 2 = the two-stream step captured "
                         "(side stream forked / joined inside the capture)")
    ap.add_argument("--full-length", action="store_true", help="all rows have n = L (worst case, no padding)")
    ap.add_argument("--allreduce", default="bucketed", choices=["bucketed", "single"],
                    help="data parallel: gradient buckets overlapped with backward / optimizer (default) or ONE all-reduce of the whole buffer")
    ap.add_argument("--time-budget", type=float, default=420.0,
                    help="seconds after which the optional legs (sub-records, their PMC passes, the plugin-path leg) are skipped and say so")
    ap.add_argument("--workload", default="cfg2", choices=sorted(WORKLOADS),
                    help="cfg2 = the headline config (default); cfg3 = 6L/d256/dff1024 continuous; cfg5 = 8L/d512/dff2048 L=512 bf16")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus and world > 1:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
    if args.gpus > 1 and world == 1:
        raise SystemExit("launch with torch.distributed.run for --gpus > 1")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device (no CPU fallback)")
    torch.cuda.set_device(local_rank)
    pg = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        pg = dist.group.WORLD
    rccl_ranks = None

    from sketchformer_amd import build, engine, synthetic, _lib
    if rank == 0:
        build.build_library(verbose=False)
    if world > 1:
        dist.barrier()
        # the rank count the RCCL communicator itself reports after a real collective (not the env's WORLD_SIZE)
        ones = torch.ones(1, device="cuda")
        dist.all_reduce(ones)
        rccl_ranks = int(ones.item())
        assert rccl_ranks == dist.get_world_size() == world, (rccl_ranks, dist.get_world_size(), world)

    w = WORKLOADS[args.workload]
    B, L, d, dff, N, V = args.batch, w["L"], w["d"], w["dff"], w["N"], w["V"]
    cont = w["cont"]
    cfg_kwargs = dict(batch=B, seq_len=L, d_model=d, num_heads=8, dff=dff, num_layers=N, vocab_size=None if cont else V,
                      n_classes=CN, lowerdim=U, dropout_rate=0.1, seed=1234 + rank, continuous=cont, act_dtype=w["act"])
    eng = engine.TrainEngine(engine.make_config(use_graph=args.graph, **cfg_kwargs), init_seed=0, process_group=pg)
    if args.graph == 2:        # the two-stream capture is opt-in (include/skf.h: SKF_MODEL_TWO_STREAM_GRAPH); this leg runs in a child process
        eng.set_flags(_lib.MODEL_TWO_STREAM_GRAPH)
    eng.dp_mode = args.allreduce
    xs, ys = make_batch(synthetic, w, B, rank, args.full_length)
    x = torch.from_numpy(xs).cuda()
    y = torch.from_numpy(ys).cuda()

    for _ in range(args.warmup):
        eng.train_step(x, y)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
        torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        eng.train_step(x, y)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
        torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    metrics = eng.step_metrics()
    assert np.isfinite(metrics["total_loss"]), metrics

    ms_per_step = 1e3 * elapsed / args.steps
    value = world * B * L * args.steps / elapsed
    f_step = step_flops(B, L, d, dff, N, V, U, CN, cont)
    prec = _lib.default_precision()
    peak = PEAK_BF16_MFMA_TFLOPS if w["act"] == "bf16" else PEAK_F32_MFMA_TFLOPS
    bytes_step = step_bytes(B, L, d, dff, N, V, eng.n_floats, 2 if w["act"] == "bf16" else 4)
    arithmetic = ("bf16 operands on the bf16 matrix cores, fp32 accumulate; fp32 master weights / Adam" if w["act"] == "bf16" else
                  {0: "fp32 MFMA", 6: "fp32 operands split exactly into 3 bf16 pieces, 6 products on the bf16 matrix cores, fp32 "
                                     "accumulate (error below the fp32-MFMA kernel's)", 3: "bf16x3 (opt-in fast mode)"}[prec])
    out = {
        "metric": "stroke-tokens/sec training step, d_model=128 L=200 B=128, 1/2/4/8 GPU",
        "value": value, "unit": "stroke-tokens/sec", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": w["act"], "data": "synthetic",
        "config": {"workload": w["text"], "global_batch": B * world, "per_gpu_batch": B, "seq_len": L, "parallelism": "dp%d" % world,
                   "hip_graph": args.graph, "dense_gemm_arithmetic": arithmetic, "gradient_allreduce": args.allreduce if world > 1 else None,
                   "pad_fraction": float((xs[..., 4] == 1).mean() if cont else (xs == 0).mean())},
        "step_mfma_frac": f_step / (elapsed / args.steps) / (peak * 1e12), "step_mfma_peak_tflops": peak,
        "step_tflops": f_step / (elapsed / args.steps) / 1e12,
        "achieved_hbm": bytes_step / (elapsed / args.steps) / (PEAK_HBM_GBS * 1e9), "algorithmic_bytes_per_step": bytes_step,
        "final_total_loss": metrics["total_loss"],
    }
    single = rank == 0 and world == 1
    extras = single and not args.no_extras
    _progress("headline: %.3f ms/step" % ms_per_step)
    if extras and not args.full_length:
        xf, yf = make_batch(synthetic, w, B, rank, True)
        e1 = timed(eng, torch.from_numpy(xf).cuda(), torch.from_numpy(yf).cuda(), args.steps, 5)
        out["full_length"] = {"ms_per_step": 1e3 * e1 / args.steps, "value": B * L * args.steps / e1, "pad_fraction": 0.0,
                              "step_mfma_frac": f_step / (e1 / args.steps) / (peak * 1e12),
                              "what": "the same step with every row at n = L (SURVEY 8(d) worst case)"}
    if extras and w["act"] == "f32" and prec != 0:
        # the same step with the Dense matmuls on v_mfma_f32_16x16x4_f32, timed the same way: reported beside the headline
        eng0 = engine.TrainEngine(engine.make_config(use_graph=args.graph, gemm_precision=0, **cfg_kwargs), init_seed=0)
        e1 = timed(eng0, x, y, args.steps, 5)
        del eng0
        out["fp32_mfma_mode"] = {"ms_per_step": 1e3 * e1 / args.steps, "value": B * L * args.steps / e1,
                                 "step_mfma_frac": f_step / (e1 / args.steps) / (PEAK_F32_MFMA_TFLOPS * 1e12)}
    if extras and not args.graph and not args.full_length:
        # the same step replayed from hipGraphs: `hip_graph_mode` = the two-stream step captured (use_graph = 2, round 5: the side stream
        # is forked / joined inside the capture, its launches become parallel branches; bit-equal to the eager step),
        # `one_stream_ms_per_step` inside it = the single-stream capture of rounds 1-4.  The eager step stays the default.
        # (each capture mode in a process of its own: hipGraphLaunch of a multi-branch graph has crashed inside the HIP runtime -
        #  hip::Graph::UpdateStreams - in processes that had built and destroyed many models before; a child cannot take the line down)
        try:
            rec = {}
            for mode, key in ((2, "ms_per_step"), (1, "one_stream_ms_per_step")):
                cmd = [sys.executable, os.path.abspath(__file__), "--graph", str(mode), "--steps", str(args.steps), "--warmup", "5", "--batch", str(B),
                       "--workload", args.workload, "--no-extras", "--no-cpu-baseline", "--no-profile"]
                env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
                child = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, timeout=300, check=False)
                lines = [ln for ln in child.stdout.decode(errors="replace").splitlines() if ln.startswith("{")]
                if child.returncode != 0 or not lines:
                    raise RuntimeError("use_graph=%d child exited with %d" % (mode, child.returncode))
                rec[key] = float(json.loads(lines[-1])["ms_per_step"])
            rec["value"] = B * L / (rec["ms_per_step"] * 1e-3)
            rec["step_mfma_frac"] = f_step / (rec["ms_per_step"] * 1e-3) / (peak * 1e12)
            out["hip_graph_mode"] = rec
        except Exception as exc:      # noqa: BLE001
            out["hip_graph_mode"] = {"error": str(exc)[:80]}
    del eng
    torch.cuda.empty_cache()
    _progress("full-length / fp32-MFMA / hipGraph legs done")
    if rank == 0 and not args.no_profile:
        rows, _ = kernel_profile(engine, cfg_kwargs, x, y)
        rows.sort(key=lambda r: -r["ms"])
        roof = roofline_of(rows, w["act"])
        out["kernels"] = kernel_table(rows)
        out["kernels_note"] = ("`kernels`: in-library launch profiler (HIP events around every launch on its stream; the events serialise "
                               "the two streams of the step); `kernels_concurrent`: rocprofv3 kernel trace of the un-instrumented step")
        # attention FLOPs are counted dense in `tflops`; padded key tiles are skipped (exactly): second figure over the work done
        if not cont and not args.full_length:
            lens = (xs != 0).sum(1)
            dense = float(B * L * L)
            done = float(sum(L * (int(n + 15) // 16 * 16) for n in lens))
            out["attention_work_fraction"] = {"self_attention_key_tiles_visited": done / dense,
                                              "what": "share of (query, key) pairs in non-skipped 16-key tiles, encoder self-attention (the attn_* "
                                                      "`tflops` of the kernel tables already count visited tiles only; `tflops_dense_counted` does not)"}
        _progress("per-kernel profile done")
        if extras:
            views = rocprof_views(args.workload)
            _progress("rocprofv3 child runs (kernel trace + 2 PMC passes) done")
            apply_concurrent(roof, views)
            if views.get("traffic_per_step"):
                # step-level HBM bytes by PMC (sum over the kernels of the traced step of launches x bytes per launch) beside the
                # algorithmic bytes of the step (every saved activation written once and read once + weights / optimizer)
                roof["traffic_per_step"] = views["traffic_per_step"]["bytes"]
                roof["traffic_per_step_launches_covered"] = views["traffic_per_step"]["launches_covered"]
                roof["algorithmic_bytes_per_step"] = bytes_step
            if views["kernels_concurrent"]:
                out["kernels_concurrent"] = views["kernels_concurrent"]
                out["kernels_per_step"] = views.get("kernels_per_step")
                out["rocprof_wall_per_step_us"] = views["wall_per_step_us"]
            if "full_length" in out:
                # the no-padding step has its own per-kernel table and roofline object (HIP events; no list-driven skipping there)
                xf, yf = make_batch(synthetic, w, B, rank, True)
                rows_f, _ = kernel_profile(engine, cfg_kwargs, torch.from_numpy(xf).cuda(), torch.from_numpy(yf).cuda())
                rows_f.sort(key=lambda r: -r["ms"])
                out["full_length"]["kernels"] = kernel_table(rows_f)[:12]
                out["full_length"]["roofline"] = roofline_of(rows_f, w["act"])
        if roof["traffic"] is None:
            roof["traffic"] = committed_traffic(roof["kernel"])
            if roof["traffic"] is not None:
                roof["traffic_source"] = "committed profiles/*pmc_traffic.json (rocprofv3 not runnable in this invocation)"
        out["roofline"] = roof
    if single and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline()
        _progress("cpu baseline done")
    if extras and args.workload == "cfg2" and not args.graph:
        # optional legs, most informative first; each is skipped (and says so) once the time budget is spent, and the PMC child
        # runs of a sub-record (~1 min) only while half of the budget is left
        skipped = lambda: {"skipped": "time budget (%.0f s of --time-budget %.0f s spent)" % (_elapsed(), args.time_budget)}  # noqa: E731
        for name, st, wu in (("cfg5", 8, 3), ("cfg3", 20, 5), ("cfg2grid", 20, 5)):
            if _elapsed() > args.time_budget:
                out[name] = skipped()
                continue
            try:
                out[name] = sub_record(engine, synthetic, name, 128, st, wu,
                                       traffic=name != "cfg2grid" and _elapsed() < 0.6 * args.time_budget)
            except Exception as e:      # a sub-record must never cost the headline line
                out[name] = {"error": "%s: %s" % (type(e).__name__, e)}
            _progress("sub-record %s done" % name)
        out["plugin_path"] = plugin_path_record(args.steps, args.warmup, B) if _elapsed() < args.time_budget else skipped()
        _progress("plugin path done")
    if rank == 0:
        out["rccl_ranks"] = rccl_ranks
        for path in detail_paths():
            try:
                with open(path, "w") as f:
                    json.dump(out, f, indent=1)
                _progress("full record -> %s" % path)
            except OSError as e:
                _progress("could not write %s: %s" % (path, e))
        print(compact_line(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
