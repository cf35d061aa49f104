#!/usr/bin/env python
"""Headline benchmark: stroke-tokens/sec of the sketch-transformer-tf2 train step
(forward + backward + Keras-Adam/WarmupDecay) on synthetic QuickDraw-shaped batches.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Workload (BASELINE.json configs[1] / cfg 2): 4 layers, 8 heads, d_model 128, dff 512,
L=200, per-GPU B=128, V=1004, 345 classes, dropout 0.1, fp32.  Weak scaling: every rank
runs B=128 (global batch 128*N); one RCCL all-reduce of the flat fp32 gradient buffer
per step.  Inputs are resident in HBM before the timed region.  Prints ONE JSON line.
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_F32_MFMA_TFLOPS = 157.3   # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32 / 16x16x4_f32 dense peak
PEAK_BF16_MFMA_TFLOPS = 2500.0  # MI355X_MICROARCH.md: dense bf16 MFMA peak (v_mfma_f32_16x16x32_bf16 / 32x32x16)
PEAK_HBM_GBS = 8000.0


def step_flops(B, L, d, dff, N, V, U, Cn):
    """Algorithmic FLOPs of one train step, SURVEY.md section 8(d): F_step = 3 * F_fwd."""
    Le, Ld, Lk = L, L - 1, L
    f_enc = N * (8 * B * Le * d * d + 4 * B * Le * Le * d + 4 * B * Le * d * dff)
    f_dec = N * (8 * B * Ld * d * d + 4 * B * Ld * Ld * d + 4 * B * Ld * d * d + 4 * B * Lk * d * d
                 + 4 * B * Ld * Lk * d + 4 * B * Ld * d * dff)
    f_out = 2 * B * Ld * d * V
    f_bott = 2 * B * Le * d * U + 2 * B * Le * U + 2 * B * Le * d
    f_cls = 2 * B * d * Cn + 2 * B * d * L
    return 3 * (f_enc + f_dec + f_out + f_bott + f_cls)


def cpu_baseline(seconds_budget=25.0):
    """The CPU restatement of the TF2 reference (oracle/, numpy float32, 'port'), timed on this host on a
    bounded sample: cfg 1 (C=1) at B=16 rows instead of 128 (same L, model and step; tokens/s is per-row linear)."""
    import oracle
    from sketchformer_amd import synthetic
    B, L = 16, 200
    cfg = oracle.Config(n_classes=1)
    x, y = synthetic.token_batch(B, L, cfg.vocab_size, 1, seed=0)
    state = oracle.TrainState.create(oracle.init_params(cfg, 0, np.float32))
    rng = np.random.RandomState(0)
    drops = {n: rng.rand(B, L if t == "enc" else L - 1, cfg.d_model) >= cfg.dropout_rate
             for n, t in oracle.dropout_sites(cfg)}
    oracle.train_step(state, cfg, x, x, y, drops)          # warm-up
    times = []
    t_start = time.perf_counter()
    while len(times) < 5 and time.perf_counter() - t_start < seconds_budget:
        t0 = time.perf_counter()
        oracle.train_step(state, cfg, x, x, y, drops)
        times.append(time.perf_counter() - t0)
    med = float(np.median(times))
    try:
        from threadpoolctl import threadpool_info
        cores = max([p.get("num_threads", 1) for p in threadpool_info()] or [1])
    except Exception:
        cores = os.cpu_count() or 1
    return {"value": B * L / med, "unit": "stroke-tokens/sec", "cores": int(cores), "kind": "port",
            "sample": "numpy float32 oracle, cfg1 (4L/8H/d128/dff512, L=200, V=1004, C=1), B=16 of 128 rows, "
                      "median of %d full train steps (%.2f s each), host has %d logical cores"
                      % (len(times), med, os.cpu_count() or 1)}


def _gemm_precision():
    from sketchformer_amd import _lib
    return int(_lib.default_precision())


def pmc_traffic(tag):
    """HBM bytes per launch of the kernel behind a profiler tag, from the committed rocprofv3 PMC passes
    (profiles/*pmc_traffic.json: FETCH_SIZE and WRITE_SIZE collected in separate runs, gfx950 corrections applied)."""
    import glob
    import re
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "*pmc_traffic.json")))
    if not files:
        return None
    table = json.load(open(files[-1]))
    m = re.match(r"gemm_ws(x?)<K(\d+),CW(\d+)", tag)
    if m:   # kernel template is <K, columns per lane = CW/16, ...>
        prefix = "gemm_ws%s_kernel<%s, %d," % (m.group(1), m.group(2), int(m.group(3)) // 16)
    else:
        prefix = {"wgrad<64x64>": "wgrad_kernel", "attn_bwd<dh16>": "attn_bwd_kernel<16", "attn_fwd<dh16>": "attn_fwd_kernel<16",
                  "ln_fwd": "ln_fwd_kernel", "ln_bwd": "ln_bwd_kernel"}.get(tag)
    if not prefix:
        return None
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
    del eng
    torch.cuda.empty_cache()
    return rows


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--batch", type=int, default=128, help="per-GPU batch")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-profile", action="store_true")
    ap.add_argument("--graph", action="store_true", help="replay the step from hipGraphs instead of eager launches + wgrad side stream")
    ap.add_argument("--full-length", action="store_true", help="all rows have n = L (worst case, no padding)")
    ap.add_argument("--workload", default="cfg2", choices=["cfg2", "cfg3"],
                    help="cfg2 = the headline config (default); cfg3 = 6L/d256/dff1024 continuous stroke-5 (parity-test config)")
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

    from sketchformer_amd import build, engine, synthetic
    if rank == 0:
        build.build_library(verbose=False)
    if world > 1:
        dist.barrier()

    B, L, d, dff, N, V, U, Cn = args.batch, 200, 128, 512, 4, 1004, 256, 345
    cont = args.workload == "cfg3"
    if cont:
        d, dff, N, V = 256, 1024, 6, 5
    cfg_kwargs = dict(batch=B, seq_len=L, d_model=d, num_heads=8, dff=dff, num_layers=N, vocab_size=None if cont else V,
                      n_classes=Cn, lowerdim=U, dropout_rate=0.1, seed=1234 + rank, continuous=cont)
    eng = engine.TrainEngine(engine.make_config(use_graph=args.graph, **cfg_kwargs), init_seed=0, process_group=pg)
    if cont:
        xs, ys = synthetic.continuous_batch(B, L, Cn, seed=rank, full=args.full_length)
    else:
        xs, ys = synthetic.token_batch(B, L, V, Cn, seed=rank, full=args.full_length)
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
    f_step = step_flops(B, L, d, dff, N, V, U, Cn)
    out = {
        "metric": "stroke-tokens/sec training step, d_model=128 L=200 B=128, 1/2/4/8 GPU",
        "value": value, "unit": "stroke-tokens/sec", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f32", "data": "synthetic",
        "config": {"workload": ("cfg3: sketch-transformer-tf2 6L/8H/d256/dff1024 L=200 continuous stroke-5 C=345 dropout=0.1, "
                                "fwd+bwd+Adam(WarmupDecay)") if cont else
                               ("cfg2: sketch-transformer-tf2 4L/8H/d128/dff512 L=200 V=1004 C=345 dropout=0.1, "
                                "fwd+bwd+Adam(WarmupDecay)"), "global_batch": B * world, "per_gpu_batch": B,
                   "seq_len": L, "parallelism": "dp%d" % world, "hip_graph": args.graph,
                   "dense_gemm_arithmetic": {0: "fp32 MFMA", 6: "fp32 operands split exactly into 3 bf16 pieces, 6 products on the bf16 "
                                                "matrix cores, fp32 accumulate (error below the fp32-MFMA kernel's)",
                                             3: "bf16x3 (opt-in fast mode)"}[_gemm_precision()],
                   "pad_fraction": float((xs[..., 4] == 1).mean() if cont else (xs == 0).mean())},
        "step_mfma_frac": f_step / (elapsed / args.steps) / (PEAK_F32_MFMA_TFLOPS * 1e12),
        "step_tflops": f_step / (elapsed / args.steps) / 1e12,
        "final_total_loss": metrics["total_loss"],
    }
    if rank == 0 and world == 1 and _gemm_precision() != 0 and not args.no_profile:
        # the same step with the Dense matmuls on v_mfma_f32_16x16x4_f32 (SKF_GEMM_PRECISION=f32), timed the same way on the
        # same engine after the headline region: reported beside it, never part of `value`
        eng0 = engine.TrainEngine(engine.make_config(use_graph=args.graph, gemm_precision=0, **cfg_kwargs), init_seed=0)
        for _ in range(max(2, min(args.warmup, 5))):
            eng0.train_step(x, y)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for _ in range(args.steps):
            eng0.train_step(x, y)
        torch.cuda.synchronize()
        e1 = time.perf_counter() - t1
        del eng0
        out["fp32_mfma_mode"] = {"ms_per_step": 1e3 * e1 / args.steps, "value": B * L * args.steps / e1,
                                 "step_mfma_frac": f_step / (e1 / args.steps) / (PEAK_F32_MFMA_TFLOPS * 1e12)}
    if rank == 0 and not args.no_profile:
        rows = kernel_profile(engine, cfg_kwargs, x, y)
        rows.sort(key=lambda r: -r["ms"])
        top = rows[0]
        is_mfma = top["flops"] > 0
        if is_mfma:
            ach = top["flops"] / (top["ms"] * 1e-3) / 1e12
            roof = {"bound": "mfma", "achieved": ach, "peak": PEAK_F32_MFMA_TFLOPS, "unit": "TFLOP/s",
                    "frac": ach / PEAK_F32_MFMA_TFLOPS, "traffic": None}
        else:
            ach = top["bytes"] / (top["ms"] * 1e-3) / 1e9
            roof = {"bound": "hbm", "achieved": ach, "peak": PEAK_HBM_GBS, "unit": "GB/s", "frac": ach / PEAK_HBM_GBS,
                    "traffic": None}
        if "bf16x" in top["tag"]:   # split-operand kernel: fp32-equivalent FLOPs priced against the fp32 MFMA peak (the dtype of
            n_prod = int(top["tag"].split("bf16x")[1].rstrip(">"))   # the path); the bf16 matrix cores execute n_prod x as many
            roof["issued_bf16_tflops"] = ach * n_prod
            roof["issued_frac_of_bf16_peak"] = ach * n_prod / PEAK_BF16_MFMA_TFLOPS
        roof["traffic"] = pmc_traffic(top["tag"])
        roof["algorithmic_per_launch"] = (top["flops"] if is_mfma else top["bytes"]) / top["count"]
        roof.update({"kernel": top["tag"], "launches_per_step": top["count"] // 3, "avg_launch_us": top["avg_us"],
                     "per_step_ms": top["per_step_ms"]})
        out["roofline"] = roof
        out["kernels"] = [{"tag": r["tag"], "launches_per_step": r["count"] // 3, "avg_us": round(r["avg_us"], 2),
                           "per_step_ms": round(r["per_step_ms"], 4),
                           "tflops": round(r["flops"] / (r["ms"] * 1e-3) / 1e12, 2) if r["flops"] else None,
                           "gbs": round(r["bytes"] / (r["ms"] * 1e-3) / 1e9, 1) if r["bytes"] else None}
                          for r in rows]
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline()
    if rank == 0:
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
