#!/usr/bin/env python
"""K = 128 weight-stationary Dense kernel per N: launch time, and (under rocprofv3 --pmc, one N per process) HBM bytes.

    python tools/wsx_traffic.py time                      # us per launch, N = 128 / 384 / 512 / 1004, remap rule on / forced off
    python tools/wsx_traffic.py run <N>                   # 20 launches of one N (the command rocprofv3 wraps)
    python tools/wsx_traffic.py pmc <out.json>            # drives rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE per N and tabulates
"""
import csv, glob, json, os, subprocess, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
M, K = 25600, 128
NS = (128, 384, 512, 1004)


def launches(N, n):
    import torch
    from sketchformer_amd import ops
    a = torch.randn(M, K, device="cuda"); b = torch.randn(K, N, device="cuda"); bias = torch.randn(N, device="cuda")
    out = torch.empty(M, N, device="cuda")
    for _ in range(n):
        ops.gemm(a, b, bias=bias, out=out)
    torch.cuda.synchronize()
    return lambda: ops.gemm(a, b, bias=bias, out=out)


def time_mode():
    import ctypes as C, torch
    from sketchformer_amd import _lib
    lib = _lib.load()
    for N in NS:
        fn = launches(N, 5)
        lib.skf_profiler_enable(1)
        for _ in range(50):
            fn()
        torch.cuda.synchronize()
        buf = C.create_string_buffer(1 << 16)
        lib.skf_profiler_report(buf, len(buf)); lib.skf_profiler_enable(0)
        for r in json.loads(buf.value.decode()):
            print("N=%4d  %-28s %6.2f us" % (N, r["tag"], r["ms"] / r["count"] * 1e3))


def pmc_mode(out):
    res = {}
    for N in NS:
        vals = {}
        for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
            d = "/tmp/wsx_pmc_%d_%s" % (N, ctr)
            subprocess.run("rm -rf %s; cd /tmp && TMPDIR=/tmp rocprofv3 --pmc %s -d %s -o r --output-format csv -- python %s run %d" %
                           (d, ctr, d, os.path.abspath(__file__), N), shell=True, check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
            f = glob.glob(d + "/**/*counter_collection.csv", recursive=True)[0]
            v = [float(r["Counter_Value"]) for r in csv.DictReader(open(f)) if r["Counter_Name"] == ctr and "gemm_wsx" in r["Kernel_Name"]]
            vals[ctr] = sum(v) / len(v)
        meas = (2.0 * vals["FETCH_SIZE"] + vals["WRITE_SIZE"]) * 1024.0       # gfx950: FETCH_SIZE doubled, x1024 (MI355X_MICROARCH.md)
        alg = 4.0 * (M * K + K * N + M * N)
        res[str(N)] = {"fetch_bytes": 2048.0 * vals["FETCH_SIZE"], "write_bytes": 1024.0 * vals["WRITE_SIZE"], "measured_bytes": meas,
                       "algorithmic_bytes": alg, "ratio": meas / alg, "fetch_over_algorithmic_reads": 2048.0 * vals["FETCH_SIZE"] / (4.0 * (M * K + K * N))}
        print(N, res[str(N)])
    res["what"] = "gemm_wsx<K128> M=25600, per-launch HBM bytes (rocprofv3 --pmc, separate FETCH_SIZE / WRITE_SIZE passes) vs algorithmic"
    res["xcd_env"] = os.environ.get("SKF_WS_XCD", "rule")
    json.dump(res, open(out, "w"), indent=1)


if __name__ == "__main__":
    if sys.argv[1] == "time":
        time_mode()
    elif sys.argv[1] == "run":
        launches(int(sys.argv[2]), 20)
    else:
        pmc_mode(sys.argv[2])
