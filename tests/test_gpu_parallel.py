"""Data-parallel train step end to end.  K6 of SURVEY 8(c) on the device path: W ranks x B_local rows == one engine
at batch W*B_local, replicas stay bit-identical.
  * one GPU visible: two processes share cuda:0 and all-reduce through gloo (RCCL refuses two ranks on one device; the
    schedule under test - gradient buckets, ready events, communication stream, per-bucket optimizer - is backend
    independent), eager and hipGraph-replayed steps, bucketed and single whole-buffer all-reduce;
  * >= 2 GPUs visible: the same test on the real backend ("nccl" = RCCL over xGMI), one rank per device - skipped
    otherwise (the driver's 8-GPU box runs it)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu

KW = dict(seq_len=24, d_model=64, num_heads=4, dff=128, num_layers=2, vocab_size=52, n_classes=7, lowerdim=32,
          dropout_rate=0.0, use_graph=False, seed=5)
STEPS, B_LOCAL, START = 3, 4, 3000


def _batches(world):
    from sketchformer_amd import synthetic
    return [synthetic.token_batch(B_LOCAL * world, KW["seq_len"], KW["vocab_size"], KW["n_classes"], seed=70 + s) for s in range(STEPS)]


def _worker(rank, world, port, out, backend, use_graph, dp_mode):
    local = rank if backend == "nccl" else 0
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(local), HSA_ENABLE_IPC_MODE_LEGACY="0")
    import faulthandler
    faulthandler.dump_traceback_later(120, exit=True)     # a wedged collective must not hold the GPU box
    import torch.distributed as dist
    from sketchformer_amd import engine, parallel
    torch.cuda.set_device(local)
    _, _, _, pg = parallel.init_from_env(backend=backend)
    kw = dict(KW, use_graph=use_graph)
    eng = engine.TrainEngine(engine.make_config(batch=B_LOCAL, **kw), init_seed=1, process_group=pg)
    eng.dp_mode = dp_mode
    assert eng.world_size == world and len(eng.grad_buckets()) == (1 if use_graph else 2)
    eng.assert_replicas_equal()
    eng.state[0] = START
    for x, y in _batches(world):
        xs, ys = parallel.shard_batch(x, y, rank, world)
        eng.train_step(xs, ys)
    torch.cuda.synchronize()
    flat = eng.params.detach().cpu()
    gathered = [torch.zeros_like(flat) for _ in range(world)]
    dist.all_gather(gathered, flat, group=pg)
    if rank == 0:   # (a pipe would block on this much data until the parent reads it: the parent is still joining us)
        np.savez(out, flat=flat.numpy(), spread=float(max((g - flat).abs().max() for g in gathered)), iters=eng.iterations)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("backend,use_graph,dp_mode", [("gloo", False, "bucketed"), ("gloo", True, "bucketed"),
                                                       ("gloo", False, "single"), ("nccl", False, "bucketed"),
                                                       ("nccl", True, "bucketed"), ("nccl", False, "single")])
def test_two_ranks_equal_one_engine_at_double_batch(tmp_path, backend, use_graph, dp_mode):
    from sketchformer_amd import engine
    world = 2
    if backend == "nccl" and torch.cuda.device_count() < 2:
        pytest.skip("the RCCL run needs >= 2 visible GPUs (one rank per device)")
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    out = str(tmp_path / "rank0.npz")
    mp.spawn(_worker, args=(world, port, out, backend, use_graph, dp_mode), nprocs=world, join=True)
    res = np.load(out)
    flat, spread, iters = res["flat"], float(res["spread"]), int(res["iters"])
    assert spread == 0.0 and iters == START + STEPS          # replicas stay identical
    ref = engine.TrainEngine(engine.make_config(batch=B_LOCAL * world, **KW), init_seed=1)
    ref.state[0] = START
    for x, y in _batches(world):
        ref.train_step(x, y)
    torch.cuda.synchronize()
    assert ref.iterations == START + STEPS, ref.iterations
    want = ref.params.detach().cpu().numpy()
    ref2 = engine.TrainEngine(engine.make_config(batch=B_LOCAL * world, **KW), init_seed=1)      # the reference itself is reproducible
    ref2.state[0] = START
    for x, y in _batches(world):
        ref2.train_step(x, y)
    torch.cuda.synchronize()
    assert np.array_equal(want, ref2.params.detach().cpu().numpy()), "the single-engine reference differs between two runs"
    moved = np.abs(want - engine.keras_init(ref.entries, ref.n_floats, 1)).max()
    assert moved > 1e-3                                                     # the optimizer really moved the weights
    # wk biases have an analytically zero gradient; Adam turns their rounding noise into +-lr steps (see
    # test_adam_trajectory_matches_oracle) - they cannot influence the loss and are left out of the comparison
    worst = 0.0
    for e in ref.entries:
        if e["name"].endswith("wk/bias"):
            continue
        idx = (e["offset"] + np.arange(e["rows"])[:, None] * e["row_stride"] + np.arange(e["cols"])[None, :]).reshape(-1)
        worst = max(worst, float(np.abs(flat[idx] - want[idx]).max()))
    assert worst < 5e-3 * moved, (worst, moved)
