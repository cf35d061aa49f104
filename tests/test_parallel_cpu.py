"""Data-parallel path on CPU (gloo, world_size 2): K6 of SURVEY.md section 8(c) - the all-reduced, 1/W-scaled
flat gradient of W ranks with B_local rows each equals the single-device gradient at batch W*B_local, and the
Keras-Adam update from it is identical on every rank.  Gradients come from the oracle (no GPU here); the
sharding / flat-buffer all-reduce / scale code is the product's (sketchformer_amd.parallel)."""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import oracle
from sketchformer_amd import parallel, synthetic

CFG = dict(num_layers=1, d_model=16, dff=32, num_heads=1, dropout_rate=0.0, lowerdim=8, vocab_size=24, n_classes=5,
           seq_len=10, max_pos=16)


def _flat(G, names):
    return torch.from_numpy(np.concatenate([G[n].reshape(-1) for n in names]))


def _worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    r, w, _, pg = parallel.init_from_env(backend="gloo")
    assert (r, w) == (rank, world)
    cfg = oracle.Config(**CFG)
    x, y = synthetic.token_batch(4 * world, cfg.seq_len, cfg.vocab_size, cfg.n_classes, seed=5)
    P = oracle.init_params(cfg, seed=1)
    names = [n for n, _, _ in oracle.param_specs(cfg)]
    xs, ys = parallel.shard_batch(x, y, rank, world)
    _, _, G = oracle.loss_and_grads(P, cfg, xs, xs, ys)
    flat = _flat(G, names)
    scale = parallel.allreduce_flat_gradients(flat, pg)
    assert scale == 1.0 / world
    flat *= scale
    # every rank applies the same Adam step
    w0 = _flat(P, names).numpy().copy()
    m, v = np.zeros_like(w0), np.zeros_like(w0)
    oracle.adam_update(w0, flat.numpy(), m, v, iterations=4000, lr=float(oracle.warmup_decay(4000, cfg.d_model)))
    gathered = [torch.zeros_like(flat) for _ in range(world)]
    dist.all_gather(gathered, torch.from_numpy(w0), group=pg)
    if rank == 0:
        _, _, Gfull = oracle.loss_and_grads(P, cfg, x, x, y)
        out.put((float((flat - _flat(Gfull, names)).abs().max()), float(_flat(Gfull, names).abs().max()),
                 float(max((g - gathered[0]).abs().max() for g in gathered))))
    dist.barrier()
    dist.destroy_process_group()


def test_data_parallel_gradient_equals_large_batch_gradient():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    out = ctx.SimpleQueue()
    mp.spawn(_worker, args=(2, port, out), nprocs=2, join=True)
    err, scale, spread = out.get()
    assert err < 1e-12 * max(scale, 1.0), (err, scale)
    assert spread == 0.0


def test_single_process_is_identity():
    g = torch.arange(8, dtype=torch.float32)
    assert parallel.allreduce_flat_gradients(g, None) == 1.0
    assert torch.equal(g, torch.arange(8, dtype=torch.float32))
    x, y = np.arange(12).reshape(6, 2), np.arange(6)
    xs, ys = parallel.shard_batch(x, y, 1, 3)
    assert xs.tolist() == [[2, 3], [8, 9]] and ys.tolist() == [1, 4]


def _bucket_worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    _, _, _, pg = parallel.init_from_env(backend="gloo")
    g = torch.Generator().manual_seed(100 + rank)
    flat = torch.randn(1003, generator=g, dtype=torch.float64)
    whole = flat.clone()
    parallel.allreduce_flat_gradients(whole, pg)
    # the engine's schedule: buckets in production order (tail of the buffer first), async work handles, wait per bucket
    buckets = [(600, 403), (0, 600)]
    works = [parallel.allreduce_bucket(flat[o:o + n], pg) for o, n in buckets]
    assert all(w is not None for w in works)
    for w in works:
        w.wait()
    if rank == 0:
        out.put(float((flat - whole).abs().max()))
    dist.barrier()
    dist.destroy_process_group()


def test_bucketed_allreduce_equals_whole_buffer_allreduce():
    """N>1 path of TrainEngine.apply_gradients: all-reducing the gradient buckets (slices of the flat buffer, started
    asynchronously in production order) gives exactly the all-reduce of the whole buffer."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    out = ctx.SimpleQueue()
    mp.spawn(_bucket_worker, args=(2, port, out), nprocs=2, join=True)
    assert out.get() == 0.0
    assert parallel.allreduce_bucket(torch.zeros(4), None) is None          # single process: nothing to reduce
