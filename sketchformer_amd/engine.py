"""TrainEngine: owns the flat device buffers of one replica and drives the
C-ABI train step (skf_model_*).  Host-side mirror of what
``Transformer.build_model`` + ``prepare_model_trainer`` + ``train_on_batch`` set up
in the reference (models/sketchformer.py:63-129, 313-359).

Data parallelism (new functionality): one process per
GPU; every rank runs forward+backward on its shard of the global batch, the flat
fp32 gradient buffer is summed with ONE all-reduce (RCCL over xGMI on GPUs) and
the 1/world_size scale is fused into the Adam sweep.
"""
import ctypes as C
import os
import math

import numpy as np
import torch

from . import _lib
from ._lib import SkfConfig, SkfParamEntry

METRIC_NAMES = ("recon_loss", "recon_acc", "class_loss", "class_acc", "total_loss")


def positional_encoding(position, d_model):
    """builders/utils.py:12-32: float64 numpy angles with an np.float32(d_model)
    divisor, sin on even / cos on odd columns, cast to float32.  -> (position, d_model)."""
    pos = np.arange(position)[:, np.newaxis]
    i = np.arange(d_model)[np.newaxis, :]
    rates = 1 / np.power(10000, (2 * (i // 2)) / np.float32(d_model))
    ang = pos * rates
    ang[:, 0::2] = np.sin(ang[:, 0::2])
    ang[:, 1::2] = np.cos(ang[:, 1::2])
    return ang.astype(np.float32)


def make_config(batch, seq_len=200, d_model=128, num_heads=8, dff=512, num_layers=4, vocab_size=1004, n_classes=345,
                lowerdim=256, attn_version=1, continuous=False, blind_decoder_mask=True, dropout_rate=0.1,
                recon_weight=1.0, class_weight=1.0, lr_scheduler="WarmupDecay", lr=0.01, seed=0, use_graph=True,
                max_pos=1000, optimizer="Adam", class_buffer_layers=0, class_dropout=0.1, do_classification=True,
                do_reconstruction=True, gemm_precision=None, act_dtype="f32"):
    """gemm_precision: SKF_PREC_* arithmetic of the Dense / attention matmuls (0 fp32 MFMA, 6 bf16x6, 3 bf16x3);
    None = _lib.default_precision() (bf16x6 unless SKF_GEMM_PRECISION says otherwise).
    act_dtype: "f32" (the reference's arithmetic, cfg 1-4) or "bf16" (BASELINE cfg 5: bf16 activations / weight images /
    MFMA with fp32 accumulation, fp32 master weights and Adam)."""
    cfg = SkfConfig()
    cfg.act_dtype = {"f32": 0, "fp32": 0, "float32": 0, "bf16": 1, "bfloat16": 1}[str(act_dtype).lower()]
    cfg.gemm_precision = _lib.default_precision() if gemm_precision is None else int(gemm_precision)
    cfg.batch, cfg.seq_len, cfg.d_model, cfg.num_heads, cfg.dff, cfg.num_layers = batch, seq_len, d_model, num_heads, dff, num_layers
    cfg.vocab_size, cfg.n_classes, cfg.lowerdim, cfg.attn_version = vocab_size or 0, n_classes, lowerdim, attn_version
    cfg.continuous, cfg.blind_decoder_mask, cfg.max_pos = int(continuous), int(blind_decoder_mask), max_pos
    cfg.dropout_rate, cfg.recon_weight, cfg.class_weight = dropout_rate, recon_weight, class_weight
    name = lr_scheduler.lower()
    if name in ("warmupdecay", "warmup-decay"):
        # models/sketchformer.py:113-114: warmup_steps=5000 hard-coded (hparams warmup_steps / lr ignored)
        cfg.schedule, cfg.sched_p0, cfg.sched_p1 = 0, float(d_model), float(5000 ** -1.5)
    elif name == "step-decay":
        # models/sketchformer.py:116-118 passes the non-existent kwarg min_lr= -> TypeError in the reference
        raise TypeError("__init__() got an unexpected keyword argument 'min_lr'")
    else:
        raise ValueError("unknown lr_scheduler %r" % lr_scheduler)
    cfg.beta1, cfg.beta2, cfg.eps = 0.9, 0.98, 1e-9   # models/sketchformer.py:122-124
    cfg.seed, cfg.use_graph = seed, int(use_graph)
    # models/sketchformer.py:120-126: Adam(schedule, 0.9, 0.98, 1e-9) or SGD(schedule, momentum=0.9)
    if optimizer.lower() == "adam":
        cfg.optimizer, cfg.momentum = 0, 0.0
    elif optimizer.lower() == "sgd":
        cfg.optimizer, cfg.momentum = 1, 0.9
    else:
        raise ValueError("unknown optimizer %r" % optimizer)
    cfg.class_buffer_layers, cfg.class_dropout = int(class_buffer_layers), float(class_dropout)
    cfg.do_classification, cfg.do_reconstruction = int(bool(do_classification)), int(bool(do_reconstruction))
    return cfg


def keras_init(entries, total, seed=0):
    """Keras default initialisers into a flat float32 numpy buffer
    (glorot_uniform Dense kernels, zeros biases, uniform(+-0.05) embeddings,
    ones/zeros LayerNorm, RandomNormal(0.05)/uniform(+-0.05) SelfAttnV1)."""
    rng = np.random.RandomState(seed)
    flat = np.zeros(total, dtype=np.float32)
    for e in entries:
        name = e["name"]
        r, c = e["rows"], e["cols"]
        view = strided_view(flat, e)
        if name.endswith("/kernel"):
            lim = math.sqrt(6.0 / (r + c))
            view[...] = rng.uniform(-lim, lim, size=(r, c))
        elif name.endswith("embedding") or name.endswith("V_attn"):
            view[...] = rng.uniform(-0.05, 0.05, size=(r, c))
        elif name.endswith("W_attn"):
            view[...] = rng.normal(0.0, 0.05, size=(r, c))
        elif name.endswith("/gamma"):
            view[...] = 1.0
    return flat


def strided_view(flat, e):
    """2-D numpy view of one variable inside a flat numpy buffer."""
    return np.lib.stride_tricks.as_strided(flat[e["offset"]:], shape=(e["rows"], e["cols"]),
                                           strides=(e["row_stride"] * flat.itemsize, flat.itemsize))


def param_entries(cfg):
    lib = _lib.load()
    n = lib.skf_model_param_entries(C.byref(cfg), None, 0)
    if n < 0:
        _lib.check(n, "skf_model_param_entries")
    arr = (SkfParamEntry * n)()
    lib.skf_model_param_entries(C.byref(cfg), arr, n)
    return [{"name": a.name.decode(), "offset": int(a.offset), "rows": int(a.rows), "cols": int(a.cols),
             "row_stride": int(a.row_stride)} for a in arr]


# shapes as the reference's trainable variables report them (1-D variables, (U,1) V_attn, (1,L) expand kernel)
def logical_shape(e):
    n = e["name"]
    if n.endswith(("/bias", "/gamma", "/beta", "b_attn")):
        return (e["cols"],)
    return (e["rows"], e["cols"])


class TrainEngine:
    """One replica of the sketch-transformer-tf2 train step on one GPU."""

    def __init__(self, cfg, device=None, init_seed=0, process_group=None):
        if not torch.cuda.is_available():
            raise _lib.SkfError("TrainEngine needs a HIP device: torch.cuda.is_available() is False (no CPU fallback)")
        self.lib = _lib.load()
        self.cfg = cfg
        self.device = torch.device(device if device is not None else "cuda:%d" % torch.cuda.current_device())
        _lib.call("skf_config_validate", C.byref(cfg))
        self.entries = param_entries(cfg)
        self.by_name = {e["name"]: e for e in self.entries}
        self.n_floats = int(self.lib.skf_model_param_floats(C.byref(cfg)))
        flat = keras_init(self.entries, self.n_floats, init_seed)
        dev = self.device
        self._params = torch.from_numpy(flat).to(dev)
        self._grads = torch.zeros(self.n_floats, dtype=torch.float32, device=dev)
        self._adam_m = torch.zeros_like(self._grads)
        self._adam_v = torch.zeros_like(self._grads)
        self.pos = torch.from_numpy(positional_encoding(cfg.max_pos, cfg.d_model)).to(dev)
        self._metrics = torch.zeros(32, dtype=torch.float32, device=dev)
        self._state = torch.zeros(int(self.lib.skf_step_state_bytes()) // 8 + 1, dtype=torch.int64, device=dev)
        ws_bytes = int(self.lib.skf_model_workspace_bytes(C.byref(cfg)))
        self._workspace = torch.empty(ws_bytes + 256, dtype=torch.uint8, device=dev)
        off = (-self._workspace.data_ptr()) % 256
        self._ws_ptr = self._workspace.data_ptr() + off
        self._ws_bytes = ws_bytes
        handle = C.c_void_p()
        _lib.call("skf_model_create", C.byref(cfg), C.byref(handle))
        self.handle = handle
        _lib.call("skf_model_bind", handle, self._p(self._params), self._p(self._grads), self._p(self._adam_m),
                  self._p(self._adam_v), self._p(self.pos), C.c_void_p(self._ws_ptr), ws_bytes, self._p(self._metrics),
                  self._p(self._state))
        # a dedicated non-default stream: hipGraph capture is illegal on the legacy default stream
        self.stream = torch.cuda.Stream(device=dev)
        self._depth = 0
        self._dirty = False
        self._comm = None                # communication stream of the data-parallel gradient buckets
        self.pg = process_group
        self.world_size = torch.distributed.get_world_size(process_group) if process_group is not None else 1
        self.rank = torch.distributed.get_rank(process_group) if process_group is not None else 0
        # "bucketed": the flat gradient buffer is all-reduced in the pieces backward finishes (overlapping the rest of
        # backward / the optimizer sweep of the previous piece); "single": ONE all-reduce of the whole buffer after
        # backward (the literal north-star schedule) - bench.py --allreduce single|bucketed A/Bs the two
        self.dp_mode = "bucketed"

    # The buffers live on the engine's stream.  A step no longer makes the caller's stream wait for it (that hand-over, and the
    # one back at the start of the next step, were two cross-stream hops = ~25-40 us of idle GPU between steps,
    # profiles/r03l_gaps.txt): it only marks the results unpublished, and whoever READS a buffer through the attributes below
    # (or any accessor of this class) first orders the current stream behind the engine's.
    def _publish(self):
        if self._dirty:
            torch.cuda.current_stream(self.device).wait_stream(self.stream)
            self._dirty = False

    def _published(self, t):
        self._publish()
        return t

    params = property(lambda self: self._published(self._params))
    grads = property(lambda self: self._published(self._grads))
    adam_m = property(lambda self: self._published(self._adam_m))
    adam_v = property(lambda self: self._published(self._adam_v))
    metrics = property(lambda self: self._published(self._metrics))
    state = property(lambda self: self._published(self._state))
    workspace = property(lambda self: self._published(self._workspace))

    def __del__(self):
        h = getattr(self, "handle", None)
        if h is not None and h.value:
            try:
                self.lib.skf_model_destroy(h)
            except Exception:
                pass

    @staticmethod
    def _p(t):
        return C.c_void_p(t.data_ptr())

    def _stream(self):
        return C.c_void_p(self.stream.cuda_stream)

    def _enter(self):
        # order after whatever the caller queued on its current stream (H2D copies of the batch, set()).  Nested calls (train_step
        # brackets forward_backward + apply_gradients) hand over once: every hand-over is a cross-stream event pair, and the one
        # that used to sit between the backward and the optimizer was ~40 us of idle GPU per step (profiles/r03l_gaps.txt)
        if self._depth == 0:
            self.stream.wait_stream(torch.cuda.current_stream(self.device))
        self._depth += 1

    def _leave(self):
        # let the caller's stream observe the step's results
        self._depth -= 1
        if self._depth == 0:
            self._dirty = True               # published lazily (see _publish)

    def synchronize(self):
        self.stream.synchronize()

    # ---- parameters by (Keras-style) name
    def _view(self, flat, name):
        e = self.by_name[name]
        return torch.as_strided(flat, (e["rows"], e["cols"]), (e["row_stride"], 1), e["offset"])

    def get(self, name, which="params"):
        e = self.by_name[name]
        return self._view(getattr(self, which), name).detach().cpu().numpy().reshape(logical_shape(e)).copy()

    def set(self, name, value, which="params"):
        v = self._view(getattr(self, which), name)
        v.copy_(torch.as_tensor(np.asarray(value, dtype=np.float32).reshape(v.shape)).to(self.device))

    def state_dict_numpy(self, which="params"):
        return {e["name"]: self.get(e["name"], which) for e in self.entries}

    def load_numpy(self, params, which="params"):
        for k, v in params.items():
            self.set(k, v, which)

    # ---- steps
    def _dev_tokens(self, x, limit=None, what="token id"):
        """int64 device copy.  Host inputs (the loader's numpy batches) are range-checked against ``limit`` here - a
        tokenizer / vocab_size (or class count) mismatch must fail loudly instead of training on wrong rows; tensors
        that already live on the device are trusted (checking them would cost a host sync per step)."""
        t = torch.as_tensor(x)
        if limit is not None and not t.is_cuda and t.numel():
            lo, hi = int(t.min()), int(t.max())
            if lo < 0 or hi >= limit:
                raise ValueError("%s out of range: values span [%d, %d], valid range is [0, %d) - tokenizer / dataset "
                                 "does not match the model (vocab_size / n_classes)" % (what, lo, hi, limit))
        if t.dtype != torch.int64:
            t = t.to(torch.int64)
        return self._own(t, t.to(self.device, non_blocking=True).contiguous())

    _clone_inputs = True

    def _own(self, given, staged):
        """The step reads its inputs asynchronously on the engine's stream and no longer makes the caller's stream wait for it
        (_publish): a device tensor the caller passes in - and may refill IN PLACE for the next batch - is therefore copied here,
        on the caller's stream (the hand-over of _enter orders the step behind the copy).  Host inputs already became fresh device
        copies.  ~2 us of copy kernels on the caller's stream, nothing on the engine's."""
        if self._clone_inputs and torch.is_tensor(given) and given.is_cuda and staged.data_ptr() == given.data_ptr():
            return staged.clone()
        return staged

    def _dev_input(self, x):
        """(B,L) int64 tokens, or (B,L,5) float32 stroke-5 rows in continuous mode (the reference casts the
        loader's float64 to float32 at the tf.function boundary, models/sketchformer.py:317-319)."""
        if not self.cfg.continuous:
            return self._dev_tokens(x, self.cfg.vocab_size)
        t = torch.as_tensor(x)
        if t.dim() != 3 or t.shape[-1] != 5:
            raise ValueError("continuous mode expects (B, L, 5) stroke-5 input")
        return self._own(t, t.to(self.device, dtype=torch.float32, non_blocking=True).contiguous())

    def _hold(self, *tensors):
        """The engine's stream reads these caller-owned tensors asynchronously: tell the caching allocator, so that a tensor the
        caller drops right after the call is not handed out again (and overwritten from the caller's stream) before the step has
        staged it (the step used to make the caller's stream wait instead: see _publish)."""
        for t in tensors:
            if isinstance(t, torch.Tensor) and t.is_cuda:
                t.record_stream(self.stream)

    def _ld(self, t):
        return t.stride(0) // 5 if self.cfg.continuous else t.stride(0)

    def forward(self, inp, tar=None, training=False):
        """Transformer.call: fills the internal buffers (see ``buffer``)."""
        inp = self._dev_input(inp)
        tar = inp if tar is None else self._dev_input(tar)
        self._enter()
        try:
            self._hold(inp, tar)
            _lib.call("skf_model_forward", self.handle, self._p(inp), self._p(tar), self._ld(tar), int(training), self._stream())
        finally:
            self._leave()

    def set_flags(self, flags):
        """skf_model_set_flags: per-model switches (``_lib.MODEL_DECODE_LAYERWISE``)."""
        _lib.check(self.lib.skf_model_set_flags(self.handle, int(flags)), "skf_model_set_flags")

    def encode(self, inp):
        """encode_from_seq (models/sketchformer.py:162-168): encoder + bottleneck + classifier, dropout off.
        Results in the buffers 'embedding', 'class_probs', 'enc_output'."""
        inp = self._dev_input(inp)
        self._enter()
        try:
            self._hold(inp)
            _lib.call("skf_model_encode", self.handle, self._p(inp), self._stream())
        finally:
            self._leave()

    def greedy_decode(self, embedding=None, expected_len=None, n_valid=None, sos=0, eos=0, max_steps=None):
        """predict_from_embedding (models/sketchformer.py:255-311) with a K/V cache.  embedding: (B,d) array / tensor
        or None (= the model's own 'embedding' buffer, i.e. right after ``encode``).  Returns the reconstruction as a
        host array: (n_valid, T) int32 tokens incl. the SOS column, or (n_valid, T, 5) float32 stroke-5 rows."""
        B, L = self.cfg.batch, self.cfg.seq_len
        n_valid = B if n_valid is None else int(n_valid)
        max_steps = L if max_steps is None else int(max_steps)
        emb_ptr = None
        if embedding is not None:
            e = torch.as_tensor(np.asarray(embedding, dtype=np.float32) if not torch.is_tensor(embedding) else embedding)
            e = e.to(self.device, dtype=torch.float32).contiguous()
            if self.cfg.lowerdim == 0:      # no bottleneck: the "embedding" is the encoder output (B, L, d)
                want = (B, L, self.cfg.d_model)
            else:                           # SelfAttnV2 projects to lowerdim
                want = (B, self.cfg.lowerdim if self.cfg.attn_version == 2 else self.cfg.d_model)
            if tuple(e.shape) != want:
                raise ValueError("embedding must have shape %r" % (want,))
            emb_ptr = self._p(e)
        lim = None
        if expected_len is not None:
            arr = np.zeros(B, dtype=np.int32)
            v = np.asarray(expected_len).astype(np.int32).reshape(-1)
            arr[:len(v)] = v
            arr[len(v):] = L
            lim = (C.c_int * B)(*arr.tolist())
        if self.cfg.continuous:
            out = torch.zeros(B, max_steps + 1, 5, dtype=torch.float32, device=self.device)
        else:
            out = torch.zeros(B, max_steps + 1, dtype=torch.int64, device=self.device)
        n_out = C.c_int(0)
        self._enter()
        try:
            _lib.call("skf_model_greedy_decode", self.handle, emb_ptr, lim, n_valid, int(sos), int(eos), max_steps, self._p(out),
                      C.byref(n_out), self._stream())
        finally:
            self._leave()
        self.synchronize()
        res = out[:n_valid, :n_out.value].cpu().numpy()
        return res if self.cfg.continuous else res.astype(np.int32)

    def _stage(self, inp, tar, labels):
        """caller-side arguments -> device tensors (enqueued on the CALLER's stream: must precede _enter).  Device tensors of the
        caller are NOT cloned here (round 6): the library copies them into its workspace first thing on the engine's stream and
        _forward_backward_staged makes the caller's stream wait for exactly that copy (skf_model_wait_inputs_staged) - the two
        clone kernels per step sat on the caller's stream in front of the hand-over, ~19 us of idle GPU at the head of every step."""
        self._clone_inputs = os.environ.get("SKF_CLONE_INPUTS") == "1"      # (A/B knob of the Python layer: the round-5 behaviour)
        try:
            inp = self._dev_input(inp)
            tar = inp if tar is None else self._dev_input(tar)
            labels = self._dev_tokens(labels, self.cfg.n_classes if self.cfg.do_classification and self.cfg.lowerdim else None,
                                      "class label")
        finally:
            self._clone_inputs = True
        return inp, tar, labels

    def _forward_backward_staged(self, inp, tar, labels):
        self._hold(inp, tar, labels)
        _lib.call("skf_model_forward_backward", self.handle, self._p(inp), self._p(tar), self._ld(tar), self._p(labels),
                  self._stream())
        # the caller may refill its device tensors as soon as the staging copy has read them: its stream waits for that copy only
        _lib.call("skf_model_wait_inputs_staged", self.handle, torch.cuda.current_stream(self.device).cuda_stream)

    def forward_backward(self, inp, tar, labels):
        staged = self._stage(inp, tar, labels)
        self._enter()
        try:
            self._forward_backward_staged(*staged)
        finally:
            self._leave()

    def grad_buckets(self):
        """[(offset, count)] slices of the flat gradient buffer in the order they become final during backward."""
        off, cnt = (C.c_size_t * 2)(), (C.c_size_t * 2)()
        n = int(self.lib.skf_model_grad_buckets(self.handle, 2, off, cnt))      # returns the count (negative = error)
        if n <= 0:
            _lib.check(n if n < 0 else -1, "skf_model_grad_buckets")
        return [(int(off[i]), int(cnt[i])) for i in range(n)]

    def apply_gradients(self, bucketed=None):
        """optimizer.apply_gradients (models/sketchformer.py:348).  Data parallel: every gradient bucket is all-reduced
        on a communication stream as soon as backward has finished it (bucket 0 = decoder part, overlaps the encoder
        backward; bucket 1 overlaps the optimizer sweep of bucket 0), then the optimizer runs per bucket with the 1/W
        factor fused.  ``bucketed=True`` forces that schedule without a process group (single-GPU test of the plumbing)."""
        from . import parallel
        if bucketed is None:
            bucketed = self.world_size > 1 and self.dp_mode == "bucketed"
        self._enter()
        try:
            if not bucketed:
                scale = 1.0
                if self.world_size > 1:       # one all-reduce of the whole flat buffer, ordered after backward on the step's stream
                    with torch.cuda.stream(self.stream):
                        scale = parallel.allreduce_flat_gradients(self._grads, self.pg)
                _lib.call("skf_model_apply_gradients", self.handle, scale, self._stream())
                return
            if self._comm is None:
                self._comm = torch.cuda.Stream(device=self.device)
            buckets = self.grad_buckets()
            scale = 1.0 / max(self.world_size, 1)
            works = []
            if self.cfg.use_graph:
                # a step replayed from a hipGraph records no per-bucket events (one bucket): the communication stream must be
                # ordered after the whole captured forward/backward, or the all-reduce would race with the gradient writes
                self._comm.wait_stream(self.stream)
            for i, (off, cnt) in enumerate(buckets):
                _lib.call("skf_model_wait_grad_bucket", self.handle, i, C.c_void_p(self._comm.cuda_stream))
                with torch.cuda.stream(self._comm):
                    works.append(parallel.allreduce_bucket(self._grads[off:off + cnt], self.pg))
            for i, (off, cnt) in enumerate(buckets):
                with torch.cuda.stream(self.stream):
                    if works[i] is not None:
                        works[i].wait()                      # engine stream waits for this bucket's all-reduce
                    else:
                        self.stream.wait_stream(self._comm)
                _lib.call("skf_model_apply_gradients_range", self.handle, off, cnt, scale, int(i == len(buckets) - 1), self._stream())
        finally:
            self._leave()

    def train_step(self, inp, labels, tar=None):
        """model_trainer(inp, tar, lab) (models/sketchformer.py:325-349); no host sync."""
        staged = self._stage(inp, tar, labels)       # host-to-device copies of the batch go to the caller's stream FIRST:
        self._enter()                                # the hand-over below is what orders the step behind them
        try:
            self._forward_backward_staged(*staged)
            self.apply_gradients()
        finally:
            self._leave()

    def buffer(self, name):
        """(rows, cols) view of an internal activation.  fp32 models: a float32 view of the workspace; bf16 models keep
        most activations in bf16 (row pitch padded to 8 elements): a bfloat16 view - call .float() for arithmetic."""
        ptr, rows, cols, ld, is16 = C.c_void_p(), C.c_int(), C.c_int(), C.c_int(), C.c_int()
        _lib.call("skf_model_buffer_info", self.handle, name.encode(), C.byref(ptr), C.byref(rows), C.byref(cols), C.byref(ld),
                  C.byref(is16))
        esz = 2 if is16.value else 4
        lo = self._ws_ptr - self.workspace.data_ptr()
        ws = self.workspace[lo:lo + self._ws_bytes].view(torch.bfloat16 if is16.value else torch.float32)
        off = (ptr.value - self._ws_ptr) // esz
        return ws[off:off + rows.value * ld.value].view(rows.value, ld.value)[:, :cols.value]

    def step_metrics(self):
        """This step's five scalars (host sync)."""
        m = self.metrics[:5].cpu().numpy()
        return dict(zip(METRIC_NAMES, (float(x) for x in m)))

    def running_metrics(self):
        """Keras running metrics (never reset during train(), core/models.py:183-197)."""
        m = self.metrics.cpu().numpy()
        return {n: float(m[8 + i] / m[16 + i]) if m[16 + i] else 0.0 for i, n in enumerate(METRIC_NAMES)}

    def reset_metrics(self):
        self.metrics.zero_()

    def fold_metric_accumulators_into_rank0(self):
        """Data parallel, before a checkpoint: the running-metric (sum, count) accumulators of all ranks are summed into rank
        0's buffer and zeroed elsewhere (collective).  Later reads still sum over the ranks, so nothing is counted twice, and
        the file rank 0 writes carries every rank's history (restore: rank 0 loads it, the others start from zero)."""
        if self.world_size <= 1:
            return
        self._enter()               # after whatever the caller's stream wrote into the buffer (load_state_dict); marks it unpublished
        try:
            with torch.cuda.stream(self.stream):
                acc = torch.cat([self._metrics[8:13], self._metrics[16:21]]).contiguous()
                torch.distributed.all_reduce(acc, op=torch.distributed.ReduceOp.SUM, group=self.pg)
                if self.rank == 0:
                    self._metrics[8:13], self._metrics[16:21] = acc[:5], acc[5:]
                else:
                    self.zero_metric_accumulators()
        finally:
            self._leave()

    def zero_metric_accumulators(self):
        self._enter()
        try:
            with torch.cuda.stream(self.stream):
                self._metrics[8:13] = 0.0
                self._metrics[16:21] = 0.0
        finally:
            self._leave()

    def metrics_snapshot(self):
        """Device copy of the 32 metric floats as they stand after the steps queued so far (no host sync): what
        ``train_on_batch`` hands back; ``resolve_metrics`` turns a list of them into floats with ONE read-back."""
        with torch.cuda.stream(self.stream):
            snap = self._metrics.clone()
        return snap

    def resolve_metrics(self, snaps, reduce=True):
        """[snapshot] -> [running-metric dict].  One device->host copy for the whole list; under data parallelism the
        (sum, count) accumulators are summed over the ranks first (all-reduce only when metrics are read),
        so every rank must call this with the same number of snapshots.  reduce=False: this rank's own running values,
        no collective (what a rank-specific read such as ``if rank == 0: log(res['total_loss'])`` gets)."""
        if not snaps:
            return []
        with torch.cuda.stream(self.stream):
            m = torch.stack(list(snaps))
            if self.world_size > 1 and reduce:
                acc = torch.cat([m[:, 8:13], m[:, 16:21]], dim=1).contiguous()
                torch.distributed.all_reduce(acc, op=torch.distributed.ReduceOp.SUM, group=self.pg)
                m = m.clone()
                m[:, 8:13], m[:, 16:21] = acc[:, :5], acc[:, 5:]
        self.stream.synchronize()
        rows = m.cpu().numpy()
        return [{n: float(r[8 + i] / r[16 + i]) if r[16 + i] else 0.0 for i, n in enumerate(METRIC_NAMES)} for r in rows]

    def assert_replicas_equal(self):
        """Data parallelism keeps W copies of the parameters / optimizer state in step only if they start equal: compare
        every rank's flat buffers with rank 0's (one broadcast each) and raise on any difference."""
        if self.world_size <= 1:
            return
        import torch.distributed as dist
        src = dist.get_global_rank(self.pg, 0) if self.pg is not None else 0
        bad = torch.zeros(1, dtype=torch.int32, device=self.device)
        for name in ("params", "adam_m", "adam_v"):
            mine = getattr(self, name)
            ref = mine.clone()
            dist.broadcast(ref, src=src, group=self.pg)
            if not torch.equal(ref, mine):
                bad += 1
        its = self.state[:1].clone()
        dist.broadcast(its, src=src, group=self.pg)
        if int(its.item()) != self.iterations:
            bad += 1
        dist.all_reduce(bad, op=dist.ReduceOp.SUM, group=self.pg)
        if int(bad.item()):
            raise _lib.SkfError("data-parallel replicas differ at start (parameters / optimizer state / iterations): "
                                "every rank must build the model with the same init_seed and restore the same checkpoint")

    @property
    def iterations(self):
        return int(self.state[0].item())
