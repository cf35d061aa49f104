"""``sketch-transformer-tf2`` on MI355X: the plugin surface of models/sketchformer.py (:17-365) over
TrainEngine (the C-ABI HIP train step).  Same registry name, hparams, ctor, ``train_on_batch`` contract
and metric names; the arithmetic runs in libskf.so - there is no CPU / eager fallback.
"""
import zlib
from collections.abc import Mapping

import numpy as np

from .. import builders
from ..core.models import BaseModel
from .evaluation_mixin import TransformerMetricsMixin
from ..utils.hparams import HParams


class DeferredMetrics(Mapping):
    """What ``train_on_batch`` returns: the reference's {metric name: float} dict (models/sketchformer.py:351-359), read
    back from the device only when somebody looks.  The reference pays one host sync per step for ``.numpy()``
    (builders/keras_metrics.py:41-42); here the step only snapshots the 32 device floats, ``BaseModel.train`` resolves
    all pending snapshots with ONE copy when it prints (``log_every``), and a caller that indexes / iterates the mapping
    right away gets the same floats the reference would have returned (one sync, like the reference).  Under data
    parallelism ``resolve_all`` (what ``BaseModel.train`` calls on EVERY rank) all-reduces the (sum, count) accumulators;
    indexing / iterating / printing a single result never runs a collective - it reads this rank's own running values, so a
    rank-specific read (``if rank == 0: log(res['total_loss'])``) cannot deadlock."""

    def __init__(self, engine, snapshot, drop):
        self._engine, self._snap, self._drop, self._vals = engine, snapshot, drop, None

    def resolve_with(self, vals):
        self._vals = {k: v for k, v in vals.items() if k not in self._drop}
        self._snap = None

    def _resolved(self):
        if self._vals is None:
            self.resolve_with(self._engine.resolve_metrics([self._snap], reduce=False)[0])
        return self._vals

    @staticmethod
    def resolve_all(pending):
        """One read-back (and, data parallel, one all-reduce) for a list of unresolved results of the same engine."""
        todo = [p for p in pending if p._vals is None]
        if todo:
            for p, vals in zip(todo, todo[0]._engine.resolve_metrics([p._snap for p in todo])):
                p.resolve_with(vals)

    def __getitem__(self, k):
        return self._resolved()[k]

    def __iter__(self):
        return iter(self._resolved())

    def __len__(self):
        return len(self._resolved())

    def __repr__(self):
        return repr(self._resolved())


class Transformer(BaseModel, TransformerMetricsMixin):
    name = 'sketch-transformer-tf2'
    quick_metrics = ['recon_loss', 'recon_acc', 'class_loss', 'class_acc', 'total_loss']
    slow_metrics = ["sketch-reconstruction", "val-clas-acc", "tsne", "tsne-predicted"]

    @classmethod
    def specific_default_hparams(cls):
        """models/sketchformer.py:25-53 (names, defaults and types unchanged)."""
        return HParams(
            num_layers=4, d_model=128, dff=512, num_heads=8, dropout_rate=0.1,
            lowerdim=256, attn_version=1,
            do_classification=True, class_weight=1.0, class_buffer_layers=0, class_dropout=0.1,
            do_reconstruction=True, recon_weight=1.0, blind_decoder_mask=True,
            is_training=True, optimizer='Adam', lr=0.01, lr_scheduler='WarmupDecay', warmup_steps=10000,
        )

    def __init__(self, hps, dataset, out_dir, experiment_id, device=None, process_group=None, init_seed=0):
        self.losses_manager = builders.losses.LossManager()
        self.metrics_manager = builders.keras_metrics.MetricManager()
        self.vocab_size = dataset.tokenizer.VOCAB_SIZE if not dataset.hps['use_continuous_data'] else None
        self.seq_len = dataset.hps['max_seq_len']
        self._device, self._pg, self._init_seed = device, process_group, init_seed
        super().__init__(hps, dataset, out_dir, experiment_id, process_group=process_group)

    def build_model(self):
        from .. import engine
        h = self.hps
        if h['optimizer'].lower() not in ('adam', 'sgd'):
            raise ValueError("optimizer=%r: the reference builds Adam or SGD (models/sketchformer.py:120-126)" % h['optimizer'])
        # models/sketchformer.py:76-108: decoder / losses / metrics are only registered for the heads that exist
        self._has_cls = bool(h['lowerdim']) and bool(h['do_classification'])
        if h['do_classification'] and not h['lowerdim']:
            raise ValueError("do_classification needs lowerdim > 0 (models/sketchformer.py:96-108: the class head lives "
                             "inside the bottleneck block; the reference fails on the unregistered 'class' loss)")
        if h['do_reconstruction']:
            if self.dataset.hps['use_continuous_data']:
                self.losses_manager.add_continuous_reconstruction_loss('recon', weight=h['recon_weight'])
                self.metrics_manager.add_mean_metric('recon_loss')
            else:
                self.losses_manager.add_reconstruction_loss('recon', weight=h['recon_weight'])
                self.metrics_manager.add_mean_metric('recon_loss')
                self.metrics_manager.add_sparse_categorical_accuracy('recon_acc')
        if self._has_cls:
            self.losses_manager.add_sparse_categorical_crossentropy('class', weight=h['class_weight'])
            self.metrics_manager.add_mean_metric('class_loss')
            self.metrics_manager.add_sparse_categorical_accuracy('class_acc')
        self.metrics_manager.add_mean_metric('total_loss')
        # WarmupDecay is built with warmup_steps=5000 whatever the hparams say (models/sketchformer.py:113-114)
        self.learning_rate = (builders.schedulers.WarmupDecay(h['d_model'], warmup_steps=5000)
                              if h['lr_scheduler'].lower() in ('warmupdecay', 'warmup-decay') else None)
        cfg = engine.make_config(
            batch=h['batch_size'], seq_len=self.seq_len, d_model=h['d_model'], num_heads=h['num_heads'], dff=h['dff'],
            num_layers=h['num_layers'], vocab_size=self.vocab_size or 0, n_classes=self.dataset.n_classes,
            lowerdim=h['lowerdim'], attn_version=h['attn_version'], continuous=self.dataset.hps['use_continuous_data'],
            blind_decoder_mask=h['blind_decoder_mask'], dropout_rate=h['dropout_rate'], recon_weight=h['recon_weight'],
            class_weight=h['class_weight'], lr_scheduler=h['lr_scheduler'], lr=h['lr'], use_graph=False,
            optimizer=h['optimizer'], class_buffer_layers=h['class_buffer_layers'], class_dropout=h['class_dropout'],
            do_classification=h['do_classification'], do_reconstruction=h['do_reconstruction'],
            # dropout key = hash(seed, iterations): the seed differs per experiment id and per rank (ranks must
            # draw independent masks, or W ranks at B rows are not one step at W*B rows); the weight init seed does NOT
            # depend on the rank - replicas start identical (checked below)
            seed=(zlib.crc32(str(self.experiment_id).encode()) + 0x9e3779b1 * self.rank) & 0xffffffff)
        self.engine = engine.TrainEngine(cfg, device=self._device, init_seed=self._init_seed, process_group=self._pg)
        self.engine.assert_replicas_equal()
        self.trainable_variables = [e["name"] for e in self.engine.entries]
        drop = set()
        if self.dataset.hps['use_continuous_data'] or not h['do_reconstruction']:
            drop.add('recon_acc')
        if not h['do_reconstruction']:
            drop.add('recon_loss')
        if not self._has_cls:
            drop.update(('class_loss', 'class_acc'))
        self._dropped_metrics = drop

    # ---- the train step (models/sketchformer.py:351-359)
    def train_on_batch(self, batch):
        data, labels = batch
        self.engine.train_step(data, labels)
        # the Keras running metrics of this step, as a mapping that is read back lazily (no host sync here)
        return DeferredMetrics(self.engine, self.engine.metrics_snapshot(), self._dropped_metrics)

    def prepare_for_start_of_epoch(self):
        pass

    def prepare_for_end_of_epoch(self):
        self.engine.reset_metrics()

    # ---- inference API (models/sketchformer.py:162-311); inputs are padded to the engine's batch size
    def _pad_batch(self, x):
        x = np.asarray(x)
        if x.ndim == (2 if self.engine.cfg.continuous else 1):
            x = x[None]
        n, B = x.shape[0], self.engine.cfg.batch
        if n > B:
            raise ValueError("at most batch_size=%d sequences per call" % B)
        pad = np.zeros((B,) + x.shape[1:], dtype=np.float32 if self.engine.cfg.continuous else np.int64)
        if self.engine.cfg.continuous:
            pad[..., 4] = 1.0                      # stroke-5 padding rows
        pad[:n] = x
        return pad, n

    def encode_from_seq(self, inp_seq):
        pad, n = self._pad_batch(inp_seq)
        self.engine.encode(pad)
        self.engine.synchronize()
        B = self.engine.cfg.batch
        enc = self.engine.buffer('enc_output').view(B, self.seq_len, -1)[:n].cpu().numpy()
        emb = self.engine.buffer('embedding')[:n].cpu().numpy() if self.hps['lowerdim'] else enc
        return {'enc_output': enc, 'embedding': emb,
                'class': self.engine.buffer('class_probs')[:n].cpu().numpy() if self._has_cls else None}

    def predict_class(self, inp_seq):
        out = self.encode_from_seq(inp_seq)
        if self._has_cls:
            out['class'] = out['class'].argmax(-1).astype(np.int32)
        return out

    def make_dummy_input(self, expected_len, nattn, batch_size):
        """models/sketchformer.py:230-253: fake encoder input, only its padding mask matters (first nattn positions real)."""
        if self.engine.cfg.continuous:
            d = np.zeros((batch_size, self.seq_len, 5), dtype=np.float32)
            d[:, int(nattn):, 4] = 1.0
            return d
        d = np.zeros((batch_size, self.seq_len), dtype=np.float32)
        if expected_len is None:
            d[:, :int(nattn)] = 1.0
        else:
            for b, n in enumerate(np.asarray(nattn).reshape(-1)):
                d[b, :int(n)] = 1.0
        return d

    def predict_from_embedding(self, emb, expected_len=None):
        """Greedy reconstruction from the bottleneck (models/sketchformer.py:255-311), KV-cached on the device.
        Returns {'recon', 'class', 'attn_weights'}; attention weights are never materialised here (no consumer in the
        reference reads them: evaluation_mixin.py:25-35, experiments/*.py) -> None."""
        if not self.hps['do_reconstruction']:
            raise ValueError("do_reconstruction is off")
        emb = np.asarray(emb, dtype=np.float32)
        if emb.ndim == (1 if self.hps['lowerdim'] else 2):     # one embedding: (E,) - or (L, d) without a bottleneck
            emb = emb[None]
        n, B = emb.shape[0], self.engine.cfg.batch
        if n > B:
            raise ValueError("at most batch_size=%d embeddings per call" % B)
        pad = np.zeros((B,) + emb.shape[1:], dtype=np.float32)
        pad[:n] = emb
        tok = self.dataset.tokenizer
        if self.hps['blind_decoder_mask']:
            expected_len = None                     # "will be ignored if blind_decoder_mask=True"
        recon = self.engine.greedy_decode(pad, expected_len=expected_len, n_valid=n,
                                          sos=getattr(tok, 'SOS', 0) if tok is not None else 0,
                                          eos=getattr(tok, 'EOS', 0) if tok is not None else 0)
        out = {'recon': recon, 'attn_weights': None}
        if self._has_cls:
            out['class'] = self.engine.buffer('class_probs')[:n].cpu().numpy().argmax(-1).astype(np.int32)
        return out

    def predict(self, inp_seq):
        """models/sketchformer.py:201-221."""
        out = self.encode_from_seq(inp_seq)
        if self._has_cls:
            out['class'] = out['class'].argmax(-1).astype(np.int32)
        if self.hps['do_reconstruction']:
            x = np.asarray(inp_seq)
            if self.hps['blind_decoder_mask']:
                tlen = None
            elif self.engine.cfg.continuous:
                tlen = np.sum(x[..., -1] != 1, axis=-1).reshape(-1)
            else:
                tlen = np.sum(x > 0, axis=-1).reshape(-1)
            dec = self.predict_from_embedding(out['embedding'], tlen)
            out['recon'] = dec['recon']
            out['attn_weights'] = dec['attn_weights']
        return out

    def load_reference_checkpoint(self, prefix):
        """Weights (+ Adam slots, optimizer.iterations, current_step) from a checkpoint written by the reference's
        tf.train.Checkpoint(transformer=..., optimizer=...) (core/models.py:321-344), read without TensorFlow."""
        import torch
        from ..utils import tf_checkpoint
        params, m, v, scalars = tf_checkpoint.load_reference_checkpoint(prefix, self.engine.entries)
        e = self.engine
        e.load_numpy(params)
        e.adam_m.zero_()
        e.adam_v.zero_()
        e.load_numpy(m, "adam_m")
        e.load_numpy(v, "adam_v")
        if 'iterations' in scalars:
            e.state[0] = int(scalars['iterations'])
        self.current_step = int(scalars.get('current_step', scalars.get('iterations', 0)))
        torch.cuda.synchronize()
        e.assert_replicas_equal()

    # ---- checkpoint payload
    def state_dict(self):
        e = self.engine
        e.synchronize()
        return {'params': e.params.cpu(), 'adam_m': e.adam_m.cpu(), 'adam_v': e.adam_v.cpu(), 'metrics': e.metrics.cpu(),
                'iterations': e.iterations, 'entries': e.entries}

    def prepare_metrics_for_save(self):
        self.engine.fold_metric_accumulators_into_rank0()

    def load_state_dict(self, state):
        e = self.engine
        e.params.copy_(state['params'])
        e.adam_m.copy_(state['adam_m'])
        e.adam_v.copy_(state['adam_v'])
        e.metrics.copy_(state['metrics'])
        if e.rank != 0:      # the file holds the accumulators of ALL ranks (folded into rank 0 before the save): one copy only
            e.zero_metric_accumulators()
        e.state[0] = int(state['iterations'])
        e.assert_replicas_equal()
