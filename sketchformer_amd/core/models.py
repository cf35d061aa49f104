"""BaseModel: the training-loop / hparams / output-directory / checkpoint contract of the reference's
core/models.py (:16-358) without TensorFlow.  The loop calls ``train_on_batch`` exactly like the reference
(core/models.py:163-197); the checkpoint is this project's own format (the flat parameter / Adam buffers +
step counter via torch.save) instead of a TF object-graph bundle; slow metrics, plots and the Slack notifier
are outside the accelerated path and are not built.
"""
import glob
import os
import pprint
import socket
from abc import ABCMeta, abstractmethod

from ..utils import hparams as hp
from .metrics import QuickMetric


class BaseModel(object, metaclass=ABCMeta):

    @classmethod
    def base_default_hparams(cls):
        return hp.HParams(
            batch_size=128, num_epochs=10, save_every=1., safety_save=.5, autograph=True, log_every=100,
            notify_every=1000, slack_config='token.secret', goal='No description')

    def __init__(self, hps, dataset, outdir, experiment_id, process_group=None):
        self.hps = hps if isinstance(hps, dict) else dict(hps.values())
        self.dataset = dataset
        # data parallelism (one process per GPU): every rank runs the same loop on its own shard; rank 0 alone
        # prints, writes config / checkpoints / plots, the others only take part in the collectives
        self.process_group = process_group
        if process_group is not None:
            import torch.distributed as dist
            self.rank, self.world_size = dist.get_rank(process_group), dist.get_world_size(process_group)
        else:
            self.rank, self.world_size = 0, 1
        self._pending_quick = []
        self.host = socket.gethostname()
        self.experiment_id = experiment_id
        self.batches_per_epoch = self.dataset.n_samples // self.hps['batch_size']
        for attr, what in (('name', 'a reference name'), ('quick_metrics', 'quick metric names'),
                           ('slow_metrics', 'slow metric names')):
            if not hasattr(self, attr):
                raise Exception("You must give your model %s (class attribute %r)" % (what, attr))
        self.quick_metrics = {q: QuickMetric() for q in self.quick_metrics}
        self.slow_metrics = {m: None for m in self.slow_metrics}     # evaluation plug-ins: not part of the hot path
        self.out_dir = os.path.join(outdir, self.identifier)
        self.plots_out_dir = os.path.join(self.out_dir, 'plots')
        self.wgt_out_dir = os.path.join(self.out_dir, 'weights')
        self.tmp_out_dir = os.path.join(self.out_dir, 'tmp')
        for d in (self.out_dir, self.plots_out_dir, self.wgt_out_dir, self.tmp_out_dir):
            os.makedirs(d, exist_ok=True)
        self.config_filepath = self.get_config_filepath(outdir, self.experiment_id)
        self.current_step = 0
        self.epoch = 0
        self._metric_inputs = {}
        self.build_model()
        self.prepare_checkpointing()

    @property
    def identifier(self):
        return "{}-{}".format(self.name, self.experiment_id)

    @classmethod
    def default_hparams(cls):
        return hp.combine_hparams_into_one(cls.specific_default_hparams(), cls.base_default_hparams())

    @classmethod
    def get_config_filepath(cls, output_dir, exp_id):
        return os.path.join(output_dir, "{}-{}".format(cls.name, exp_id), 'config.json')

    @classmethod
    def parse_hparams(cls, base, specific):
        hps = cls.default_hparams()
        if base is not None:
            hps = hps.parse(base)
        if specific is not None:
            hps = hps.parse(specific)
        return hps

    @classmethod
    @abstractmethod
    def specific_default_hparams(cls):
        pass

    @abstractmethod
    def build_model(self):
        pass

    @abstractmethod
    def train_on_batch(self, batch):
        pass

    @abstractmethod
    def prepare_for_start_of_epoch(self):
        pass

    @abstractmethod
    def prepare_for_end_of_epoch(self):
        pass

    # ---- the loop (core/models.py:163-197)
    def _barrier(self):
        if self.world_size > 1:
            import torch.distributed as dist
            dist.barrier(group=self.process_group)

    def train(self, max_steps=None):
        if self.rank == 0:
            print("*Training started on {}*\n*Goal:* {}\nParams:\n{}".format(self.host, self.hps['goal'], pprint.pformat(self.hps)))
        total_steps = self.batches_per_epoch * self.hps['num_epochs']
        if max_steps is not None:
            total_steps = min(total_steps, self.current_step + max_steps)
        self.epoch = self.current_step // max(self.batches_per_epoch, 1)
        it = self.dataset.batch_iterator(split_name='train', batch_size=self.hps['batch_size'], stop_at_end_of_split=False)
        for _ in range(total_steps - self.current_step):
            self.current_step += 1
            quick = self.train_on_batch(next(it))
            self.update_quick_metrics_history(quick)
            self.status_report()
            self.save_checkpoint_if_its_time()
            if self.current_step // self.batches_per_epoch > self.epoch:
                self.epoch = self.current_step // self.batches_per_epoch
                self.status_report(end_of_epoch=True)
        self.flush_quick_metrics()

    def update_quick_metrics_history(self, new_metrics):
        """core/models.py:199-201.  A model may return a lazily read mapping (``resolve_all`` on its type: one device
        read-back for many steps); those are queued and enter the histories, in order, when the loop next prints."""
        if hasattr(type(new_metrics), 'resolve_all'):
            self._pending_quick.append(new_metrics)
            return
        self.flush_quick_metrics()
        for name, value in new_metrics.items():
            self.quick_metrics[name].append_to_history(value)

    def flush_quick_metrics(self):
        pending, self._pending_quick = self._pending_quick, []
        if not pending:
            return
        type(pending[0]).resolve_all(pending)
        for res in pending:
            for name, value in res.items():
                self.quick_metrics[name].append_to_history(value)

    def status_report(self, end_of_epoch=False):
        """core/models.py:203-225: builds and returns the log string on EVERY call.  The device metrics are only read back
        (flush_quick_metrics: collective under data parallelism, every rank gets here at the same step) on the steps that
        print; on the others the string carries the last resolved values.  At notify_every / end of epoch the slow metrics are
        computed, plotted and appended like the reference does (rank 0; SKF_TRAIN_SLOW_METRICS=0 skips them)."""
        cur_iter = self.current_step % self.batches_per_epoch
        printing = (cur_iter % self.hps['log_every'] == 0) or (cur_iter % self.hps['notify_every'] == 0) or end_of_epoch
        if printing:
            self.flush_quick_metrics()
        log = "Epoch {} Batch {}/{}".format(self.epoch, cur_iter, self.batches_per_epoch)
        for k, m in self.quick_metrics.items():
            log = "{}|{}={:4.4f}".format(log, k, m.last_value)
        if ((cur_iter != 0 and cur_iter % self.hps['notify_every'] == 0) or end_of_epoch) and self.rank == 0 \
                and os.environ.get("SKF_TRAIN_SLOW_METRICS", "1") != "0":
            log = self.prepare_plot_send_slow_metrics(log)
        if printing and self.rank == 0:
            print(log)
        return log

    def prepare_plot_send_slow_metrics(self, msg):
        """core/models.py:227-235 without the Slack notifier: compute every slow metric, one PNG, their values in the log.
        A failing metric is reported and skipped (the reference's metric workers swallow exceptions, core/metrics.py:127-133)."""
        try:
            self.compute_all_metrics()
            for metric in self.slow_metrics.values():
                if metric is not None:
                    metric.wait()
            self.plot_metrics(self.slow_metrics, "plots_step{}.png".format(self.current_step))
            for sm, metric in self.slow_metrics.items():
                if metric is not None:
                    msg = "{}, {}={}".format(msg, sm, metric.last_value_repr)
        except Exception as e:      # noqa: BLE001
            print("[slow metrics] skipped: {}: {}".format(type(e).__name__, e))
        return msg

    # ---- slow metrics (core/models.py:218-296 of the reference): computed on demand (evaluate-metrics.py /
    # compute_all_metrics) and by the training loop at notify_every / end of epoch; plotting = one PNG per call, no Slack
    def build_slow_metrics(self, names=None):
        from .. import metrics
        names = list(self.slow_metrics) if names is None else names
        built = {m: metrics.build_metric_by_name(m, self.hps) for m in names}
        for m in built:
            if m in self.slow_metrics:
                self.slow_metrics[m] = built[m]
        return built

    def gather_data_for_metric(self, data_type):
        if data_type not in self._metric_inputs:
            gather = getattr(self, "compute_{}".format(data_type), None)
            if gather is None:
                raise AttributeError("One of your metrics requires a compute_{} method".format(data_type))
            self._metric_inputs[data_type] = gather()
        return self._metric_inputs[data_type]

    def compute_metrics_from(self, chosen_metrics):
        self._metric_inputs = {}
        for metric in chosen_metrics.values():
            metric.computation_worker(self.gather_data_for_metric(metric.input_type))
        self._metric_inputs = {}

    def compute_all_metrics(self):
        self._metric_inputs = {}
        for m in [m for m, v in self.slow_metrics.items() if v is None]:
            self.build_slow_metrics([m])
        for metric in self.slow_metrics.values():
            metric.compute_in_parallel(self.gather_data_for_metric(metric.input_type))
        self._metric_inputs = {}

    def plot_metrics(self, metrics_dict, filename):
        """One PNG with every quick-metric history and every ready slow metric (utils/plots.PlotManager in the reference)."""
        import matplotlib
        matplotlib.use("Agg")
        import matplotlib.pyplot as plt
        import numpy as np
        panels = [(m, 'lines', q.history) for m, q in self.quick_metrics.items() if len(q.history) > 1]
        panels += [(m, v.plot_type, v.get_data_for_plot()) for m, v in metrics_dict.items() if v is not None and v.is_ready_for_plot()]
        if not panels:
            return None
        cols = min(3, len(panels))
        rows = (len(panels) + cols - 1) // cols
        fig, axes = plt.subplots(rows, cols, figsize=(5 * cols, 4 * rows), squeeze=False)
        for ax in axes.reshape(-1)[len(panels):]:
            ax.axis("off")
        for ax, (name, kind, data) in zip(axes.reshape(-1), panels):
            ax.set_title(name)
            if kind == 'lines':
                ax.plot(np.asarray(data, dtype=np.float64))
            elif kind == 'scatter':
                d = np.asarray(data, dtype=np.float64)
                ax.scatter(d[:, 0], d[:, 1], c=d[:, 2], s=6, cmap="tab10")
            elif kind == 'hist':
                ax.hist(np.asarray(data, dtype=np.float64).reshape(-1), bins=30)
            elif kind == 'image':
                ax.imshow(plt.imread(data))
                ax.axis("off")
        path = os.path.join(self.plots_out_dir, filename)
        fig.savefig(path, dpi=70)
        plt.close(fig)
        return path

    def plot_and_send_notification_for(self, metrics_list):
        for metric in metrics_list.values():
            metric.wait()
        return self.plot_metrics(metrics_list, "evaluation_plots.png")

    def clean_up_tmp_dir(self):
        for name in os.listdir(self.tmp_out_dir):
            path = os.path.join(self.tmp_out_dir, name)
            if os.path.isfile(path):
                os.unlink(path)

    def prepare_checkpointing(self):
        self._safety = sorted(glob.glob(os.path.join(self.wgt_out_dir, 'ckpt-*.pt')),
                              key=lambda p: int(os.path.basename(p)[5:-3]))

    def prepare_metrics_for_save(self):
        """Hook for models whose running metrics live per rank (data parallelism): called on every rank before rank 0 writes."""

    @abstractmethod
    def state_dict(self):
        pass

    @abstractmethod
    def load_state_dict(self, state):
        pass

    # ---- checkpoints (core/models.py:321-358; own on-disk format).  Data parallel: the replicas are identical, so rank 0
    # alone writes (temp file + atomic rename) and rotates; a barrier on each side keeps the other ranks from running
    # ahead into the next collective while the file is written, and every rank restores from the same file.
    def _save(self, path):
        import torch
        # every rank: queued metric snapshots enter the histories, and (data parallel) the per-rank running-metric accumulators
        # are summed into rank 0's copy so that the file holds the history of ALL ranks (prepare_metrics_for_save)
        self.flush_quick_metrics()
        self.prepare_metrics_for_save()
        self._barrier()
        if self.rank == 0:
            state = self.state_dict()
            state['current_step'] = self.current_step
            tmp = path + '.tmp'
            torch.save(state, tmp)
            os.replace(tmp, path)
        self._barrier()

    def restore_checkpoint_if_exists(self, checkpoint):
        import torch
        if checkpoint is None:
            return
        if checkpoint == 'latest':
            if not self._safety:
                if self.rank == 0:
                    print("[Checkpoint] Not found")
                return
            checkpoint = self._safety[-1]
        if os.path.exists(checkpoint + '.index'):       # a TensorFlow checkpoint prefix written by the reference
            self.load_reference_checkpoint(checkpoint)
            if self.rank == 0:
                print("[Checkpoint] Restored reference (TensorFlow) checkpoint, step #{}".format(self.current_step))
            return
        state = torch.load(checkpoint, map_location='cpu', weights_only=False)
        self.load_state_dict(state)
        self.current_step = int(state['current_step'])
        if self.rank == 0:
            print("[Checkpoint] Restored, step #{}".format(self.current_step))

    def save_checkpoint_if_its_time(self):
        safety_save = max(int(self.hps['safety_save'] * self.batches_per_epoch), 1)
        save_every = max(int(self.hps['save_every'] * self.batches_per_epoch), 1)
        if (self.current_step + 1) % safety_save == 0:
            n = int(os.path.basename(self._safety[-1])[5:-3]) + 1 if self._safety else 1
            path = os.path.join(self.wgt_out_dir, 'ckpt-%d.pt' % n)
            self._save(path)
            self._safety.append(path)
            while len(self._safety) > 2:          # CheckpointManager(max_to_keep=2)
                old = self._safety.pop(0)
                if self.rank == 0:
                    try:
                        os.remove(old)
                    except FileNotFoundError:
                        pass
            if self.rank == 0:
                print('Saving safety checkpoint for step {} at {}'.format(self.current_step + 1, path))
        if (self.current_step + 1) % save_every == 0:
            path = "{}/step{}.pt".format(self.wgt_out_dir, self.current_step)
            self._save(path)
            if self.rank == 0:
                print('Saving fixed checkpoint for step {} at {}'.format(self.current_step + 1, path))
