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

    def __init__(self, hps, dataset, outdir, experiment_id):
        self.hps = hps if isinstance(hps, dict) else dict(hps.values())
        self.dataset = dataset
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
    def train(self, max_steps=None):
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

    def update_quick_metrics_history(self, new_metrics):
        for name, value in new_metrics.items():
            self.quick_metrics[name].append_to_history(value)

    def status_report(self, end_of_epoch=False):
        cur_iter = self.current_step % self.batches_per_epoch
        log = "Epoch {} Batch {}/{}".format(self.epoch, cur_iter, self.batches_per_epoch)
        for k, m in self.quick_metrics.items():
            log = "{}|{}={:4.4f}".format(log, k, m.last_value)
        if (cur_iter % self.hps['log_every'] == 0) or (cur_iter % self.hps['notify_every'] == 0) or end_of_epoch:
            print(log)
        return log

    # ---- checkpoints (core/models.py:321-358; own on-disk format)
    def prepare_checkpointing(self):
        self._safety = sorted(glob.glob(os.path.join(self.wgt_out_dir, 'ckpt-*.pt')),
                              key=lambda p: int(os.path.basename(p)[5:-3]))

    @abstractmethod
    def state_dict(self):
        pass

    @abstractmethod
    def load_state_dict(self, state):
        pass

    def _save(self, path):
        import torch
        state = self.state_dict()
        state['current_step'] = self.current_step
        torch.save(state, path)

    def restore_checkpoint_if_exists(self, checkpoint):
        import torch
        if checkpoint is None:
            return
        if checkpoint == 'latest':
            if not self._safety:
                print("[Checkpoint] Not found")
                return
            checkpoint = self._safety[-1]
        if os.path.exists(checkpoint + '.index'):       # a TensorFlow checkpoint prefix written by the reference
            self.load_reference_checkpoint(checkpoint)
            print("[Checkpoint] Restored reference (TensorFlow) checkpoint, step #{}".format(self.current_step))
            return
        state = torch.load(checkpoint, map_location='cpu', weights_only=False)
        self.load_state_dict(state)
        self.current_step = int(state['current_step'])
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
                os.remove(self._safety.pop(0))
            print('Saving safety checkpoint for step {} at {}'.format(self.current_step + 1, path))
        if (self.current_step + 1) % save_every == 0:
            path = "{}/step{}.pt".format(self.wgt_out_dir, self.current_step)
            self._save(path)
            print('Saving fixed checkpoint for step {} at {}'.format(self.current_step + 1, path))
