"""core/experiments.py of the reference: long-running evaluations with their own hparams; subclasses give `name`,
`requires_model`, `specific_default_hparams()` and `compute(model)`.  (No Slack notifier here.)"""
import os

from ..utils import hparams as hp


class Experiment(object):

    @classmethod
    def base_default_hparams(cls):
        return hp.HParams(slack_config='token.secret')

    def __init__(self, hps, experiment_sub_id, outdir):
        self.hps = hps if isinstance(hps, dict) else dict(hps.values())
        if not hasattr(self, 'name'):
            raise Exception("You must give your experiment a reference name")
        if not hasattr(self, 'requires_model'):
            raise Exception("You must advertise if your experiment requires a trained model")
        self.identifier = "{}-{}".format(self.name, experiment_sub_id)
        self.out_dir = os.path.join(outdir, self.identifier)
        os.makedirs(self.out_dir, exist_ok=True)

    @classmethod
    def default_hparams(cls):
        return hp.combine_hparams_into_one(cls.specific_default_hparams(), cls.base_default_hparams())

    @classmethod
    def parse_hparams(cls, new_hps):
        hps = cls.default_hparams()
        if new_hps is not None:
            hps = hps.parse(new_hps)
        return hps

    @classmethod
    def specific_default_hparams(cls):
        raise NotImplementedError

    def compute(self, model=None):
        raise NotImplementedError
