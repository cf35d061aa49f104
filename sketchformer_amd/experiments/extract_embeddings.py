"""extract-embeddings (experiments/extract_embeddings.py of the reference): greedy reconstructions of n validation
sketches + class predictions and embeddings of a whole split, saved as one .npz with the reference's keys."""
import os

import numpy as np

from ..core.experiments import Experiment
from ..metrics.samples import stroke5_to_stroke3
from ..utils import hparams as hp


class ExtractSketchEmbeddings(Experiment):
    name = "extract-embeddings"
    requires_model = True

    @classmethod
    def specific_default_hparams(cls):
        return hp.HParams(batch_size=256, target_file='embeddings.npz', set_type='valid', n_samples_to_reconstruct=32)

    def compute(self, model=None):
        bs = min(self.hps['batch_size'], model.hps['batch_size'])          # the engine's batch is its capacity per call
        x, _ = model.dataset.get_n_samples_from('valid', n=self.hps['n_samples_to_reconstruct'], shuffled=True, seeded=True)
        recon = []
        for i in range(0, len(x), bs):
            recon.extend(list(model.predict(x[i:i + bs])['recon']))
        all_x, all_y = model.dataset.get_all_data_from(self.hps['set_type'])
        pred_y, pred_z = [], []
        for i in range(0, len(all_x), bs):
            res = model.predict_class(all_x[i:i + bs])
            pred_y.append(res['class'])
            pred_z.append(res['embedding'])
        if model.dataset.hps['use_continuous_data']:
            sk = [stroke5_to_stroke3(s) for s in x]
            rsk = [stroke5_to_stroke3(s[1:]) for s in recon]
        else:
            tok = model.dataset.tokenizer
            sk = [tok.decode_single(s) for s in x]
            rsk = [tok.decode_single(s) for s in recon]
        obj = lambda lst: np.array(lst + [None], dtype=object)[:-1]          # ragged lists -> object arrays  # noqa: E731
        target = self.hps['target_file']
        if not os.path.isabs(target):
            target = os.path.join(self.out_dir, target)
        np.savez(target, y=np.asarray(all_y).reshape(-1), sketches=obj(sk), recon_sketches=obj(rsk),
                 pred_y=np.concatenate(pred_y, axis=0), embeddings=np.concatenate(pred_z, axis=0))
        return target
