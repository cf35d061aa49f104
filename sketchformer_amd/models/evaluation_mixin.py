"""models/evaluation_mixin.py of the reference: the model data the slow metrics ask for
(`predictions_on_validation_set` / `predictions_on_test_set`)."""
import os
import time

import numpy as np


class TransformerMetricsMixin(object):

    def compute_predictions_on_validation_set(self):
        return self.compute_predictions_on_set('valid')

    def compute_predictions_on_test_set(self):
        return self.compute_predictions_on_set('test')

    def compute_predictions_on_set(self, set_type, n_reconstruct=32):
        """(x, all_y, pred_x, pred_y, pred_z, tokenizer, plot_filepath, tmp_filepath, is_continuous): greedy
        reconstructions of 32 seeded validation sketches, class predictions + embeddings of the whole split."""
        bs = self.hps['batch_size']
        x, _ = self.dataset.get_n_samples_from('valid', n=n_reconstruct, shuffled=True, seeded=True)
        pred_x, pred_y, pred_z = [], [], []
        if self.hps['do_reconstruction']:
            L = self.seq_len + 1
            for i in range(0, len(x), bs):
                r = np.asarray(self.predict(x[i:i + bs])['recon'])
                pad = np.zeros((r.shape[0], L) + r.shape[2:], dtype=r.dtype)    # batches stop at different lengths
                pad[:, :r.shape[1]] = r
                pred_x.append(pad)
        all_x, all_y = self.dataset.get_all_data_from(set_type)
        for i in range(0, len(all_x), bs):
            res = self.predict_class(all_x[i:i + bs])
            if res.get('class') is not None:
                pred_y.append(res['class'])
            pred_z.append(res['embedding'])
        cat = lambda parts: np.concatenate(parts, axis=0) if parts else np.zeros((0,))  # noqa: E731
        stamp = time.strftime("%Y%m%d-%H%M%S")
        return (x, np.asarray(all_y).reshape(-1), cat(pred_x), cat(pred_y), cat(pred_z),
                None if self.dataset.hps['use_continuous_data'] else self.dataset.tokenizer,
                os.path.join(self.plots_out_dir, stamp + "_{}.svg"), os.path.join(self.tmp_out_dir, "converted_{}.png"),
                self.dataset.hps['use_continuous_data'])
