"""val-clas-acc / test-clas-acc (metrics/classification.py of the reference): accuracy of predict_class on a split."""
import numpy as np

from ..core.metrics import HistoryMetric


def _accuracy(y, pred):
    return float(np.mean(np.asarray(y).reshape(-1) == np.asarray(pred).reshape(-1)))


class PrecomputedValidationAccuracy(HistoryMetric):
    name = 'val-clas-acc'
    input_type = 'predictions_on_validation_set'

    def compute(self, input_data):
        x, y, pred_x, pred_y, pred_z, tokenizer, plot_filepath, tmp_filepath, _ = input_data
        return _accuracy(y, pred_y)


class PrecomputedTestAccuracy(HistoryMetric):
    name = 'test-clas-acc'
    input_type = 'predictions_on_test_set'

    def compute(self, input_data):
        x, y, pred_x, pred_y, pred_z, tokenizer, plot_filepath, tmp_filepath, _ = input_data
        return _accuracy(y, pred_y)
