"""builders/keras_metrics.py of the reference (:13-42): running Mean / SparseCategoricalAccuracy."""
import torch


class _Mean(object):
    def __init__(self):
        self.reset_states()

    def reset_states(self):
        self.total, self.count = 0.0, 0.0

    def __call__(self, value):
        self.total += float(value)
        self.count += 1.0

    def result(self):
        return self.total / self.count if self.count else 0.0


class _SparseCategoricalAccuracy(_Mean):
    def __call__(self, y_true, y_pred):
        pred = torch.as_tensor(y_pred).argmax(-1).reshape(-1)
        true = torch.as_tensor(y_true, device=pred.device).reshape(-1)
        self.total += float((pred == true).sum())
        self.count += float(true.numel())


class MetricManager(object):
    def __init__(self):
        self.metric_names, self.metric_fns = [], {}

    def add_mean_metric(self, name):
        self.metric_names.append(name)
        self.metric_fns[name] = _Mean()

    def add_sparse_categorical_accuracy(self, name):
        self.metric_names.append(name)
        self.metric_fns[name] = _SparseCategoricalAccuracy()

    def compute(self, name, *args):
        assert name in self.metric_names, 'Error! {} metric not found.'.format(name)
        self.metric_fns[name](*args)

    def reset(self):
        for m in self.metric_fns.values():
            m.reset_states()

    def get_results(self):
        return [self.metric_fns[m].result() for m in self.metric_names]

    def get_results_as_dict(self):
        return {m: self.metric_fns[m].result() for m in self.metric_names}
