"""QuickMetric: per-step history of one scalar (core/metrics.py:7-52 of the reference).
The threaded SlowMetric family (plots, t-SNE, sklearn) is outside the accelerated path."""


class QuickMetric(object):
    def __init__(self):
        self.history = []

    @property
    def last_value(self):
        return self.history[-1] if self.history else 0.0

    @property
    def last_value_repr(self):
        return "{:4.4f}".format(self.last_value)

    def append_to_history(self, new_value):
        self.history.append(float(new_value))

    def get_mean_of_latest(self, n=100):
        h = self.history[-n:]
        return sum(h) / len(h) if h else 0.0

    def save(self, directory):
        pass
