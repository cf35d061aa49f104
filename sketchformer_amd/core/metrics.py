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


# ---------------------------------------------------------------------------------------------------------------
# Slow metrics (core/metrics.py:53-293 of the reference): computed now and then from the current model, in a worker
# thread; a metric names the model data it needs (`input_type` -> BaseModel.compute_<input_type>()) and implements
# `compute(input_data)`.  Exceptions in a worker are printed and swallowed like in the reference (:127-133).
import pickle          # noqa: E402
import threading       # noqa: E402

import numpy as np     # noqa: E402


class SlowMetric(object):
    plot_type = None
    input_type = None
    _state_attr = None                       # name of the attribute that holds the latest result

    def __init__(self, params):
        self.hps = params
        self.thread = threading.Thread()
        self.thread.start()

    def compute_in_parallel(self, input_data):
        self.thread.join()                   # one computation of a metric at a time
        self.thread = threading.Thread(target=self.computation_worker, args=(input_data,))
        self.thread.start()

    def wait(self):
        self.thread.join()

    def compute(self, input_data):
        raise NotImplementedError

    def _fallback(self):
        return getattr(self, self._state_attr)

    def computation_worker(self, input_data):
        try:
            result = self.compute(input_data)
        except Exception as e:  # noqa: BLE001
            print("Exception while computing metrics: {}".format(repr(e)))
            result = self._fallback()
        self._store(result)

    def _store(self, result):
        setattr(self, self._state_attr, result)

    def get_data_for_plot(self):
        return getattr(self, self._state_attr)

    def is_ready_for_plot(self):
        return getattr(self, self._state_attr) is not None

    @property
    def last_value_repr(self):
        return 'plotted' if self.is_ready_for_plot() else 'waiting'

    def save(self, filepath):
        if self.is_ready_for_plot():
            with open(filepath, 'wb') as f:
                pickle.dump({self._state_attr: getattr(self, self._state_attr)}, f)

    def load(self, filepath):
        with open(filepath, 'rb') as f:
            setattr(self, self._state_attr, pickle.load(f)[self._state_attr])


class HistoryMetric(SlowMetric):
    plot_type = 'lines'
    _state_attr = 'last_value'

    def __init__(self, params):
        super().__init__(params)
        self.history, self.last_value = [], None

    def _fallback(self):
        return 0 if self.last_value is None else self.last_value

    def _store(self, result):
        self.last_value = result
        self.history.append(result)

    def get_data_for_plot(self):
        return self.history

    def is_ready_for_plot(self):
        return bool(self.history)

    @property
    def last_value_repr(self):
        return str(self.last_value) if self.last_value is not None else 'waiting'

    def save(self, filepath):
        if self.last_value is not None:
            with open(filepath, 'wb') as f:
                pickle.dump({"last_value": self.last_value, "history": self.history}, f)

    def load(self, filepath):
        with open(filepath, 'rb') as f:
            d = pickle.load(f)
        self.last_value, self.history = d["last_value"], d["history"]


class ProjectionMetric(SlowMetric):
    plot_type = 'scatter'
    _state_attr = 'current_projection'

    def __init__(self, params):
        super().__init__(params)
        self.current_projection = None

    def _fallback(self):
        return np.array([[0., 0., 0.]]) if self.current_projection is None else self.current_projection


class ImageMetric(SlowMetric):
    plot_type = 'image'
    _state_attr = 'current_image'

    def __init__(self, params):
        super().__init__(params)
        self.current_image = None

    @property
    def last_value_repr(self):
        return 'image-grid' if self.is_ready_for_plot() else 'waiting'


class HistogramMetric(SlowMetric):
    plot_type = 'hist'
    _state_attr = 'current_hist'

    def __init__(self, params):
        super().__init__(params)
        self.current_hist = None

    def _fallback(self):
        return np.array([0.]) if self.current_hist is None else self.current_hist
