"""builders/losses.py of the reference (:9-85): named, weighted losses; computed by the fused softmax-CE kernel."""
import torch

from .. import ops


class LossManager(object):
    def __init__(self):
        self.loss_names, self.loss_fns, self.loss_weights = [], {}, {}

    def _add_loss(self, name, weight, func):
        self.loss_names.append(name)
        self.loss_weights[name] = weight
        self.loss_fns[name] = func

    def add_sparse_categorical_crossentropy(self, name='class', weight=1.0):
        def fn(labels, probs_or_logits):
            # the reference feeds the softmax output; in graph mode TF recovers the logits, so logits are expected here
            x = probs_or_logits.detach().clone().contiguous().view(-1, probs_or_logits.shape[-1])
            lab = torch.as_tensor(labels, device=x.device).view(-1, 1).to(torch.int64).contiguous()
            row_loss, _, _ = ops.softmax_ce(x, lab, tgt_cols=1, write_grad=False)
            return row_loss.mean()
        self._add_loss(name, weight, fn)

    def add_reconstruction_loss(self, name='recon', weight=1.0):
        def fn(real, pred):
            x = pred.detach().clone().contiguous().view(-1, pred.shape[-1])
            tgt = torch.as_tensor(real, device=x.device).to(torch.int64).contiguous()
            row_loss, _, _ = ops.softmax_ce(x, tgt, tgt_cols=tgt.shape[1], mask_pad=True, write_grad=False)
            return row_loss.sum() / row_loss.numel()       # mean over ALL positions, padded ones contribute 0
        self._add_loss(name, weight, fn)

    def add_continuous_reconstruction_loss(self, name='recon', weight=1.0):
        def fn(real, pred):
            import ctypes as C
            from .. import _lib
            p = pred.detach().clone().to(torch.float32).contiguous().view(-1, 5)
            t = torch.as_tensor(real, device=p.device).to(torch.float32).contiguous()
            rows = p.shape[0]
            buf = [torch.empty(rows, dtype=torch.float32, device=p.device) for _ in range(3)]
            scal = torch.zeros(4, dtype=torch.float32, device=p.device)
            _lib.call("skf_continuous_loss", C.c_void_p(p.data_ptr()), C.c_void_p(t.data_ptr()), t.shape[1], t.shape[1], 0,
                      rows, 1.0, *(C.c_void_p(b.data_ptr()) for b in buf), C.c_void_p(scal.data_ptr()), 0,
                      C.c_void_p(torch.cuda.current_stream().cuda_stream))
            return scal[3]
        self._add_loss(name, weight, fn)

    # builders/losses.py:68-75.  tf.keras.losses.MAE / MSE reduce the last axis and keep the others; tf.reduce_mean reduces everything.
    def add_mae_loss(self, name, weight=1.):
        self._add_loss(name, weight, lambda y_true, y_pred: ops.row_mean(torch.as_tensor(y_pred), y_true, mode=1))

    def add_mean_loss(self, name, weight=1.):
        def fn(x):
            x = torch.as_tensor(x).detach().to(torch.float32).contiguous().view(1, -1)
            return ops.row_mean(x)[0]
        self._add_loss(name, weight, fn)

    def add_mse_loss(self, name, weight=1.):
        self._add_loss(name, weight, lambda y_true, y_pred: ops.row_mean(torch.as_tensor(y_pred), y_true, mode=2))

    def compute_all_loss(self, rp_dict):
        return {n: self.loss_weights[n] * self.loss_fns[n](*rp_dict[n]) for n in self.loss_names}

    def compute_loss(self, name, *args):
        assert name in self.loss_names, "Error! Loss name {} not found".format(name)
        return self.loss_weights[name] * self.loss_fns[name](*args)
