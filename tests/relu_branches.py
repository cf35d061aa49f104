"""ReLU branches of the device for oracle runs (test infrastructure).

relu(z) has a kink at z = 0: a pre-activation within fp32 rounding distance of 0 takes one branch in the float64 oracle and
possibly the other on the device, and that ONE hidden unit then shifts a whole column of dW1 by x[r,:]*dh[r,j] - far above
the 1e-3 gradient bar at a 800-row batch, although both evaluations are correct to rounding.  (Observed at the BASELINE
sizes: about one unit in 10^6.)  The parity tests therefore hand the oracle the branch the device took - exactly as they
hand it the device's dropout keep-masks - and check that every unit where that differs from the oracle's own `pre > 0`
really sits on the kink."""
import numpy as np

import oracle


class device_relu_branches:
    """with device_relu_branches(eng, ocfg, B) as chk: <run the oracle>; chk.flips = units overridden on the kink."""

    def __init__(self, eng, ocfg, B, kink=2e-5, max_flips=64):
        # (a bf16 model evaluates the pre-activations to ~2^-8 relative: pass kink ~ 5e-2 and a max_flips fraction there)
        self.eng, self.ocfg, self.B, self.kink, self.max_flips = eng, ocfg, B, kink, max_flips
        self.flips = 0

    def __enter__(self):
        c = self.ocfg
        oracle.RELU_MASKS.clear()
        oracle.RELU_PRE.clear()
        for side, L in (("encoder", c.seq_len), ("decoder", c.seq_len - 1)):
            if side == "decoder" and not c.do_reconstruction:
                continue
            for i in range(c.num_layers):
                h = self.eng.buffer("%s/layer%d/ffn_h" % (side, i)).float().cpu().numpy()
                oracle.RELU_MASKS["%s/layer%d/ffn" % (side, i)] = (h > 0).reshape(self.B, L, c.dff)
        return self

    def __exit__(self, *exc):
        try:
            if exc[0] is None:
                for k, mask in oracle.RELU_MASKS.items():
                    pre = oracle.RELU_PRE[k]
                    diff = mask != (pre > 0)
                    self.flips += int(diff.sum())
                    # a unit may only differ where the oracle's own pre-activation is numerically zero
                    assert np.all(np.abs(pre[diff]) < self.kink * max(1.0, np.abs(pre).max())), \
                        (k, int(diff.sum()), float(np.abs(pre[diff]).max()))
                assert self.flips <= self.max_flips, self.flips
        finally:
            oracle.RELU_MASKS.clear()
            oracle.RELU_PRE.clear()
        return False
