"""Second witness for the oracle: torch.nn.functional forward + torch.autograd gradients in
float64.  The code lives in oracle/torch_restatement.py (bench.py times the same restatement in fp32 as the CPU baseline)."""
from oracle.torch_restatement import loss_and_grads  # noqa: F401
