"""Front-ends with the names and signatures of the reference's ``builders`` package, backed by the HIP
kernels of libskf.so (no CPU path).  Tensors are torch CUDA(HIP) tensors."""
from . import keras_metrics, losses, schedulers, utils  # noqa: F401
from . import layers  # noqa: F401
