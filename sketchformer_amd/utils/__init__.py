from . import hparams  # noqa: F401
from .tokenizer import GridTokenizer, Tokenizer  # noqa: F401
