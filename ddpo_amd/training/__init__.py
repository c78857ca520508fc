from . import prompts, callbacks  # noqa: F401
from .prompts import make_prompts  # noqa: F401
from .callbacks import callback_fns, evaluate_callbacks  # noqa: F401
