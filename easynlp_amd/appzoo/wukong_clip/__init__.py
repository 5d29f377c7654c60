from .evaluator import WukongCLIPEvaluator  # noqa: F401
from .model import WukongCLIP  # noqa: F401
