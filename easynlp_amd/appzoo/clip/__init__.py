from .data import DevicePrefetcher  # noqa: F401
from .evaluator import CLIPEvaluator  # noqa: F401
from .model import CLIPApp  # noqa: F401
from .predictor import CLIPPredictor  # noqa: F401
