from .data import WukongCLIPDataset  # noqa: F401
from .evaluator import WukongCLIPEvaluator  # noqa: F401
from .model import WukongCLIP  # noqa: F401
from .predictor import WukongCLIPPredictor  # noqa: F401
from .tokenizer import FullTokenizer  # noqa: F401
