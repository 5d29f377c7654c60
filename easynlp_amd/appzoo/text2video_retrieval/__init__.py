from .data import Text2VideoRetrievalDataset  # noqa: F401
from .evaluator import Text2VideoRetrievalEvaluator  # noqa: F401
from .model import Text2VideoRetrieval  # noqa: F401
from .predictor import Text2VideoRetrievalPredictor  # noqa: F401
