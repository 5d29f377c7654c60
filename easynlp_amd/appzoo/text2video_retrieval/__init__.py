from .evaluator import Text2VideoRetrievalEvaluator  # noqa: F401
from .model import Text2VideoRetrieval  # noqa: F401
