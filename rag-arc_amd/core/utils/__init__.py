from .data_model import Document  # noqa: F401
from .fusion import FusionMethod, HipRRFusion, RetrievalResult, RRFusion  # noqa: F401
