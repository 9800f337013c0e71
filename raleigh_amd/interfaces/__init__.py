from .partial_hevp import partial_hevp  # noqa: F401
from .pca import pca, pca_error  # noqa: F401
from .lra import LowerRankApproximation  # noqa: F401
from .truncated_svd import truncated_svd  # noqa: F401
