"""Name registry of the hot path (mirror of the reference's model_zoo package)."""
from .base_model import BaseModel  # noqa: F401
from .deepctr import DeepCTR  # noqa: F401
from .deep_mtl_ctr import DeepMTLCTR  # noqa: F401
from .maml import MAML  # noqa: F401
from .reptile import Reptile  # noqa: F401
from .domain_negotiation import DomainNegotiation  # noqa: F401
from .mamdr import MAMDR  # noqa: F401
from .star import Star  # noqa: F401
from .mldg import MLDG  # noqa: F401
from .uncertainty_weight import UncertaintyWeight  # noqa: F401
from .pcgrad import PCGrad  # noqa: F401
