from .common.get_model import get_model, save_state_dict  # noqa: F401
from .common.utils import get_unit  # noqa: F401
