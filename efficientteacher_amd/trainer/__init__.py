from .ssod_trainer import SSODTrainer  # noqa: F401
from .trainer import Trainer  # noqa: F401
