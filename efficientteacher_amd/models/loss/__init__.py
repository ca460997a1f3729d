from .loss import ComputeLoss, DomainLoss, TargetLoss  # noqa: F401
from .ssod.ssod_loss import ComputeStudentMatchLoss  # noqa: F401
from .tal_loss import ComputeTalLoss  # noqa: F401


def build_ssod_loss(model, cfg):
    # reference models/loss/__init__.py: build_ssod_loss dispatches on cfg.SSOD.loss_type
    if cfg.SSOD.loss_type == 'ComputeStudentMatchLoss':
        if cfg.Loss.type == 'ComputeTalLoss':       # EXTENSION: the anchor-free head (no counterpart in the reference, see there)
            from .ssod.tal_ssod_loss import ComputeStudentMatchTalLoss
            return ComputeStudentMatchTalLoss(model, cfg)
        return ComputeStudentMatchLoss(model, cfg)
    raise NotImplementedError(cfg.SSOD.loss_type)
