"""pssr2_amd — MI355X-native hot path of PSSR2 (ResUNet train/infer, MS-SSIM+L1 loss, crappifiers)."""
__version__ = "0.1.0"

from .crappifiers import AdditiveGaussian, Blur, Crappifier, MultiCrappifier, Poisson, SaltPepper  # noqa: E402,F401


def __getattr__(name):
    # torch-dependent symbols are imported lazily so that `import pssr2_amd` stays cheap
    if name in ("ResUNet",):
        from .models import ResUNet
        return ResUNet
    if name in ("RDResUNet",):
        from .models import RDResUNet
        return RDResUNet
    if name in ("ResUNetA", "RDResUNetA"):
        from . import models
        return getattr(models, name)
    if name in ("SSIMLoss",):
        from .util import SSIMLoss
        return SSIMLoss
    if name in ("train_paired",):
        from .train import train_paired
        return train_paired
    if name in ("predict_images",):
        from .predict import predict_images
        return predict_images
    if name in ("FusedAdamW",):
        from .optim import FusedAdamW
        return FusedAdamW
    raise AttributeError(name)
