"""camal_amd — MI355X-native implementation of CamAL's detection + CAM-localization hot path.

Public surface mirrors the reference (adrienpetralia/CamAL):
    from camal_amd import CamAL, CamALResNet, CAM
Everything heavy is imported lazily so that ``import camal_amd`` works without a GPU.
"""
__all__ = ["CamAL", "CamALResNet", "ResNet", "CAM", "Ensemble", "NILMmetrics", "Classifmetrics", "BasedClassifTrainer", "TSLoader",
           "TSDataset", "NILMscaler"]


def __getattr__(name):
    if name == "CamAL":
        from .core import CamAL
        return CamAL
    if name == "CamALResNet":
        from .resnet import CamALResNet
        return CamALResNet
    if name == "ResNet":
        from .resnet import ResNet
        return ResNet
    if name == "CAM":
        from .cam import CAM
        return CAM
    if name == "Ensemble":
        from .engine import Ensemble
        return Ensemble
    if name in ("NILMmetrics", "Classifmetrics"):
        from . import metrics
        return getattr(metrics, name)
    if name in ("BasedClassifTrainer", "TSLoader", "TSDataset"):
        from . import trainer
        return getattr(trainer, name)
    if name == "NILMscaler":
        from .windows import NILMscaler
        return NILMscaler
    raise AttributeError(name)
