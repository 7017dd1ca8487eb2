from .resnet import *
from . import resnet

__all__ = resnet.__all__
