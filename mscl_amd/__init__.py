"""mscl_amd -- MI355X-native MSCL training hot path.

Importing the package registers the reference's model names (MSCLWithAug, MoCoV2, TPNMoCo, BaseMoCo, MoCoHead,
MSCLWithAugMxHead, MSCLWithAugPosHeadV2, CrossEntropyLoss_torch, SyncMoCoAugmentV5, IdentityAug; Recognizer3D, I3DHead,
CrossEntropyLoss for the fine-tune / evaluation consumer)."""
__version__ = '0.1.0'

from . import augment, heads, necks, recognizer3d, recognizers          # noqa: F401  (registration side effects)
from .config import Config                                # noqa: F401
from .optim import ClipSGD                                # noqa: F401
from .registry import (BACKBONES, HEADS, LOSSES, NECKS, RECOGNIZERS, SSL_AUGS, build_head, build_loss,   # noqa: F401
                       build_model, build_neck, build_recognizer, build_ssl_aug)
