"""spr_pick_amd — MI355X-native joint denoise + particle-pick hot path of nextpyp/spr_pick.

Python mirrors of the reference's plugin surface (``Denoiser``, ``JointNetwork``,
``non_maximum_suppression``) over hand-written gfx950 kernels in libsprk.so (include/sprk.h), and the
``joint train`` / ``joint eval`` counterpart (``python -m spr_pick_amd``; train.py, eval.py, cli.py).
"""
from . import cfg, params  # noqa: F401
from .algorithms import nms_device, non_maximum_suppression  # noqa: F401
from .datasets import DetectionDataset  # noqa: F401
from .denoiser import Denoiser, PuLoss  # noqa: F401
from .networks import (BasicConv2d, Detector, DualNetwork, DualNetworkShallow, JointNetwork,  # noqa: F401
                       LinearClassifier, ResidA, ResNet8, ShiftConv2d)

__version__ = "0.1.0"
