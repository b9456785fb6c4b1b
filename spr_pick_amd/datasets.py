"""Batch layout constants of the reference's ``DetectionDataset`` (datasets/image_wrapper.py:17-22,
:279-289): the list a DataLoader hands to ``Denoiser.run_pipeline``."""
from enum import Enum, auto


class DetectionDataset:
    INPUT = 0
    TARGET = 1
    HM = 2
    HM_SMALL = 3
    METADATA = -1

    class Metadata(Enum):
        IMAGE = auto()
        HM = auto()
        HM_SMALL = auto()
        IMAGE_SHAPE = auto()
        AUG_IMG = auto()
        INDEXES = auto()
        TARGET = auto()
        GT = auto()
        NAME = auto()

    @staticmethod
    def make_batch(inp, target, hm=None, hm_small=None, metadata=None):
        """Assemble the list layout from bare tensors (used by the bench / tests)."""
        import torch
        hm = hm if hm is not None else torch.zeros(0)
        hm_small = hm_small if hm_small is not None else torch.zeros(0)
        md = {DetectionDataset.Metadata.GT: [], DetectionDataset.Metadata.INDEXES: None,
              DetectionDataset.Metadata.IMAGE_SHAPE: None}
        if metadata:
            md.update(metadata)
        return [inp, target, hm, hm_small, md]
