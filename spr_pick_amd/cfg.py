"""Default configuration and inference of derived keys — behaviour of the reference's
``spr_pick/cfg.py`` (base :10-43, infer_pipeline :157-171, infer_blindspot :174-188,
infer :191-199, config_name :202-217) restated for the joint pipeline."""
import os

from .params import ConfigValue as C
from .params import DatasetType, NoiseAlgorithm, Pipeline

DEFAULT_RUN_DIR = "hi_runs"


def base():
    return {
        C.ITERATIONS: 200000, C.DETECTLOSS: None, C.TRAIN_MINIBATCH_SIZE: 16, C.TEST_MINIBATCH_SIZE: 1,
        C.IMAGE_CHANNELS: 1, C.TRAIN_PATCH_SIZE: 64, C.LEARNING_RATE: 1e-5,
        C.LR_RAMPDOWN_FRACTION: 0.7, C.LR_RAMPUP_FRACTION: 0.2,
        C.EVAL_INTERVAL: 3200, C.PRINT_INTERVAL: 1280, C.SNAPSHOT_INTERVAL: 3200,
        C.DATALOADER_WORKERS: 4, C.PIN_DATA_MEMORY: False, C.DIAGONAL_COVARIANCE: False,
        C.TRAIN_DATA_PATH: None, C.TRAIN_GT_PATH: None, C.TRAIN_LABEL_PATH: None, C.TRAIN_DATASET_TYPE: None,
        C.TEST_DATA_PATH: None, C.TEST_LABEL_PATH: None, C.TEST_GT_PATH: None, C.TEST_DATASET_TYPE: None,
        C.JOINT_LR: 1e-5, C.ALPHA: 0.8, C.NMS: 15, C.NUM_EVAL: 1, C.NOISE_STYLE: None, C.TAU: 0.01, C.BB: 24,
    }


def infer_pipeline(algorithm):
    if algorithm == NoiseAlgorithm.SELFSUPERVISED_DENOISING:
        return Pipeline.SSDN
    if algorithm in (NoiseAlgorithm.SELFSUPERVISED_DENOISING_MEAN_ONLY, NoiseAlgorithm.NOISE_TO_NOISE,
                     NoiseAlgorithm.NOISE_TO_CLEAN):
        return Pipeline.MSE
    if algorithm == NoiseAlgorithm.NOISE_TO_VOID:
        return Pipeline.MASK_MSE
    raise NotImplementedError("Algorithm does not have a default pipeline.")


def infer_blindspot(algorithm):
    if algorithm in (NoiseAlgorithm.SELFSUPERVISED_DENOISING, NoiseAlgorithm.SELFSUPERVISED_DENOISING_MEAN_ONLY):
        return True
    if algorithm in (NoiseAlgorithm.NOISE_TO_NOISE, NoiseAlgorithm.NOISE_TO_CLEAN, NoiseAlgorithm.NOISE_TO_VOID):
        return False
    raise NotImplementedError("Not known if algorithm requires blindspot.")


def _dataset_type(path):
    if os.path.isdir(path):
        return DatasetType.FOLDER
    return DatasetType.TXT if path.endswith(".txt") else DatasetType.HDF5


def infer_datasets(cfg):
    for data, kind in ((C.TRAIN_DATA_PATH, C.TRAIN_DATASET_TYPE), (C.TEST_DATA_PATH, C.TEST_DATASET_TYPE)):
        if cfg.get(data) is not None and cfg.get(kind) is None:
            cfg[kind] = _dataset_type(cfg[data])


def infer(cfg, model_only=False):
    if cfg.get(C.PIPELINE) is None:
        cfg[C.PIPELINE] = infer_pipeline(cfg[C.ALGORITHM])
    if cfg.get(C.BLINDSPOT) is None:
        cfg[C.BLINDSPOT] = infer_blindspot(cfg[C.ALGORITHM])
    if not model_only:
        infer_datasets(cfg)
    return cfg


def test_length(cfg):
    return cfg[C.NUM_EVAL]


def config_name(cfg):
    cfg = infer(cfg)
    parts = [cfg[C.ALGORITHM].value]
    if cfg[C.PIPELINE] != infer_pipeline(cfg[C.ALGORITHM]):
        parts.append(cfg[C.PIPELINE].value + "_pipeline")
    parts.append(cfg[C.NOISE_STYLE])
    return "-".join(parts)
