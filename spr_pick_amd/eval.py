"""``DenoiserEvaluator`` — ``joint eval`` (eval.py:29-144 of the reference): load a ``.training`` or
``.wt`` file, run every requested micrograph through the filled network and write, per micrograph,
the image outputs and ``{name}_scores.txt`` into ``<runs_dir>/%05d-eval-<cfg>/eval_imgs``.
With WORLD_SIZE > 1 micrograph i is processed by rank i % world; there is no collective."""
import logging
import os

from . import cfg as cfg_mod
from . import checkpoint
from .denoiser import Denoiser
from .datasets import DetectionDataset
from .params import PipelineOutput
from .train import DenoiserTrainer, setup_logging
from .utils import separator

logger = logging.getLogger("joint.eval")


class DenoiserEvaluator(DenoiserTrainer):
    def __init__(self, target_path, runs_dir=cfg_mod.DEFAULT_RUN_DIR, run_dir=None, device=None):
        super().__init__({}, "joint", runs_dir=runs_dir, run_dir=run_dir, device=device)
        state_dict = checkpoint.load(target_path)
        if "denoiser" in state_dict:
            self.load_state_dict(state_dict, restore_optimizer=False)
        else:
            self.denoiser = Denoiser.from_state_dict(state_dict, mode=self.mode, device=self.device)
        self.cfg = self.denoiser.cfg
        self.init_state()

    def evaluate(self):
        self.reset_metrics(train=False)
        if self.denoiser is None:
            raise RuntimeError("Denoiser not initialised for evaluation")
        os.makedirs(self.run_dir_path, exist_ok=True)
        if self.rank == 0:
            setup_logging(self.run_dir_path)
        logger.info(separator())
        logger.info("Loading Test Dataset...")
        self.testfeed = self.test_data()
        logger.info("Loaded Test Dataset.")
        logger.info(separator())
        logger.info("EVALUATION STARTED")
        logger.info(separator())
        self._evaluate(self.testfeed, self.evaluation_output_callback())
        logger.info(self.eval_state_str("EVALUATION RESULT"))
        logger.info(separator())
        logger.info("EVALUATION FINISHED")
        logger.info(separator())

    @property
    def run_dir(self):
        if self._run_dir is None:
            self._run_dir = "{:05d}-eval-{}".format(self.next_run_id(), self.config_name())
            if self.world > 1:
                import torch
                box = [self._run_dir]
                torch.distributed.broadcast_object_list(box, src=0)
                self._run_dir = box[0]
        return self._run_dir

    def evaluation_output_callback(self):
        def callback(output_0_index, outputs):
            inp = outputs[PipelineOutput.INPUTS][DetectionDataset.INPUT]
            output_dir = os.path.join(self.run_dir_path, "eval_imgs")
            self.save_image_outputs(outputs, output_dir, "{name}_{desc}.png", "{name}_{desc}.txt",
                                    batch_indexes=range(inp.shape[0]))
        return callback
