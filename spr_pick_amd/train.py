"""``DenoiserTrainer`` — the iteration loop behind ``joint train start|resume``,
restating spr_pick/train.py with the data side on the device:

* one iteration == one image; the interval checks (eval / print / snapshot) run at the top of
  the loop on ``iteration % interval == 0``, the loop leaves once ``iteration >= ITERATIONS``
  (train.py:161-373);
* learning rate AS EXECUTED by the reference (train.py:430-442): ``compute_ramped_lrate(i, ITERATIONS,
  LR_RAMPDOWN_FRACTION, LR_RAMPUP_FRACTION, 1e-4)`` — the two fractions arrive swapped, so the ramp-up
  covers the first 70 % and the ramp-down the last 20 %, and the base rate is 1e-4 whatever ``--lr``
  says; Adam betas (0.9, 0.99) (train.py:128-140);
* run directory ``%05d-train-<alg>-<noise_style>-iter<k>-<alpha>-<tau>-<mode>`` under ``runs_dir``
  (train.py:846-961), checkpoints ``training_jt/model_%08d.training``, final weights
  ``final-<cfg>.wt``, validation outputs ``val_imgs_joint/{name}_{iter:08}_{desc}.png|txt``
  (train.py:444-585).

Deviations, all deliberate: patches come from the device feed (spr_pick_amd.feed) instead of a
DataLoader; metrics go to ``metrics.tsv`` (tensorboard is not in this image); ``resume`` works
(the reference's resume_run omits the required ``mode`` argument and its CLI then reads an
undefined ``cfg``); with WORLD_SIZE > 1 every rank trains on TRAIN_MINIBATCH_SIZE / world patches
of its own sampler stream and gradients are averaged in one RCCL all-reduce — the iteration
counter advances by the global batch so schedules match a single-process run."""
import glob
import logging
import math
import os
import re
import time
from collections import defaultdict

import numpy as np
import torch

from . import cfg as cfg_mod
from . import checkpoint, distributed, feed as feed_mod, graph_step, outputs as outputs_mod, picks
from .algorithms import nms_device
from .datasets import DetectionDataset
from .denoiser import Denoiser
from .params import ConfigValue, DatasetType, HistoryValue, Pipeline, PipelineOutput, StateValue
from .utils import Metric, MetricDict, TrackedTime, compute_ramped_lrate, seconds_to_dhms, separator

logger = logging.getLogger("joint.train")
BASE_LR = 1e-4
NMS_THRESHOLD = 0.02
EVAL_NOISE_SEED = 7_000_000      # + micrograph index = seed of that micrograph's evaluation noise stream


def setup_logging(run_dir_path, filename="log.txt"):
    root = logging.getLogger("joint")
    root.setLevel(logging.INFO)
    for h in list(root.handlers):
        root.removeHandler(h)
    fmt = logging.Formatter("%(asctime)s %(levelname)-8s %(message)s")
    for h in (logging.StreamHandler(), logging.FileHandler(os.path.join(run_dir_path, filename))):
        h.setFormatter(fmt)
        root.addHandler(h)


tensor_to_png = outputs_mod.tensor_to_png      # save_tensor_image (utils/data.py:71-93,143-147), see outputs.py


class DenoiserTrainer:
    def __init__(self, cfg, mode, state=None, runs_dir=cfg_mod.DEFAULT_RUN_DIR, run_dir=None, alpha=0.5, tau=0.01,
                 bb=32, device=None, seed=0, graph=True):
        self.runs_dir = os.path.abspath(runs_dir)
        self._run_dir = run_dir
        self.cfg = cfg
        if self.cfg:
            cfg_mod.infer(self.cfg)
        self.state = state if state is not None else {}
        self.mode = mode
        self.alpha, self.tau, self.bb = alpha, tau, bb
        self.seed = seed
        self.rank, self.world, local = distributed.init_from_env()
        self.device = torch.device(device) if device else torch.device("cuda", local)
        if self.device.type == "cuda" and torch.cuda.is_available():
            torch.cuda.set_device(self.device)      # one process drives one GPU (kernels go to ITS streams)
        self._denoiser = None
        self._optimizer = None
        self._stepper = None
        self.graph = graph and os.environ.get("SPRK_GRAPH", "1") != "0"
        self._metrics_file = None
        self._eval_modes_logged = set()
        self.trainfeed, self.testfeed = None, None
        self._train_mode = False
        self.writer = outputs_mod.OutputWriter()       # PNG / score files leave the loop through a thread pool
        self.timing = {}                                # wall clocks of the phases (seconds), for reports

    # ---- model / optimiser ---------------------------------------------------------------------
    @property
    def denoiser(self):
        return self._denoiser

    @denoiser.setter
    def denoiser(self, denoiser):
        self._denoiser = denoiser
        # operand precision of the U-Nets' MFMA convolutions: not a flag of the reference's CLI, so the drop-in
        # command line stays as it is and the choice travels in the environment (f32 | bf16 | f16; BASELINE configs[4])
        dt = os.environ.get("SPRK_CONV_DTYPE", "f32")
        if dt != "f32":
            n = denoiser.set_conv_dtype(dt)
            logger.info("MFMA operand precision %s requested for %d convolution layers (SPRK_CONV_DTYPE)", dt, n)
        self.init_optimiser()

    def init_optimiser(self):
        for _, p in self.denoiser.named_parameters():
            p.requires_grad = True
        params = [p for p in self.denoiser.parameters() if p.requires_grad]
        # Adam(beta = (0.9, 0.99)) as train.py:128-140, fused + capturable (state and learning rate on the device)
        self._optimizer = graph_step.make_adam(params, lr=BASE_LR, betas=(0.9, 0.99))
        self._stepper = None      # built at the first training step (needs the per-rank batch shape)

    def new_target(self):
        torch.manual_seed(self.seed)            # same initial weights on every rank
        self.denoiser = Denoiser(self.cfg, device=self.device, mode=self.mode)
        self.init_state()
        self.seed_streams()

    def seed_streams(self):
        """Per-rank random streams for everything drawn AFTER the (identical) initial weights: the
        reparameterisation noise on the device and the flip-axis draw of the pipeline (global NumPy state).
        On resume the streams are keyed by the sample counter too, so a resumed run does not replay the
        draws (and, through PatchFeed's seed, the patches) of its first segment."""
        it = int(self.state.get(StateValue.ITERATION, 0) or 0)
        s = (self.seed + 1000 + self.rank + 7919 * it) % (2 ** 31 - 1)
        # only the streams that need a per-rank / per-segment key: torch's CPU generator is left alone (new run: it
        # continues from the weight initialisation; resume: it continues from the checkpoint's "rng" entry, as the
        # reference restores it, train.py:936)
        torch.cuda.manual_seed(s)
        np.random.seed(s)

    def init_state(self):
        self.state[StateValue.INITIALISED] = True
        self.state[StateValue.ITERATION] = 0
        self.state[StateValue.HISTORY] = {HistoryValue.TRAIN: MetricDict(), HistoryValue.EVAL: MetricDict(),
                                          HistoryValue.TIMINGS: defaultdict(TrackedTime)}
        self.reset_metrics()

    @property
    def learning_rate(self):
        c = self.cfg
        return compute_ramped_lrate(self.state[StateValue.ITERATION], c[ConfigValue.ITERATIONS],
                                    c[ConfigValue.LR_RAMPDOWN_FRACTION], c[ConfigValue.LR_RAMPUP_FRACTION], BASE_LR)

    @property
    def optimizer(self):
        lr = self.learning_rate
        graph_step.set_lr(self._optimizer, lr)
        return self._optimizer

    # ---- the loop ------------------------------------------------------------------------------
    def train(self):
        if self.mode not in ("joint", "denoise"):
            raise NotImplementedError("Unsupported training mode: %r" % self.mode)
        if self.denoiser is None:
            self.new_target()
        denoiser = self.denoiser
        t_setup = time.perf_counter()
        os.makedirs(self.run_dir_path, exist_ok=True)
        if self.rank == 0:
            setup_logging(self.run_dir_path)
        logger.info(separator())
        logger.info("Loading Training Dataset...")
        self.trainfeed = self.train_data()
        logger.info("Loaded Training Dataset.")
        if self.cfg[ConfigValue.TEST_DATA_PATH]:
            logger.info("Loading Validation Dataset...")
            self.testfeed = self.test_data()
            logger.info("Loaded Validation Dataset.")
        logger.info(separator())
        logger.info("TRAINING STARTED")
        logger.info(separator())

        c = self.cfg
        history = self.state[StateValue.HISTORY]
        train_history = history[HistoryValue.TRAIN]
        joint = self.mode == "joint"
        self.timing["setup_s"] = time.perf_counter() - t_setup
        t_loop, it_loop, t_other = time.perf_counter(), self.state[StateValue.ITERATION], 0.0
        while True:
            iteration = self.state[StateValue.ITERATION]
            t_side = time.perf_counter()
            if iteration % c[ConfigValue.EVAL_INTERVAL] == 0 and self.testfeed is not None:
                torch.cuda.empty_cache()
                self._evaluate(self.testfeed, output_callback=self.validation_output_callback(0))
            if iteration % c[ConfigValue.PRINT_INTERVAL] == 0:
                history[HistoryValue.TIMINGS]["total"].update()
                last_print = history[HistoryValue.TIMINGS]["last_print"]
                last_print.update()
                samples = history[HistoryValue.EVAL]["n"] + history[HistoryValue.TRAIN]["n"]
                self.update_eta(samples, last_print.total)
                logger.info(self.state_str(eval_prefix="VALID"))
                self.write_metrics(eval_prefix="valid")
                last_print.total = 0
                self.reset_metrics()
            if iteration % c[ConfigValue.SNAPSHOT_INTERVAL] == 0:
                self.snapshot()
            if iteration % c[ConfigValue.EVAL_INTERVAL] == 0 or iteration % c[ConfigValue.SNAPSHOT_INTERVAL] == 0:
                t_other += time.perf_counter() - t_side       # validation passes and checkpoints: not the step loop
            if iteration >= c[ConfigValue.ITERATIONS]:
                break

            data = self.trainfeed.next_batch()
            image_count = data[DetectionDataset.INPUT].shape[0] * self.world
            if not self._train_mode:            # (Module.train() walks ~340 modules: 1.3 ms per step when called blindly)
                denoiser.train()
                denoiser.unfill()
                self._train_mode = True
            optimizer = self.optimizer          # sets the ramped learning rate (a device scalar)
            if self._stepper is None:
                inp0 = data[DetectionDataset.INPUT]
                self._stepper = graph_step.GraphedTrainStep(denoiser, inp0.shape[0], inp0.shape[-1], self.alpha, self.tau,
                                                            world=self.world, mode=self.mode, graph=self.graph)
            # zero_grad + forward + mean(loss).backward(): replayed from a HIP graph (graph_step.py), the gradients
            # land in one flat buffer; then the in-place all-reduce over the ranks and the Adam update
            outputs = self._stepper(data[DetectionDataset.INPUT], data[DetectionDataset.TARGET])
            self._stepper.grads.all_reduce(self.world)
            optimizer.step()

            with torch.no_grad():
                train_history["n"] += image_count
                train_history["loss"] += outputs[PipelineOutput.LOSS]
                if joint:
                    train_history["denoise_loss"] += outputs[PipelineOutput.DENOISE_LOSS]
                    train_history["detect_loss"] += outputs[PipelineOutput.DETECT_LOSS].unsqueeze(0)
                    train_history["aug_loss"] += outputs[PipelineOutput.AUG_LOSS].unsqueeze(0)
                for key in (PipelineOutput.NOISE_STD_DEV, PipelineOutput.MODEL_STD_DEV):
                    if key in outputs:
                        train_history[key.value].add(outputs[key], scale=255.0)
            self.state[StateValue.ITERATION] += image_count

        st = self._stepper
        if self.device.type == "cuda":
            torch.cuda.synchronize(self.device)
        self.timing["loop_s"] = time.perf_counter() - t_loop - t_other
        self.timing["loop_images"] = self.state[StateValue.ITERATION] - it_loop
        self.timing["validation_and_checkpoint_s"] = t_other
        if st is not None:
            self.timing["execution"] = ("eager launches (capture failed: %s)" % st.fallback_reason if st.fallback_reason else
                                        ("HIP-graph replay, %d kernels per step" % (st.kernels_per_step or 0)
                                         if st.use_graph else "eager launches"))
            self.timing["graph_fallback"] = st.fallback_reason
            logger.info("training loop: %d images in %.1f s = %.1f patches/s (validation + checkpoints %.1f s beside it)",
                        self.timing["loop_images"], self.timing["loop_s"],
                        self.timing["loop_images"] / max(self.timing["loop_s"], 1e-9), t_other)
            logger.info("step execution: %s; gradient collectives: %d%s",
                        "eager launches (capture failed: %s)" % st.fallback_reason if st.fallback_reason else
                        ("HIP-graph replay (%d kernels per step)" % (st.kernels_per_step or 0) if st.use_graph else "eager launches"),
                        st.grads.collectives,
                        " over %s, world size %d" % (distributed.backend_name(), self.world) if st.grads.collectives else "")
        logger.info(separator())
        logger.info("TRAINING FINISHED")
        logger.info(separator())
        self.snapshot()
        self.snapshot(output_name="final-{}.wt".format(self.denoiser.config_name()), subdir="", model_only=True)

    # ---- evaluation ----------------------------------------------------------------------------
    def evaluate(self, feed, output_callback=None):
        self.reset_metrics(train=False)
        return self._evaluate(feed, output_callback)

    def _evaluate(self, feed, output_callback):
        self._train_mode = False
        self.denoiser.eval()
        self.denoiser.fill()
        cuda = self.device.type == "cuda"
        t_loop = time.perf_counter()
        spans = []
        gen = torch.Generator(device=self.device) if cuda else None
        with torch.no_grad():
            eval_history = self.state[StateValue.HISTORY][HistoryValue.EVAL]
            for idx, data in feed:
                inp = data[DetectionDataset.INPUT]
                image_count = inp.shape[0]
                tile, halo = self._eval_tile(inp)
                shape = tuple(int(v) for v in inp.shape[-2:])
                if (shape, tile) not in self._eval_modes_logged:
                    self._eval_modes_logged.add((shape, tile))
                    logger.info("evaluation of %dx%d inputs: %s", shape[0], shape[1],
                                "whole image" if tile is None else "halo-tiled, tile %d halo %d" % (tile, halo))
                eps = None
                if cuda and self.mode == "joint":
                    # the reparameterisation noise of micrograph k comes from a Philox stream seeded with k (SURVEY.md
                    # §8d): the picks of a micrograph do not depend on which rank evaluates it, in which order, or on
                    # what was drawn before — and two operand precisions can be compared on the same noise
                    k = int(data[DetectionDataset.METADATA][DetectionDataset.Metadata.INDEXES][0])
                    gen.manual_seed(EVAL_NOISE_SEED + k)
                    eps = torch.randn((inp.shape[0], 1) + shape, dtype=torch.float32, device=self.device, generator=gen)
                if cuda:
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record()
                try:
                    outputs = self.denoiser.run_pipeline(data, train=False, tile=tile, halo=halo or 0, eps=eps)
                except torch.OutOfMemoryError as e:
                    raise RuntimeError("out of device memory evaluating a %dx%d input %s; set SPRK_EVAL_TILE=tile[,halo] "
                                       "(e.g. 2048,512) to evaluate in smaller halo-tiled windows" % (
                                           shape[0], shape[1], "whole" if tile is None else "in windows of %d" % (tile + 2 * halo))) from e
                eval_history["n"] += image_count
                clean = outputs[PipelineOutput.INPUTS][DetectionDataset.METADATA][DetectionDataset.Metadata.GT]
                if len(clean) > 0:   # PSNR against the clean reference images (train.py:404-413)
                    for key, name in self.img_outputs(prefix="psnr").items():
                        eval_history[name] += self.calculate_psnr(outputs, key)
                if output_callback:
                    output_callback(idx, outputs)
                if cuda:
                    e1.record()
                    spans.append((e0, e1))
        self.writer.drain()
        if cuda:
            torch.cuda.synchronize(self.device)
            self.timing["eval_device_s"] = self.timing.get("eval_device_s", 0.0) + sum(a.elapsed_time(b) for a, b in spans) * 1e-3
        self.timing["eval_loop_s"] = self.timing.get("eval_loop_s", 0.0) + time.perf_counter() - t_loop
        self.timing["eval_micrographs"] = self.timing.get("eval_micrographs", 0) + len(spans)
        self.denoiser.unfill()

    # Whole-image evaluation while the working set (WORKING_SET_BYTES_PER_PIXEL, measured: 57 GB peak at 4096^2 + margin)
    # fits in WHOLE_IMAGE_HBM_FRACTION of the device's TOTAL memory (288 GB on MI355X -> 4096^2 = 16.8 Mpix is whole, as
    # BASELINE configs[2] needs), halo-tiled above.  The choice depends on the micrograph's SIZE and the device MODEL only —
    # never on what else holds HBM at the moment — so the same checkpoint and micrograph always produce the same
    # *_scores.txt on the same kind of GPU; the mode is written to the log.
    WORKING_SET_BYTES_PER_PIXEL = 5000
    WHOLE_IMAGE_HBM_FRACTION = 0.30
    EVAL_TILE, EVAL_HALO = 3072, 512     # windows of 4096^2: the whole-image kernels, 1.78x the pixels

    def whole_image_max_pixels(self):
        if self.device.type != "cuda" or not torch.cuda.is_available():
            return 4096 * 4096
        total = torch.cuda.get_device_properties(self.device).total_memory
        return int(total * self.WHOLE_IMAGE_HBM_FRACTION / self.WORKING_SET_BYTES_PER_PIXEL)

    def _eval_tile(self, inp):
        """(tile, halo) for halo-tiled evaluation, or (None, None) for the whole-image path.  SPRK_EVAL_TILE =
        "tile[,halo]" forces tiling (0 = never tile)."""
        H, W = int(inp.shape[-2]), int(inp.shape[-1])
        env = os.environ.get("SPRK_EVAL_TILE")
        if env is not None:
            parts = [int(v) for v in env.split(",")]
            t, h = parts[0], (parts[1] if len(parts) > 1 else self.EVAL_HALO)
            return (t, h) if t > 0 and min(H, W) >= t + 2 * h else (None, None)
        limit = self.whole_image_max_pixels()
        if self.mode != "joint" or H * W <= limit:
            return None, None
        tile, halo = self.EVAL_TILE, self.EVAL_HALO
        while tile + 2 * halo > min(H, W) and tile > 1024:       # a long, narrow image: smaller windows
            tile -= 1024
        if min(H, W) < tile + 2 * halo:
            raise RuntimeError("a %dx%d input exceeds the whole-image limit of this device (%d pixels) and is too narrow for "
                               "halo tiling (tile %d + 2 x halo %d); set SPRK_EVAL_TILE=tile,halo" % (H, W, limit, tile, halo))
        return tile, halo

    def img_outputs(self, prefix=None):
        """Image outputs of the configured pipeline -> metric names (train.py:763-779)."""
        outputs = {PipelineOutput.IMG_DENOISED: "out"}
        if self.cfg[ConfigValue.PIPELINE] == Pipeline.SSDN:
            outputs[PipelineOutput.IMG_MU] = "mu_out"
        return {k: "_".join((prefix, v)) if prefix else v for k, v in outputs.items()}

    @staticmethod
    def calculate_psnr(outputs, key):
        """Per image: 20 log10(1) - 10 log10(mean squared error) between the un-padded output and the clean
        reference (train.py:781-814, utils/data.py:124-135) -> tensor [B]."""
        metadata = outputs[PipelineOutput.INPUTS][DetectionDataset.METADATA]
        clean = metadata[DetectionDataset.Metadata.GT]
        shapes = metadata[DetectionDataset.Metadata.IMAGE_SHAPE]
        vals = []
        for img, ref, shape in zip(outputs[key], clean, shapes):
            c, h, w = (int(v) for v in shape)
            mse = torch.mean((img[:c, :h, :w] - ref.to(img.device)) ** 2)
            vals.append(-10.0 * torch.log10(mse))
        return torch.stack(vals)

    def validation_output_callback(self, output_index):
        def callback(output_0_index, outputs):
            inp = outputs[PipelineOutput.INPUTS][DetectionDataset.INPUT]
            bi = output_index - output_0_index
            if 0 <= bi < inp.shape[0]:
                output_dir = os.path.join(self.run_dir_path, "val_imgs_" + self.mode)
                scorefmt = "{name}_{iter:08}_{desc}.txt" if self.mode == "joint" else None
                self._save_image_outputs(outputs, output_dir, "{name}_{iter:08}_{desc}.png", bi, scorefmt)
        return callback

    def save_image_outputs(self, outputs, output_dir, fileformat, scoreformat, batch_indexes=None):
        if batch_indexes is None:
            batch_indexes = range(outputs[PipelineOutput.INPUTS][DetectionDataset.INPUT].shape[0])
        for bi in batch_indexes:
            self._save_image_outputs(outputs, output_dir, fileformat, bi, scoreformat)

    def _save_image_outputs(self, outputs, output_dir, fileformat, batch_index, scoreformat=None):
        """Un-pad every image output of one batch item and write it; run the NMS on the un-padded
        score map and write ``*_scores.txt`` (train.py:500-585)."""
        os.makedirs(output_dir, exist_ok=True)
        metadata = outputs[PipelineOutput.INPUTS][DetectionDataset.METADATA]
        name = metadata[DetectionDataset.Metadata.NAME][batch_index]
        shape = [int(v) for v in metadata[DetectionDataset.Metadata.IMAGE_SHAPE][batch_index]]

        def path(fmt, desc):
            index = metadata[DetectionDataset.Metadata.INDEXES][batch_index]
            return os.path.join(output_dir, fmt.format(iter=self.state[StateValue.ITERATION], index=int(index),
                                                       desc=desc, name=name))

        def unpad(t):
            t = t[batch_index]
            return t[tuple(slice(0, n) for n in shape[-t.dim():])]

        images = [(PipelineOutput.INPUTS, "nsy"), (PipelineOutput.IMG_DENOISED, "out"), (PipelineOutput.IMG_MU, "out-mu"),
                  (PipelineOutput.TARGET, "out-target"), (PipelineOutput.MODEL_STD_DEV, "out-std"),
                  (PipelineOutput.DETECT, "pred_tar")]
        for key, desc in images:
            if key not in outputs:
                continue
            t = outputs[key][DetectionDataset.INPUT] if key == PipelineOutput.INPUTS else outputs[key]
            if not torch.is_tensor(t) or t.numel() == 0:
                continue
            self.writer.png(unpad(t), path(fileformat, desc))
        if PipelineOutput.DETECT in outputs and scoreformat is not None:
            score_map = unpad(outputs[PipelineOutput.DETECT])[0].contiguous()
            scores, coords = nms_device(score_map, self.cfg[ConfigValue.NMS], NMS_THRESHOLD)
            self.writer.call(picks.write_scores, path(scoreformat, "scores"), name, scores.cpu().numpy(),
                             coords.cpu().numpy(), tuple(score_map.shape))

    # ---- checkpoints ---------------------------------------------------------------------------
    def snapshot(self, output_name=None, subdir=None, model_only=False):
        if self.rank != 0:
            return
        if subdir is None:
            tag = "jt" if self.mode == "joint" else "dn"
            subdir = ("model_" if model_only else "training_") + tag
        output_dir = os.path.join(self.run_dir_path, subdir)
        os.makedirs(output_dir, exist_ok=True)
        iteration = self.state[StateValue.ITERATION]
        if model_only:
            checkpoint.save(self.denoiser.state_dict(), os.path.join(output_dir, output_name or
                                                                     "model_{:08d}.wt".format(iteration)))
        else:
            checkpoint.save(self.state_dict(), os.path.join(output_dir, output_name or
                                                            "model_{:08d}.training".format(iteration)))

    def state_dict(self):
        return {"denoiser": self.denoiser.state_dict(), "state": self.state,
                "optimizer": self.optimizer.state_dict(), "rng": torch.get_rng_state()}

    def load_state_dict(self, state_dict, restore_optimizer=True):
        if isinstance(state_dict, str):
            state_dict = checkpoint.load(state_dict)
        self.denoiser = Denoiser.from_state_dict(state_dict["denoiser"], mode=self.mode, device=self.device)
        self.cfg = self.denoiser.cfg
        self.state = state_dict["state"]
        if restore_optimizer and state_dict.get("optimizer"):
            # the reference leaves this commented out (train.py:931) and restarts Adam's moments on resume
            try:
                self._optimizer.load_state_dict(state_dict["optimizer"])
            except ValueError as e:
                logger.warning("optimizer state not restored: %s", e)
        torch.set_rng_state(state_dict["rng"])
        self.seed_streams()       # device + NumPy streams (and the patch sampler below) continue, not replay

    # ---- metrics / logging ---------------------------------------------------------------------
    def reset_metrics(self, eval=True, train=True):
        def reset(d):
            d["n"] = 0
            for v in d.values():
                if isinstance(v, Metric):
                    v.reset()
        history = self.state[StateValue.HISTORY]
        if train:
            reset(history[HistoryValue.TRAIN])
        if eval:
            reset(history[HistoryValue.EVAL])

    def write_metrics(self, eval_prefix="eval"):
        """(tag, iteration, value) rows appended to <run>/metrics.tsv — the scalars the reference sends
        to tensorboard (train.py:618-647)."""
        if self.rank != 0:
            return
        it = self.state[StateValue.ITERATION]
        rows = []
        for prefix, key in (("train", HistoryValue.TRAIN), (eval_prefix, HistoryValue.EVAL)):
            for name, metric in self.state[StateValue.HISTORY][key].items():
                if isinstance(metric, Metric) and not metric.empty():
                    rows.append((prefix + "/" + name, it, float(torch.as_tensor(metric.accumulated()).mean())))
            if prefix == "train":
                rows.append(("train/learning_rate", it, self.learning_rate))
        with open(os.path.join(self.run_dir_path, "metrics.tsv"), "a") as f:
            for tag, i, v in rows:
                f.write("%s\t%d\t%.9g\n" % (tag, i, v))

    @staticmethod
    def _metric_strs(metrics):
        out = []
        for key, metric in metrics.items():
            if isinstance(metric, Metric) and not metric.empty():
                out.append("{}={:8.2f}".format(key, float(torch.as_tensor(metric.accumulated()).mean())))
        return out

    def train_state_str(self):
        history = self.state[StateValue.HISTORY]
        timings = history[HistoryValue.TIMINGS]
        eta = timings.get("eta", None)
        eta_str = "???" if not isinstance(eta, int) else ("<1s" if eta < 1 else seconds_to_dhms(eta))
        summary = "[{:08d}] {:>5} | ".format(self.state[StateValue.ITERATION], "TRAIN")
        strs = self._metric_strs(history[HistoryValue.TRAIN])
        summary += ", ".join(strs)
        if strs:
            summary += " | "
        return summary + "[{} ~ ETA: {}]".format(seconds_to_dhms(timings["total"].total, trim=False), eta_str)

    def eval_state_str(self, prefix="EVAL"):
        return "{} | ".format(prefix) + ", ".join(self._metric_strs(self.state[StateValue.HISTORY][HistoryValue.EVAL]))

    def state_str(self, eval_prefix="EVAL"):
        s = self.train_state_str()
        if self.state[StateValue.HISTORY][HistoryValue.EVAL]["n"] > 0:
            s = os.linesep.join([s, self.eval_state_str(prefix="{:10} {:>5}".format("", eval_prefix))])
        return s

    def update_eta(self, samples, elapsed, smoothing_factor=0.95):
        """Smoothed remaining-time estimate (train.py:804-838).  The reference reads ``timings["eta"]`` from
        a defaultdict when no sample has been seen yet, which plants a TrackedTime under that key and
        breaks the next update when there is no validation set; here a missing estimate stays missing."""
        timings = self.state[StateValue.HISTORY][HistoryValue.TIMINGS]
        previous = timings.get("eta", None)
        if samples <= 0:
            return previous
        r = self.cfg[ConfigValue.ITERATIONS] - self.state[StateValue.ITERATION]
        if self.testfeed is not None:
            r += len(self.testfeed) * math.ceil(r / self.cfg[ConfigValue.EVAL_INTERVAL])
        new_eta = elapsed / samples * r
        if isinstance(previous, (int, float)):
            new_eta = smoothing_factor * new_eta + (1 - smoothing_factor) * previous
        timings["eta"] = new_eta
        return new_eta

    # ---- run directory -------------------------------------------------------------------------
    @property
    def run_dir_path(self):
        return os.path.join(self.runs_dir, self.run_dir)

    @property
    def run_dir(self):
        if self._run_dir is None:
            self._run_dir = "{:05d}-train-{}".format(self.next_run_id(), self.config_name())
            if self.world > 1:                     # rank 0 names the directory, everybody uses it
                box = [self._run_dir]
                torch.distributed.broadcast_object_list(box, src=0)
                self._run_dir = box[0]
        return self._run_dir

    def next_run_id(self):
        ids = []
        if os.path.exists(self.runs_dir):
            for path, _, _ in os.walk(self.runs_dir):
                try:
                    ids.append(int(path.split(os.sep)[-1].split("-")[0]))
                except ValueError:
                    continue
        return max(ids) + 1 if ids else 0

    def config_name(self):
        iterations = self.state.get(StateValue.ITERATION, 0) or self.cfg[ConfigValue.ITERATIONS]
        if iterations >= 1000000:
            iter_str = "iter%dm" % (iterations // 1000000)
        elif iterations >= 1000:
            iter_str = "iter%dk" % (iterations // 1000)
        else:
            iter_str = "iter%d" % iterations
        parts = [cfg_mod.config_name(self.cfg), iter_str]
        for key in (ConfigValue.TEST_DATASET_NAME, ConfigValue.TRAIN_DATASET_NAME):
            if self.cfg.get(key) is not None:
                parts.insert(0, self.cfg[key])
        parts += [str(self.cfg[ConfigValue.ALPHA]), str(self.cfg[ConfigValue.TAU]), self.mode]
        return "-".join(parts)

    # ---- data ----------------------------------------------------------------------------------
    def _require_txt(self, kind):
        if self.cfg[kind] != DatasetType.TXT:
            raise NotImplementedError("only micrograph lists (.txt tables of image_name/path) are built; "
                                      "HDF5 and image-folder datasets are outside the joint picking path")

    def train_data(self):
        c = self.cfg
        self._require_txt(ConfigValue.TRAIN_DATASET_TYPE)
        # (--train_gt is loaded by the reference's dataset but never read by its training loop: ignored here)
        groups, names = feed_mod.load_micrographs(c[ConfigValue.TRAIN_DATA_PATH], c[ConfigValue.TRAIN_LABEL_PATH],
                                                  radius=3, bb=c[ConfigValue.BB])
        batch = c[ConfigValue.TRAIN_MINIBATCH_SIZE]
        if batch % self.world:
            raise ValueError("train batch size %d is not divisible by the %d ranks" % (batch, self.world))
        return feed_mod.PatchFeed(groups, names, batch // self.world, patch=c[ConfigValue.TRAIN_PATCH_SIZE],
                                  device=self.device, balance=0.1,
                                  seed=self.seed + self.rank + 7919 * int(self.state.get(StateValue.ITERATION, 0) or 0),
                                  size=c[ConfigValue.ITERATIONS] * batch)

    def test_data(self):
        c = self.cfg
        self._require_txt(ConfigValue.TEST_DATASET_TYPE)
        gt = feed_mod.load_reference_images(c[ConfigValue.TEST_GT_PATH]) if c.get(ConfigValue.TEST_GT_PATH) else None
        groups, names = feed_mod.load_micrographs(c[ConfigValue.TEST_DATA_PATH], c.get(ConfigValue.TEST_LABEL_PATH),
                                                  radius=3, bb=c[ConfigValue.BB])
        return feed_mod.MicrographFeed(groups, names, count=cfg_mod.test_length(c), device=self.device,
                                       rank=self.rank, world=self.world, gt=gt)

    def set_train_data(self, path):
        self.cfg[ConfigValue.TRAIN_DATA_PATH] = path
        self.cfg[ConfigValue.TRAIN_DATASET_TYPE] = None
        cfg_mod.infer_datasets(self.cfg)

    def set_test_data(self, path):
        self.cfg[ConfigValue.TEST_DATA_PATH] = path
        self.cfg[ConfigValue.TEST_DATASET_TYPE] = None
        cfg_mod.infer_datasets(self.cfg)

    def set_train_label(self, path):
        self.cfg[ConfigValue.TRAIN_LABEL_PATH] = path

    def set_test_label(self, path):
        self.cfg[ConfigValue.TEST_LABEL_PATH] = path

    def set_train_gt_data(self, path):
        self.cfg[ConfigValue.TRAIN_GT_PATH] = path

    def set_test_gt_data(self, path):
        self.cfg[ConfigValue.TEST_GT_PATH] = path


def resume_run(run_dir, iteration=None, mode=None, device=None):
    """Newest (or the given) ``*.training`` file of a run directory -> trainer that continues in it."""
    run_dir = os.path.abspath(run_dir)
    found = {}
    for sub, m in (("training_jt", "joint"), ("training_dn", "denoise")):
        if mode is not None and m != mode:
            continue
        for path in glob.glob(os.path.join(run_dir, sub, "*.training")):
            digits = re.findall(r"\d+", os.path.basename(path))
            if digits:
                found[(int(digits[0]), m)] = path
    if not found:
        raise ValueError("Run directory contains no training files.")
    if iteration is None:
        iteration = max(k[0] for k in found)
    key = next((k for k in sorted(found, reverse=True) if k[0] == iteration), None)
    if key is None:
        raise ValueError("No training file for iteration %d" % iteration)
    logger.info("Loading from '%s'...", found[key])
    trainer = DenoiserTrainer(None, key[1], runs_dir=os.path.join(run_dir, ".."), run_dir=os.path.basename(run_dir),
                              device=device)
    trainer.load_state_dict(found[key])
    trainer.alpha = trainer.cfg[ConfigValue.ALPHA]
    trainer.tau = trainer.cfg[ConfigValue.TAU]
    for timing in trainer.state[StateValue.HISTORY][HistoryValue.TIMINGS].values():
        if isinstance(timing, TrackedTime):
            timing.forget()
    trainer.reset_metrics()      # partial sums in the file live on the CPU; start the print window afresh
    return trainer
