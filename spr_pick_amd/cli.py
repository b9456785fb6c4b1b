"""``joint`` command line (cli/cli.py, cli/cmds/train.py:24-300, cli/cmds/eval.py:15-71 of the
reference): ``joint train start|resume`` and ``joint eval`` with the same flags and the same
flag -> configuration mapping.  Run as ``python -m spr_pick_amd ...``; for several GPUs launch it
under ``python -m torch.distributed.run --nproc-per-node N -m spr_pick_amd -- ...`` (keep the ``--``)."""
import argparse

from . import __version__, cfg as cfg_mod
from .params import ConfigValue, NoiseAlgorithm, NoiseValue


def _shared_train_args(p, start):
    p.add_argument("--train_dataset", "-t", required=start, help="Tab-separated table (image_name, path) of training micrographs.")
    p.add_argument("--alpha", "-ap", type=float, required=start, help="alpha value")
    p.add_argument("--tau", "-tau", type=float, required=start,
                   help="tau value for positive unlabeled learning - percentage of positives")
    p.add_argument("--train_gt", "-gt", help="Path to ground truth dataset")
    p.add_argument("--train_label", "-l", required=start, help="Particle coordinates table (image_name, x_coord, y_coord).")
    p.add_argument("--validation_dataset", "-v", help="Table of validation micrographs.")
    p.add_argument("--validation_label", "-vl", help="Validation particle coordinates.")
    p.add_argument("--validation_gt", "-vgt", help="Path to validation ground truth dataset")
    p.add_argument("--iterations", "-iter", required=start, type=int, help="Number of joint training iterations")
    p.add_argument("--num", "-num", type=int, default=1, help="Number of eval samples during training")
    p.add_argument("--lr", "-lr", type=float, help="learning rate")
    p.add_argument("--nms", "-nms", type=int, help="non_maximum suppression radius")
    p.add_argument("--bb", "-bb", type=int, help="bounding box radius for particle of interests")
    p.add_argument("--eval_interval", type=int, help="Iterations between evaluations.")
    p.add_argument("--checkpoint_interval", type=int, help="Iterations between checkpoints.")
    p.add_argument("--print_interval", type=int, help="Iterations between progress lines.")
    p.add_argument("--train_batch_size", type=int, help="Batch size to use for training images.")
    p.add_argument("--validation_batch_size", type=int, help="Batch size to use for validation images.")
    p.add_argument("--patch_size", type=int, help="Patch size to use for training (square).")
    p.add_argument("--fraction", help="percent fraction of frames.")


def build_parser():
    parser = argparse.ArgumentParser(prog="joint", description="Joint denoising + particle picking on MI355X "
                                     "(train / evaluate), drop-in for spr_pick's `joint` command.")
    parser.add_argument("--version", action="version", version="%(prog)s v" + __version__)
    cmds = parser.add_subparsers(dest="command", required=True)

    train = cmds.add_parser("train", help="Train or resume training of a Denoiser model.")
    actions = train.add_subparsers(dest="train_cmd", required=True)
    start = actions.add_parser("start", help="start a new run")
    _shared_train_args(start, True)
    start.add_argument("--algorithm", "-a", required=True, choices=[a.value for a in NoiseAlgorithm],
                       help="The algorithm to train.")
    start.add_argument("--noise_style", "-n", required=True, help="Noise style, e.g. 'gaussian'.")
    start.add_argument("--noise_value", choices=[v.value for v in NoiseValue],
                       help="[joint] Whether the noise value should be estimated.")
    start.add_argument("--dn_only", action="store_true", help="denoising only")
    start.add_argument("--runs_dir", default=cfg_mod.DEFAULT_RUN_DIR,
                       help="Directory in which the output directory is generated.")
    resume = actions.add_parser("resume", help="Resume a run from its latest *.training file.")
    resume.add_argument("run_dir", help="Path to run directory to resume.")
    _shared_train_args(resume, False)

    ev = cmds.add_parser("eval", help="Evaluate a pre-trained model.")
    ev.add_argument("--model", "-m", required=True, help="Path to model weights or training file.")
    ev.add_argument("--dataset", "-d", required=True, help="Table of micrographs to evaluate.")
    ev.add_argument("--runs_dir", default=cfg_mod.DEFAULT_RUN_DIR,
                    help="Directory in which the output directory is generated.")
    ev.add_argument("--batch_size", type=int, help="Batch size to use, defaults to that used while training.")
    ev.add_argument("--gt_dataset", "-g", help="ground truth image")
    ev.add_argument("--nms", "-nms", type=int, help="non maximum suppression radius")
    ev.add_argument("--num", "-num", type=int, default=10, help="Number of micrographs to evaluate")
    return parser


def run_train(args, parser):
    from .train import DenoiserTrainer, resume_run
    if args["train_cmd"] == "start":
        if args["algorithm"] == "ssdn" and args.get("noise_value") is None:
            parser.error("SSDN requires --noise_value")
        cfg = cfg_mod.base()
        cfg[ConfigValue.ALGORITHM] = NoiseAlgorithm(args["algorithm"])
        cfg[ConfigValue.NOISE_STYLE] = args["noise_style"]
        if args.get("noise_value") is not None:
            cfg[ConfigValue.NOISE_VALUE] = NoiseValue(args["noise_value"])
        for flag, key in (("lr", ConfigValue.LEARNING_RATE), ("bb", ConfigValue.BB), ("nms", ConfigValue.NMS)):
            if args.get(flag) is not None:
                cfg[key] = args[flag]
        if args["dn_only"]:
            trainer = DenoiserTrainer(cfg, mode="denoise", runs_dir=args["runs_dir"])
        else:
            trainer = DenoiserTrainer(cfg, mode="joint", alpha=args["alpha"], tau=args["tau"],
                                      runs_dir=args["runs_dir"])
    else:
        trainer = resume_run(args["run_dir"])
        cfg = trainer.cfg
        if args.get("alpha") is not None:
            trainer.alpha = args["alpha"]
        if args.get("tau") is not None:
            trainer.tau = args["tau"]

    for flag, setter in (("train_dataset", trainer.set_train_data), ("train_gt", trainer.set_train_gt_data),
                         ("train_label", trainer.set_train_label), ("validation_dataset", trainer.set_test_data),
                         ("validation_gt", trainer.set_test_gt_data), ("validation_label", trainer.set_test_label)):
        if args.get(flag) is not None:
            setter(args[flag])
    for flag, key in (("iterations", ConfigValue.ITERATIONS), ("num", ConfigValue.NUM_EVAL),
                      ("eval_interval", ConfigValue.EVAL_INTERVAL), ("checkpoint_interval", ConfigValue.SNAPSHOT_INTERVAL),
                      ("print_interval", ConfigValue.PRINT_INTERVAL), ("train_batch_size", ConfigValue.TRAIN_MINIBATCH_SIZE),
                      ("validation_batch_size", ConfigValue.TEST_MINIBATCH_SIZE), ("patch_size", ConfigValue.TRAIN_PATCH_SIZE),
                      ("alpha", ConfigValue.ALPHA), ("tau", ConfigValue.TAU)):
        if args.get(flag) is not None:
            cfg[key] = args[flag]
    trainer.train()
    return trainer


def run_eval(args):
    from .eval import DenoiserEvaluator
    evaluator = DenoiserEvaluator(args["model"], runs_dir=args["runs_dir"])
    for flag, key in (("batch_size", ConfigValue.TEST_MINIBATCH_SIZE), ("nms", ConfigValue.NMS),
                      ("num", ConfigValue.NUM_EVAL)):
        if args.get(flag) is not None:
            evaluator.cfg[key] = args[flag]
    evaluator.set_test_data(args["dataset"])
    evaluator.set_test_gt_data(args["gt_dataset"])
    evaluator.evaluate()
    return evaluator


def start(argv=None):
    import sys
    argv = list(sys.argv[1:] if argv is None else argv)
    if argv[:1] == ["--"]:      # `torch.distributed.run ... -m spr_pick_amd -- eval --num 8`: the separator keeps
        argv = argv[1:]         # the launcher's own parser from prefix-matching our flags (--num ~ --numa-binding)
    parser = build_parser()
    args = vars(parser.parse_args(argv))
    if args["command"] == "train":
        return run_train(args, parser)
    return run_eval(args)
