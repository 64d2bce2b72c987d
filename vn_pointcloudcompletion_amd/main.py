"""CLI with the reference's surface (/root/reference/main.py:23-123):
``python main.py [-n NAME] [--resume] [--ckpt_path P] [-from EPOCH] [-epochs N] {train,overfit,test}``.
Under ``python -m torch.distributed.run --nproc-per-node N main.py ... train`` every rank runs this entry point and the
train loop shards each batch over the ranks (distributed.py)."""
from __future__ import annotations

import argparse
import logging
import logging.config
import os

import torch

from . import distributed as D
from .experiments import LOGGING_CONFIG, create_experiment, load_config, load_env, setup_loggers, store_config


def build_parser():
    parser = argparse.ArgumentParser(description="PSGN")
    parser.add_argument("-n", "--name", nargs="?", type=str, default=None, help="name of the experiment directory to use")
    parser.add_argument("--resume", action="store_true", default=False, help="Resume training specified by the exp_name")
    parser.add_argument("--ckpt_path", type=str, default=None, help="The path of pretrained model")
    parser.add_argument("-from", "--from_epoch", nargs="?", default=0, help="use model from checkpoint at epoch")
    parser.add_argument("-epochs", "--num_epochs", nargs="?", type=str, default=None,
                        help="how many epochs the model should be trained")
    sub = parser.add_subparsers(help="commands", dest="command")
    sub.add_parser("train", help="run training")
    sub.add_parser("overfit", help="run overfitting")
    sub.add_parser("test", help="run testing")
    return parser


def main(argv=None):
    load_env(".env")
    args = build_parser().parse_args(argv)
    logging.config.dictConfig(LOGGING_CONFIG)
    log = logging.getLogger("root")
    if args.command is None:
        log.warning("No command specified")
        return
    rank, local_rank, world = D.init_from_env()
    if torch.cuda.is_available():
        device = f"cuda:{local_rank}" if world > 1 else "cuda:0"
        log.info(f"Using device: {device}")
    else:
        raise RuntimeError("no HIP device: the MI355X path has no CPU fallback (the reference would fall back to 'cpu', "
                           "main.py:85-87)")
    config = load_config(args.name)
    config.device = device
    if args.from_epoch != 0:
        config.checkpoint = int(args.from_epoch) if str(args.from_epoch).isdigit() else str(args.from_epoch)
    else:
        config.checkpoint = 0
    if args.num_epochs is not None:
        config.max_epochs = int(args.num_epochs)
    if args.command == "train":
        config.overfit = False
    elif args.command == "overfit":
        config.overfit = True
    if not args.resume:
        if rank == 0:
            config = create_experiment(config)
        if world > 1:
            import torch.distributed as dist
            box = [config.exp_dir, getattr(config, "run_name", "")]
            dist.broadcast_object_list(box, src=0)
            config.exp_dir, config.run_name = box
    elif not getattr(config, "exp_dir", ""):
        config.exp_dir = os.path.join(os.environ["OUTPUT_DIR"], args.name)
    if rank == 0:
        store_config(config)
        setup_loggers("train", log_path=os.path.join(config.exp_dir, "train.log"))
    if args.command == "test":
        from .test import test
        test(config, args)
    else:
        from .train import train
        train(config, args)


if __name__ == "__main__":
    main()
