#!/usr/bin/env python3
"""Training harness with the command line of the reference's `train.py` (flags :62-182,
`--config` file action :25-37, options dump :208-213, model dispatch :240-246, epoch/iteration
loop and checkpoint cadence :279-329) driving the MI355X-native hot path in `3dgan_amd/`.

Differences that are deliberate:
  * `--data` is accepted as an explicit alias of `--dataset` (README.md:50 relies on prefix matching).
  * `--n_gpus N` > 1 re-launches this script as N processes (one per GPU, RCCL all-reduce of the
    gradients) instead of building N in-graph towers; each process is one tower.
  * TensorFlow's Supervisor/Saver is replaced by `<dir>/checkpoint-<n>.npz` keyed by the reference's
    variable names; resume-from-`--dir` and `--epochs +n` behave as in train.py:273-282.
  * There is no CPU fallback (`--n_gpus 0` is an error here; in the reference it is a NameError).
"""
import argparse
import glob
import importlib
import os
import random
import re
import subprocess
import sys
import time
import uuid

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def _arguments():
    return importlib.import_module('3dgan_amd.arguments')


def build_parser():
    """The general (pass-1) parser: every gen-1 flag of train.py:62-182 and every gen-2 flag of
    hem/util/arguments.py:32-150; plugin flags are merged by parse_args (3dgan_amd/arguments.py)."""
    return _arguments().build_parser()


def parse_args(argv=None):
    """gen-2 three-pass parse (hem/util/arguments.py:153-163) incl. `@file` configs; gen-1 `--config FILE` too."""
    return _arguments().parse_args(argv, warn=lambda m: message(m))


def resolve_defaults(args):
    """train.py:107-111 default 5; the pix2pix plugin overrides it to 1 (hem/models/pix2pix.py:65-68)."""
    if args.n_disc_train is None:
        args.n_disc_train = 1 if args.model == 'pix2pix' else 5
    return args


def message(s):
    print('\033[1m\033[92m{}\033[0m'.format(s))


def maybe_relaunch(args, argv):
    """One process per GPU: when asked for several GPUs and not yet under a launcher, start
    torch.distributed.run as a child (before anything touches the GPU) and exit with its code."""
    if args.n_gpus <= 1 or 'RANK' in os.environ:
        return
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(args.n_gpus),
           '--master-addr', '127.0.0.1', '--master-port', str(29500 + os.getpid() % 1000), os.path.abspath(__file__)] + argv
    sys.exit(subprocess.call(cmd))


def latest_checkpoint(d):
    best, best_n = None, -1
    for f in glob.glob(os.path.join(d, 'checkpoint-*.npz')):
        m = re.search(r'checkpoint-(\d+)\.npz$', f)
        if m and int(m.group(1)) > best_n:
            best, best_n = f, int(m.group(1))
    return best


def main(argv=None):
    argv = sys.argv[1:] if argv is None else argv
    args = parse_args(argv)
    maybe_relaunch(args, argv)
    resolve_defaults(args)
    if args.n_gpus < 1:
        raise SystemExit('--n_gpus 0: this build has no CPU path (the reference raises NameError here, SURVEY App. C-8)')

    import numpy as np
    import torch
    from tqdm import tqdm
    K = importlib.import_module('3dgan_amd.kernels')
    rt = importlib.import_module('3dgan_amd.runtime')
    util = importlib.import_module('3dgan_amd.util')
    datasets = importlib.import_module('3dgan_amd.datasets')
    ckpt = importlib.import_module('3dgan_amd.checkpoint')
    summaries = importlib.import_module('3dgan_amd.summaries')

    if args.seed is None:
        args.seed = int.from_bytes(os.urandom(4), 'little')          # train.py:193-194 (kept an int, App. C-12)
    random.seed(args.seed)
    world = rt.init_distributed()
    sess = rt.Session(dtype=K.BF16 if args.precision == 'bf16' else K.F32, seed=args.seed,
                      check_numerics=args.check_numerics)
    chief = sess.rank == 0

    if chief:
        message('Parsing options...')
        os.makedirs(args.dir, exist_ok=True)
        with open(os.path.join(args.dir, 'options.config'), 'w') as f:      # train.py:208-213
            for a in vars(args):
                v = getattr(args, a)
                if a == 'config':
                    continue
                f.write('{} {}\n'.format(a, v))
                print('    {} = {}'.format(a, v))

    if chief:
        message('Initializing input pipeline...')
    x, x_count, image_shape = datasets.get_dataset(args, sess)              # train.py:219
    args.image_shape = image_shape
    if args.epoch_size <= 0:
        iter_per_epoch = int(x_count / (args.batch_size * args.n_gpus))     # train.py:222
    else:
        iter_per_epoch = args.epoch_size

    if chief:
        message('Initializing model...')
    models = importlib.import_module('3dgan_amd.models')
    model_funcs = models.model_funcs()                                      # train.py:240-244
    if args.model not in model_funcs:
        raise SystemExit('unknown --model %r (available: %s)' % (args.model, ', '.join(sorted(model_funcs))))
    train_func = model_funcs[args.model](x, args, sess)                     # train.py:246
    replica = train_func.replica

    # resume (tf.train.Supervisor restores the newest checkpoint in --dir, train.py:254-259,273)
    last = latest_checkpoint(args.dir)
    if last is not None:
        ckpt.restore(last, replica, sess)
        if chief:
            message('Restored {}'.format(last))
    for store in replica.stores():
        rt.broadcast_store(store)
    replica.refresh()

    start_time = time.time()
    current_epoch = sess.global_epoch
    max_epochs = current_epoch + int(args.epochs[1:]) if args.epochs[0] == '+' else int(args.epochs)   # train.py:279-282
    status = None
    if sess.global_step == 0 and chief:
        message('Generating baseline checkpoint...')
        ckpt.save(os.path.join(args.dir, 'checkpoint-0.npz'), replica, sess)                           # train.py:288-291
    if chief:
        message('Starting training...')
    try:
        for epoch in range(current_epoch, max_epochs):
            it = range(iter_per_epoch)
            pbar = tqdm(it, desc='Epoch {:3d}'.format(epoch + 1), unit='batch') if chief else it
            for i in pbar:
                prev_status = status
                profiled = args.profile and chief and epoch == current_epoch and i == min(3, iter_per_epoch - 1)
                if profiled:                     # --profile: one iteration run eagerly with the library's per-launch HIP events
                    graphs, replica.use_graphs = getattr(replica, 'use_graphs', False), False
                    K.timing_begin()
                status = train_func(sess, args)                                                        # train.py:307
                if profiled:
                    write_profile(os.path.join(args.dir, 'profile.txt'), K.timing_end())
                    replica.use_graphs = graphs
                if chief:
                    pbar.set_postfix(util.format_for_terminal(dict(status), prev_status))
            sess.global_epoch += 1                                                                     # train.py:322
            if chief:
                real, fake = replica.samples(args.examples) if hasattr(replica, 'samples') else (None, None)
                summaries.write_epoch(os.path.join(args.dir, 'summaries'), sess.global_epoch, status or {}, real, fake,
                                      args.examples)                                                   # models/gan.py:93-107
                ckpt.save(os.path.join(args.dir, 'checkpoint-{}.npz'.format(sess.global_epoch)), replica, sess)   # :329
                ckpt.prune(args.dir, getattr(args, 'max_to_keep', 0))                                   # gen-2 --max_to_keep
    except Exception as e:
        # gen-2's convention (hem/util/training.py:173-175): report and leave with -1, the status repeat.sh restarts on
        # (it resumes from the newest checkpoint in --dir); the library's status text travels in the exception
        print('Caught unexpected exception during training:', type(e).__name__, e, flush=True)
        sys.exit(-1)
    if chief:
        message('\nTraining complete! Elapsed time: {}s'.format(int(time.time() - start_time)))
    if world > 1:
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()


def write_profile(path, records):
    """--profile (a dead flag in the reference: train.py:264-265 builds RunOptions that no sess.run receives): the conv GEMM
    launches of one training iteration, per kernel -- launches, milliseconds, algorithmic TFLOP/s."""
    acc = {}
    for name, ms, flops in records:
        e = acc.setdefault(name, [0, 0.0, 0.0])
        e[0] += 1
        e[1] += ms
        e[2] += flops
    with open(path, 'w') as f:
        f.write('%-48s %8s %10s %9s\n' % ('kernel', 'launches', 'ms', 'TFLOP/s'))
        for name, (n, ms, fl) in sorted(acc.items(), key=lambda kv: -kv[1][1]):
            f.write('%-48s %8d %10.3f %9.1f\n' % (name, n, ms, fl / (ms * 1e-3) / 1e12 if ms > 0 else 0.0))
        f.write('%-48s %8d %10.3f\n' % ('total (conv GEMM kernels of one iteration)', sum(v[0] for v in acc.values()),
                                       sum(v[1] for v in acc.values())))


if __name__ == '__main__':
    main()
