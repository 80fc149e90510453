"""Command line of both generations of the reference's harness.

gen-1 `train.py:62-182`: one flat argparse parser, `--config FILE` action (:25-37).
gen-2 `hem/util/arguments.py:10-179`: `@file` arguments (one `key value...` per line, `#` comments and blank lines
skipped, `--` prefixed: `hem/util/misc.py:72-82`), and a THREE-PASS parse -- general flags, then the flags the chosen
dataset plugin declares in `arguments()`, then the chosen model plugin's (:153-163); whatever is left over is reported
as a warning, not an error (:161-163 -- the reference's own `examples/pix2pix/noise.config` relies on that: it carries
the retired flag `add_noise1`).  `conflict_handler='resolve'` lets a plugin re-declare a general flag (e.g. pix2pix's
`--examples` / `--n_disc_train`, nyuv2's `--resize`).

This module is the union: every gen-1 flag, every gen-2 flag, both config-file forms, plugins merged the gen-2 way.
"""
import argparse
import multiprocessing
import os
import sys
import uuid

from . import plugins

GEN1_MODELS = ('gan', 'wgan', 'iwgan', 'vae', 'cnn')                 # train.py:240-244 (function models, no plugin class)
DATASET_ALIASES = {'floorplans': 'floorplan'}                        # gen-1 default name (train.py:159) -> gen-2 plugin name


class CustomArgumentParser(argparse.ArgumentParser):
    """hem/util/misc.py:72-82: `@file` lines are `key value value ...`; comments and blank lines are skipped."""

    def convert_arg_line_to_args(self, arg_line):
        words = arg_line.split()
        if not words or words[0].startswith('#'):
            return []
        flag, values = words[0], words[1:]
        return ['--' + flag, *values]


class load_args_from_file(argparse.Action):
    """train.py:25-37 (`--config FILE`): the file is a flat stream of whitespace separated `key value` pairs (every even
    token is a key and gets its `--` when it lacks one); the pairs are parsed by the same parser, and only what they set
    to a truthy value is taken over, so a flag already given on the command line keeps its value unless the file names it too."""

    def __call__(self, parser, namespace, values, option_string=None):
        tokens = []
        for line in values.read().splitlines():
            if not line.lstrip().startswith('#'):
                tokens.extend(line.split())
        for k in range(0, len(tokens) - len(tokens) % 2, 2):
            if not tokens[k].startswith('--'):
                tokens[k] = '--' + tokens[k]
        own_dest = option_string.strip('-')
        parsed, _unknown = parser.parse_known_args(tokens, namespace=namespace)
        for dest, value in vars(parsed).items():
            if dest != own_dest and value:
                setattr(namespace, dest, value)


def build_parser():
    parser = CustomArgumentParser(description='Autoencoder training harness.',
                                  formatter_class=argparse.ArgumentDefaultsHelpFormatter,
                                  fromfile_prefix_chars='@', conflict_handler='resolve',
                                  epilog='Example: python train.py @path/to/config_file --dir workspace/model_test --lr 0.1')
    parser._action_groups.pop()
    model_args = parser.add_argument_group('Model')
    data_args = parser.add_argument_group('Data')
    optimizer_args = parser.add_argument_group('Optimizer')
    train_args = parser.add_argument_group('Training')
    misc_args = parser.add_argument_group('Miscellaneous')
    add = misc_args.add_argument
    add('--config', type=open, action=load_args_from_file,
        help='gen-1: read a file of `key value` pairs; command line arguments overwrite it.')
    add('--seed', type=int, help='Randomized each execution if not set.')
    add('--n_gpus', type=int, default=1, help='Number of GPUs (one replica process per GPU).')
    add('--profile', default=False, action='store_true', help='Write <dir>/profile.txt: the conv GEMM kernels of one training iteration (a dead flag in the reference).')
    add('--check_numerics', default=False, action='store_true', help='Fail with the variable name on NaN/Inf gradients.')
    add('--precision', default='bf16', choices=['bf16', 'f32'],
        help='bf16 MFMA with f32 accumulate / master weights (throughput) or exact f32 (parity).')
    add = train_args.add_argument
    add('--epochs', default='3', help='Max epochs, or `+n` for n more than the restored checkpoint.')
    add('--batch_size', type=int, default=256, help='Batch size to use, per device.')
    add('--epoch_size', type=int, default=-1, help='Iterations per epoch; default: the whole dataset.')
    add('--examples', type=int, default=64, help='Number of examples to generate when sampling.')
    add('--dir', type=str, default='workspace/{}'.format(uuid.uuid4()), help='Checkpoints, logs; resumes if populated.')
    add('--n_disc_train', type=int, default=None,
        help='Discriminator steps per generator step (default 5; 1 for --model pix2pix, its plugin default).')
    add('--max_to_keep', type=int, default=0, help='gen-2: most recent checkpoints to keep; 0 keeps every one.')
    add('--test_epochs', nargs='*', default=[], type=int,
        help='gen-2: epochs at which to run the test split (accepted; the test pass belongs to the thesis harness).')
    add = optimizer_args.add_argument
    add('--optimizer', type=lambda s: s.lower(), default='rmsprop')
    add('--lr', type=float, default=0.001)
    add('--loss', type=lambda s: s.lower(), default='l1', help='Parsed but unused, as in the reference.')
    add('--momentum', type=float, default=0.01)
    add('--decay', type=float, default=0.9)
    add('--centered', default=False, action='store_true')
    add('--beta1', type=float, default=0.9)
    add('--beta2', type=float, default=0.999)
    add = model_args.add_argument
    add('--model', type=lambda s: s.lower(), default='fc', help='gan | wgan | iwgan | vae | cnn | any model plugin (pix2pix).')
    add('--latent_size', type=int, default=200)
    # opt-ins for the reference's defects; every default reproduces the reference's effective behaviour
    add('--wgan_clip', type=float, default=0.0,
        help='App. C-3: clamp the critic to [-c, c] before each critic step (the clip op of models/gan.py:142-148 never runs in the reference); 0 = off.')
    add('--gp_per_sample', default=False, action='store_true',
        help='App. C-4: per-sample gradient-penalty norms instead of one norm over the whole batch tensor (models/gan.py:229).')
    add('--vae_full_elbo', default=False, action='store_true',
        help='App. C-7: differentiate reconstruction + KL instead of the reconstruction term alone (models/vae.py:41).')
    add('--mean_loss', default=False, action='store_true',
        help='App. C-11: report the mean loss over replicas instead of the last replica\'s.')
    add = data_args.add_argument
    add('--dataset', '--data', dest='dataset', type=lambda s: s.lower(), default='floorplans',
        help='Any dataset plugin: cifar | mnist | floorplan(s) | nyuv2 | synthetic.')
    add('--resize', type=int, nargs=2, help='Resize input images to w x h.')
    add('--shuffle', default=True, action='store_true')
    add('--buffer_size', type=int, default=10000)
    add('--streaming', default=False, action='store_true',
        help='Keep the dataset in host memory and serve it through the reference\'s pipeline shape (repeat -> shuffle(buffer_size) -> '
             'batch over a pinned-host ring with asynchronous copies) even when it fits the HBM budget.')
    add('--hbm_budget_gb', type=float, default=64.0,
        help='Datasets whose float32 form is larger than this stay in host memory (streaming input pipeline).')
    add('--grayscale', default=False, action='store_true')
    add('--cache_dir', default=None, help='Cache decoded datasets here.')
    add('--data_dir', default='data', help='gen-1: where the dataset files live (data.py:37-39).')
    add('--raw_dataset_dir', default='/tmp', help='gen-2: location of raw dataset files (conversion scripts only).')
    add('--dataset_dir', default='datasets', help='gen-2: location of prepared tfrecord files.')
    add('--n_threads', type=int, default=multiprocessing.cpu_count(), help='gen-2: input pipeline threads (accepted).')
    return parser


def dataset_plugin_name(name):
    return DATASET_ALIASES.get(name, name)


def parse_args(argv=None, display=False, warn=None):
    """hem/util/arguments.py:10-179.  Returns the namespace; unknown arguments are warned about, not fatal (:161-163)."""
    argv = sys.argv[1:] if argv is None else list(argv)
    parser = build_parser()
    args, leftover = parser.parse_known_args(argv)
    # pass 2: the dataset plugin's flags (:153-156)
    dsets = plugins.data_plugins()
    dname = dataset_plugin_name(args.dataset)
    if dname in dsets:
        for k, v in dsets[dname].arguments().items():
            parser.add_argument(k, **v)
        args, leftover = parser.parse_known_args(leftover, namespace=args)
    # pass 3: the model plugin's flags (:158-162)
    models = plugins.model_plugins()
    if args.model in models:
        for k, v in models[args.model].arguments().items():
            kw = dict(v)
            dest = kw.get('dest') or k.lstrip('-').replace('-', '_')
            if hasattr(args, dest):
                if getattr(args, dest) is None:
                    delattr(args, dest)                  # declared by an earlier pass but never given: argparse applies a default
                                                         # only to a dest the namespace lacks, so make room for the plugin's
                else:
                    kw['default'] = getattr(args, dest)  # given (or defaulted to a value) earlier: the plugin must not undo it
            parser.add_argument(k, **kw)
        args, leftover = parser.parse_known_args(leftover, namespace=args)
    elif args.model not in GEN1_MODELS and args.model != 'fc':
        raise SystemExit('unknown --model %r (available: %s)' % (args.model, ', '.join(sorted(list(GEN1_MODELS) + list(models)))))
    if leftover:
        (warn or (lambda m: sys.stderr.write(m + '\n')))('WARNING: unknown and unused arguments provided: {}'.format(leftover))
    args.unknown_args = leftover
    if args.n_disc_train is None:                        # train.py:107-111 default 5 (plugins bring their own)
        args.n_disc_train = 5
    if display:
        for a in vars(args):
            print('    {} = {}'.format(a, getattr(args, a)))
    return args
