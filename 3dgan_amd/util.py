"""Replica / optimizer / reporting utilities with the names of the reference's `util.py`
(tower_scope_range :54-77, average_gradients :118-147, init_optimizer :150-183,
collection_to_dict :187-193, format_for_terminal :196-212)."""
from . import engine
from .ops.input import batch_slice


def tower_scope_range(x, n_gpus, batch_size, session=None):
    """util.py:54-77.  The reference yields one (slice, scope, gpu_id) per in-graph tower; here
    each process is exactly one tower, so the generator yields once with this rank's id.
    `x` may be a global batch (sliced like ops/input.py:24) or a per-rank source."""
    rank = session.rank if session is not None else 0
    if hasattr(x, 'shape') and x.shape[0] == batch_size * max(n_gpus, 1):
        x = batch_slice(x, batch_size, rank)
    yield x, 'tower_%d' % rank, rank


def average_gradients(session, store):
    """util.py:118-147: mean over towers of every gradient == all-reduce(sum) / n on the flat
    bucket; the division is folded into the optimizer kernel.  Returns that scale."""
    return session.allreduce_mean_scale(store.grads)


def init_optimizer(args, store):
    """util.py:150-183, every branch."""
    o = args.optimizer
    if o == 'rmsprop':
        return engine.RMSProp(store, args.lr, decay=args.decay, momentum=args.momentum, centered=args.centered)
    if o == 'adam':
        return engine.Adam(store, args.lr, args.beta1, args.beta2)
    if o == 'momentum':
        return engine.Momentum(store, args.lr, args.momentum)
    if o == 'sgd':
        return engine.Momentum(store, args.lr, 0.0)
    if o == 'adadelta':
        return engine.Adadelta(store, args.lr)
    if o in ('adagrad', 'padagrad'):   # ProximalAdagrad with zero l1/l2 strengths is plain Adagrad
        return engine.Adagrad(store, args.lr)
    if o == 'ftrl':
        return engine.Ftrl(store, args.lr)
    if o == 'pgd':
        return None                    # the reference forgets the `return` (util.py:171-172)
    raise ValueError('unknown optimizer %r' % o)


def collection_to_dict(collection):
    """util.py:187-193: key = last path component of the tensor name, ':0' stripped."""
    d = {}
    for c in collection:
        name, value = (c.name, c) if hasattr(c, 'name') else c
        d[name.split('/')[-1].split(':')[0]] = value
    return d


def format_for_terminal(results, prev_results):
    """util.py:196-212: the tqdm postfix.  Without an earlier result every value becomes its '{:3f}' text -- IN the dict
    that was passed in, as the reference does (its callers rely on nothing else reading it afterwards); with one, a new dict
    with the value followed by the direction it moved in: (+), (-) or (~)."""
    if not prev_results:
        for key, value in list(results.items()):
            results[key] = '{:3f}'.format(value)
        return results
    shown = {}
    for key in prev_results:
        change = float(results[key]) - float(prev_results[key])
        mark = '~' if change == 0 else ('+' if change > 0 else '-')
        shown[key] = '{:3f}({})'.format(results[key], mark)
    return shown


def chunks(x, n):
    """hem/util/misc.py chunks (pinned by hem/util/test_misc.py:22-30)."""
    for i in range(0, len(x), n):
        yield x[i:i + n]
