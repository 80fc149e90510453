"""Checkpoint / resume in the reference's variable namespace (train.py:254-259,288-292,329:
tf.train.Saver(max_to_keep=0) under a Supervisor).  One `.npz` per save: every variable under
its reference name (`generator/vars/fc1/weights`, ...), optimizer slots as `<opt>/<slot>`,
the `global_step` / `global_epoch` counters (train.py:201-202) and -- what TF's unseeded RNG
ops never had -- the position of the device Philox streams (`rng/draws`), so that a resumed run
with a fixed --seed continues the z / alpha / eps / dropout streams instead of replaying them.

The archive is written to `<path>.tmp` and renamed into place: a kill during the save leaves
the previous checkpoint as the newest readable one (`repeat.sh` restarts the reference the same way).
"""
import os

import numpy as np
import torch


def save(path, replica, sess):
    out = {}
    for store in replica.stores():
        out.update(store.state_dict())
    for name, opt in replica.optimizers().items():
        if opt is None:
            continue
        out['%s/t' % name] = np.array(opt.t)
        for slot, t in opt.state_tensors().items():
            out['%s/%s' % (name, slot)] = t.detach().cpu().numpy()
    out['global_step'] = np.array(sess.global_step)
    out['global_epoch'] = np.array(sess.global_epoch)
    out['rng/draws'] = np.array(sess.rng_state())
    tmp = path + '.tmp'
    with open(tmp, 'wb') as f:                       # a file object: np.savez would append '.npz' to a str path
        np.savez(f, **out)
        f.flush()
        os.fsync(f.fileno())
    os.replace(tmp, path)


def restore(path, replica, sess):
    z = np.load(path)
    for store in replica.stores():
        store.load({k: z[k] for k in store.index})
    for name, opt in replica.optimizers().items():
        if opt is None or '%s/t' % name not in z:
            continue
        opt.set_step_count(int(z['%s/t' % name]))
        for slot, t in opt.state_tensors().items():
            t.copy_(torch.as_tensor(z['%s/%s' % (name, slot)]))
    sess.global_step = int(z['global_step'])
    sess.global_epoch = int(z['global_epoch'])
    if 'rng/draws' in z:
        sess.set_rng_state(int(z['rng/draws']))
    if hasattr(replica, 'refresh'):
        replica.refresh()                            # packed GEMM operands follow the restored masters


def prune(directory, max_to_keep):
    """gen-2's `--max_to_keep N` (hem/util/misc.py:148: tf.train.Saver(max_to_keep=N)): keep the N most recent `checkpoint-<n>.npz`
    of `directory`; 0 keeps every one (gen-1's train.py:259)."""
    import glob
    import re
    if not max_to_keep or max_to_keep <= 0:
        return []
    found = []
    for f in glob.glob(os.path.join(directory, 'checkpoint-*.npz')):
        m = re.search(r'checkpoint-(\d+)\.npz$', f)
        if m:
            found.append((int(m.group(1)), f))
    found.sort()
    gone = [f for _, f in found[:-max_to_keep]]
    for f in gone:
        os.remove(f)
    return gone
