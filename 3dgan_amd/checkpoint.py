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
