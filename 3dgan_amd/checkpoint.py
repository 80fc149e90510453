"""Checkpoint / resume in the reference's variable namespace (train.py:254-259,288-292,329:
tf.train.Saver(max_to_keep=0) under a Supervisor).  One `.npz` per save: every variable under
its reference name (`generator/vars/fc1/weights`, ...), optimizer slots as `<opt>/<slot>`,
and the `global_step` / `global_epoch` counters (train.py:201-202)."""
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
    np.savez(path, **out)


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
