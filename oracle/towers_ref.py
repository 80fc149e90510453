"""CPU restatement of the reference's multi-tower training step: n independent replicas of the model graph on n
different batch shards, the arithmetic mean of their gradients variable by variable, ONE optimizer step on the
shared variables.

TEST INFRASTRUCTURE ONLY (see oracle/tf_ops.py header).  PARITY UNPINNED against TensorFlow, like the single-tower
oracles this file composes (oracle/gan_ref.py, vae_ref.py, pix2pix_ref.py).

Follows (relative to /root/reference):
  util.py:54-77        tower_scope_range: one tower per GPU, tower i gets rows [i*B, (i+1)*B) of the global batch
                       (ops/input.py:11-25); variables are shared (reuse after tower 0)
  models/gan.py:55-70  per tower: OWN z (:246) and alpha (:224) draws, own batch statistics, own whole-batch penalty norm
  models/gan.py:65-68  per-tower compute_gradients for the two nets
  util.py:118-147      average_gradients: mean over the tower axis
  models/gan.py:76-81  one apply_gradients per net on the averaged gradients
  util.py:187-193      the reported loss dict keeps the LAST tower's tensors
  models/vae.py:31-47, hem/models/pix2pix.py:101-133: the same tower loop around their own losses

This is what SURVEY section 4 names as the oracle of a multi-rank run: "n independent replicas, then mean" (batch
norm and the penalty norm are per tower, so it is NOT one replica on the concatenated batch).

`follow` (every step method): a free-running comparison over several optimizer steps is ill-conditioned however exact
either side is -- Adam / RMSProp move a variable whose gradient is rounding residue by +-lr whatever its size, the two
runs' variables part by ~lr in those elements, and a later step whose (l)relu masks or penalty norm sit close to a kink
answers with 1e-3-sized gradient differences (this file's own float32 run leaves its float64 run by 4.8e-3 on
discriminator/vars/c3/weights at the eighth optimizer step of tests/test_gpu_distributed.py's iwgan schedule).  A caller
that holds the compared run's mean gradients of the step passes them as `follow`: the oracle still computes and records
ITS tower mean at its current variables (`last_*_grads`, what the caller compares), but the optimizer step is taken with
the followed gradients, so the oracle's variables stay within optimizer rounding of the compared run's and every step is
a same-state comparison.
"""
import numpy as np

from . import gan_ref as G
from . import tf_ops as T


class GanTowers:
    """models/gan.py:39-91 with n towers.  Every step takes one (x, z, alpha) triple PER TOWER."""

    def __init__(self, P, cfg, args):
        self.P, self.cfg, self.args = P, cfg, args
        self.g_opt, self.d_opt = T.init_optimizer(args), T.init_optimizer(args)     # models/gan.py:46
        self.last_d_grads = self.last_g_grads = None

    @staticmethod
    def rescale(x01):
        return 2.0 * (x01.reshape(x01.shape[0], -1) - 0.5)                           # models/gan.py:49-50

    def d_grads(self, xs, zs, alphas):
        """The tower mean of the critic's gradients at the current variables (no update)."""
        tower = []
        for r, (x, z, a) in enumerate(zip(xs, zs, alphas)):
            if G.KINK is not None:
                G.KINK.tower = r
            tower.append(G.d_loss_and_grads(self.P, self.rescale(x), z, a, self.cfg)[1])
        return T.average_gradients(tower)                                            # models/gan.py:77

    def d_step(self, xs, zs, alphas, follow=None, grads=None):
        self.last_d_grads = grads if grads is not None else self.d_grads(xs, zs, alphas)
        self.d_opt.apply(self.P, follow if follow is not None else self.last_d_grads)    # :81

    def g_grads(self, xs, zs, alphas, want_report=True):
        tower, rep = [], None
        for r, (x, z, a) in enumerate(zip(xs, zs, alphas)):
            if G.KINK is not None:
                G.KINK.tower = r
            gl, gg, _ = G.g_loss_and_grads(self.P, z, self.cfg)
            tower.append(gg)
            if want_report:
                dl, _, _ = G.d_loss_and_grads(self.P, self.rescale(x), z, a, self.cfg, want_grads=False)
                rep = {'g_loss': float(gl), 'd_loss': float(dl)}
        return T.average_gradients(tower), rep                                       # :76

    def g_step(self, xs, zs, alphas, follow=None, grads=None):
        """[g_train_op, losses] (models/gan.py:172): the losses are those of the LAST tower's batch, on the
        variables before this step's update."""
        if grads is None:
            self.last_g_grads, rep = self.g_grads(xs, zs, alphas)
        else:
            self.last_g_grads, rep = grads
        self.g_opt.apply(self.P, follow if follow is not None else self.last_g_grads)    # :80
        return rep

    def gan_step(self, xs, zs):
        """_train_gan (models/gan.py:110-131): one batch per tower, both nets updated from the same forward."""
        dt, gt, rep = [], [], None
        for x, z in zip(xs, zs):
            gl, gg, _ = G.g_loss_and_grads(self.P, z, self.cfg)
            dl, dg, _ = G.d_loss_and_grads(self.P, self.rescale(x), z, None, self.cfg)
            dt.append(dg)
            gt.append(gg)
            rep = {'g_loss': float(gl), 'd_loss': float(dl)}
        self.last_d_grads, self.last_g_grads = T.average_gradients(dt), T.average_gradients(gt)
        self.d_opt.apply(self.P, self.last_d_grads)
        self.g_opt.apply(self.P, self.last_g_grads)
        return rep


class VaeTowers:
    """models/vae.py:25-51 with n towers (default_training: one optimizer step per call)."""

    def __init__(self, P, args):
        from . import vae_ref as V
        self.V, self.P, self.opt = V, P, T.init_optimizer(args)
        self.last_grads = None

    def step(self, xs, epss, follow=None):
        tower, rep = [], None
        for x, e in zip(xs, epss):
            losses, c = self.V.forward(self.P, x, e)
            tower.append(self.V.backward(self.P, c))
            rep = {k: float(v) for k, v in losses.items()}
        self.last_grads = T.average_gradients(tower)
        self.opt.apply(self.P, follow if follow is not None else self.last_grads)
        return rep


class Pix2pixTowers:
    """hem/models/pix2pix.py:101-133,151-156 with n towers, on torch autograd (P: dict of float64 leaf tensors)."""

    def __init__(self, P, args):
        from . import pix2pix_ref as PR
        from . import torch_ref as TR
        self.PR, self.TR, self.P, self.args = PR, TR, P, args
        self.g_opt, self.d_opt = TR.make_optimizer(args), TR.make_optimizer(args)
        self.last_d_grads = self.last_g_grads = None

    def _mean(self, tower):
        return {k: sum(t[k] for t in tower) / len(tower) for k in tower[0]}

    def d_step(self, pairs, follow=None):
        tower = []
        for x01, y01 in pairs:
            _, d_total, _ = self.PR.losses(self.P, x01, y01, self.args)
            tower.append(self.TR.grads_of(d_total, self.P, 'discriminator/'))
        self.last_d_grads = self._mean(tower)
        self.d_opt.apply(self.P, follow if follow is not None else self.last_d_grads)

    def g_step(self, pairs, follow=None):
        tower = []
        for x01, y01 in pairs:
            g_total, _, _ = self.PR.losses(self.P, x01, y01, self.args)
            tower.append(self.TR.grads_of(g_total, self.P, 'generator/'))
        self.last_g_grads = self._mean(tower)
        self.g_opt.apply(self.P, follow if follow is not None else self.last_g_grads)

    def report(self, pairs):
        import torch
        with torch.no_grad():
            return self.PR.losses(self.P, *pairs[-1], self.args)[2]                  # the last tower's batch
