"""GAN / WGAN / IWGAN on MI355X -- the builder surface of the reference's `models/gan.py`
(gan :39-91, losses :178-211, gradient_penalty :214-231, generator :234-254,
discriminator :257-287, step policies :110-175) with the TensorFlow tower code replaced by
a hand-scheduled sequence of HIP kernels per replica.

What is kept: `gan(x, args)` returns `train_func(sess, args) -> {'g_loss', 'd_loss'}`; the
generator/discriminator are written against dense/conv2d/deconv2d + arg_scope exactly as in
the reference; variable names, loss definitions, per-step batch consumption
((n_disc_train + 1) batches for wgan/iwgan) and the reference's *effective* semantics
(no clipping, no BN moving averages, whole-batch penalty norm, last
replica's losses reported).

What is MI355X-native: D(x), D(g) and D(x_hat) run as ONE batched pass over a 3B-image
buffer [x | g | x_hat] (the iwgan critic has no batch norm, so this is exact); the penalty's
second-order term is the hand-derived tangent pass of engine.SeqNet.tangent_backward; all
scalars stay on the device until the end of the step; gradients live in one flat bucket per net
(one RCCL all-reduce + one fused optimizer launch).
"""
import os

import torch

from .. import _lib
from .. import kernels as K
from .. import engine
from ..ops.layers import (dense, conv2d, deconv2d, flatten, reshape, random_normal, arg_scope,
                          variable_scope, placeholder, reset_graph)
from ..ops.activations import lrelu, relu, tanh, sigmoid
from ..util import tower_scope_range, average_gradients, init_optimizer, collection_to_dict

GP_LAMBDA = 10.0          # l_term, models/gan.py:199


# ------------------------------------------------------------------------------ builders
def generator(batch_size, latent_size, args, reuse=False):
    """models/gan.py:234-254.  The reference hard-wires a 4x4 base and 64x64x3 output (:241,247-248);
    the base is generalised to s0 = H/16 so 32x32 data needs no --resize;
    --resize 64 64 reproduces the literal network."""
    h, w, c = args.image_shape
    s0h, s0w = h // 16, w // 16
    output_dim = h * w * c
    with arg_scope([dense, deconv2d], reuse=reuse, use_batch_norm=True, activation=relu):
        z = random_normal([batch_size, latent_size])
        y = dense(z, latent_size, s0h * s0w * 4 * latent_size, name='fc1')
        y = reshape(y, [-1, s0h, s0w, 4 * latent_size])
        y = deconv2d(y, 4 * latent_size, 2 * latent_size, 5, 2, name='dc1')
        y = deconv2d(y, 2 * latent_size, latent_size, 5, 2, name='dc2')
        y = deconv2d(y, latent_size, int(latent_size / 2), 5, 2, name='dc3')
        y = deconv2d(y, int(latent_size / 2), c, 5, 2, name='dc4', activation=tanh, use_batch_norm=False)
        y = reshape(y, [-1, output_dim])
    return y


def discriminator(x, args, reuse=False):
    """models/gan.py:257-287 (the reshape to [-1, 4*4*4*L] is kept literally: SURVEY App. C-2)."""
    use_bn = False if args.model == 'iwgan' else True
    final_activation = None if args.model in ['wgan', 'iwgan'] else sigmoid
    h, w, c = args.image_shape
    with arg_scope([conv2d], use_batch_norm=use_bn, activation=lrelu, reuse=reuse):
        x = reshape(x, [-1, h, w, c])
        x = conv2d(x, c, args.latent_size, 5, 2, name='c1', use_batch_norm=False)
        x = conv2d(x, args.latent_size, args.latent_size * 2, 5, 2, name='c2')
        x = conv2d(x, args.latent_size * 2, args.latent_size * 4, 5, 2, name='c3')
        x = reshape(x, [-1, 4 * 4 * 4 * args.latent_size])
        x = dense(x, 4 * 4 * 4 * args.latent_size, 1, use_batch_norm=False, activation=final_activation,
                  name='fc2', reuse=reuse)
        x = reshape(x, [-1])
    return x


# ------------------------------------------------------------------------------ one replica
class GanReplica(engine.GraphRunner):
    """The per-GPU replica ("tower") of models/gan.py:55-70 plus its optimizers (:46,79-81)."""

    # device scalar slots
    S_DREAL, S_DFAKE, S_SUMSQ, S_GP, S_GPCOEF, S_GLOSS_AUX, S_GAN_D, S_GAN_G = 0, 1, 2, 3, 4, 5, 6, 7

    def __init__(self, x_source, args, sess):
        self.args, self.sess, self.x_source = args, sess, x_source
        self.model = args.model
        B, L = args.batch_size, args.latent_size
        h, w, c = args.image_shape
        dev, dt = sess.device, sess.dtype
        self.B = B
        self.iwgan = self.model == 'iwgan'
        self.display_d_loss = getattr(args, 'display_d_loss', True)
        self.gp_per_sample = bool(getattr(args, 'gp_per_sample', False))
        self._pen_rows = None

        # ---- build the graph exactly as models/gan.py:55-63 does for one tower
        reset_graph()
        x_sym = placeholder((None, h * w * c))
        for _x, scope, gpu_id in tower_scope_range(x_sym, args.n_gpus, B, sess):
            with variable_scope('generator') as gnet:
                g_sym = generator(B, L, args, reuse=False)
            with variable_scope('discriminator') as dnet:
                d_real = discriminator(_x, args, reuse=False)
                d_fake = discriminator(g_sym, args, reuse=True)
                if self.iwgan:                     # gradient_penalty's third pass (models/gan.py:227)
                    d_hat = discriminator(placeholder((None, h * w * c)), args, reuse=True)
        self.rows_per_image = d_real.rows_per_image
        self.gnet, self.dnet = gnet, dnet

        # ---- bind to HBM.  D input buffer holds the three passes: slot 0 = x, 1 = g, 2 = x_hat
        self.nslots = 3 if self.iwgan else 2
        self.ws = K.Workspace(dev)
        self.d_store, self.g_store = engine.ParamStore(dev), engine.ParamStore(dev)
        self.D = engine.SeqNet(dnet, self.nslots * B, (h, w, c), dt, dev, self.d_store,
                               n_bn_passes=(1 if self.iwgan else 2), need_input_grad=True,
                               tangent_capacity=(B if self.iwgan else 0), ws=self.ws)
        # the generator writes straight into slot 1 and reads dL/dg from the same slot of D.dx
        self.G = engine.SeqNet(gnet, B, (1, 1, L), dt, dev, self.g_store, ws=self.ws,
                               out_act=self.D.x.view(B, B), out_grad=self.D.dx.view(B, B))
        # the generator passes of an iteration's n_disc_train critic steps run through the SAME generator variables
        # (models/gan.py:150-155,169-173): a second, forward-only binding takes them as one pass over n_disc_train x B
        # latent vectors with per-batch batch-norm statistics (_generate_ahead); each critic step copies its batch into slot 1
        nd = int(getattr(args, 'n_disc_train', 1) or 1)
        self.G_ahead, self._ahead = None, None
        if self.model != 'gan' and nd > 1 and os.environ.get('TDG_G_AHEAD', '1') != '0':        # (=0, diagnostics: one pass per critic step)
            self.n_ahead = nd
            self.G_ahead = engine.SeqNet(gnet, nd * B, (1, 1, L), dt, dev, self.g_store, ws=self.ws, forward_only=True,
                                         out_act=K.Act(nd * B, h, w, c, dt, dev, cs=self.D.x.cs))
            self.G_ahead.share_filters(self.G)
        self.D.declare_variables()
        self.G.declare_variables()
        self.d_store.allocate()
        self.g_store.allocate()
        gen = torch.Generator().manual_seed(sess.seed)
        self.G.init_variables(gen)
        self.D.init_variables(gen)
        self.g_opt, self.d_opt = init_optimizer(args, self.g_store), init_optimizer(args, self.d_store)   # :46
        self.img_elems = self.D.x.image_elems        # channel-padded image size in HBM
        # hipGraph replay of the two step bodies (launch-bound otherwise: ~600 small launches per iteration)
        self.init_graphs(args, sess)
        # the conv layer with the largest filter: its gradient slice is exchanged first (d_step)
        self._d_big_layer = max(self.D.conv_layers(), key=lambda L: self.d_store[L.wname].numel())
        self.x_stage = torch.zeros(B, h, w, c, dtype=torch.float32, device=dev)
        self._seed_g = None
        self.alpha = torch.zeros(B, dtype=torch.float32, device=dev)
        self.scal = torch.zeros(16, dtype=torch.float32, device=dev)
        self.refresh()

    # -- variables ---------------------------------------------------------------------------------
    def refresh(self):
        self.G.repack()
        self.D.repack()

    def load_variables(self, arrays):
        self.g_store.load(arrays)
        self.d_store.load(arrays)
        self.refresh()

    def stores(self):
        return [self.g_store, self.d_store]

    def optimizers(self):
        return {'optimizers/generator': self.g_opt, 'optimizers/discriminator': self.d_opt}

    def variables(self):
        d = self.g_store.state_dict()
        d.update(self.d_store.state_dict())
        return d

    def gradients(self):
        d = self.g_store.grads_dict()
        d.update(self.d_store.grads_dict())
        return d

    # -- pieces ------------------------------------------------------------------------------------
    def _load_real(self, x01):
        """models/gan.py:49-50: x = 2 * (flatten(x) - 0.5) into slot 0."""
        h, w, c = self.args.image_shape
        n = self.B * h * w * c
        if x01.dtype != torch.float32 or x01.numel() != n:
            raise ValueError('expected a float32 batch of %d values in [0,1], got %s %s' % (n, x01.dtype, tuple(x01.shape)))
        self.x_stage.copy_(x01.reshape(self.x_stage.shape))          # fixed address: the step bodies may be graph-captured


    def _rescale_real(self):
        h, w, c = self.args.image_shape
        _lib.call('tdg_affine_cast_rows', self.sess.dtype, K.ptr(self.x_stage), self.B * h * w, c, self.D.x.cs, 2.0, -0.5,
                  self.D.x.ptr(0), K.stream())

    def _generate(self, backward_follows=True):
        self.sess.random_normal(self.G.x, self.B, 'z')                 # models/gan.py:246
        self.G.forward(0, self.B, keep_pre=backward_follows)           # g lands in D.x slot 1

    def _generate_ahead(self):
        """The generator passes of the next n_disc_train critic steps: one draw of n_disc_train x B latent vectors, one
        batched forward pass, batch norm per batch of B (what n_disc_train runs of d_train_op compute one by one)."""
        self.sess.random_normal(self.G_ahead.x, self.n_ahead * self.B, 'z')
        self.G_ahead.forward_groups(self.n_ahead, self.B)

    def _ahead_ok(self, n_steps):
        """Tests that inject or stage ONE batch of z per critic step keep the pass-per-step form."""
        if self.G_ahead is None or n_steps != self.n_ahead or self.sess.inject.get('z'):
            return False
        staged = self.sess.staged.get('z')
        return staged is None or staged.numel() >= self.n_ahead * self.B * self.G_ahead.x.image_elems

    def _begin_ahead(self, n_steps):
        """Start of an iteration of n_steps critic steps: their generator passes, if this iteration may take them ahead."""
        ahead = self._ahead_ok(n_steps)
        if ahead:
            self._run('g_ahead', self._generate_ahead)
        return ahead

    def _stage_ahead(self):
        """Critic step number self._ahead of the iteration: its g (generated ahead) into slot 1.  Returns the suffix of the
        captured bodies' names: a body either holds the generator pass or it does not."""
        if self._ahead is None:
            return ''
        B = self.B
        self.D.x.view(B, B).buf.copy_(self.G_ahead.layers[-1].h.view(self._ahead * B, B).buf)
        return '+ahead'

    def samples(self, n):
        """(inputs, fake) as float32 NHWC in [-1, 1]: the first n images of the staged real batch and of a fresh
        generator pass (models/gan.py:99-100 `x[0:args.examples]`, `g[0:args.examples]`)."""
        h, w, c = self.args.image_shape
        n = min(n, self.B)
        self._rescale_real()
        self._generate(backward_follows=False)
        view = self.D.x.buf[:2 * self.B * h * w * self.D.x.cs].view(2, self.B, h, w, self.D.x.cs)
        both = view[:, :n, :, :, :c].float().cpu().numpy()
        return both[0], both[1]

    def _interpolate(self, out_slot=2):
        """models/gan.py:224-226 into slot 2 (or, elementwise in place, over g in slot 1 once nothing reads g any more)."""
        B = self.B
        self.sess.random_uniform(self.alpha, B, 'alpha')
        _lib.call('tdg_gp_interp', self.sess.dtype, self.D.x.ptr(0), self.D.x.ptr(B), K.ptr(self.alpha), B,
                  self.img_elems, self.D.x.ptr(out_slot * B), K.stream())

    def _d_forward(self, first_slot, nslots):
        B = self.B
        if self.iwgan:
            self.D.forward(first_slot * B, nslots * B)                 # BN-free critic: one batched pass
        else:
            for s in range(first_slot, first_slot + nslots):           # per-pass batch statistics and betas
                self.D.forward(s * B, B, bn_pass=s)
        return self.D.layers[-1].out

    def _means(self, scores):
        R = self.B * self.rows_per_image
        # scores = [D(x) | D(g) | ...] and the slots S_DREAL, S_DFAKE are adjacent: both means in one launch
        _lib.call('tdg_mean_segments_f32', K.ptr(scores, 0), 2, R, K.ptr(self.scal, 4 * self.S_DREAL), K.stream())

    def _seeds(self, *values):
        """Constant dL/d(score) seeds of slots 0..len-1 (None: a slot this pass does not read).  Every distinct
        assignment is a prefilled buffer that the critic's seed tensor is switched to; it is built once, in the eager
        warm-up pass of the body, so a training iteration carries no fill launches for them (17 before)."""
        bufs = self.__dict__.setdefault('_seed_bufs', {})
        buf = bufs.get(values)
        if buf is None:
            R = self.B * self.rows_per_image
            buf = torch.zeros_like(self.D.layers[-1].seed)
            for slot, v in enumerate(values):
                if v is not None:
                    buf[slot * R:(slot + 1) * R] = v
            bufs[values] = buf
        self.D.layers[-1].seed = buf

    def _penalty_from_v(self, tangent_seed=False, slot=2):
        """slopes = sqrt(sum over the WHOLE batch tensor) (models/gan.py:229), penalty (:230).  With `tangent_seed` also
        u = d(lambda * penalty)/dv into the tangent pass's input.  --gp_per_sample (opt-in, App. C-4): one norm per image,
        penalty = mean_i (|v_i| - 1)^2."""
        B = self.B
        if self.gp_per_sample:
            if self._pen_rows is None:
                self._pen_rows = torch.zeros(B, dtype=torch.float32, device=self.sess.device)
                self._u_sink = K.Act(B, *self.args.image_shape, self.sess.dtype, self.sess.device) if not self.D.tangent_capacity else None
            u = self.D.tan_in if self.D.tangent_capacity else self._u_sink
            _lib.call('tdg_gp_rows', self.sess.dtype, self.D.dx.ptr(slot * B), B, self.img_elems, GP_LAMBDA, K.ptr(self._pen_rows),
                      u.ptr(0), K.stream())
            _lib.call('tdg_mean_f32', K.ptr(self._pen_rows), B, K.ptr(self.scal, 4 * self.S_GP), K.stream())
            return
        w = self.ws.ensure(4096)                                      # sum of squares and the penalty scalars in one launch
        _lib.call('tdg_gp_sumsq', self.sess.dtype, self.D.dx.ptr(slot * B), B * self.img_elems, K.ptr(self.scal, 4 * self.S_SUMSQ), GP_LAMBDA,
                  K.ptr(self.scal, 4 * self.S_GP), K.ptr(w), w.numel(), K.stream())
        if tangent_seed:
            _lib.call('tdg_scale_by_dev', self.sess.dtype, self.D.dx.ptr(slot * B), B * self.img_elems,
                      K.ptr(self.scal, 4 * self.S_GPCOEF), self.D.tan_in.ptr(0), K.stream())

    def big_slice(self):
        """[lo, hi) of the critic's flat bucket holding its largest filter and that layer's bias (adjacent variables)."""
        big, store = self._d_big_layer, self.d_store
        lo, hi = store.index[big.wname][0], store.index[big.bname][0] + (big.spec.out_size + 3) // 4 * 4
        # the variables declared behind it belong to fc2 (a row kernel: its gradients are complete before any conv filter
        # gradient is taken), so the slice runs to the end of the bucket: two collectives per critic step instead of three
        # (the third was a 51 KB one paying a full collective latency)
        if all(L.rowdot for L in self.D.layers[big.idx + 1:]) and self.D.n_bn_passes == 1 and not any(L.spec.normed for L in self.D.layers):
            hi = store.size
        return lo, hi

    # -- steps -------------------------------------------------------------------------------------
    def d_step(self, x01):
        """One run of d_train_op (models/gan.py:152,171): d_loss gradients w.r.t. D, averaged, applied.

        Several replicas (iwgan): the critic's largest filter (c3: 80 % of its parameters) gets its last contribution
        -- the tangent-pass filter gradient -- FIRST, and its slice of the flat bucket starts its RCCL all-reduce while
        the remaining tangent-pass filter gradients are still being computed (second captured body); the small rest of
        the bucket follows.  One replica: a single captured body, no exchange."""
        self._load_real(x01)
        sess, store = self.sess, self.d_store
        tag = self._stage_ahead()
        if sess.world_size > 1 and self.iwgan and os.environ.get('TDG_DSPLIT', '1') != '0':      # (TDG_DSPLIT=0, diagnostics: one body, one exchange)
            lo, hi = self.big_slice()
            self._run('d_grads_a' + tag, self._d_grads_a)
            self._scale = sess.allreduce_split(store.grads, lo, hi, between=lambda: self._run('d_grads_b', self._d_grads_b))
            sess.assert_finite(store, 'd_step')                       # after EVERY slice is summed: a NaN/Inf on one
                                                                      # replica is in every replica's bucket by now
        elif sess.world_size == 1 and not sess.check_numerics and os.environ.get('TDG_ONE_BODY', '1') != '0':
            # one replica, nothing between the gradients and their update (no exchange, no finite check): ONE captured body --
            # every boundary between two graph launches is ~9 us of idle GPU, 6 of them per iteration here
            self._scale = 1.0
            self._run('d_step' + tag, self._d_grads_and_apply)
            self.sess.global_step += 1
            return
        else:
            self._run('d_grads' + tag, self._d_grads)
            sess.assert_finite(store, 'd_step')
            self._scale = average_gradients(sess, store)              # models/gan.py:77 (RCCL, outside the graphs)
        self._run('d_apply', self._d_apply)
        self.sess.global_step += 1

    def _d_grads_and_apply(self):
        self._d_grads()
        self._d_apply()

    def _g_grads_and_apply(self):
        self._g_grads()
        self._g_apply()

    def _d_apply(self):
        self.d_opt.step(self._scale)                                  # models/gan.py:81
        self.D.repack()

    def _g_apply(self):
        self.g_opt.step(self._scale)                                  # models/gan.py:80
        self.G.repack()

    def _d_grads(self):
        self._d_grads_a(whole=True)

    def _clip_critic(self):
        """--wgan_clip c (opt-in): clamp every critic variable to [-c, c] BEFORE the critic step -- what
        models/gan.py:142-148 evidently intends; in the reference the clip ops never run (the `control_dependencies` block
        wraps an already-created op), which is this build's default too."""
        c = float(getattr(self.args, 'wgan_clip', 0.0) or 0.0)
        if c > 0.0:
            _lib.call('tdg_clamp', K.ptr(self.d_store.params), self.d_store.size, -c, c, K.stream())
            self.D.repack()

    def _d_grads_a(self, whole=False):
        B, R = self.B, self.B * self.rows_per_image
        self._clip_critic()
        self._rescale_real()
        if self._ahead is None:
            self._generate(backward_follows=False)                     # the critic step does not back-propagate into G
        if self.iwgan:
            self._interpolate()
        scores = self._d_forward(0, self.nslots)
        self._means(scores)
        # seeds: d/d(d_real) of -mean(d_real), d/d(d_fake) of mean(d_fake), tf.gradients(d_interpolates, ...) (:228)
        self._seeds(-1.0 / R, 1.0 / R, 1.0) if self.iwgan else self._seeds(-1.0 / R, 1.0 / R)
        if self.iwgan:
            # first-order backward with the conv filter gradients deferred: they are taken together with the tangent-pass
            # ones, one GEMM per layer over [D(x) rows | D(g) rows | tangent rows] (engine.SeqNet.merged_wgrad)
            self.D.backward(0, 3 * B, want_params=True, want_dx=True, param_images=(0, 2 * B), dx_images=(2 * B, B),
                            defer_wgrad=True)
            # u = d(lambda * penalty)/dv = lambda * 2 (s-1)/s * v, then the tangent pass
            self._penalty_from_v(tangent_seed=True)
            self.D.tangent_forward(2 * B, B, acc=True)
            convs = list(reversed(self.D.conv_layers())) if whole else [self._d_big_layer]   # several replicas: the
            self.D.merged_wgrad(2 * B, B, convs)                      # largest filter first, the rest in _d_grads_b
        else:
            self.D.backward(0, B, bn_pass=0, want_params=True, acc=False)
            self.D.backward(B, B, bn_pass=1, want_params=True, acc=True)

    def _d_grads_b(self):
        if self.iwgan:
            rest = [L for L in reversed(self.D.conv_layers()) if L is not self._d_big_layer]
            self.D.merged_wgrad(2 * self.B, self.B, rest)

    def g_step(self, x01):
        """One run of [g_train_op, losses] (models/gan.py:153,172).

        Several replicas (iwgan with the reported d_loss): the generator's gradients need only D(g) -- the critic passes
        on x and x_hat and the penalty's backward exist for the REPORTED d_loss -- so the gradient path runs first, its
        bucket starts its RCCL all-reduce asynchronously, and the display-only part (3 of the step's 5 critic passes)
        runs underneath as a second captured body.  Rows are independent in the BN-free critic: same numbers."""
        if self.display_d_loss:
            self._load_real(x01)
        sess, store = self.sess, self.g_store
        if sess.world_size > 1 and self.iwgan and self.display_d_loss and os.environ.get('TDG_GSPLIT', '1') != '0':   # (diagnostics)
            self._run('g_grads_a', self._g_grads_critical)
            work = sess.allreduce_async(store.grads)
            self._run('g_grads_b', self._g_display_d_loss)
            work.wait()
            sess.assert_finite(store, 'g_step')
            self._scale = 1.0 / sess.world_size
        elif sess.world_size == 1 and not sess.check_numerics and os.environ.get('TDG_ONE_BODY', '1') != '0':
            self._scale = 1.0
            self._run('g_step', self._g_grads_and_apply)
            self.sess.global_step += 1
            return
        else:
            self._run('g_grads', self._g_grads)
            sess.assert_finite(store, 'g_step')
            self._scale = average_gradients(sess, store)              # models/gan.py:76
        self._run('g_apply', self._g_apply)
        self.sess.global_step += 1

    def _g_grads_critical(self):
        """g_loss = -mean(D(g)) and its gradient w.r.t. the generator: slot 1 only."""
        B, R = self.B, self.B * self.rows_per_image
        self._generate()
        scores = self._d_forward(1, 1)
        _lib.call('tdg_mean_f32', K.ptr(scores, 4 * R), R, K.ptr(self.scal, 4 * self.S_DFAKE), K.stream())
        self._seeds(None, -1.0 / R)
        self.D.backward(B, B, want_params=False, want_dx=True)
        self.G.backward(0, B, want_params=True)

    def _g_display_d_loss(self):
        """The reported d_loss (models/gan.py:199-205 fetched beside g_train_op): D(x), D(x_hat), the penalty."""
        B, R = self.B, self.B * self.rows_per_image
        self._rescale_real()
        # g (slot 1) has served the gradient path above: x_hat takes its place, so that D(x) and D(x_hat) are ONE batched pass
        # over the adjacent slots 0 and 1 (two separate 512-image passes cost 0.1 ms more per iteration)
        self._interpolate(out_slot=1)
        scores = self._d_forward(0, 2)
        _lib.call('tdg_mean_f32', K.ptr(scores, 0), R, K.ptr(self.scal, 4 * self.S_DREAL), K.stream())
        self._seeds(None, 1.0)
        self.D.backward(B, B, want_params=False, want_dx=True)
        self._penalty_from_v(slot=1)

    def _g_grads(self):
        B, R = self.B, self.B * self.rows_per_image
        self._generate()
        if self.display_d_loss:
            self._rescale_real()
            if self.iwgan:
                self._interpolate()
            scores = self._d_forward(0, self.nslots)
        else:
            scores = self._d_forward(1, 1)
        self._means(scores)
        if self.iwgan and self.display_d_loss:
            self._seeds(None, -1.0 / R, 1.0)                          # d/d(d_fake) of g_loss = -mean(d_fake); x_hat's 1
            self.D.backward(B, 2 * B, want_params=False, want_dx=True)
            self._penalty_from_v()
        else:
            self._seeds(None, -1.0 / R)
            self.D.backward(B, B, bn_pass=1, want_params=False, want_dx=True)
        self.G.backward(0, B, want_params=True)                       # seed = D.dx slot 1 (aliased)

    # -- vanilla GAN: one batch, both updates from the same forward (models/gan.py:110-131) -------------
    def _gan_grads(self):
        B, R = self.B, self.B * self.rows_per_image
        last = self.D.layers[-1]
        self._rescale_real()
        self._generate()
        scores = self._d_forward(0, 2)                               # D(x) with beta set 0, D(g) with beta set 1
        if self._seed_g is None:
            self._seed_g = torch.zeros(R, dtype=torch.float32, device=self.sess.device)
        _lib.call('tdg_gan_logloss', K.ptr(scores, 0), K.ptr(scores, 4 * R), R, K.ptr(last.seed, 0), K.ptr(last.seed, 4 * R),
                  K.ptr(self._seed_g), K.ptr(self.scal, 4 * self.S_GAN_D), K.stream())
        self.D.backward(0, B, bn_pass=0, want_params=True, acc=False)
        self.D.backward(B, B, bn_pass=1, want_params=True, acc=True)
        # generator: same forward, g_loss seed on the fake scores, gradient w.r.t. g only
        last.seed[R:2 * R].copy_(self._seed_g)
        self.D.backward(B, B, bn_pass=1, want_params=False, want_dx=True)
        self.G.backward(0, B, want_params=True)

    def _gan_apply(self):
        self.d_opt.step(self._scale)
        self.g_opt.step(self._scale)
        self.D.repack()
        self.G.repack()

    def gan_step(self, x01):
        self._load_real(x01)
        self._run('gan_grads', self._gan_grads)
        self.sess.assert_finite(self.d_store, 'gan_step')
        self.sess.assert_finite(self.g_store, 'gan_step')
        self._scale = average_gradients(self.sess, self.d_store)
        average_gradients(self.sess, self.g_store)
        self._run('gan_apply', self._gan_apply)
        self.sess.global_step += 2                                    # both apply ops bump it (models/gan.py:80-81)

    def losses(self):
        """Host read-back of the device scalars (one sync): models/gan.py:193-205."""
        s = self.sess.report_scalars(self.scal, mean=getattr(self.args, 'mean_loss', False)).cpu().tolist()
        tower = self.sess.world_size - 1          # the dict keeps the LAST tower's tensors (util.py:187-193, App. C-11)
        if self.model == 'gan':
            return collection_to_dict([('tower_%d/g_loss:0' % tower, s[self.S_GAN_G]),
                                       ('tower_%d/d_loss:0' % tower, s[self.S_GAN_D])])
        g_loss = -s[self.S_DFAKE]
        d_loss = s[self.S_DFAKE] - s[self.S_DREAL]
        if self.iwgan:
            d_loss += GP_LAMBDA * s[self.S_GP]
        return collection_to_dict([('tower_%d/g_loss:0' % tower, g_loss),
                                   ('tower_%d/d_loss:0' % tower, d_loss)])

    def train_func(self, sess=None, args=None):
        """_train_wgan / _train_iwgan helper (models/gan.py:150-155,169-173): n_disc_train D steps,
        then one G step, each on a fresh batch; returns the loss dict of the G step's batch."""
        args = args or self.args
        if self.model == 'gan':                                       # _train_gan: one run of both train ops
            self.gan_step(self.x_source.next_batch())
            return self.losses()
        ahead = self._begin_ahead(args.n_disc_train)
        try:
            for i in range(args.n_disc_train):
                self._ahead = i if ahead else None
                self.d_step(self.x_source.next_batch())
        finally:
            self._ahead = None
        self.g_step(self.x_source.next_batch())
        return self.losses()


def gan(x, args, sess=None):
    """models/gan.py:39-91.  `x` is the per-replica input source (`.next_batch()` -> device float32
    [B, H, W, C] in [0,1]); returns the training function train.py:246,307 expects."""
    from ..runtime import Session
    sess = sess or Session(dtype=getattr(args, 'dtype_code', K.BF16), seed=getattr(args, 'seed', 0) or 0)
    replica = GanReplica(x, args, sess)

    def train_func(sess_=None, args_=None):
        return replica.train_func(sess_, args_)
    train_func.replica = replica
    return train_func
