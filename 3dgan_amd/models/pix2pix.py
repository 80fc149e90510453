"""pix2pix conditional GAN on MI355X -- the reference's gen-2 plugin `hem/models/pix2pix.py`
(arguments :36-78, __init__ :81-148, train :151-156, generator :160-228, discriminator :232-259,
loss :263-304) on the HIP kernels.

Kept: the plugin contract (`name`, `arguments()`, `__init__(x_y, args)`, `train(sess, args, feed_dict)`
-> loss dict), the generator/discriminator written against conv2d/deconv2d + arg_scope with the
reference's variable names (`generator/enocder/vars/1/weights` -- sic --, `generator/decoder/vars/8/bias`,
`discriminator/vars/m5/weights`), init N(0, 0.02) for weights and biases, batch norm on every decoder
layer, the hard-coded L1 weight 10.0 (C-9), three fresh batches per `train()`.

MI355X-native: activations are NHWC; every skip `tf.concat([y, e_k], axis=1)` is a ZERO-COPY concat --
the decoder layer and the encoder layer each write their channel window of one buffer, and in the backward
pass the encoder's two gradient paths (next encoder layer + skip) are summed by the `accumulate` epilogue of
the backward-data GEMM; D(x,y) and D(x,G(x)) run as one batched pass over [x|y ; x|g]; G(x) lands directly
in channel 3 of D's input and dL/dG(x) is read from channel 3 of D's input gradient.

`--dropout` (keep probability of decoder layers 1-3, hem/models/pix2pix.py:204-208) and `--noise input|latent|end`
(a U(-1,1) channel concatenated to the generator input / the 1x1 bottleneck / the last decoder layer's input,
:183-186,204-206,223-225) are executed by the U-Net below: each noise tensor is one more channel window of a zero-copy
concat, drawn per generator pass from the device Philox stream (keys 'noise_input', 'noise_latent', 'noise_end').
"""
import torch

from .. import _lib
from .. import kernels as K
from .. import engine
from ..ops.layers import conv2d, deconv2d, concat, arg_scope, variable_scope, placeholder, reset_graph, random_uniform
from ..ops.activations import Activation, tanh
from .._lib import ACT_LRELU
from ..util import tower_scope_range, average_gradients, init_optimizer, collection_to_dict
from .ModelPlugin import ModelPlugin

L1_WEIGHT = 10.0            # l_term, hem/models/pix2pix.py:284 (the --lambda flag is ignored by the reference)


def _lrelu(leak):
    return Activation('lrelu', ACT_LRELU, leak)


class pix2pix(ModelPlugin, engine.GraphRunner):
    name = 'pix2pix'

    @staticmethod
    def arguments():
        """hem/models/pix2pix.py:36-78 (argparse kwargs per flag)."""
        return {
            '--skip_layers': {'action': 'store_true', 'default': 'false', 'help': 'Adds skip layers to the generator.'},
            '--noise': {'type': str, 'nargs': '*', 'choices': ['input', 'latent', 'end'], 'default': []},
            '--dropout': {'type': float, 'default': 0},
            '--batch_norm_disc': {'action': 'store_true', 'default': False},
            '--batch_norm_gen': {'action': 'store_true', 'default': False},
            '--examples': {'type': int, 'default': 64},
            '--n_disc_train': {'type': int, 'default': 1},
            '--add_l1': {'action': 'store_true', 'default': False},
            '--lambda': {'type': float, 'default': 10.0},
        }

    # ------------------------------------------------------------------------------ builders
    @staticmethod
    def generator(x, args, reuse=False):
        """hem/models/pix2pix.py:160-228 (NHWC; 256x256x3 -> 256x256x1)."""
        init = 'normal0.02'
        with arg_scope([conv2d], reuse=reuse, use_batch_norm=args.batch_norm_gen, filter_size=4, stride=2, init=init,
                       activation=_lrelu(0.2)):
            with variable_scope('enocder'):
                if 'input' in args.noise:                                                  # :183-186
                    noise = random_uniform([args.batch_size, 256, 256, 1], minval=-1.0, maxval=1.0)
                    e1 = conv2d(concat([x, noise]), 4, 64, name='1', use_batch_norm=False)
                else:
                    e1 = conv2d(x, 3, 64, name='1', use_batch_norm=False)
                e2 = conv2d(e1, 64, 128, name='2')
                e3 = conv2d(e2, 128, 256, name='3')
                e4 = conv2d(e3, 256, 512, name='4')
                e5 = conv2d(e4, 512, 512, name='5')
                e6 = conv2d(e5, 512, 512, name='6')
                e7 = conv2d(e6, 512, 512, name='7')
                e8 = conv2d(e7, 512, 512, name='8')
        with arg_scope([deconv2d, conv2d], reuse=reuse, use_batch_norm=True, filter_size=4, stride=2, init=init,
                       activation=_lrelu(0.0)):
            with variable_scope('decoder'):
                if 'latent' in args.noise:                                                 # :204-206
                    noise = random_uniform([args.batch_size, 1, 1, 512], minval=-1.0, maxval=1.0)
                    y = deconv2d(concat([e8, noise]), 1024, 512, name='1', dropout=args.dropout)
                else:
                    y = deconv2d(e8, 512, 512, name='1', dropout=args.dropout)
                y = concat([y, e7])
                y = deconv2d(y, 1024, 512, name='2', dropout=args.dropout)
                y = concat([y, e6])
                y = deconv2d(y, 1024, 512, name='3', dropout=args.dropout)
                y = concat([y, e5])
                y = deconv2d(y, 1024, 512, name='4')
                y = concat([y, e4])
                y = deconv2d(y, 1024, 256, name='5')
                y = concat([y, e3])
                y = deconv2d(y, 512, 128, name='6')
                y = concat([y, e2])
                y = deconv2d(y, 256, 64, name='7')
                y = concat([y, e1])
                if 'end' in args.noise:                                                    # :223-225
                    noise = random_uniform([args.batch_size, 128, 128, 1], minval=-1.0, maxval=1.0)
                    y = deconv2d(concat([y, noise]), 129, 1, name='8', activation=tanh)
                else:
                    y = deconv2d(y, 128, 1, name='8', activation=tanh)
        return y

    @staticmethod
    def discriminator(x, y, args, reuse=False):
        """hem/models/pix2pix.py:232-259 (PatchGAN); returns the logits [B, 8, 8, 1]."""
        with arg_scope([conv2d], reuse=reuse, use_batch_norm=args.batch_norm_disc, activation=_lrelu(0.2), init='normal0.02',
                       filter_size=4, stride=2):
            x_y = concat([x, y])
            h = conv2d(x_y, 4, 64, name='m1', use_batch_norm=False)
            h = conv2d(h, 64, 128, name='m2')
            h = conv2d(h, 128, 256, name='m3')
            h = conv2d(h, 256, 512, name='m4')
            h = conv2d(h, 512, 1, name='m5', activation=None)
        return h

    # ------------------------------------------------------------------------------ construction
    S_DREAL, S_DFAKE, S_GFAKE, S_L1, S_RMSE = 0, 1, 2, 3, 4

    def __init__(self, x_y, args, sess=None):
        from ..runtime import Session
        self.args, self.x_y = args, x_y
        self.sess = sess = sess or Session(dtype=getattr(args, 'dtype_code', K.BF16), seed=getattr(args, 'seed', 0) or 0)
        for flag, default in (('noise', []), ('dropout', 0), ('batch_norm_gen', False), ('batch_norm_disc', False), ('add_l1', False)):
            if not hasattr(args, flag):
                setattr(args, flag, default)
        B = self.B = args.batch_size
        H = W = 256
        dev, dt = sess.device, sess.dtype

        reset_graph()
        xs, ys = placeholder((None, H, W, 3)), placeholder((None, H, W, 1))
        for _xy, scope, gpu_id in tower_scope_range((xs, ys), args.n_gpus, B, sess):
            with variable_scope('generator'):
                g = pix2pix.generator(_xy[0], args, reuse=False)
            with variable_scope('discriminator') as dnet:
                d_real = pix2pix.discriminator(_xy[0], _xy[1], args, reuse=False)
                d_fake = pix2pix.discriminator(_xy[0], g, args, reuse=True)
        from ..ops import layers as Lyr
        self.enet, self.dec_net, self.dnet = Lyr._nets['generator/enocder'], Lyr._nets['generator/decoder'], dnet

        self.ws = K.Workspace(dev)
        self.g_store, self.d_store = engine.ParamStore(dev), engine.ParamStore(dev)
        # D input holds both passes: images [0,B) = [x | y], images [B,2B) = [x | G(x)]
        self.D = engine.SeqNet(dnet, 2 * B, (H, W, 4), dt, dev, self.d_store, n_bn_passes=(2 if args.batch_norm_disc else 1),
                               need_input_grad=True, ws=self.ws)
        self.D.declare_variables()
        self.U = UNet(self.enet, self.dec_net, B, H, W, dt, dev, self.g_store, self.ws,
                      x_in=self.D.x.view(0, B).window(0, 3),
                      g_out=self.D.x.view(B, B).window(3, 1),
                      g_grad=self.D.dx.view(B, B).window(3, 1), sess=sess)
        self.d_store.allocate()
        self.g_store.allocate()
        gen = torch.Generator().manual_seed(sess.seed)
        self.U.init_variables(gen)
        self.D.init_variables(gen)
        self.g_opt, self.d_opt = init_optimizer(args, self.g_store), init_optimizer(args, self.d_store)
        self.x_stage = torch.zeros(B, H, W, 3, dtype=torch.float32, device=dev)
        self.y_stage = torch.zeros(B, H, W, 1, dtype=torch.float32, device=dev)
        self.scal = torch.zeros(8, dtype=torch.float32, device=dev)
        self.init_graphs(args, sess)               # the step bodies below are captured into hipGraphs and replayed
        self.refresh()

    # ---- variables -----------------------------------------------------------------------------------
    def stores(self):
        return [self.g_store, self.d_store]

    def optimizers(self):
        return {'optimizers/generator': self.g_opt, 'optimizers/discriminator': self.d_opt}

    def refresh(self):
        self.U.repack()
        self.D.repack()

    def load_variables(self, arrays):
        self.g_store.load(arrays)
        self.d_store.load(arrays)
        self.refresh()

    def variables(self):
        d = self.g_store.state_dict()
        d.update(self.d_store.state_dict())
        return d

    def gradients(self):
        d = self.g_store.grads_dict()
        d.update(self.d_store.grads_dict())
        return d

    # ---- pieces ------------------------------------------------------------------------------------------
    def _stage(self, batch):
        """The batch at fixed device addresses (the step bodies may be graph-captured)."""
        x01, y01 = batch
        self.x_stage.copy_(x01.reshape(self.x_stage.shape))
        self.y_stage.copy_(y01.reshape(self.y_stage.shape))

    def _rescale(self):
        """hem.rescale((0,1) -> (-1,1)) of both halves (hem/models/pix2pix.py:103-104) into D's input slots."""
        B, dt = self.B, self.sess.dtype
        rows, cs = B * 256 * 256, self.D.x.cs
        if cs == 4:                                # [x | y] and [x | (G(x): written by the generator pass)] in one pass
            _lib.call('tdg_affine_cast_pair', dt, K.ptr(self.x_stage), 3, K.ptr(self.y_stage), 1, rows, 2.0, -0.5, self.D.x.ptr(0),
                      self.D.x.ptr(B), K.stream())
        else:
            for img0 in (0, B):
                _lib.call('tdg_affine_cast_rows', dt, K.ptr(self.x_stage), rows, 3, cs, 2.0, -0.5, self.D.x.ptr(img0), K.stream())
            _lib.call('tdg_affine_cast_rows', dt, K.ptr(self.y_stage), rows, 1, cs, 2.0, -0.5, self.D.x.window(3, 1).ptr(0), K.stream())
        if self.U.xn is not None:                  # --noise input: the generator reads [x | noise] from its own buffer
            _lib.call('tdg_affine_cast_rows', dt, K.ptr(self.x_stage), rows, 3, self.U.xn.cs, 2.0, -0.5, self.U.xn.ptr(0), K.stream())

    def _load(self, batch):
        self._stage(batch)
        self._rescale()

    def _d_forward(self, first, count):
        B = self.B
        if self.args.batch_norm_disc:
            for s in range(first, first + count):
                self.D.forward(s * B, B, bn_pass=s)
        else:
            self.D.forward(first * B, count * B)
        return self.D.layers[-1].h

    def _xent(self, mode):
        last = self.D.layers[-1]
        rows = self.B * last.h.h * last.h.w
        _lib.call('tdg_p2p_xent', self.sess.dtype, last.h.ptr(0), rows, last.h.cs, mode, last.gout.ptr(0),
                  K.ptr(self.scal, 4 * self.S_DREAL), K.stream())

    def _l1(self, with_grad):
        B, dt = self.B, self.sess.dtype
        y = self.D.x.view(0, B).window(3, 1)
        g = self.D.x.view(B, B).window(3, 1)
        dg = self.D.dx.view(B, B).window(3, 1)
        w = self.ws.ensure(4096)
        _lib.call('tdg_p2p_l1', dt, y.ptr(0), g.ptr(0), B * 256 * 256, y.cs, L1_WEIGHT,
                  dg.ptr(0) if with_grad else None, dg.cs, K.ptr(self.scal, 4 * self.S_L1), K.ptr(w), w.numel(), K.stream())

    # ---- steps ---------------------------------------------------------------------------------------------
    def d_step(self, batch):
        self._stage(batch)
        self._run('d_grads', self._d_grads)
        self.sess.assert_finite(self.d_store, 'd_step')
        self._scale = average_gradients(self.sess, self.d_store)      # RCCL, outside the graphs
        self._run('d_apply', self._d_apply)
        self.sess.global_step += 1

    def _d_grads(self):
        B = self.B
        self._rescale()
        self.U.forward(backward_follows=False)
        self._d_forward(0, 2)
        self._xent(1)
        if self.args.batch_norm_disc:
            self.D.backward(0, B, bn_pass=0, want_params=True, acc=False)
            self.D.backward(B, B, bn_pass=1, want_params=True, acc=True)
        else:
            self.D.backward(0, 2 * B, want_params=True)

    def _d_apply(self):
        self.d_opt.step(self._scale)
        self.D.repack()

    def g_step(self, batch):
        self._stage(batch)
        self._run('g_grads', self._g_grads)
        self.sess.assert_finite(self.g_store, 'g_step')
        self._scale = average_gradients(self.sess, self.g_store)
        self._run('g_apply', self._g_apply)
        self.sess.global_step += 1

    def _g_grads(self):
        B = self.B
        self._rescale()
        self.U.forward()
        self._d_forward(1, 1)                  # only D(x, G(x)): sess.run(g_train_op) evaluates nothing of the real pass
        self._xent(2)                          # (hem/models/pix2pix.py:153; the losses are fetched by report() on a third batch)
        if self.args.batch_norm_disc:
            self.D.backward(B, B, bn_pass=1, want_params=False, want_dx=True)
        else:
            self.D.backward(B, B, want_params=False, want_dx=True)
        if self.args.add_l1:
            self._l1(True)
        self.U.backward()

    def _g_apply(self):
        self.g_opt.step(self._scale)
        self.U.repack()

    def _report_body(self):
        self._rescale()
        self.U.forward(backward_follows=False)
        self._d_forward(0, 2)
        self._xent(0)
        self._l1(False)

    def report(self, batch):
        """sess.run(all_losses) on a third batch (hem/models/pix2pix.py:155)."""
        self._stage(batch)
        self._run('report', self._report_body)
        s = self.sess.report_scalars(self.scal, mean=getattr(self.args, 'mean_loss', False)).cpu().tolist()
        r = self.sess.world_size - 1     # the dict keeps the LAST tower's tensors (util.py:187-193, App. C-11), as models/gan.py does
        g_total = s[self.S_GFAKE] + (L1_WEIGHT * s[self.S_L1] if self.args.add_l1 else 0.0)
        g_name = 'loss/generator/add:0' if self.args.add_l1 else 'loss/generator/g_fake:0'
        return collection_to_dict([('tower_%d/loss/generator/l1:0' % r, s[self.S_L1]), ('tower_%d/%s' % (r, g_name), g_total),
                                   ('tower_%d/loss/generator/total:0' % r, g_total),
                                   ('tower_%d/loss/discriminator/d_real:0' % r, s[self.S_DREAL]),
                                   ('tower_%d/loss/discriminator/d_fake:0' % r, s[self.S_DFAKE]),
                                   ('tower_%d/loss/discriminator/total:0' % r, s[self.S_DREAL] + s[self.S_DFAKE]),
                                   ('tower_%d/loss/rmse:0' % r, s[self.S_RMSE])])

    def train(self, sess=None, args=None, feed_dict=None):
        """hem/models/pix2pix.py:151-156."""
        args = args or self.args
        for _ in range(args.n_disc_train):
            self.d_step(self.x_y.next_batch())
        self.g_step(self.x_y.next_batch())
        return self.report(self.x_y.next_batch())


# ------------------------------------------------------------------------------------------------------
class UNet:
    """The 8-down / 8-up generator with zero-copy skip concats (see module docstring).

    cat[i] (i = 2..8) is decoder layer i's input [d_{i-1} | e_{9-i}]; gcat[i] its gradient.  Encoder layer k
    (k <= 7) writes its activation into the right window of cat[9-k] and receives its gradient -- skip path
    first, main path accumulated on top -- in the right window of gcat[9-k].
    """

    def __init__(self, enet, dnet, B, H, W, dtype, device, store, ws, x_in, g_out, g_grad, sess=None):
        self.B, self.dtype, self.device, self.store, self.ws = B, dtype, device, store, ws
        self.sess = sess
        self.enet, self.dnet = enet, dnet
        E, Dc = enet.layers, dnet.layers
        assert len(E) == 8 and len(Dc) == 8
        self.enc_bn = [l.use_bn for l in E]
        A = lambda h, w, c: K.Act(B, h, w, c, dtype, device)
        # --noise (hem/models/pix2pix.py:183-186,204-206,223-225): read off the layer widths the builder recorded
        self.noise_input = E[0].in_size == x_in.c + 1
        self.noise_latent = Dc[0].in_size == 2 * E[7].out_size
        self.noise_end = Dc[7].in_size == Dc[6].out_size + E[0].out_size + 1
        self.xn = A(H, W, x_in.c + 1) if self.noise_input else None
        if self.noise_input:
            x_in = self.xn
        # spatial size of e_k
        es = [(H >> k, W >> k) for k in range(1, 9)]
        self.cat, self.gcat = {}, {}
        for i in range(2, 9):
            cd, ce = Dc[i - 2].out_size, E[8 - i].out_size
            extra = 1 if (i == 8 and self.noise_end) else 0
            if Dc[i - 1].in_size != cd + ce + extra:
                raise ValueError('decoder layer %d expects %d input channels, skip concat provides %d' % (i, Dc[i - 1].in_size, cd + ce + extra))
            h, w = es[8 - i]
            self.cat[i], self.gcat[i] = A(h, w, cd + ce + extra), A(h, w, cd + ce + extra)
        self.x_in = x_in
        # encoder activations / gradients
        self.e_h, self.e_g, self.e_pre, self.e_delta = {}, {}, {}, {}
        for k in range(1, 9):
            co = E[k - 1].out_size
            h, w = es[k - 1]
            if k <= 7:
                cd = Dc[7 - k].out_size
                self.e_h[k] = self.cat[9 - k].window(cd, co)
                self.e_g[k] = self.gcat[9 - k].window(cd, co)         # dL/d(e_k output); == delta when no batch norm
            elif self.noise_latent:                # [e8 | noise]: e8 is the left window of decoder layer 1's input
                self.lat, self.glat = A(h, w, 2 * co), A(h, w, 2 * co)
                self.e_h[k], self.e_g[k] = self.lat.window(0, co), self.glat.window(0, co)
            else:
                self.e_h[k], self.e_g[k] = A(h, w, co), A(h, w, co)
            if self.enc_bn[k - 1]:
                self.e_pre[k], self.e_delta[k] = A(h, w, co), A(h, w, co)
            else:
                self.e_delta[k] = self.e_g[k]
        # decoder
        self.d_pre, self.d_delta, self.d_h, self.d_g = {}, {}, {}, {}
        for i in range(1, 9):
            co = Dc[i - 1].out_size
            h, w = (es[7 - i] if i <= 7 else (H, W))
            self.d_pre[i], self.d_delta[i] = A(h, w, co), A(h, w, co)
            if i <= 7:
                self.d_h[i], self.d_g[i] = self.cat[i + 1].window(0, co), self.gcat[i + 1].window(0, co)
            else:
                self.d_h[i], self.d_g[i] = g_out, g_grad
        # convs (descriptors carry the strides of the buffers each GEMM form touches)
        self.e_conv, self.d_conv = {}, {}
        for k in range(1, 9):
            spec = E[k - 1]
            big = self.x_in if k == 1 else self.e_h[k - 1]
            small = self.e_pre[k] if self.enc_bn[k - 1] else self.e_h[k]
            self.e_conv[k] = K.Conv(big, small, spec.k, spec.k, spec.stride, 1, 1)
        for i in range(1, 9):
            spec = Dc[i - 1]
            small = self._d_in(i)
            self.d_conv[i] = K.Conv(self.d_pre[i], small, spec.k, spec.k, spec.stride, 1, 1)
        self.d_stats = {i: torch.zeros(2 * Dc[i - 1].out_size, dtype=torch.float32, device=device) for i in range(1, 9)}
        # tf.nn.dropout(h, keep_prob=dropout) on decoder layers built with dropout > 0 (hem/models/pix2pix.py:204-208,
        # hem/ops/layers.py:207): the uniform draws of the pass, kept for the backward
        self.d_keep = {i: float(getattr(Dc[i - 1], 'dropout', 0) or 0) for i in range(1, 9)}
        self.d_u = {i: torch.zeros(B * self.d_pre[i].h * self.d_pre[i].w * Dc[i - 1].out_size, dtype=torch.float32, device=device)
                    for i in range(1, 9) if self.d_keep[i] > 0}
        self.e_stats = {k: torch.zeros(2 * E[k - 1].out_size, dtype=torch.float32, device=device) for k in range(1, 9)}
        # noise channel windows and the f32 staging of their uniform draws
        self.noise = {}
        if self.noise_input:
            self.noise['noise_input'] = self.xn.window(x_in.c - 1, 1)
        if self.noise_latent:
            self.noise['noise_latent'] = self.lat.window(E[7].out_size, E[7].out_size)
        if self.noise_end:
            self.noise['noise_end'] = self.cat[8].window(Dc[7].in_size - 1, 1)
        self.noise_u = {k: torch.zeros(B * a.h * a.w * a.c, dtype=torch.float32, device=device) for k, a in self.noise.items()}
        # variables
        nb = 0
        self.e_bn_name = {}
        for k in range(1, 9):
            store.declare(enet.var_name(E[k - 1], 'weights'), E[k - 1].filter_shape)
            store.declare(enet.var_name(E[k - 1], 'bias'), (E[k - 1].out_size,))
            if self.enc_bn[k - 1]:
                self.e_bn_name[k] = enet.bn_name(0, k - 1)
                store.declare(self.e_bn_name[k], (E[k - 1].out_size,))
        self.d_bn_name = {}
        for i in range(1, 9):
            store.declare(dnet.var_name(Dc[i - 1], 'weights'), Dc[i - 1].filter_shape)
            store.declare(dnet.var_name(Dc[i - 1], 'bias'), (Dc[i - 1].out_size,))
            self.d_bn_name[i] = dnet.bn_name(0, i - 1)
            store.declare(self.d_bn_name[i], (Dc[i - 1].out_size,))

    def init_variables(self, gen):
        for net in (self.enet, self.dnet):
            for l in net.layers:
                for which, shape in (('weights', l.filter_shape), ('bias', (l.out_size,))):
                    self.store[net.var_name(l, which)].copy_(torch.randn(shape, generator=gen) * 0.02)

    def repack(self):
        if getattr(self, '_pack_jobs', None) is None:
            jl = [self.e_conv[k].pack_job(self.store[self.enet.var_name(self.enet.layers[k - 1], 'weights')]) for k in range(1, 9)]
            jl += [self.d_conv[i].pack_job(self.store[self.dnet.var_name(self.dnet.layers[i - 1], 'weights')]) for i in range(1, 9)]
            self._pack_jobs = K.make_pack_jobs(jl)
        K.pack_all(self._pack_jobs)

    def _d_in(self, i):
        """Input tensor of decoder layer i: [e8 (| noise)] for i = 1, the skip concat otherwise."""
        if i > 1:
            return self.cat[i]
        return self.lat if self.noise_latent else self.e_h[8]

    def draw_noise(self):
        """tf.random_uniform(minval=-1, maxval=1) into every noise window (one draw per generator pass, as in TF)."""
        for key, a in self.noise.items():
            u = self.noise_u[key]
            self.sess.random_uniform(u, u.numel(), key)
            _lib.call('tdg_affine_cast_rows', self.dtype, K.ptr(u), self.B * a.h * a.w, a.c, a.cs, 2.0, -0.5, a.ptr(0), K.stream())

    # ---- forward: G(x) into g_out ---------------------------------------------------------------------------------
    def forward(self, backward_follows=True):
        """backward_follows=False (the critic step's and the loss fetch's generator pass): batch-norm layers write only their
        activation, not the normalised pre-activation the backward pass would read."""
        B, st = self.B, self.store
        E, Dc = self.enet.layers, self.dnet.layers
        self._keep_pre = backward_follows
        self.draw_noise()
        for k in range(1, 9):
            spec, conv = E[k - 1], self.e_conv[k]
            src = self.x_in if k == 1 else self.e_h[k - 1]
            bias = st[self.enet.var_name(spec, 'bias')]
            if self.enc_bn[k - 1]:
                epi = K.colsum_epilogue(self.ws, self.e_pre[k].rows, spec.out_size, K.COL_BN, bias=bias)
                conv.fwd(src.ptr(), self.e_pre[k].ptr(), B, epi)
                self._bn_fwd(epi, self.e_pre[k], spec, st[self.e_bn_name[k]], self.e_h[k], self.e_stats[k], bias)
            else:
                conv.fwd(src.ptr(), self.e_h[k].ptr(), B, K.epilogue(bias=bias, act=spec.act.code, leak=spec.act.leak))
        for i in range(1, 9):
            spec, conv = Dc[i - 1], self.d_conv[i]
            src = self._d_in(i)
            bias = st[self.dnet.var_name(spec, 'bias')]
            epi = K.colsum_epilogue(self.ws, self.d_pre[i].rows, spec.out_size, K.COL_BN, bias=bias)
            conv.bwd_data(src.ptr(), self.d_pre[i].ptr(), B, epi)
            self._bn_fwd(epi, self.d_pre[i], spec, st[self.d_bn_name[i]], self.d_h[i], self.d_stats[i], bias)
            if self.d_keep[i] > 0:
                self.sess.random_uniform(self.d_u[i], self.d_u[i].numel(), 'dropout')
                self._dropout(self.d_h[i], i)

    def _bn_fwd(self, epi, pre, spec, beta, h, stats, bias):
        """Batch norm + activation of a layer whose GEMM has just stored `pre`: statistics from the epilogue's column
        partials when the launch provided them, else by the separate pass."""
        if K.nblk(epi):
            K.bn_fwd_from_partials(epi, pre, spec.out_size, beta, spec.act.code, pre if self._keep_pre else None, h, stats, bias,
                                   leak=spec.act.leak)
        else:
            K.bn_fwd(self.ws, pre, spec.out_size, beta, spec.act.code, pre, h, stats, leak=spec.act.leak)

    def _dropout(self, act, i):
        rows = self.B * act.h * act.w
        _lib.call('tdg_dropout', self.dtype, act.ptr(), rows, self.dnet.layers[i - 1].out_size, act.cs, K.ptr(self.d_u[i]),
                  self.d_keep[i], K.stream())

    # ---- backward from dL/dG(x) in g_grad -------------------------------------------------------------------------
    def backward(self):
        B, st, g = self.B, self.store, self.store.grad
        E, Dc = self.enet.layers, self.dnet.layers
        for i in range(8, 0, -1):
            spec, conv = Dc[i - 1], self.d_conv[i]
            if self.d_keep[i] > 0:
                self._dropout(self.d_g[i], i)                                     # d(dropout)/dh = the same mask / keep
            K.bn_bwd(self.ws, self.d_g[i], self.d_pre[i], spec.out_size, st[self.d_bn_name[i]], self.d_stats[i], spec.act.code,
                     self.d_delta[i], g(self.d_bn_name[i]), leak=spec.act.leak, dbias=g(self.dnet.var_name(spec, 'bias')))
            src = self._d_in(i)
            conv.bwd_filter(self.d_delta[i].ptr(), src.ptr(), g(self.dnet.var_name(spec, 'weights')), B, 0.0)
            if i > 1:
                conv.fwd(self.d_delta[i].ptr(), self.gcat[i].ptr(), B)            # first writer of gcat[i] (both windows)
            else:
                conv.fwd(self.d_delta[i].ptr(), self.e_g[8].ptr(), B, self._into_encoder(8, accumulate=False))
        for k in range(8, 0, -1):
            spec, conv = E[k - 1], self.e_conv[k]
            if self.enc_bn[k - 1]:
                K.bn_bwd(self.ws, self.e_g[k], self.e_pre[k], spec.out_size, st[self.e_bn_name[k]], self.e_stats[k], spec.act.code,
                         self.e_delta[k], g(self.e_bn_name[k]), leak=spec.act.leak, dbias=g(self.enet.var_name(spec, 'bias')))
            delta = self.e_delta[k]
            if not self.enc_bn[k - 1]:
                K.bias_grad(self.ws, delta, spec.out_size, g(self.enet.var_name(spec, 'bias')))
            src = self.x_in if k == 1 else self.e_h[k - 1]
            conv.bwd_filter(src.ptr(), delta.ptr(), g(self.enet.var_name(spec, 'weights')), B, 0.0)
            if k > 1:
                conv.bwd_data(delta.ptr(), self.e_g[k - 1].ptr(), B, self._into_encoder(k - 1, accumulate=True))

    def _into_encoder(self, k, accumulate):
        """Epilogue of the GEMM that delivers a gradient to encoder layer k's output: add to the skip gradient
        already there, and -- without batch norm -- apply lrelu'(e_k) so the result is delta_k directly."""
        spec = self.enet.layers[k - 1]
        if self.enc_bn[k - 1]:
            return K.epilogue(accumulate=accumulate)
        return K.epilogue(mask_mode=K.MASK_LRELU, leak=spec.act.leak, mask_src=self.e_h[k].ptr(), accumulate=accumulate)
