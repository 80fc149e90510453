"""Convolutional autoencoder on MI355X -- the builder surface of the reference's `models/cnn.py`
(cnn :20-57, loss :75-79, latent :82-93, encoder :96-113, decoder :116-134) on the HIP kernels.

Kept: `cnn(x, args)` returns `train_func(sess, args) -> {'loss'}` (util.py:22-28 default_training: one batch per
call); encoder/latent/decoder are written against dense/conv2d/deconv2d + arg_scope as in the reference; variable
names (`encoder/vars/c1/weights`, `latent/vars/d1/weights`, `decoder/vars/dc4/bias`, ...); inputs rescaled to [-1,1]
(:31); loss = mean |x - d| with TF's AbsGrad (sign(0) = 0).  Like the reference it is hard-wired to 64x64 inputs
through `32*4*4` (:92).

MI355X-native: the same implicit-GEMM / LDS-DMA kernel set as the GAN path; the latent dense layer is a 1x1 GEMM on
the flattened encoder output; the L1 mean and its seed are one pass (`tdg_l1_loss`).
"""
import torch

from .. import _lib
from .. import kernels as K
from .. import engine
from ..ops.layers import dense, conv2d, deconv2d, flatten, reshape, arg_scope, variable_scope, placeholder, reset_graph
from ..ops.activations import lrelu, relu, tanh
from ..util import tower_scope_range, average_gradients, init_optimizer, collection_to_dict


def encoder(x, reuse=False):
    """models/cnn.py:96-113."""
    with arg_scope([conv2d], reuse=reuse, activation=lrelu):
        x = conv2d(x, x.shape[-1], 64, 5, 2, name='c1')
        x = conv2d(x, 64, 128, 5, 2, name='c2')
        x = conv2d(x, 128, 256, 5, 2, name='c3')
        x = conv2d(x, 256, 256, 5, 2, name='c4')
        x = conv2d(x, 256, 96, 1, name='c5')
        x = conv2d(x, 96, 32, 1, name='c6')
    return x


def latent(x, latent_size, reuse=False):
    """models/cnn.py:82-93."""
    with arg_scope([dense], reuse=reuse):
        x = flatten(x)
        x = dense(x, 32 * 4 * 4, latent_size, name='d1')
    return x


def decoder(x, latent_size, out_channels=3, reuse=False):
    """models/cnn.py:116-134."""
    with arg_scope([dense, conv2d, deconv2d], reuse=reuse, activation=relu):
        x = dense(x, latent_size, 32 * 4 * 4, name='d1')
        x = reshape(x, [-1, 4, 4, 32])
        x = conv2d(x, 32, 96, 1, name='c1')
        x = conv2d(x, 96, 256, 1, name='c2')
        x = deconv2d(x, 256, 256, 5, 2, name='dc1')
        x = deconv2d(x, 256, 128, 5, 2, name='dc2')
        x = deconv2d(x, 128, 64, 5, 2, name='dc3')
        x = deconv2d(x, 64, out_channels, 5, 2, name='dc4', activation=tanh)
    return x


class CnnReplica(engine.GraphRunner):
    def __init__(self, x_source, args, sess):
        self.args, self.sess, self.x_source = args, sess, x_source
        B, L = args.batch_size, args.latent_size
        h, w, c = args.image_shape
        if (h, w) != (64, 64):
            raise ValueError('models/cnn.py is hard-wired to 64x64 inputs (32*4*4 features, :92); use --resize 64 64')
        dev, dt = sess.device, sess.dtype
        self.B, self.L = B, L

        reset_graph()
        x_sym = placeholder((None, h, w, c))
        for _x, scope, gpu_id in tower_scope_range(x_sym, args.n_gpus, B, sess):
            with variable_scope('encoder') as enet:
                e = encoder(_x, reuse=False)
            with variable_scope('latent') as lnet:
                z = latent(e, L, reuse=False)
            with variable_scope('decoder') as dnet:
                d = decoder(placeholder((None, L)), L, c, reuse=False)
        self.enet, self.lnet, self.dnet = enet, lnet, dnet

        self.ws = K.Workspace(dev)
        self.store = engine.ParamStore(dev)                       # ONE optimizer over every variable (models/cnn.py:27,51)
        self.E = engine.SeqNet(enet, B, (h, w, c), dt, dev, self.store, ws=self.ws)
        self.Dn = engine.SeqNet(dnet, B, (1, 1, L), dt, dev, self.store, need_input_grad=True, ws=self.ws)
        self.E.declare_variables()
        (self.lat,) = lnet.layers
        self.wname, self.bname = lnet.var_name(self.lat, 'weights'), lnet.var_name(self.lat, 'bias')
        self.store.declare(self.wname, self.lat.filter_shape)
        self.store.declare(self.bname, (self.lat.out_size,))
        self.Dn.declare_variables()
        self.store.allocate()
        gen = torch.Generator().manual_seed(sess.seed)
        self.E.init_variables(gen)
        for name, shape in ((self.wname, self.lat.filter_shape), (self.bname, (self.lat.out_size,))):
            cpu = torch.empty(shape)
            engine.xavier_uniform_(cpu, shape, gen)
            self.store[name].copy_(cpu)
        self.Dn.init_variables(gen)
        self.opt = init_optimizer(args, self.store)

        # latent dense: flat [B,512] -> z [B,L], written straight into the decoder's input
        e_last = self.E.layers[-1]
        self.flat = K.Act(B, 1, 1, 512, dt, dev, 512, e_last.h.buf)
        self.dflat = K.Act(B, 1, 1, 512, dt, dev, 512, e_last.gout.buf)
        self.lat_conv = K.Conv(self.flat, self.Dn.x, 1, 1, 1, 0, 0)
        self.x_stage = torch.zeros(B, h, w, c, dtype=torch.float32, device=dev)
        self.scal = torch.zeros(4, dtype=torch.float32, device=dev)
        self.init_graphs(args, sess)
        self.refresh()

    # ---- variables -----------------------------------------------------------------------------------
    def stores(self):
        return [self.store]

    def optimizers(self):
        return {'optimizers/cnn': self.opt}

    def refresh(self):
        self.E.repack()
        self.Dn.repack()
        self.lat_conv.pack(self.store[self.wname].view(1, 1, 512, self.L), fwd=True, bwd=True)

    def load_variables(self, arrays):
        self.store.load(arrays)
        self.refresh()

    def variables(self):
        return self.store.state_dict()

    def gradients(self):
        return self.store.grads_dict()

    # ---- one training step (util.py:22-28 default_training) -------------------------------------------------
    def step(self, x01):
        self.x_stage.copy_(x01.reshape(self.x_stage.shape))      # fixed address: the bodies below may be graph-captured
        self._run('grads', self._grads)
        self.sess.assert_finite(self.store, 'cnn_step')
        self._scale = average_gradients(self.sess, self.store)   # RCCL, outside the graphs
        self._run('apply', self._apply)
        self.sess.global_step += 1

    def _apply(self):
        self.opt.step(self._scale)
        self.refresh()

    def _grads(self):
        B, L, dt = self.B, self.L, self.sess.dtype
        h, w, c = self.args.image_shape
        _lib.call('tdg_affine_cast_rows', dt, K.ptr(self.x_stage), B * h * w, c, self.E.x.cs, 2.0, -0.5, self.E.x.ptr(0),
                  K.stream())                                                                    # models/cnn.py:31
        self.E.forward(0, B)
        self.lat_conv.fwd(self.flat.ptr(), self.Dn.x.ptr(), B, K.epilogue(bias=self.store[self.bname]))
        d = self.Dn.forward(0, B)
        last = self.Dn.layers[-1]
        wsb = self.ws.ensure(4096)
        _lib.call('tdg_l1_loss', dt, K.ptr(self.x_stage), d.ptr(), B * h * w, c, d.cs, 2.0, -0.5, last.gout.ptr(),
                  K.ptr(self.scal), K.ptr(wsb), wsb.numel(), K.stream())                         # :75-79
        self.Dn.backward(0, B, want_params=True, want_dx=True)
        g = self.store.grad
        K.bias_grad(self.ws, self.Dn.dx, L, g(self.bname), rows=B)
        self.lat_conv.bwd_filter(self.flat.ptr(), self.Dn.dx.ptr(), g(self.wname).view(1, 1, 512, L), B, 0.0)
        self.lat_conv.bwd_data(self.Dn.dx.ptr(), self.dflat.ptr(), B)     # dL/d(c6 output); E.backward applies lrelu'
        self.E.backward(0, B, want_params=True)

    def losses(self):
        s = self.sess.report_scalars(self.scal, mean=getattr(self.args, 'mean_loss', False)).cpu().tolist()
        return collection_to_dict([('tower_%d/loss/loss:0' % (self.sess.world_size - 1), s[0])])

    def samples(self, n):
        """(inputs, outputs) as float32 NHWC in [-1, 1] (models/cnn.py:60-67)."""
        h, w, c = self.args.image_shape
        n = min(n, self.B)
        last = self.Dn.layers[-1].h
        x = self.E.x.buf[:self.B * h * w * self.E.x.cs].view(self.B, h, w, self.E.x.cs)[:n, :, :, :c]
        d = last.buf[:self.B * h * w * last.cs].view(self.B, h, w, last.cs)[:n, :, :, :c]
        return x.float().cpu().numpy(), d.float().cpu().numpy()

    def train_func(self, sess=None, args=None):
        self.step(self.x_source.next_batch())
        return self.losses()


def cnn(x, args, sess=None):
    """models/cnn.py:20-57."""
    from ..runtime import Session
    sess = sess or Session(dtype=getattr(args, 'dtype_code', K.BF16), seed=getattr(args, 'seed', 0) or 0)
    replica = CnnReplica(x, args, sess)

    def train_func(sess_=None, args_=None):
        return replica.train_func(sess_, args_)
    train_func.replica = replica
    return train_func
