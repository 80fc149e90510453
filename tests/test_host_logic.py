"""Host-side mirror of the reference interface: builders, arg_scope, variable names, CLI,
config-file action, TFRecord reader, checkpoints -- everything that runs without a GPU."""
import io
import os
from types import SimpleNamespace

import numpy as np
import pytest
import torch

from conftest import pkg, ROOT
from oracle import gan_ref as G


def _build(model, shape=(32, 32, 3), L=8, B=4):
    Lyr, gan = pkg('ops.layers'), pkg('models.gan')
    Lyr.reset_graph()
    args = SimpleNamespace(model=model, latent_size=L, image_shape=shape, batch_size=B)
    with Lyr.variable_scope('generator') as gnet:
        g = gan.generator(B, L, args)
    with Lyr.variable_scope('discriminator') as dnet:
        d_real = gan.discriminator(Lyr.placeholder((None, int(np.prod(shape)))), args)
        d_fake = gan.discriminator(g, args, reuse=True)
    return gnet, dnet, d_real, d_fake


@pytest.mark.parametrize('model', ['iwgan', 'wgan'])
def test_builders_create_the_reference_variables(model):
    """Names and shapes == the oracle's statement of models/gan.py + ops/layers.py."""
    gnet, dnet, d_real, _ = _build(model)
    eng = pkg('engine')
    store_g, store_d = eng.ParamStore('cpu'), eng.ParamStore('cpu')
    # declare without binding buffers: replicate SeqNet.declare_variables' naming
    for net, store, passes in ((gnet, store_g, 1), (dnet, store_d, 1 if model == 'iwgan' else 2)):
        for l in net.layers:
            store.declare(net.var_name(l, 'weights'), l.filter_shape)
            store.declare(net.var_name(l, 'bias'), (l.out_size,))
        for p in range(passes):
            for i, l in enumerate(net.layers):
                if l.use_bn:
                    store.declare(net.bn_name(p, i), (l.out_size,))
    got = {k: v[1] for k, v in list(store_g.index.items()) + list(store_d.index.items())}
    want = {k: tuple(v) for k, v in G.param_shapes(G.make_cfg(model, (32, 32, 3), 8, 4)).items()}
    assert got == want
    assert len(dnet.passes) == 2 and d_real.rows_per_image == 1


def test_literal_64_gives_four_rows_per_image():
    _, _, d_real, _ = _build('iwgan', (64, 64, 3))
    assert d_real.rows_per_image == 4


def test_reuse_rules_match_tf_variable_scopes():
    Lyr = pkg('ops.layers')
    Lyr.reset_graph()
    x = Lyr.placeholder((None, 16))
    with Lyr.variable_scope('net'):
        Lyr.dense(x, 16, 4, name='fc')
        with pytest.raises(ValueError):
            Lyr.dense(x, 16, 4, name='fc')                   # exists, reuse not set
        Lyr.dense(x, 16, 4, name='fc', reuse=True)
    Lyr.reset_graph()
    with Lyr.variable_scope('net'):
        with pytest.raises(ValueError):
            Lyr.dense(x, 16, 4, name='fc', reuse=True)       # does not exist yet
    with pytest.raises(RuntimeError):
        Lyr.dense(x, 16, 4, name='fc')                       # outside any variable scope


def test_arg_scope_defaults_and_overrides():
    Lyr, A = pkg('ops.layers'), pkg('ops.activations')
    Lyr.reset_graph()
    x = Lyr.placeholder((None, 8, 8, 4))
    with Lyr.variable_scope('n') as net:
        with Lyr.arg_scope([Lyr.conv2d], use_batch_norm=True, activation=A.lrelu):
            Lyr.conv2d(x, 4, 8, 5, 2, name='a')
            Lyr.conv2d(x, 4, 8, 5, 2, name='b', use_batch_norm=False, activation=A.tanh)
        Lyr.conv2d(x, 4, 8, name='c')
    a, b, c = net.layers
    assert a.use_bn and a.act is A.lrelu and a.out_shape == (4, 4, 8)
    assert not b.use_bn and b.act is A.tanh
    assert not c.use_bn and c.act is None and (c.k, c.stride) == (3, 1)
    assert a.filter_shape == (5, 5, 4, 8)
    with Lyr.variable_scope('m') as net2:
        y = Lyr.deconv2d(x, 4, 6, 5, 2, name='d')
    assert y.shape == (None, 16, 16, 6) and net2.layers[0].filter_shape == (5, 5, 6, 4)     # [k,k,Cout,Cin]


def test_cli_defaults_alias_and_config_file(tmp_path):
    import importlib.util
    spec = importlib.util.spec_from_file_location('train_cli', os.path.join(ROOT, 'train.py'))
    train = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(train)
    p = train.build_parser()
    a = train.resolve_defaults(p.parse_args([]))
    assert train.resolve_defaults(p.parse_args(['--model', 'pix2pix'])).n_disc_train == 1
    assert train.resolve_defaults(p.parse_args(['--model', 'pix2pix', '--n_disc_train', '3'])).n_disc_train == 3
    # ... and through the REAL three-pass parse (the plugin re-declares --n_disc_train with its own default of 1,
    # hem/models/pix2pix.py:65-68; an earlier pass must not leave a None that hides it, nor may the gen-1 fallback of 5 win)
    assert train.parse_args(['--model', 'pix2pix', '--dataset', 'synthetic']).n_disc_train == 1
    assert train.parse_args(['--model', 'pix2pix', '--dataset', 'synthetic', '--n_disc_train', '3']).n_disc_train == 3
    assert train.parse_args(['--model', 'iwgan', '--dataset', 'synthetic']).n_disc_train == 5
    # train.py:62-182 defaults
    assert (a.n_gpus, a.batch_size, a.n_disc_train, a.optimizer, a.lr, a.momentum, a.decay) == (1, 256, 5, 'rmsprop', 0.001, 0.01, 0.9)
    assert (a.beta1, a.beta2, a.latent_size, a.dataset, a.epochs, a.buffer_size) == (0.9, 0.999, 200, 'floorplans', '3', 10000)
    assert p.parse_args(['--data', 'CIFAR']).dataset == 'cifar'            # README.md:50 spelling
    assert p.parse_args(['--dataset', 'mnist']).dataset == 'mnist'
    cfg = tmp_path / 'iwgan.config'
    cfg.write_text('model\t\tiwgan\nepochs\t\t20\nbatch_size \t256\nn_gpus \t\t2\noptimizer\tadam\nlr\t\t1e-4\nbeta1\t\t0.5\nbeta2\t\t0.9\n')
    a = p.parse_args(['--config', str(cfg), '--batch_size', '64'])          # examples/iwgan.config; CLI wins
    assert (a.model, a.optimizer, a.lr, a.beta1, a.beta2, a.n_gpus, a.batch_size) == ('iwgan', 'adam', 1e-4, 0.5, 0.9, 2, 64)


def test_tfrecord_round_trip_and_cifar_layout(tmp_path):
    tfr = pkg('tfrecord')
    rng = np.random.default_rng(0)
    imgs = rng.integers(0, 256, (5, 32, 32, 3), dtype=np.uint8)
    path = str(tmp_path / 'cifar.32.train.tfrecords')
    tfr.write_records(path, [tfr.make_example({'image': im.tobytes(), 'label': int(i)}) for i, im in enumerate(imgs)])
    recs = list(tfr.read_records(path, verify=True))
    assert len(recs) == 5
    ex = tfr.parse_example(recs[3])
    assert ex['label'] == [3] and len(ex['image']) == 3072
    assert np.array_equal(tfr.load_image_tfrecords(path, (32, 32, 3)), imgs)
    assert tfr.crc32c(b'123456789') == 0xE3069283                          # CRC-32C check value


def test_format_for_terminal_and_tower_scope():
    util = pkg('util')
    first = util.format_for_terminal({'g_loss': 1.0, 'd_loss': -2.0}, None)
    assert first == {'g_loss': '1.000000', 'd_loss': '-2.000000'}
    nxt = util.format_for_terminal({'g_loss': 0.5, 'd_loss': -2.0}, {'g_loss': 1.0, 'd_loss': -2.0})
    assert nxt == {'g_loss': '0.500000(-)', 'd_loss': '-2.000000(~)'}
    sess = SimpleNamespace(rank=1)
    x = np.arange(12).reshape(6, 2)
    (xs, scope, gpu_id), = list(util.tower_scope_range(x, 3, 2, sess))
    assert scope == 'tower_1' and gpu_id == 1 and np.array_equal(xs, x[2:4])   # ops/input.py:24


def test_param_store_checkpoint_round_trip(tmp_path):
    eng, ck = pkg('engine'), pkg('checkpoint')
    s = eng.ParamStore('cpu')
    s.declare('generator/vars/fc1/weights', (3, 5))
    s.declare('generator/vars/fc1/bias', (5,))
    s.allocate()
    assert s.index['generator/vars/fc1/bias'][0] % 4 == 0 and s.size % 4 == 0
    s['generator/vars/fc1/weights'].copy_(torch.arange(15.).view(3, 5))
    opt = eng.Optimizer(s)
    opt.set_step_count(7)
    rep = SimpleNamespace(stores=lambda: [s], optimizers=lambda: {'optimizers/generator': opt})
    draws = [41]
    sess = SimpleNamespace(global_step=12, global_epoch=2, rng_state=lambda: draws[0], set_rng_state=lambda n: draws.__setitem__(0, n))
    path = str(tmp_path / 'checkpoint-2.npz')
    ck.save(path, rep, sess)
    assert os.path.exists(path) and not os.path.exists(path + '.tmp')       # written aside, renamed into place
    s.params.zero_()
    opt.t, sess.global_step, sess.global_epoch, draws[0] = 0, 0, 0, 0
    ck.restore(path, rep, sess)
    assert torch.equal(s['generator/vars/fc1/weights'], torch.arange(15.).view(3, 5))
    assert (opt.t, sess.global_step, sess.global_epoch, draws[0]) == (7, 12, 2, 41)
    # a save that dies half-way leaves the previous archive as the newest readable one
    with open(path + '.tmp', 'wb') as f:
        f.write(b'PK truncated')
    ck.restore(path, rep, sess)


def test_montage_layout_and_png_roundtrip(tmp_path):
    """ops/summaries.py:97-124: image j*m + r lands at block row j, block column r; PNG bytes decode back."""
    import struct
    import zlib
    S = pkg('summaries')
    m, n, H, W, C = 4, 2, 3, 5, 3
    x = np.arange(m * n * H * W * C, dtype=np.float32).reshape(m * n, H, W, C)
    out = S.montage(x, m, n)
    assert out.shape == (n * H, m * W, C)
    for j in range(n):
        for r in range(m):
            assert np.array_equal(out[j * H:(j + 1) * H, r * W:(r + 1) * W], x[j * m + r])
    assert S.factorization(64) == (8, 8) and S.factorization(12) == (3, 4) and S.factorization(7) == (1, 7)
    assert S.montage(np.zeros((6, 2, 2))).shape == (3 * 2, 2 * 2, 1)
    img = np.linspace(0, 1, 4 * 6 * 3).reshape(4, 6, 3)
    b = S.png_bytes(img)
    assert b[:8] == b'\x89PNG\r\n\x1a\n'
    pos, idat, hdr = 8, b'', None
    while pos < len(b):
        ln, tag = struct.unpack('>I', b[pos:pos + 4])[0], b[pos + 4:pos + 8]
        data = b[pos + 8:pos + 8 + ln]
        assert struct.unpack('>I', b[pos + 8 + ln:pos + 12 + ln])[0] == zlib.crc32(tag + data) & 0xffffffff
        if tag == b'IHDR':
            hdr = struct.unpack('>IIBBBBB', data)
        if tag == b'IDAT':
            idat += data
        pos += 12 + ln
    assert hdr == (6, 4, 8, 2, 0, 0, 0)
    raw = zlib.decompress(idat)
    rows = [raw[r * (1 + 18):(r + 1) * (1 + 18)] for r in range(4)]
    assert all(r[0] == 0 for r in rows)
    dec = np.frombuffer(b''.join(r[1:] for r in rows), dtype=np.uint8).reshape(4, 6, 3)
    assert np.array_equal(dec, np.rint(img * 255).astype(np.uint8))
    files = S.write_epoch(str(tmp_path), 1, {'g_loss': 1.5, 'd_loss': -0.25}, np.zeros((64, 8, 8, 3)), np.ones((64, 8, 8, 3)))
    assert len(files) == 3 and open(files[0]).read().splitlines() == ['epoch,d_loss,g_loss', '1,-0.25,1.5']


def test_deconv2d_rejects_inconsistent_output_shape():
    Lm = pkg('ops.layers')
    Lm.reset_graph()
    with Lm.variable_scope('g'):
        x = Lm.placeholder((None, 5, 5, 8))
        with pytest.raises(ValueError):
            Lm.deconv2d(x, 8, 4, 5, 2, output_shape=(2, 4, 16, 16), padding='VALID', name='bad')   # 16 -> 6, not 5
        with pytest.raises(ValueError):
            Lm.deconv2d(x, 8, 4, 5, 2, output_shape=(2, 3, 14, 14), padding='VALID', name='bad_c')
        y = Lm.deconv2d(x, 8, 4, 5, 2, output_shape=(2, 4, 13, 14), padding='VALID', name='ok')
        assert y.shape[1:] == (13, 14, 4)


def test_lambda_activation_is_traced_to_a_fused_epilogue():
    """`activation=lambda x: hem.lrelu(x, leak=...)` (hem/models/paper_cgan.py:233) must fuse like the token itself."""
    Lm, act, lib = pkg('ops.layers'), pkg('ops.activations'), pkg('_lib')
    Lm.reset_graph()
    with Lm.variable_scope('g') as net:
        x = Lm.placeholder((None, 8, 8, 4))
        Lm.conv2d(x, 4, 8, 3, 1, activation=lambda t: act.lrelu(t, leak=0.1), name='a')
        Lm.conv2d(x, 4, 8, 3, 1, activation=lambda t: act.relu(t), name='b')
        with pytest.raises(NotImplementedError):
            Lm.conv2d(x, 4, 8, 3, 1, activation=lambda t: t, name='c')
    a, b = net.layers[0], net.layers[1]
    assert (a.act.code, a.act.leak) == (lib.ACT_LRELU, 0.1) and b.act.code == lib.ACT_RELU


def test_checkpoint_retention(tmp_path):
    """gen-2 --max_to_keep N keeps the N newest checkpoint-<n>.npz (numeric order, not lexical); 0 keeps all."""
    ck = pkg('checkpoint')
    for n in (0, 1, 2, 9, 10, 11):
        (tmp_path / ('checkpoint-%d.npz' % n)).write_bytes(b'x')
    (tmp_path / 'options.config').write_text('a 1\n')
    assert ck.prune(str(tmp_path), 0) == []
    gone = ck.prune(str(tmp_path), 2)
    assert sorted(os.path.basename(f) for f in gone) == ['checkpoint-0.npz', 'checkpoint-1.npz', 'checkpoint-2.npz', 'checkpoint-9.npz']
    assert sorted(os.listdir(tmp_path)) == ['checkpoint-10.npz', 'checkpoint-11.npz', 'options.config']
