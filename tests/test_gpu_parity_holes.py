"""GPU parity cases the first round left open ("close the cheap parity holes"):

* the reference's LITERAL 64x64 network (models/gan.py:280-286: `[-1, 64L]` reshape => 4 scores per image),
* the reference's own rmse vectors (hem/ops/test_losses.py:6-27) through the HIP loss kernel,
* --check_numerics (hem/util/training.py:52-53) naming the offending variable,
* train -> save -> restore -> continue == uninterrupted (train.py:273-292), bit for bit.
"""
import os
from types import SimpleNamespace

import numpy as np
import pytest
import torch

from conftest import pkg
from oracle import gan_ref as G
from test_gpu_gan_step import build, relerr, relerr_where_significant, make_args

pytestmark = pytest.mark.gpu


# ------------------------------------------------------------------------------------------------ literal 64x64
@pytest.mark.parametrize('model', ['iwgan', 'wgan'])
def test_literal_64x64_d_and_g_step_f32(model):
    """image_shape (64, 64, 3): D's last conv is [B, 8, 8, 4L], reshaped to [-1, 64L] => 4B rows, fc2 emits 4 scores per
    image (each from a 2-row strip), the means run over 4B values and the penalty's gradient flows through all four
    (SURVEY App. C-2).  D step + G step against the NumPy oracle at the north-star's 1e-3."""
    args, cfg, P, batches, zs, alphas, sess, rep = build(model, 0, B=4, L=8, shape=(64, 64, 3))
    assert rep.rows_per_image == 4
    assert rep.D.layers[-1].out.numel() == rep.nslots * 4 * 4            # nslots x B x 4 scores on the device path
    tr = G.GanTrainer({k: v.copy() for k, v in P.items()}, cfg, args)
    sess.inject = {'z': [zs[0]], 'alpha': [alphas[0]]}
    rep.d_step(rep.x_source.next_batch())
    x = tr.rescale(batches[0].astype(np.float64))
    loss, grads, aux = G.d_loss_and_grads(P, x, zs[0].astype(np.float64), alphas[0].astype(np.float64), cfg)
    assert aux['d_real'].size == 4 * 4                                    # the oracle reproduces [4B] too
    got = rep.gradients()
    for k, g in grads.items():
        if k.endswith('/bias') and cfg.d_bn and ('/c2/' in k or '/c3/' in k):
            continue
        assert relerr(got[k], g) < 1e-3, k
    s = rep.scal.cpu().numpy()
    assert abs(s[rep.S_DREAL] - aux['d_real'].mean()) < 1e-4
    assert abs(s[rep.S_DFAKE] - aux['d_fake'].mean()) < 1e-4
    if model == 'iwgan':
        assert abs(s[rep.S_GP] - aux['gp']) < 1e-3 * max(1.0, aux['gp'])
    tr.d_step(batches[0].astype(np.float64), zs[0].astype(np.float64), alphas[0].astype(np.float64))
    sess.inject = {'z': [zs[1]], 'alpha': [alphas[1]]}
    rep.g_step(rep.x_source.next_batch())
    _, ggrads, _ = G.g_loss_and_grads(tr.P, zs[1].astype(np.float64), cfg)
    got = rep.gradients()
    for k, g in ggrads.items():
        if k.endswith('/bias') and 'dc4' not in k:
            continue
        assert relerr(got[k], g) < 1e-3, k
    ref = tr.g_step(batches[1].astype(np.float64), zs[1].astype(np.float64), alphas[1].astype(np.float64))
    out = rep.losses()
    assert abs(out['g_loss'] - ref['g_loss']) < 1e-3 * max(1, abs(ref['g_loss']))
    assert abs(out['d_loss'] - ref['d_loss']) < 1e-3 * max(1, abs(ref['d_loss']))


# ------------------------------------------------------------------------------------------------ rmse known answers
@pytest.mark.parametrize('dtype', [0, 1])
def test_reference_rmse_vectors_through_the_hip_loss_kernel(dtype):
    """hem/ops/test_losses.py:6-27: rmse(1, 1) = 0, rmse(1, 0) = 1, rmse(-1, 1) = 2, rmse(1, -1) = 2 on (1, 64, 64, 3)
    tensors.  `tdg_p2p_l1` rescales its [-1, 1] operands to [0, 1] first (hem/models/pix2pix.py:299 via hem.rescale), so
    the vectors are fed pre-rescaled: v -> 2 v - 1."""
    K, _lib = pkg('kernels'), pkg('_lib')
    dev = torch.device('cuda:0')
    n = 64 * 64 * 3
    scal = torch.zeros(4, dtype=torch.float32, device=dev)
    ws = torch.empty(1 << 20, dtype=torch.uint8, device=dev)
    for x, x_hat, want in [(1.0, 1.0, 0.0), (1.0, 0.0, 1.0), (-1.0, 1.0, 2.0), (1.0, -1.0, 2.0)]:
        a = K.Act(1, 64, 64 * 3, 1, dtype, dev).set(np.full((1, 64, 64 * 3, 1), 2 * x - 1, np.float32))
        b = K.Act(1, 64, 64 * 3, 1, dtype, dev).set(np.full((1, 64, 64 * 3, 1), 2 * x_hat - 1, np.float32))
        _lib.call('tdg_p2p_l1', dtype, a.ptr(0), b.ptr(0), n, a.cs, 10.0, None, 0, K.ptr(scal), K.ptr(ws), ws.numel(), K.stream())
        l1, rmse = scal[:2].cpu().tolist()
        assert abs(rmse - want) < 1e-6, (x, x_hat, rmse)
        assert abs(l1 - abs(x - x_hat)) < 1e-6


# ------------------------------------------------------------------------------------------------ --check_numerics
def test_check_numerics_names_the_variable():
    """A NaN in the critic's c3 filter poisons the D gradients; with --check_numerics the step raises a
    FloatingPointError that names a variable (tf.check_numerics(g, v.name), hem/util/training.py:52-53), and the
    optimizer is NOT applied.  Without the flag the same step runs through silently, as in the reference."""
    gan, rt = pkg('models.gan'), pkg('runtime')
    dev = torch.device('cuda:0')
    for check in (True, False):
        args = make_args('iwgan', 4, 8, (32, 32, 3))
        sess = rt.Session(device=dev, dtype=0, seed=0, rank=0, world_size=1, check_numerics=check)
        rng = np.random.default_rng(0)

        class Src:
            def next_batch(self):
                return torch.tensor(rng.uniform(0, 1, (4, 32, 32, 3)).astype(np.float32), device=dev)
        rep = gan.GanReplica(Src(), args, sess)
        assert rep.use_graphs                                # the finite check runs BETWEEN the captured bodies: the graphs stay on
        rep.d_store['discriminator/vars/c3/weights'].view(-1)[5] = float('nan')
        rep.refresh()
        before = rep.d_store['discriminator/vars/c1/weights'].clone()
        if check:
            with pytest.raises(FloatingPointError) as e:
                rep.d_step(Src().next_batch())
            assert 'd_step' in str(e.value) and 'discriminator/vars/' in str(e.value)
            assert torch.equal(before, rep.d_store['discriminator/vars/c1/weights'])
            assert sess.global_step == 0
        else:
            rep.d_step(Src().next_batch())
            assert sess.global_step == 1
    # a NaN that appears AFTER the bodies were captured is caught in a replayed step as well
    args = make_args('iwgan', 4, 8, (32, 32, 3))
    sess = rt.Session(device=dev, dtype=0, seed=0, rank=0, world_size=1, check_numerics=True)
    rep = gan.GanReplica(Src(), args, sess)
    for _ in range(3):
        rep.d_step(Src().next_batch())                       # eager, capture, replay
    rep.d_store['discriminator/vars/c2/weights'].view(-1)[7] = float('inf')
    rep.refresh()
    with pytest.raises(FloatingPointError):
        rep.d_step(Src().next_batch())
    assert sess.global_step == 3


# ------------------------------------------------------------------------------------------------ resume
@pytest.mark.parametrize('optimizer', ['adam', 'rmsprop'])
def test_resume_is_bit_identical_to_uninterrupted(tmp_path, optimizer):
    """2 iterations -> save -> fresh replica -> restore -> 2 more == 4 uninterrupted iterations, bit for bit: variables,
    optimizer slots, Adam's step count, the global step and the Philox draw counter (z / alpha streams continue instead
    of replaying).  hipGraphs on, so the device-resident counters are what is saved and restored."""
    gan, rt, data, ckpt = pkg('models.gan'), pkg('runtime'), pkg('data'), pkg('checkpoint')
    dev = torch.device('cuda:0')
    B, L, shape = 8, 16, (32, 32, 3)

    def fresh():
        args = SimpleNamespace(model='iwgan', batch_size=B, latent_size=L, image_shape=shape, n_gpus=1, optimizer=optimizer,
                               lr=1e-4, beta1=0.5, beta2=0.9, decay=0.9, momentum=0.01, centered=False, n_disc_train=5,
                               display_d_loss=True, use_graphs=True)
        sess = rt.Session(device=dev, dtype=0, seed=7, rank=0, world_size=1)
        src = data.SyntheticSource(24 * B, shape, B, dev, seed=5)
        return args, sess, src, gan.GanReplica(src, args, sess)

    args, sess, src, rep = fresh()
    for _ in range(4):
        full_out = rep.train_func()
    steps = (lambda r: (r.d_opt.t, r.g_opt.t)) if optimizer == 'adam' else (lambda r: ())   # only Adam's update reads t
    full = (rep.variables(), {k: {s: t.cpu().numpy() for s, t in o.state_tensors().items()} for k, o in rep.optimizers().items()},
            steps(rep), sess.global_step, sess.rng_state())

    args, sess, src, rep = fresh()
    for _ in range(2):
        rep.train_func()
    path = str(tmp_path / 'checkpoint-2.npz')
    ckpt.save(path, rep, sess)
    assert os.path.exists(path) and not os.path.exists(path + '.tmp')
    pos = src.i
    args, sess, src, rep = fresh()                          # a new process would start exactly like this
    ckpt.restore(path, rep, sess)
    src.i = pos                                             # the input pipeline's position is the loader's to restore
    assert sess.global_step == 12 and sess.rng_state() > 0
    for _ in range(2):
        out = rep.train_func()
    assert out == full_out
    assert (steps(rep), sess.global_step, sess.rng_state()) == full[2:]
    v = rep.variables()
    for k in full[0]:
        assert np.array_equal(v[k], full[0][k]), k
    for name, o in rep.optimizers().items():
        for slot, t in o.state_tensors().items():
            assert np.array_equal(t.cpu().numpy(), full[1][name][slot]), (name, slot)


def test_wgan_clip_opt_in_clamps_before_the_critic_step():
    """--wgan_clip 0.01 (SURVEY App. C-3 opt-in): the critic's variables are clamped to [-c, c] before its gradients are
    taken (oracle: the same D step from clipped variables); the default (0) leaves them alone, like the reference, whose
    clip ops never execute."""
    from oracle import gan_ref as G
    c = 0.01
    for clip in (c, 0.0):
        args, cfg, P, batches, zs, alphas, sess, rep = build('wgan', 0, optimizer='rmsprop')
        args.wgan_clip = clip
        sess.inject = {'z': [zs[0]]}
        rep.d_step(rep.x_source.next_batch())
        Pc = {k: (np.clip(v, -c, c) if (clip and k.startswith('discriminator/')) else v) for k, v in P.items()}
        tr = G.GanTrainer({k: v.copy() for k, v in Pc.items()}, cfg, args)
        x = tr.rescale(batches[0].astype(np.float64))
        _, grads, _ = G.d_loss_and_grads(Pc, x, zs[0].astype(np.float64), None, cfg)
        got = rep.gradients()
        for k, g in grads.items():
            if k.endswith('/bias') and ('/c2/' in k or '/c3/' in k):
                continue
            assert relerr(got[k], g) < 1e-3, (clip, k)
        w = rep.d_store['discriminator/vars/c1/weights']
        if clip:
            assert float(w.abs().max()) <= c + 1e-3 + 1e-7         # clamped, then moved by one rmsprop step (lr 1e-3)
        else:
            assert float(w.abs().max()) > 10 * c


def test_gp_per_sample_opt_in():
    """--gp_per_sample (SURVEY App. C-4 opt-in): one gradient norm per image, penalty = mean_i (|v_i| - 1)^2, instead of the
    reference's single norm over the whole batch tensor (models/gan.py:229).  D step against the oracle's per-sample form;
    the default stays the whole-batch form (covered by test_d_and_g_step_f32)."""
    from oracle import gan_ref as G
    args, cfg, P, batches, zs, alphas, sess, rep = build('iwgan', 0, B=4, L=8)
    assert rep.gp_per_sample is False
    args.gp_per_sample = True
    cfg.gp_per_sample = True
    gan, rt = pkg('models.gan'), pkg('runtime')
    sess = rt.Session(device=sess.device, dtype=0, seed=0, rank=0, world_size=1)
    rep = gan.GanReplica(rep.x_source, args, sess)
    rep.load_variables({k: v.astype(np.float32) for k, v in P.items()})
    sess.inject = {'z': [zs[0]], 'alpha': [alphas[0]]}
    rep.x_source.i = 0
    rep.d_step(rep.x_source.next_batch())
    tr = G.GanTrainer({k: v.copy() for k, v in P.items()}, cfg, args)
    x = tr.rescale(batches[0].astype(np.float64))
    loss, grads, aux = G.d_loss_and_grads(P, x, zs[0].astype(np.float64), alphas[0].astype(np.float64), cfg)
    cfg2 = G.make_cfg('iwgan', (32, 32, 3), 8, 4)
    _, grads_whole, aux_whole = G.d_loss_and_grads(P, x, zs[0].astype(np.float64), alphas[0].astype(np.float64), cfg2)
    assert abs(aux['gp'] - aux_whole['gp']) > 1e-3 * abs(aux_whole['gp'])         # the two forms really differ
    got = rep.gradients()
    for k, g in grads.items():
        assert relerr(got[k], g) < 1e-3, k
    s = rep.scal.cpu().numpy()
    assert abs(s[rep.S_GP] - aux['gp']) < 1e-3 * max(1.0, aux['gp'])


def test_epoch_summaries_from_a_live_replica(tmp_path):
    """Summaries-lite on the GPU path (models/gan.py:93-107): `replica.samples(64)` returns the first 64 images of the
    staged real batch and of a fresh generator pass in [-1, 1]; `write_epoch` lays them out as the 8 x 8 `inputs` / `fake`
    montages (ops/summaries.py:97-124) and the PNG files decode back to exactly those pixels."""
    gan, rt, data, S, png = pkg('models.gan'), pkg('runtime'), pkg('data'), pkg('summaries'), pkg('png')
    dev = torch.device('cuda:0')
    B, L, shape = 64, 16, (32, 32, 3)
    args = SimpleNamespace(model='iwgan', batch_size=B, latent_size=L, image_shape=shape, n_gpus=1, optimizer='adam', lr=1e-4,
                           beta1=0.5, beta2=0.9, decay=0.9, momentum=0.01, centered=False, n_disc_train=1, display_d_loss=True)
    sess = rt.Session(device=dev, dtype=1, seed=2, rank=0, world_size=1)
    src = data.SyntheticSource(4 * B, shape, B, dev, seed=9)
    rep = gan.GanReplica(src, args, sess)
    status = rep.train_func()
    real, fake = rep.samples(64)
    assert real.shape == fake.shape == (64, 32, 32, 3)
    assert real.min() >= -1.0 and real.max() <= 1.0 and np.abs(fake).max() <= 1.0 and np.abs(fake).max() > 0
    staged = rep.x_stage.cpu().numpy()                            # the batch the last step consumed, in [0, 1]
    assert np.abs(real - (2 * staged - 1)).max() < 1e-2           # bf16 storage of 2 (x - 0.5)
    files = S.write_epoch(str(tmp_path), 1, status, real, fake, 64)
    assert [os.path.basename(f) for f in files] == ['losses.csv', 'montage-inputs-0001.png', 'montage-fake-0001.png']
    with open(files[0]) as f:
        header, row = f.read().strip().split('\n')
    assert header == 'epoch,d_loss,g_loss' and row.startswith('1,')
    for path, t in ((files[1], real), (files[2], fake)):
        img = png.decode(open(path, 'rb').read())
        assert img.shape == (8 * 32, 8 * 32, 3)
        want = np.clip(np.rint((t + 1.0) / 2.0 * 255.0), 0, 255).astype(np.uint8)
        for j in (0, 3, 7):
            for r in (0, 5):
                assert np.array_equal(img[j * 32:(j + 1) * 32, r * 32:(r + 1) * 32], want[j * 8 + r])


def test_training_failure_leaves_with_the_gen2_status(tmp_path):
    """hem/util/training.py:173-175: an exception inside the training loop is reported and the process leaves with -1 (255), the
    status repeat.sh restarts on.  Here: --check_numerics under an absurd learning rate (the second critic step sees Inf weights)."""
    import os
    import subprocess
    import sys
    from conftest import ROOT
    env = {k: v for k, v in os.environ.items() if k not in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK')}
    p = subprocess.run([sys.executable, os.path.join(ROOT, 'train.py'), '--model', 'iwgan', '--batch_size', '8', '--latent_size', '16',
                        '--optimizer', 'sgd', '--lr', '1e30', '--dataset', 'synthetic', '--epoch_size', '4', '--epochs', '1',
                        '--check_numerics', '--dir', str(tmp_path / 'ws')], env=env, timeout=600, capture_output=True, text=True)
    assert p.returncode == 255, (p.returncode, p.stdout[-800:], p.stderr[-1500:])
    assert 'Caught unexpected exception during training: FloatingPointError' in p.stdout
    assert 'discriminator/vars/' in p.stdout or 'generator/vars/' in p.stdout


def test_profile_flag_writes_a_kernel_table(tmp_path):
    """--profile (train.py:75,264-265: parsed, never used by the reference): here one iteration's conv GEMM kernels."""
    import os
    import subprocess
    import sys
    from conftest import ROOT
    env = {k: v for k, v in os.environ.items() if k not in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK')}
    d = str(tmp_path / 'ws')
    p = subprocess.run([sys.executable, os.path.join(ROOT, 'train.py'), '--model', 'iwgan', '--batch_size', '8', '--latent_size', '16',
                        '--optimizer', 'adam', '--lr', '1e-4', '--dataset', 'synthetic', '--epoch_size', '5', '--epochs', '1', '--profile',
                        '--dir', d], env=env, timeout=600, capture_output=True, text=True)
    assert p.returncode == 0, (p.stdout[-800:], p.stderr[-1500:])
    text = open(os.path.join(d, 'profile.txt')).read()
    assert 'igemm' in text and 'TFLOP/s' in text and 'total' in text
