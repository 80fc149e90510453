#!/usr/bin/env python3
"""Headline benchmark: images/sec (whole node) of CIFAR-10-shaped IWGAN training, bs=512 per
GPU (BASELINE.json metric).

One "step" = one `train_func` call = n_disc_train (5) discriminator steps + 1 generator step
on 6 fresh synthetic batches (models/gan.py:169-173), optimizer steps included;
images/sec = steps/s x B x n_gpus (the reference's own progress unit, train.py:298).

  python bench.py --gpus 1 --steps K --warmup W
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

`python bench.py --gpus N` with N > 1 and no launcher environment starts its own N ranks (a child
`torch.distributed.run`, before anything touches the GPU) and exits with the child's status.

Prints ONE JSON line on rank 0 (contract in the task statement) with `roofline` (dominant conv GEMM
kernel: HIP events recorded by the library around every conv GEMM launch, on the launch stream, in
instrumented eager steps run DIRECTLY AFTER the timed region -- hipGraph replay hides launches from
events), `cpu_baseline` (the oracle's torch-autograd port on the host cores, bounded sample, rank 0 at
N=1 only), at N=1 `config.secondary` (pix2pix bs 64, VAE bs 512 and the f32 headline step: SURVEY 8d
configs 4/5 and BASELINE.md s.4's parity row) and at N>1 `collectives` (ranks seen on the RCCL group and the
measured all-reduce time of each gradient bucket).
"""
import argparse
import importlib
import json
import os
import sys
import time
from types import SimpleNamespace

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)


def maybe_self_launch():
    """`python bench.py --gpus N` (N > 1) outside a launcher: start N ranks as a CHILD process (never exec), before this
    process has imported torch or touched the GPU, and exit with the child's return code."""
    import subprocess
    n = 1
    for i, a in enumerate(sys.argv):
        if a == '--gpus' and i + 1 < len(sys.argv):
            n = int(sys.argv[i + 1])
        elif a.startswith('--gpus='):
            n = int(a.split('=', 1)[1])
    if n > 1 and 'RANK' not in os.environ:
        import socket
        with socket.socket() as sk:
            sk.bind(('127.0.0.1', 0))
            port = sk.getsockname()[1]
        env = dict(os.environ)
        env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
        cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(n),
               '--master-addr', '127.0.0.1', '--master-port', str(port), os.path.abspath(__file__)] + sys.argv[1:]
        sys.exit(subprocess.call(cmd, env=env))


if __name__ == '__main__':
    maybe_self_launch()

import numpy as np                      # noqa: E402
import torch                            # noqa: E402
import torch.distributed as dist        # noqa: E402

# IWGAN 32x32x3, L=200, reference-faithful variant (G step re-evaluates d_loss): SURVEY 8d / BASELINE.md s.2
GFLOP_PER_IMAGE_ITERATION = 30.08
GFLOP_PER_IMAGE = {'pix2pix': 93.1, 'vae': 1.64}       # per image of one train() call / step
PEAK_TFLOPS = {'bf16': 2500.0, 'f32': 157.3}           # MI355X_MICROARCH.md, dense
SUSTAINED_TFLOPS = {'bf16': 1457.0}     # see roofline.power_bound_reference


def cpu_baseline(args):
    """The oracle's autograd port of the identical iteration on the host cores (kind "port"): batch 512, one warm-up
    D step (allocator, thread pool) + one untimed warm-up iteration when the budget allows, then >= 3 timed iterations."""
    from oracle import gan_ref as G, torch_ref as TR
    try:
        cores = len(os.sched_getaffinity(0))
    except AttributeError:
        cores = os.cpu_count() or 1
    cores = max(1, min(cores, 16))       # the GPU box grants a 16-core share per GPU
    torch.set_num_threads(cores)
    B = args.cpu_batch
    cfg = G.make_cfg('iwgan', (32, 32, 3), args.latent_size, B)
    P = TR.to_torch(G.init_params(cfg, 0, np.float32))
    a = SimpleNamespace(optimizer='adam', lr=1e-4, beta1=0.5, beta2=0.9, n_disc_train=5)
    tr = TR.TorchGanTrainer(P, cfg, a)
    rng = np.random.default_rng(1234)

    def inputs():
        bs = [torch.tensor(rng.integers(0, 256, (B, 32, 32, 3)).astype(np.float32) / 255.0) for _ in range(6)]
        zs = [torch.randn(B, args.latent_size) for _ in range(6)]
        als = [torch.rand(B, 1) for _ in range(6)]
        return bs, zs, als
    t0 = time.time()
    tr.d_step(*[v[0] for v in inputs()])           # untimed warm-up step (allocator, thread pool)
    warm = 'one D step'
    if (time.time() - t0) * 6 < 20.0:              # a whole warm-up iteration if it costs < ~20 s
        tr.train_func(*inputs())
        warm = 'one D step + one iteration'
    n, t0 = 0, time.time()
    while n < args.cpu_iters:
        tr.train_func(*inputs())
        n += 1
    dt = time.time() - t0
    return {'value': n * B / dt, 'unit': 'images/sec', 'cores': cores, 'kind': 'port',
            'sample': '%d timed iterations (5 D + 1 G steps each) at batch %d in %.1f s after %s of warm-up, f32 '
                      'torch-autograd port of the oracle, %d threads (CPU restatement, not TensorFlow)' % (n, B, dt, warm, cores)}


def cpu_baseline_pix2pix(batch=8):
    """Config 4's CPU figure: the oracle's torch-autograd port of hem/models/pix2pix.py (oracle/pix2pix_ref.py), one
    train() call (D step + G step + loss fetch on 3 batches) at a stated batch, f32, on the host cores."""
    from oracle import pix2pix_ref as PR, torch_ref as TR
    try:
        cores = len(os.sched_getaffinity(0))
    except AttributeError:
        cores = os.cpu_count() or 1
    cores = max(1, min(cores, 16))
    torch.set_num_threads(cores)
    a = SimpleNamespace(optimizer='adam', lr=1e-4, beta1=0.5, beta2=0.999, decay=0.9, momentum=0.01, n_disc_train=1,
                        batch_norm_disc=False, batch_norm_gen=False, add_l1=False, noise=[], dropout=0)
    tr = PR.Trainer(TR.to_torch(PR.init_params(a, 0, np.float32)), a)
    g = torch.Generator().manual_seed(1234)
    pair = lambda: (torch.rand(batch, 256, 256, 3, generator=g), torch.rand(batch, 256, 256, 1, generator=g) * 0.98 + 0.01)
    tr.d_step(*pair())                              # untimed warm-up (allocator, thread pool)
    t0 = time.time()
    tr.train([pair() for _ in range(3)])
    dt = time.time() - t0
    return {'value': batch / dt, 'unit': 'images/sec', 'cores': cores, 'kind': 'port',
            'sample': 'one train() call (D step + G step + loss fetch, 3 batches) at batch %d in %.1f s after one warm-up D step, '
                      'f32 torch-autograd port of the oracle, %d threads (CPU restatement, not TensorFlow)' % (batch, dt, cores)}


def cpu_baseline_vae(batch=512, iters=6):
    """Config 5's CPU figure (its per-GPU share, batch 512): the oracle's torch-autograd statement of models/vae.py:25-151
    (oracle/vae_ref.py: torch_losses; only decoder_loss is differentiated, as the reference's compute_gradients does) with an
    Adam step per call, f32, on the host cores."""
    from oracle import vae_ref as VR
    try:
        cores = len(os.sched_getaffinity(0))
    except AttributeError:
        cores = os.cpu_count() or 1
    cores = max(1, min(cores, 16))
    torch.set_num_threads(cores)
    P = {k: torch.tensor(v, requires_grad=True) for k, v in VR.init_params(200, 0, np.float32).items()}
    opt = torch.optim.Adam(list(P.values()), lr=1e-4, betas=(0.5, 0.9), eps=1e-8)
    g = torch.Generator().manual_seed(1234)

    def step():
        x, eps = torch.rand(batch, 64, 64, 3, generator=g), torch.randn(batch, 200, generator=g)
        opt.zero_grad(set_to_none=True)
        d_loss, _ = VR.torch_losses(P, x, eps)
        d_loss.backward()
        opt.step()
    step()                                          # untimed warm-up (allocator, thread pool)
    t0 = time.time()
    for _ in range(iters):
        step()
    dt = time.time() - t0
    return {'value': iters * batch / dt, 'unit': 'images/sec', 'cores': cores, 'kind': 'port',
            'sample': '%d optimizer steps at batch %d in %.1f s after one warm-up step, f32 torch-autograd port of the oracle, '
                      '%d threads (CPU restatement, not TensorFlow)' % (iters, batch, dt, cores)}


def newest_pmc_file():
    """The committed PMC summary (profiles/rNN_*_pmc_fetch_write_per_kernel.json) of the newest build."""
    d = os.path.join(ROOT, 'profiles')
    try:
        names = sorted(f for f in os.listdir(d) if f.endswith('_pmc_fetch_write_per_kernel.json'))
    except OSError:
        return None
    return os.path.join(d, names[-1]) if names else None


def _kernel_key(name):
    """(base name, leading integer template arguments) of a kernel name in any of the spellings that meet here: the
    library's own ("igemm_fwd_patch_kernel<bf16,192,208>"), rocprofv3's mangled ("_Z22igemm_fwd_patch_kernelILi192ELi208ELi0EEv6IgArgs")
    or demangled ("void igemm_fwd_patch_kernel<192, 208, 0>(IgArgs)") form.  The dtype token is dropped."""
    import re
    name = name.strip()
    if name.startswith('_Z'):
        m = re.match(r'_Z\d+([A-Za-z_0-9]+?)I(.*)', name)
        if not m:
            return name, []
        return m.group(1), [int(v) for v in re.findall(r'Li(\d+)E', m.group(2))]
    name = re.sub(r'^void\s+', '', name)
    base = name.split('<')[0].split('(')[0].strip()
    args = name[name.index('<') + 1:name.rindex('>')] if '<' in name and '>' in name else ''
    return base, [int(t.strip()) for t in args.split(',') if t.strip().isdigit()]


def pmc_traffic(symbol):
    """HBM-side bytes per launch of the kernel `symbol` from the committed rocprofv3 PMC passes (FETCH_SIZE and
    WRITE_SIZE in separate runs of this same bench, KB per dispatch averaged over the kernel's launches), with the
    gfx950 correction of MI355X_MICROARCH.md s.HBM: FETCH_SIZE counts 128-B requests as 64 B for 16-B-per-lane
    streaming reads, so bytes = (2 * FETCH_SIZE + WRITE_SIZE) * 1024.  None when the file or the kernel is absent."""
    path = newest_pmc_file()
    try:
        with open(path) as f:
            pmc = json.load(f)
    except (OSError, TypeError):
        return None, None
    want = _kernel_key(symbol)

    def find(name):
        for k, v in pmc.get(name, {}).items():
            have = _kernel_key(k)
            if have[0] == want[0] and (have[1][:len(want[1])] == want[1] or (have[1] and want[1][-len(have[1]):] == have[1])):
                return v['avg']
        return None
    fetch, write = find('FETCH_SIZE'), find('WRITE_SIZE')
    if fetch is None or write is None:
        return None, None
    return (2.0 * fetch + write) * 1024.0, os.path.relpath(path, ROOT) + ' (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes of this bench; bytes per launch; build ' + str(pmc.get('build', '?')) + ')'


def kernel_table(timer):
    sym = {}
    for name, kms, fl in timer:
        e = sym.setdefault(name, [0, 0.0, 0.0])
        e[0] += 1
        e[1] += kms
        e[2] += fl
    return sym


def roofline_of(timer, dtype, timer_steps, ms_per_step, with_traffic=True):
    """The dominant conv GEMM kernel (largest total time) of the instrumented steps against the dense MFMA peak."""
    sym = kernel_table(timer)
    gemm = {k: v for k, v in sym.items() if v[2] > 0}
    kind, (n, tot_ms, fl) = max(gemm.items(), key=lambda kv: kv[1][1])
    achieved = fl / (tot_ms * 1e-3) / 1e12
    gemm_ms = sum(v[1] for v in gemm.values())
    gemm_fl = sum(v[2] for v in gemm.values())
    traffic, traffic_src = pmc_traffic(kind) if with_traffic else (None, None)
    return {'bound': 'mfma', 'achieved': achieved, 'peak': PEAK_TFLOPS[dtype], 'unit': 'TFLOP/s',
            'frac': achieved / PEAK_TFLOPS[dtype], 'traffic': traffic, 'traffic_source': traffic_src,
            # informational: what a sustained MFMA loop reaches on this part under its power budget -- the library's 8192^3
            # bf16 GEMM runs 0.93 MFMA-busy at 1.51 GHz effective = 1457 TF (profiles/r03_c_effective_clock.txt); `frac` above
            # stays against the nominal dense peak of MI355X_MICROARCH.md
            'power_bound_reference': {'tflops': SUSTAINED_TFLOPS.get(dtype), 'frac': (achieved / SUSTAINED_TFLOPS[dtype]) if dtype in SUSTAINED_TFLOPS else None,
                                      'source': 'hipBLASLt 8192^3 bf16, SQ cycle counters: profiles/r03_c_effective_clock.txt'},
            'kernel': kind, 'launches': n, 'avg_launch_ms': tot_ms / n, 'flop_per_launch': fl / n,
            'measured': 'HIP events recorded by the library around every conv GEMM kernel launch (on the launch '
                        'stream) in %d instrumented eager steps run directly after the timed region (hipGraph '
                        'replay hides launches from events); achieved = sum of algorithmic FLOPs / sum of '
                        'durations over all launches of this kernel' % timer_steps,
            'all_conv_gemms': {'ms_per_step': gemm_ms / timer_steps, 'tflops': gemm_fl / (gemm_ms * 1e-3) / 1e12,
                               'share_of_step': gemm_ms / timer_steps / ms_per_step},
            'per_kernel': {k: {'launches': v[0], 'avg_ms': round(v[1] / v[0], 4),
                               'tflops': round(v[2] / (v[1] * 1e-3) / 1e12, 1)}
                           for k, v in sorted(sym.items(), key=lambda kv: -kv[1][1])}}


def time_calls(fn, warmup, steps):
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        out = fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps, out


def instrumented(rep, fn, K, steps):
    """Eager steps with the library's per-launch events on; returns the launch records."""
    if hasattr(rep, 'use_graphs'):
        rep.use_graphs = False
    fn()
    K.timing_begin()
    for _ in range(steps):
        fn()
    torch.cuda.synchronize()
    return K.timing_end()


def secondary_leg(model, sess_dtype, K, rt, data, models, steps=5):
    """One bounded run of a secondary model on synthetic data (SURVEY 8d config 4 / 5 at one GPU's share)."""
    B = 64 if model == 'pix2pix' else 512
    shape = (256, 256, 3) if model == 'pix2pix' else (64, 64, 3)
    sess = rt.Session(dtype=sess_dtype, seed=0, rank=0, world_size=1)
    if model == 'pix2pix':       # examples/pix2pix.config: adam 1e-4 beta1 0.5, n_disc_train 1, --skip_layers
        margs = SimpleNamespace(model=model, batch_size=B, latent_size=200, image_shape=shape, n_gpus=1, optimizer='adam', lr=1e-4,
                                beta1=0.5, beta2=0.999, decay=0.9, momentum=0.01, centered=False, n_disc_train=1,
                                skip_layers=True, noise=[], dropout=0, batch_norm_disc=False, batch_norm_gen=False,
                                add_l1=False, seed=0)
        src = data.SyntheticPairSource(2, B, sess.device)
        work = '--model pix2pix --batch_size 64 --optimizer adam --lr 1e-4 --beta1 0.5 --n_disc_train 1 --skip_layers, 256x256 synthetic rgb/depth pairs; one call = train() = D step + G step + loss fetch on 3 batches'
    else:                        # train.py defaults: rmsprop lr 1e-3 decay 0.9 momentum 0.01
        margs = SimpleNamespace(model=model, batch_size=B, latent_size=200, image_shape=shape, n_gpus=1, optimizer='rmsprop',
                                lr=1e-3, beta1=0.9, beta2=0.999, decay=0.9, momentum=0.01, centered=False, n_disc_train=1, seed=0)
        src = data.SyntheticSource(2 * B, shape, B, sess.device)
        work = '--model vae --batch_size 512 (the per-GPU share of config 5) --optimizer rmsprop defaults, 64x64x3 synthetic; one call = one optimizer step'
    train = models.model_funcs()[model](src, margs, sess)
    fn = lambda: train(sess, margs)
    # two timed rounds, the lower one reported (both kept in `ms_per_call_rounds`): a bounded leg of 5 calls is at the mercy of a
    # single host / allocator stall (seen once in round 4: 14.9 ms where every other run of the same build gave 11.8 - 12.0)
    dt_a, status = time_calls(fn, 4, steps)
    dt_b, status = time_calls(fn, 0, steps)
    dt = min(dt_a, dt_b)
    rec = instrumented(train.replica, fn, K, 1)
    ips = B / dt
    tf = ips * GFLOP_PER_IMAGE[model] / 1e3
    r = roofline_of(rec, 'bf16', 1, dt * 1e3)
    out = {'workload': work, 'ms_per_call': dt * 1e3, 'ms_per_call_rounds': [dt_a * 1e3, dt_b * 1e3], 'images_per_sec': ips, 'dtype': 'bf16',
           'call_tflops': tf, 'call_frac_of_peak': tf / PEAK_TFLOPS['bf16'],
           'dominant_kernel': {k: r[k] for k in ('kernel', 'achieved', 'frac', 'launches', 'avg_launch_ms', 'flop_per_launch',
                                                 'traffic', 'traffic_source')},
           'all_conv_gemms': r['all_conv_gemms'], 'per_kernel': r['per_kernel'],
           'final_losses': {k: float(v) for k, v in (status or {}).items()}}
    del train, fn
    torch.cuda.empty_cache()
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=10)
    ap.add_argument('--warmup', type=int, default=3)
    ap.add_argument('--batch_size', type=int, default=512)
    ap.add_argument('--latent_size', type=int, default=200)
    ap.add_argument('--dtype', default='bf16', choices=['bf16', 'f32'])
    ap.add_argument('--model', default='iwgan')
    ap.add_argument('--cpu_batch', type=int, default=512)
    ap.add_argument('--cpu_iters', type=int, default=3)
    ap.add_argument('--no_cpu_baseline', action='store_true')
    ap.add_argument('--no_secondary', action='store_true', help='skip the pix2pix / vae / f32 legs of config.secondary')
    ap.add_argument('--secondary_legs', default='pix2pix,vae,f32', help='which legs of config.secondary to run')
    ap.add_argument('--no_headline_timer', action='store_true', help='profiling passes of the secondary legs: skip the instrumented headline steps')
    ap.add_argument('--no_kernel_timer', action='store_true')
    ap.add_argument('--timer_steps', type=int, default=2)
    ap.add_argument('--no_graphs', action='store_true')
    ap.add_argument('--dump_launches', default=None, help='diagnostics: write (kernel, GFLOP, count, avg ms, TFLOP/s) per distinct launch shape to this file')
    args = ap.parse_args()

    rt = importlib.import_module('3dgan_amd.runtime')
    K = importlib.import_module('3dgan_amd.kernels')
    gan = importlib.import_module('3dgan_amd.models.gan')
    data = importlib.import_module('3dgan_amd.data')
    models = importlib.import_module('3dgan_amd.models')

    world = rt.init_distributed()
    if world != args.gpus:
        raise SystemExit('--gpus %d but WORLD_SIZE=%d' % (args.gpus, world))
    ranks_seen = dist.get_world_size() if dist.is_initialized() else 1

    def headline_replica(dtype):
        sess = rt.Session(dtype=K.BF16 if dtype == 'bf16' else K.F32, seed=0)
        margs = SimpleNamespace(model=args.model, batch_size=args.batch_size, latent_size=args.latent_size,
                                image_shape=(32, 32, 3), n_gpus=args.gpus, optimizer='adam', lr=1e-4, beta1=0.5,
                                beta2=0.9, decay=0.9, momentum=0.01, centered=False, n_disc_train=5,
                                display_d_loss=True, use_graphs=not args.no_graphs)              # examples/iwgan.config
        src = data.SyntheticSource(12 * args.batch_size, margs.image_shape, args.batch_size, sess.device, 1234, sess.rank)
        rep = gan.GanReplica(src, margs, sess)
        rt.broadcast_store(rep.g_store)
        rt.broadcast_store(rep.d_store)
        rep.refresh()
        return sess, rep

    sess, rep = headline_replica(args.dtype)

    def sync():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    for _ in range(2):                            # setup, not warm-up: the first call runs eagerly (lazy workspaces,
        rep.train_func()                          # kernel attributes), the second captures the hipGraphs
    for _ in range(args.warmup):                  # W untimed warm-up steps (graph replays)
        rep.train_func()
    sync()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        status = rep.train_func()
    sync()
    dt = time.perf_counter() - t0
    # Per-kernel HIP events: graph replay hides the individual launches from events, so the same steps
    # are repeated eagerly, directly after the timed region, with every conv GEMM launch bracketed by events
    # on the launch stream (rank 0 only; the other ranks run the same steps so collectives stay matched).
    timer = None
    if not args.no_kernel_timer and not args.no_headline_timer and args.timer_steps > 0:
        rep.use_graphs = False
        rep.train_func()
        if sess.rank == 0:
            K.timing_begin()                      # HIP events around every conv GEMM kernel launch, inside the library
        for _ in range(args.timer_steps):
            rep.train_func()
        sync()
        if sess.rank == 0:
            timer = K.timing_end()
    coll = None
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=sess.device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
        coll = measure_collectives(rep, sess, sync)
        coll['ranks_seen'] = ranks_seen

    if sess.rank == 0:
        ms = dt / args.steps * 1e3
        value = args.steps * args.batch_size * args.gpus / dt
        out = {
            'metric': 'images/sec (whole node), CIFAR-10 %s bs=%d' % (args.model.upper(), args.batch_size), 'value': value,
            'unit': 'images/sec',
            'n_gpus': args.gpus, 'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': ms,
            'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None, 'dtype': args.dtype,
            'data': 'synthetic (uint8 U{0..255}/255 images, xavier-uniform random-init weights, on-device Philox z/alpha)',
            'config': {'workload': '--model %s --dataset cifar(32x32x3 synthetic) --batch_size %d/GPU --latent_size %d '
                                   '--optimizer adam --lr 1e-4 --beta1 0.5 --beta2 0.9 --n_disc_train 5; one step = 5 D + 1 G '
                                   'optimizer steps on 6 fresh batches' % (args.model, args.batch_size, args.latent_size),
                       'global_batch': args.batch_size * args.gpus, 'parallelism': 'dp%d' % args.gpus,
                       'ranks_seen': ranks_seen,
                       'consumed_images_per_sec': value * 6,
                       'step_tflops': (value / args.gpus * GFLOP_PER_IMAGE_ITERATION / 1e3) if args.model == 'iwgan' else None,
                       'final_losses': status},
        }
        if coll is not None:
            out['collectives'] = coll
        if timer and args.dump_launches:
            shapes = {}
            for name, kms, fl in timer:
                e = shapes.setdefault((name, round(fl / 1e9, 2)), [0, 0.0])
                e[0] += 1
                e[1] += kms
            with open(args.dump_launches, 'w') as f:
                for (name, gf), (n, tot) in sorted(shapes.items(), key=lambda kv: -kv[1][1]):
                    f.write('%-48s %9.2f GFLOP  x%-3d %7.4f ms  %7.1f TF  (%.3f ms per step)\n'
                            % (name, gf, n, tot / n, gf / (tot / n) if tot else 0.0, tot / args.timer_steps))
        if timer:
            out['roofline'] = roofline_of(timer, args.dtype, args.timer_steps, ms)
        if args.gpus == 1 and not args.no_secondary:
            del rep
            torch.cuda.empty_cache()
            sec = {}
            legs = args.secondary_legs.split(',')
            for m in ('pix2pix', 'vae'):
                if m in legs:
                    sec['%s_bs%d' % (m, 64 if m == 'pix2pix' else 512)] = secondary_leg(m, K.BF16, K, rt, data, models)
            if 'pix2pix' in legs and not args.no_cpu_baseline:
                sec['pix2pix_bs64']['cpu_baseline'] = cpu_baseline_pix2pix()
            if 'vae' in legs and not args.no_cpu_baseline:
                sec['vae_bs512']['cpu_baseline'] = cpu_baseline_vae()
            if args.dtype == 'bf16' and args.model == 'iwgan' and 'f32' in legs:
                # the parity dtype on the headline workload: exact-f32 MFMA path
                s32, r32 = headline_replica('f32')
                d32, st32 = time_calls(r32.train_func, 3, 3)
                rec = instrumented(r32, r32.train_func, K, 1)
                v32 = args.batch_size / d32
                r = roofline_of(rec, 'f32', 1, d32 * 1e3, with_traffic=False)
                sec['iwgan_bs%d_f32' % args.batch_size] = {
                    'workload': 'the headline step in f32 (v_mfma_f32_16x16x4_f32, exact f32: the parity path)',
                    'ms_per_step': d32 * 1e3, 'images_per_sec': v32, 'dtype': 'f32',
                    'step_tflops': v32 * GFLOP_PER_IMAGE_ITERATION / 1e3,
                    'step_frac_of_peak': v32 * GFLOP_PER_IMAGE_ITERATION / 1e3 / PEAK_TFLOPS['f32'],
                    'dominant_kernel': {k: r[k] for k in ('kernel', 'achieved', 'frac', 'launches', 'avg_launch_ms')},
                    'final_losses': st32}
                del r32
                torch.cuda.empty_cache()
            out['config']['secondary'] = sec
        if args.gpus == 1 and not args.no_cpu_baseline:
            out['cpu_baseline'] = cpu_baseline(args)
        print(json.dumps(out))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def measure_collectives(rep, sess, sync, reps=10):
    """All-reduce time of each gradient bucket as the steps issue them (every rank takes part; max over ranks)."""
    out = {'backend': dist.get_backend()}
    store = rep.d_store
    lo, hi = rep.big_slice()
    pieces = {'d_big_slice': store.grads[lo:hi], 'd_rest': torch.cat([store.grads[:lo], store.grads[hi:]]),
              'g_bucket': rep.g_store.grads}
    for name, t in pieces.items():
        buf = torch.zeros_like(t)
        for _ in range(2):
            dist.all_reduce(buf)
        sync()
        t0 = time.perf_counter()
        for _ in range(reps):
            dist.all_reduce(buf)
        torch.cuda.synchronize()
        ms = torch.tensor([(time.perf_counter() - t0) / reps * 1e3], dtype=torch.float64, device=sess.device)
        dist.all_reduce(ms, op=dist.ReduceOp.MAX)
        out[name] = {'bytes': buf.numel() * 4, 'ms': float(ms.item()),
                     'per_step': 5 if name.startswith('d_') else 1}
    out['allreduce_ms_per_step'] = 5 * (out['d_big_slice']['ms'] + out['d_rest']['ms']) + out['g_bucket']['ms']
    out['note'] = ('standalone back-to-back all-reduces of the same buffers; in the step the big critic slice and the '
                   'generator bucket are exchanged asynchronously under the remaining backward work (DESIGN.md s.6)')
    return out


if __name__ == '__main__':
    main()
