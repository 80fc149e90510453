#!/usr/bin/env python3
"""Headline benchmark: images/sec (whole node) of CIFAR-10-shaped IWGAN training, bs=512 per
GPU (BASELINE.json metric).

One "step" = one `train_func` call = n_disc_train (5) discriminator steps + 1 generator step
on 6 fresh synthetic batches (models/gan.py:169-173), optimizer steps included;
images/sec = steps/s x B x n_gpus (the reference's own progress unit, train.py:298).

  python bench.py --gpus 1 --steps K --warmup W
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Prints ONE JSON line on rank 0 (contract in the task statement) with `roofline` (dominant
conv GEMM kernel, HIP-event timed inside the timed region) and `cpu_baseline` (the oracle's
torch-autograd port on the host cores, bounded sample, rank 0 at N=1 only).
"""
import argparse
import importlib
import json
import os
import sys
import time
from types import SimpleNamespace

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

# IWGAN 32x32x3, L=200, reference-faithful variant (G step re-evaluates d_loss): SURVEY 8d / BASELINE.md s.2
GFLOP_PER_IMAGE_ITERATION = 30.08
PEAK_TFLOPS = {'bf16': 2500.0, 'f32': 157.3}        # MI355X_MICROARCH.md, dense


def cpu_baseline(args):
    """The oracle's autograd port of the identical iteration on the host cores (kind "port")."""
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
    tr.d_step(*[v[0] for v in inputs()])           # untimed warm-up (allocator, thread pool)
    # bounded sample: whole iterations until ~12 s of CPU work are spent (at least 2, at most 8)
    n, t0 = 0, time.time()
    while n < 2 or (time.time() - t0 < 12.0 and n < 8):
        tr.train_func(*inputs())
        n += 1
    dt = time.time() - t0
    return {'value': n * B / dt, 'unit': 'images/sec', 'cores': cores, 'kind': 'port',
            'sample': '%d iterations (5 D + 1 G steps each) at batch %d in %.1f s, f32 torch-autograd port of the oracle, '
                      '%d threads (CPU restatement, not TensorFlow)' % (n, B, dt, cores)}


PMC_FILE = os.path.join(ROOT, 'profiles', 'r01_r_pmc_fetch_write_per_kernel.json')


def pmc_traffic(symbol):
    """HBM-side bytes per launch of the kernel `symbol` from the committed rocprofv3 PMC passes (FETCH_SIZE and
    WRITE_SIZE in separate runs of this same bench, KB per dispatch averaged over the kernel's launches), with the
    gfx950 correction of MI355X_MICROARCH.md §HBM: FETCH_SIZE counts 128-B requests as 64 B for 16-B-per-lane
    streaming reads, so bytes = (2 * FETCH_SIZE + WRITE_SIZE) * 1024.  None when the file or the kernel is absent."""
    try:
        with open(PMC_FILE) as f:
            pmc = json.load(f)
    except OSError:
        return None, None
    base = symbol.split('<')[0]
    parts = symbol[symbol.index('<') + 1:-1].split(',') if '<' in symbol else []
    want = base + 'I' + ('DF16b' if parts and parts[0] == 'bf16' else 'f') + ''.join('Li%sE' % v for v in parts[1:])

    def find(name):
        for k, v in pmc.get(name, {}).items():
            if want in k or (base in k and 'wgrad_dma' in base):
                return v['avg']
        return None
    fetch, write = find('FETCH_SIZE'), find('WRITE_SIZE')
    if fetch is None or write is None:
        return None, None
    return (2.0 * fetch + write) * 1024.0, os.path.relpath(PMC_FILE, ROOT) + ' (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes; bytes per launch)'


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=10)
    ap.add_argument('--warmup', type=int, default=3)
    ap.add_argument('--batch_size', type=int, default=512)
    ap.add_argument('--latent_size', type=int, default=200)
    ap.add_argument('--dtype', default='bf16', choices=['bf16', 'f32'])
    ap.add_argument('--model', default='iwgan')
    ap.add_argument('--cpu_batch', type=int, default=128)
    ap.add_argument('--no_cpu_baseline', action='store_true')
    ap.add_argument('--no_kernel_timer', action='store_true')
    ap.add_argument('--timer_steps', type=int, default=2)
    ap.add_argument('--no_graphs', action='store_true')
    ap.add_argument('--dump_launches', default=None, help='diagnostics: write (kernel, GFLOP, count, avg ms, TFLOP/s) per distinct launch shape to this file')
    args = ap.parse_args()

    rt = importlib.import_module('3dgan_amd.runtime')
    K = importlib.import_module('3dgan_amd.kernels')
    gan = importlib.import_module('3dgan_amd.models.gan')
    data = importlib.import_module('3dgan_amd.data')

    world = rt.init_distributed()
    if world != args.gpus and not (world == 1 and args.gpus == 1):
        raise SystemExit('--gpus %d but WORLD_SIZE=%d: launch with torch.distributed.run --nproc-per-node %d'
                         % (args.gpus, world, args.gpus))
    sess = rt.Session(dtype=K.BF16 if args.dtype == 'bf16' else K.F32, seed=0)
    margs = SimpleNamespace(model=args.model, batch_size=args.batch_size, latent_size=args.latent_size,
                            image_shape=(32, 32, 3), n_gpus=args.gpus, optimizer='adam', lr=1e-4, beta1=0.5,
                            beta2=0.9, decay=0.9, momentum=0.01, centered=False, n_disc_train=5,
                            display_d_loss=True, use_graphs=not args.no_graphs)              # examples/iwgan.config
    src = data.SyntheticSource(12 * args.batch_size, margs.image_shape, args.batch_size, sess.device, 1234, sess.rank)
    rep = gan.GanReplica(src, margs, sess)
    rt.broadcast_store(rep.g_store)
    rt.broadcast_store(rep.d_store)
    rep.refresh()

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
    if not args.no_kernel_timer and args.timer_steps > 0:
        rep.use_graphs = False
        rep.train_func()
        if sess.rank == 0:
            K.timing_begin()                      # HIP events around every conv GEMM kernel launch, inside the library
        for _ in range(args.timer_steps):
            rep.train_func()
        sync()
        if sess.rank == 0:
            timer = K.timing_end()
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=sess.device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

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
                       'consumed_images_per_sec': value * 6,
                       'step_tflops': (value / args.gpus * GFLOP_PER_IMAGE_ITERATION / 1e3) if args.model == 'iwgan' else None,
                       'final_losses': status},
        }
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
            sym = {}
            for name, kms, fl in timer:
                e = sym.setdefault(name, [0, 0.0, 0.0])
                e[0] += 1
                e[1] += kms
                e[2] += fl
            gemm = {k: v for k, v in sym.items() if v[2] > 0}
            # dominant kernel = the GEMM kernel symbol with the largest total time
            kind, (n, tot_ms, fl) = max(gemm.items(), key=lambda kv: kv[1][1])
            achieved = fl / (tot_ms * 1e-3) / 1e12
            gemm_ms = sum(v[1] for v in gemm.values())
            gemm_fl = sum(v[2] for v in gemm.values())
            traffic, traffic_src = pmc_traffic(kind)
            out['roofline'] = {'bound': 'mfma', 'achieved': achieved, 'peak': PEAK_TFLOPS[args.dtype], 'unit': 'TFLOP/s',
                               'frac': achieved / PEAK_TFLOPS[args.dtype], 'traffic': traffic, 'traffic_source': traffic_src,
                               'kernel': kind, 'launches': n, 'avg_launch_ms': tot_ms / n,
                               'flop_per_launch': fl / n,
                               'measured': 'HIP events recorded by the library around every conv GEMM kernel launch (on the launch '
                                           'stream) in %d instrumented eager steps run directly after the timed region (hipGraph '
                                           'replay hides launches from events); achieved = sum of algorithmic FLOPs / sum of '
                                           'durations over all launches of this kernel' % args.timer_steps,
                               'all_conv_gemms': {'ms_per_step': gemm_ms / args.timer_steps,
                                                  'tflops': gemm_fl / (gemm_ms * 1e-3) / 1e12,
                                                  'share_of_step': gemm_ms / args.timer_steps / ms},
                               'per_kernel': {k: {'launches': v[0], 'avg_ms': round(v[1] / v[0], 4),
                                                  'tflops': round(v[2] / (v[1] * 1e-3) / 1e12, 1)}
                                              for k, v in sorted(sym.items(), key=lambda kv: -kv[1][1])}}
        if args.gpus == 1 and not args.no_cpu_baseline:
            out['cpu_baseline'] = cpu_baseline(args)
        print(json.dumps(out))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
