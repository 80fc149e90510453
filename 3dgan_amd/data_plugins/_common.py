"""Shared pieces of the dataset plugins: TF-1.x `resize_images` (legacy bilinear), per-replica sharding, grayscale."""
import numpy as np
import torch

from ..data import ArraySource, StreamingSource


def resize_bilinear_tf1(x, out_h, out_w):
    """tf.image.resize_images(x, [h, w]) of TF 1.x (bilinear, align_corners=False, NO half-pixel centres:
    src = dst * in / out; data.py:21, hem/data/nyuv2.py:178-179).  x: [N, H, W, C] float tensor."""
    n, h, w, c = x.shape
    if (h, w) == (out_h, out_w):
        return x
    dev = x.device

    def axis(n_in, n_out):
        src = torch.arange(n_out, device=dev, dtype=torch.float64) * (n_in / n_out)
        lo = src.floor().long().clamp_(0, n_in - 1)
        hi = (lo + 1).clamp_(max=n_in - 1)
        return lo, hi, (src - lo.double()).to(x.dtype)
    y0, y1, fy = axis(h, out_h)
    x0, x1, fx = axis(w, out_w)
    top = x[:, y0][:, :, x0] * (1 - fx)[None, None, :, None] + x[:, y0][:, :, x1] * fx[None, None, :, None]
    bot = x[:, y1][:, :, x0] * (1 - fx)[None, None, :, None] + x[:, y1][:, :, x1] * fx[None, None, :, None]
    return top * (1 - fy)[None, :, None, None] + bot * fy[None, :, None, None]


def finish_images(imgs_u8, args, sess, pad_to_32=False):
    """uint8 [N, H, W, C] -> (source, N, shape): /255, optional MNIST pad, --resize W H, --grayscale (train.py:226-231),
    this replica's shard (rank r takes every world_size-th example: ops/input.py:24 on a shuffled stream)."""
    import torch.nn.functional as F
    n = imgs_u8.shape[0]
    # a dataset whose float32 form exceeds the HBM budget (or --streaming) stays in host memory and is served by the reference's
    # own pipeline shape: repeat -> shuffle(buffer_size) -> batch(B * n_gpus) over a pinned-host ring (data.StreamingSource)
    budget = float(getattr(args, 'hbm_budget_gb', 64.0) or 64.0) * (1 << 30)
    if getattr(args, 'streaming', False) or float(np.prod(imgs_u8.shape)) * 4.0 > budget:
        def post(x):
            if pad_to_32 and x.shape[1] == 28:
                x = F.pad(x.permute(0, 3, 1, 2), (2, 2, 2, 2)).permute(0, 2, 3, 1)
            if getattr(args, 'resize', None):
                x = resize_bilinear_tf1(x, args.resize[1], args.resize[0])
            if getattr(args, 'grayscale', False) and x.shape[-1] == 3:
                x = (x * torch.tensor([0.2989, 0.5870, 0.1140], device=x.device)).sum(-1, keepdim=True)
            return x.contiguous()
        seed = args.seed if isinstance(getattr(args, 'seed', None), int) else 0
        src = StreamingSource(imgs_u8, args.batch_size, sess.device, buffer_size=getattr(args, 'buffer_size', 10000), seed=seed,
                              rank=sess.rank, world=sess.world_size, shuffle=getattr(args, 'shuffle', True), post=post)
        probe = post(torch.zeros((1,) + tuple(imgs_u8.shape[1:]), device=sess.device))
        return src, n, tuple(probe.shape[1:])
    x = torch.from_numpy(np.ascontiguousarray(imgs_u8)).to(sess.device).float() / 255.0
    if pad_to_32 and x.shape[1] == 28:                    # MNIST: 28 -> 32 so the 2x deconv ladder fits
        x = F.pad(x.permute(0, 3, 1, 2), (2, 2, 2, 2)).permute(0, 2, 3, 1)
    if getattr(args, 'resize', None):
        w, h = args.resize
        x = resize_bilinear_tf1(x, h, w)
    if getattr(args, 'grayscale', False) and x.shape[-1] == 3:
        x = (x * torch.tensor([0.2989, 0.5870, 0.1140], device=x.device)).sum(-1, keepdim=True)   # tf.image.rgb_to_grayscale
    x = x.contiguous()
    shard = x[sess.rank::sess.world_size] if sess.world_size > 1 else x
    seed = (args.seed if isinstance(getattr(args, 'seed', None), int) else 0) + sess.rank
    src = ArraySource(shard.cpu().numpy(), args.batch_size, sess.device, shuffle_seed=seed if getattr(args, 'shuffle', True) else None)
    return src, n, tuple(x.shape[1:])
