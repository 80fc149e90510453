"""`--dataset synthetic`: the bench / smoke input of SURVEY.md section 8d (not in the reference, which has no way to run
without its private data): uint8 U{0..255} / 255 images, or (rgb, depth) pairs for --model pix2pix."""
from .DataPlugin import DataPlugin
from ..data import SyntheticSource, SyntheticPairSource


class SyntheticDataset(DataPlugin):
    name = 'synthetic'

    @staticmethod
    def arguments():
        return {'--resize': {'type': int, 'nargs': 2, 'help': 'Image size w x h of the synthetic stream (default 32 32).'},
                '--random_crop': {'type': int, 'nargs': 2, 'help': 'Accepted so that nyuv2 configs run on synthetic pairs (always 256 x 256).'}}

    @staticmethod
    def get_source(args, sess):
        B = args.batch_size
        if args.model == 'pix2pix':
            return SyntheticPairSource(4, B, sess.device, 256, 1234, sess.rank), 4 * B * sess.world_size, (256, 256, 3)
        shape = (32, 32, 3)
        if getattr(args, 'resize', None):
            shape = (args.resize[1], args.resize[0], 3)
        if getattr(args, 'grayscale', False):
            shape = shape[:2] + (1,)
        return SyntheticSource(50000, shape, B, sess.device, 1234, sess.rank), 50000, shape
