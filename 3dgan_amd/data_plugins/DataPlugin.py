"""Base class of dataset plugins (hem/data/DataPlugin.py:24-60): `name`, `arguments()` -> {flag: argparse kwargs} merged
into the command line by the 3-pass parser (3dgan_amd/arguments.py), file checks, and -- replacing the reference's
`get_datasets(args)` (TFRecordDataset objects) -- `get_source(args, sess) -> (source, n_examples, image_shape)` where
`source.next_batch()` yields this replica's next batch on the device (float32 in [0, 1], NHWC)."""
import os


class DataPlugin:
    name = None

    @staticmethod
    def arguments():
        """{flag: **kwargs for argparse.add_argument}."""
        return {}

    @staticmethod
    def check_files(storage_dir, required_files):
        """hem/data/DataPlugin.py:38-44."""
        have = os.listdir(storage_dir) if os.path.isdir(storage_dir) else []
        return all(f in have for f in required_files)

    @staticmethod
    def check_prepared_datasets(storage_dir):
        return False

    @staticmethod
    def get_source(args, sess):
        raise NotImplementedError


def dataset_dirs(args):
    """Where prepared files may live: gen-2 `--dataset_dir` (default 'datasets'), gen-1's hard-wired 'data/' (data.py:37-39)."""
    out = []
    for d in (getattr(args, 'dataset_dir', None), getattr(args, 'data_dir', None), 'data'):
        if d and d not in out:
            out.append(d)
    return out


def find_file(args, names):
    for d in dataset_dirs(args):
        for n in names:
            p = os.path.join(d, n)
            if os.path.exists(p):
                return p
    return None


def write_cache_atomically(path, write):
    """`write(file_object)` into a per-process temporary file beside `path`, then os.replace: with `--n_gpus > 1` every
    replica process decodes and caches the same dataset, and a late starter must find either no cache or a whole one."""
    os.makedirs(os.path.dirname(os.path.abspath(path)), exist_ok=True)
    tmp = '%s.%d.tmp' % (path, os.getpid())
    try:
        with open(tmp, 'wb') as f:
            write(f)
        os.replace(tmp, path)
    finally:
        if os.path.exists(tmp):
            os.remove(tmp)
