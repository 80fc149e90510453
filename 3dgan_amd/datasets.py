"""Input pipeline entry point with the contract of the reference's `data.py:34-60`
(`get_dataset(args) -> x, x_init, x_count`) / `hem.get_dataset_tensors` (hem/util/data.py:60-100): here
`(source, count, image_shape)` where `source.next_batch()` serves this replica's batches already resident in HBM.

The dataset itself is a plugin (`3dgan_amd/data_plugins/*.py`, discovered like hem/util/data.py:11-29): cifar, mnist,
floorplan (gen-1 name: floorplans), nyuv2, synthetic.  `--resize W H` and `--grayscale` (train.py:226-231) are applied
once on the device by the image plugins.
"""
from . import plugins
from .arguments import dataset_plugin_name


def get_dataset(args, sess):
    name = dataset_plugin_name(args.dataset)
    found = plugins.data_plugins()
    if name not in found:
        raise NotImplementedError('unknown --dataset %r (available: %s)' % (args.dataset, ', '.join(sorted(found))))
    return found[name].get_source(args, sess)
