"""Model builders (reference: models/gan.py, models/vae.py; hem/models/pix2pix.py)."""


def model_funcs():
    """The dispatch table of train.py:240-244."""
    from .gan import gan
    from .vae import vae
    return {'gan': gan, 'wgan': gan, 'iwgan': gan, 'vae': vae}


def get_model(name):
    """hem/models/ModelPlugin.py:4-8: plugin lookup by `name`."""
    from .pix2pix import pix2pix
    plugins = {pix2pix.name: pix2pix}
    return plugins[name]
