"""Model builders (reference: models/gan.py, models/vae.py; hem/models/pix2pix.py)."""


def model_funcs():
    """The dispatch table of train.py:240-244."""
    from .gan import gan
    from .vae import vae
    return {'gan': gan, 'wgan': gan, 'iwgan': gan, 'vae': vae}
