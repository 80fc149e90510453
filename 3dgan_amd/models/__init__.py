"""Model builders (reference: models/gan.py, models/vae.py, models/cnn.py; hem/models/pix2pix.py)."""


def model_funcs():
    """The dispatch table of train.py:240-244."""
    from .gan import gan
    from .vae import vae
    from .cnn import cnn

    def pix2pix_func(x, args, sess=None):
        """gen-2 plugins expose .train(sess, args, feed_dict); adapt to the gen-1 train_func contract."""
        model = get_model('pix2pix')(x, args, sess)

        def train_func(sess_=None, args_=None):
            return model.train(sess_, args_, None)
        train_func.replica = model
        return train_func
    return {'gan': gan, 'wgan': gan, 'iwgan': gan, 'vae': vae, 'cnn': cnn, 'pix2pix': pix2pix_func}


def get_model(name):
    """hem/models/ModelPlugin.py:4-8: plugin lookup by `name` among the classes discovered in this directory whose
    first base is named `ModelPlugin` (3dgan_amd/plugins.py)."""
    from ..plugins import get_model as _get
    return _get(name)
