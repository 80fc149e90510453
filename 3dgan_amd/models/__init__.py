"""Model builders (reference: models/gan.py, models/vae.py; hem/models/pix2pix.py)."""
