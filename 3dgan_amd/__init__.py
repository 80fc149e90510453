"""3dgan_amd -- MI355X-native training hot path of algoterranean/3dgan.

The package name starts with a digit, so import it with
`importlib.import_module('3dgan_amd')` (tests/conftest.py and train.py do).
"""
__version__ = '0.1.0'
