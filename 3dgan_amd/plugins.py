"""Plugin discovery with the semantics of the reference's gen-2 registry (`hem/util/data.py:11-29`,
`hem/models/ModelPlugin.py:4-8`): every `*.py` in a plugin directory is imported and every class DEFINED in that module
whose FIRST base class is named `ModelPlugin` / `DataPlugin` is registered under its `name` attribute.  So dropping a
new file into `3dgan_amd/models/` or `3dgan_amd/data_plugins/` is all it takes to add `--model foo` / `--dataset foo`
(and its `arguments()` to the command line: 3dgan_amd/arguments.py).
"""
import importlib
import inspect
import os

_HERE = os.path.dirname(os.path.abspath(__file__))


def search_for_plugins(plugin_dir, plugin_module, plugin_name):
    """hem/util/data.py:11-29."""
    valid = []
    files = sorted(f for f in os.listdir(plugin_dir) if f.endswith('.py') and f not in ('__init__.py', plugin_name + '.py'))
    for f in files:
        module_name = plugin_module + '.' + f[:-3]
        mod = importlib.import_module(module_name)
        for _, cls in inspect.getmembers(mod, inspect.isclass):
            if cls.__module__ == module_name and cls.__bases__ and cls.__bases__[0].__name__ == plugin_name:
                valid.append(cls)
    return {cls.name: cls for cls in valid}


def model_plugins():
    return search_for_plugins(os.path.join(_HERE, 'models'), '3dgan_amd.models', 'ModelPlugin')


def data_plugins():
    return search_for_plugins(os.path.join(_HERE, 'data_plugins'), '3dgan_amd.data_plugins', 'DataPlugin')


def get_model(name):
    """hem/models/ModelPlugin.py:4-8: the plugin class registered under `name` (KeyError for an unknown name, as there)."""
    return model_plugins()[name]


def get_dataset(name):
    """hem/util/data.py:32-35."""
    return data_plugins()[name]
