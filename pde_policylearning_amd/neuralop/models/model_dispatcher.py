"""`get_model(config)`: build a model from a configuration mapping (reference neuralop/models/model_dispatcher.py:7-94).
config['arch'] names the class, config[arch] holds its keyword arguments with `data_channels` in place of `in_channels`;
multi-grid patching multiplies the input channels by (levels + 1)."""
import inspect

from .tfno import FNO, FNO1d, FNO2d, FNO3d, TFNO, TFNO1d, TFNO2d, TFNO3d
from .uno import UNO

MODEL_ZOO = {'tfno': TFNO, 'tfno1d': TFNO1d, 'tfno2d': TFNO2d, 'tfno3d': TFNO3d,
             'fno': FNO, 'fno1d': FNO1d, 'fno2d': FNO2d, 'fno3d': FNO3d, 'uno': UNO}


def available_models():
    return list(MODEL_ZOO)


def dispatch_model(ModelClass, config):
    """ModelClass(**config), reporting arguments the class does not take and defaults the config leaves unset."""
    params = inspect.signature(ModelClass).parameters
    name = ModelClass.__name__
    for key in config:
        if key not in params:
            print(f"Given argument key={key!r} that is not in {name}'s signature.")
    for key, prm in params.items():
        if prm.default is not inspect.Parameter.empty and key not in config:
            print(f"Keyword argument {key} not specified for model {name}, using default={prm.default}.")
    return ModelClass(**config)


def get_model(config):
    arch = config['arch'].lower()
    if arch not in MODEL_ZOO:
        raise ValueError(f"Got config.arch={arch!r}, expected one of {available_models()}.")
    kwargs = dict(config.get(arch))
    channels = kwargs.pop('data_channels')
    try:
        levels = config['patching']['levels']
    except (KeyError, TypeError):
        levels = 0
    if levels:
        channels *= levels + 1
    kwargs['in_channels'] = channels
    return dispatch_model(MODEL_ZOO[arch], kwargs)
