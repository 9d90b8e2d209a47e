"""The plug-in boundary of the reference: ``ConfigParser.initialize(name, module, *args, **kwargs)``
(parse_config_dist_multi.py:73-100) restated for a plain dict config, so the same JSON that builds the reference's
``model.model.ObjectRelation`` / ``model.loss.GlobalLocalLoss`` builds the MI355X ones:

    cfg = json.load(open('configs/pt/o2t-cl-local-select-loss-cc.json'))
    model = initialize(cfg, 'arch', demovlp_amd.model)
    loss  = initialize(cfg, 'loss', demovlp_amd.loss)
"""
from __future__ import annotations

import inspect
import json


def read_json(path):
    with open(path, "rt") as f:
        return json.load(f)


def initialize(config: dict, name: str, module, *args, index=None, **kwargs):
    entry = config[name] if index is None else config[name][index]
    module_name = entry["type"]
    module_args = dict(entry["args"])
    if index is None:
        assert all(k not in module_args for k in kwargs), "Overwriting kwargs given in config file is not allowed"
        module_args.update(kwargs)
    cls = getattr(module, module_name)
    # constructor parameters missing from the sub-dict are filled from top-level config keys (:88-92)
    for param in inspect.signature(cls.__init__).parameters:
        if param not in module_args and param in config:
            module_args[param] = config[param]
    return cls(*args, **module_args)
