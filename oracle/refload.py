"""Import the reference's own modules from /root/reference (build container only).

The reference imports `cv2` (utils/other_utils.py:2) and `smplx` (model/posenet.py:5),
neither of which is installed here; both are stubbed.  `smplx.create`
returns whatever body model was registered with `set_body_model`.  Nothing on the GPU
box may call this: /root/reference does not exist there.
"""
import os
import sys
import types

REF_ROOT = '/root/reference'
_body_model = None


def set_body_model(m):
    global _body_model
    _body_model = m


def available():
    return os.path.isdir(os.path.join(REF_ROOT, 'model'))


def load():
    """Return a namespace with the reference classes/modules."""
    if not available():
        raise RuntimeError('reference tree not present')
    if REF_ROOT not in sys.path:
        sys.path.insert(0, REF_ROOT)
    if 'cv2' not in sys.modules:
        sys.modules['cv2'] = types.ModuleType('cv2')
    if 'smplx' not in sys.modules:
        import torch
        smplx = types.ModuleType('smplx')

        class _Empty(torch.nn.Module):
            pass

        def create(**kw):
            return _body_model if _body_model is not None else _Empty()
        smplx.create = create
        sys.modules['smplx'] = smplx
    import importlib
    ns = types.SimpleNamespace()
    ns.posenet = importlib.import_module('model.posenet')
    ns.trajnet = importlib.import_module('model.trajnet')
    ns.gd_posenet = importlib.import_module('diffusion.gaussian_diffusion_posenet')
    ns.gd_trajnet = importlib.import_module('diffusion.gaussian_diffusion_trajnet')
    ns.respace = importlib.import_module('diffusion.respace')
    ns.model_util = importlib.import_module('utils.model_util')
    ns.motion_repr = importlib.import_module('data_loaders.motion_representation')
    ns.quaternion = importlib.import_module('data_loaders.common.quaternion')
    ns.konia = importlib.import_module('utils.konia_transform')
    ns.other_utils = importlib.import_module('utils.other_utils')
    return ns
