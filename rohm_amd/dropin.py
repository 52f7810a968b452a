"""Zero-edit drop-in for the RoHM drivers' import block.

The drivers (`test_amass_full.py:10-15`, `test_prox_egobody.py:10-15`, `test_posenet.py`, `test_trajnet.py`) import the hot path
by module path:

    from model.posenet import PoseNet
    from diffusion import gaussian_diffusion_posenet
    from model.trajnet import TrajNet
    from diffusion import gaussian_diffusion_trajnet
    from diffusion.respace import SpacedDiffusionPoseNet, SpacedDiffusionTrajNet
    from utils.model_util import create_gaussian_diffusion

`install()` makes exactly those six module paths resolve to their `rohm_amd` counterparts and nothing else: the reference's
own `utils.other_utils`, `utils.dist_util`, `utils.fixseed`, `utils.vis_util`, `data_loaders.*`, `diffusion.logger`, ... keep
coming from the RoHM checkout the driver is run from (they are host code outside the hot path).  Nothing is
copied or patched on disk and no `sitecustomize` is involved:

    python -m rohm_amd.dropin test_amass_full.py --config cfg_files/test_cfg/amass.yaml        # runs the driver unmodified

or, inside a process, `import rohm_amd.dropin; rohm_amd.dropin.install()` before the driver's imports.

How: the six names are answered by a meta-path finder placed in front of the path finders (so they win over the checkout's files
whatever the import order) and are pre-registered in `sys.modules`.  A parent package (`model`, `diffusion`, `utils`) is only
synthesised when none is importable -- i.e. when there is no RoHM checkout on `sys.path`; otherwise the checkout's package stays
the parent and its other sub-modules stay reachable.
"""
from __future__ import annotations

import importlib
import importlib.abc
import importlib.machinery
import importlib.util
import sys
import types

# reference module path -> rohm_amd module (each mirrors names, signatures and state_dict keys of the file it stands for)
ALIASES = {
    'model.posenet': 'rohm_amd.model.posenet',                                      # model/posenet.py:11-317
    'model.trajnet': 'rohm_amd.model.trajnet',                                      # model/trajnet.py:43-400
    'diffusion.gaussian_diffusion_posenet': 'rohm_amd.diffusion.gaussian_diffusion_posenet',
    'diffusion.gaussian_diffusion_trajnet': 'rohm_amd.diffusion.gaussian_diffusion_trajnet',
    'diffusion.respace': 'rohm_amd.diffusion.respace',                              # diffusion/respace.py
    'utils.model_util': 'rohm_amd.utils.model_util',                                # utils/model_util.py:6-40
}
_STUB_FLAG = '__rohm_amd_dropin_stub__'


class _AliasFinder(importlib.abc.MetaPathFinder, importlib.abc.Loader):
    """Answers the six aliased names with the already-imported rohm_amd module (the module object itself, not a copy: classes keep
    one identity whichever path imported them)."""

    def find_spec(self, fullname, path=None, target=None):
        if fullname in ALIASES:
            return importlib.machinery.ModuleSpec(fullname, self, origin=f'alias of {ALIASES[fullname]}')
        return None

    def create_module(self, spec):
        mod = importlib.import_module(ALIASES[spec.name])
        self._own_spec = getattr(mod, '__spec__', None)
        return mod

    def exec_module(self, module):       # already executed under its own name; the import machinery re-labelled __spec__: undo
        if getattr(self, '_own_spec', None) is not None:
            module.__spec__ = self._own_spec


_finder = _AliasFinder()
_saved: dict = {}


def _parent_importable(name):
    if name in sys.modules:
        return not getattr(sys.modules[name], _STUB_FLAG, False)
    try:
        return importlib.util.find_spec(name) is not None
    except (ImportError, ValueError):
        return False


def install():
    """Route the six module paths of the drivers' import block to rohm_amd.  Idempotent; `uninstall()` undoes it.  Returns the
    mapping that is in force."""
    if _finder not in sys.meta_path:
        sys.meta_path.insert(0, _finder)
    for ref_name, ours in ALIASES.items():
        mod = importlib.import_module(ours)
        parent, _, leaf = ref_name.rpartition('.')
        if not _parent_importable(parent):
            # no RoHM checkout on sys.path: an empty package that only knows the aliased children
            pkg = sys.modules.get(parent)
            if pkg is None:
                pkg = types.ModuleType(parent)
                pkg.__path__ = []
                pkg.__package__ = parent
                setattr(pkg, _STUB_FLAG, True)
                sys.modules[parent] = pkg
        if ref_name not in _saved:
            _saved[ref_name] = sys.modules.get(ref_name)
        sys.modules[ref_name] = mod
        pkg = sys.modules.get(parent)
        if pkg is None:                        # the checkout's package (the drivers import it anyway; usually a namespace package)
            try:
                pkg = importlib.import_module(parent)
            except ImportError:
                pkg = None
        if pkg is not None:                    # `import model.posenet as m` / `model.posenet.PoseNet` resolve through the parent
            try:
                setattr(pkg, leaf, mod)
            except (AttributeError, TypeError):
                pass
    return dict(ALIASES)


def uninstall():
    """Remove the aliases (previous `sys.modules` entries come back; synthesised parents disappear)."""
    if _finder in sys.meta_path:
        sys.meta_path.remove(_finder)
    for ref_name in ALIASES:
        prev = _saved.pop(ref_name, None)
        if sys.modules.get(ref_name) is not None and sys.modules[ref_name].__name__ == ALIASES[ref_name]:
            if prev is not None:
                sys.modules[ref_name] = prev
            else:
                del sys.modules[ref_name]
        parent, _, leaf = ref_name.rpartition('.')
        pkg = sys.modules.get(parent)
        if pkg is not None:
            if getattr(pkg, leaf, None) is not None and getattr(getattr(pkg, leaf), '__name__', '') == ALIASES[ref_name]:
                try:
                    delattr(pkg, leaf)
                except AttributeError:
                    pass
            if getattr(pkg, _STUB_FLAG, False) and not any(k.startswith(parent + '.') for k in sys.modules):
                del sys.modules[parent]


def installed():
    return _finder in sys.meta_path and all(
        getattr(sys.modules.get(k), '__name__', None) == v for k, v in ALIASES.items())


def main(argv=None):
    """`python -m rohm_amd.dropin <driver.py> [driver args...]`: run a RoHM driver unmodified with the aliases in place.  The
    driver's directory goes to the front of `sys.path` exactly as `python <driver.py>` would put it."""
    import os
    import runpy
    argv = list(sys.argv[1:] if argv is None else argv)
    if not argv or argv[0] in ('-h', '--help'):
        print(main.__doc__)
        return 2
    script = argv[0]
    sys.path.insert(0, os.path.dirname(os.path.abspath(script)))
    install()
    sys.argv = argv
    runpy.run_path(script, run_name='__main__')
    return 0


if __name__ == '__main__':
    sys.exit(main())
