"""Import shim for the upstream reference (THIS container only).

`/root/reference` never travels to the GPU box; this module is used only by
`tests/golden/make_golden.py` to generate the committed golden vectors by
importing the reference's Python modules.  It stubs the third-party modules
the reference imports at module top but that are not installed here
(`wandb`, `prody`, `pymol`), and papers over two library incompatibilities
(`numpy.bool`, `ReduceLROnPlateau(verbose=...)`), as listed in SURVEY.md
Appendix C.  Nothing of the reference is copied: it is imported in place.
"""
import sys
import types
from unittest import mock

REFERENCE_ROOT = "/root/reference"


def install():
    import numpy as np
    if not hasattr(np, "bool"):
        np.bool = bool  # structure_utils.py:26 uses the removed alias
    for name in ("wandb", "prody", "pymol"):
        if name not in sys.modules:
            m = mock.MagicMock(name=name)
            sys.modules[name] = m
    # pymol.cmd is imported as a submodule
    sys.modules.setdefault("pymol.cmd", sys.modules["pymol"].cmd)
    if REFERENCE_ROOT not in sys.path:
        sys.path.insert(0, REFERENCE_ROOT)
    return types.SimpleNamespace(root=REFERENCE_ROOT)
