"""`import shoulder` -- the reference's package name, served by shoulder_amd (MI355X-native implementation).

    import shoulder
    hum = shoulder.Humerus("humerus_left.stl")          # reference README flow, unchanged

Only the names of the reference's public surface that shoulder_amd provides are forwarded (Humerus, ProximalHumerus,
HumeralHeadOsteotomy, Plot); `shoulder.bone`, `shoulder.arthroplasty`, `shoulder.plotting` and `shoulder.base` resolve
to the shoulder_amd modules of the same name.  Nothing is computed here."""
import importlib
import sys

import shoulder_amd as _impl

__version__ = _impl.__version__
_SUBMODULES = ("bone", "arthroplasty", "plotting", "base")


def __getattr__(name):
    if name in _SUBMODULES:
        mod = importlib.import_module(f"shoulder_amd.{name}")
        sys.modules[f"{__name__}.{name}"] = mod
        return mod
    return getattr(_impl, name)
