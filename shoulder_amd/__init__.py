"""shoulder_amd -- MI355X-native implementation of the `shoulder.Humerus` landmark path.

    import shoulder_amd as shoulder
    hum = shoulder.Humerus("humerus_left.stl")
    hum.apply_csys_canal_transepiconylar()
    hum.canal.axis(); hum.trans_epiconylar.axis(); hum.anatomic_neck.points(); hum.bicipital_groove.axis()

Everything numeric runs in libshoulder_hip.so (hand-written gfx950 HIP kernels) behind the C-ABI of
include/shoulder_hip.h; see DESIGN.md and INTEGRATION.md.
"""
__version__ = "0.1.0"

_LAZY = {"Humerus": "bone", "ProximalHumerus": "bone", "default_engine": "bone", "Engine": "engine", "ShoulderHipError": "engine",
         "HumeralHeadOsteotomy": "arthroplasty", "Plot": "plotting"}


def __getattr__(name):
    if name in _LAZY:
        import importlib
        return getattr(importlib.import_module(f".{_LAZY[name]}", __name__), name)
    raise AttributeError(name)
