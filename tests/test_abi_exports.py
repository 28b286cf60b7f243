"""CPU-only ABI checks: libshoulder_hip.so loads without a GPU, exports every entry point that
include/shoulder_hip.h declares, and the record layout agrees between the C header and the ctypes /
NumPy mirrors.  No compute call is made."""
import ctypes
import os
import re
import subprocess

from conftest import ROOT
from shoulder_amd import _lib


def _declared():
    src = open(os.path.join(ROOT, "include", "shoulder_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(sh_[a-z_]+)\s*\(", src)))


def test_every_declared_symbol_is_exported():
    names = _declared()
    assert len(names) >= 20
    L = ctypes.CDLL(_lib.lib_path()) if os.path.exists(_lib.lib_path()) else _lib.load()
    missing = [n for n in names if not hasattr(L, n)]
    assert not missing, missing
    assert set(_lib.EXPORTS) == set(names)


def test_record_layout_matches_header(tmp_path):
    prog = tmp_path / "sz.c"
    prog.write_text('#include <stdio.h>\n#include <stddef.h>\n#include "shoulder_hip.h"\nint main(){printf("%zu %zu %zu %zu %zu\\n", sizeof(sh_landmarks),'
                    ' offsetof(sh_landmarks, csys), offsetof(sh_landmarks, groove_points), offsetof(sh_landmarks, n_anp), sizeof(sh_params));return 0;}\n')
    exe = tmp_path / "sz"
    subprocess.check_call(["gcc", "-I", os.path.join(ROOT, "include"), str(prog), "-o", str(exe)])
    size, o_csys, o_gp, o_n, psize = map(int, subprocess.check_output([str(exe)]).split())
    dt = _lib.LANDMARKS_DTYPE
    assert size == dt.itemsize == ctypes.sizeof(_lib.Landmarks)
    assert o_csys == dt.fields["csys"][1] and o_gp == dt.fields["groove_points"][1] and o_n == dt.fields["n_anp"][1]
    assert psize == ctypes.sizeof(_lib.Params)


def test_ctx_create_fails_loudly_without_gpu():
    """No HIP device here (or a bad index): the call reports an error instead of falling back."""
    L = _lib.load()
    h = ctypes.c_void_p()
    rc = L.sh_ctx_create(9999, None, ctypes.byref(h))
    assert rc != 0 and not h.value


def test_collective_entry_points_check_their_arguments_without_a_gpu():
    """sh_comm_init_all / sh_bcast_weights / sh_gather_landmarks: exported, and an empty group is an argument error
    (RCCL itself is only loaded once a group of real contexts asks for it)."""
    L = _lib.load()
    assert L.sh_comm_init_all(None, 0) == -1 and L.sh_bcast_weights(None, 0, 0) == -1 and L.sh_gather_landmarks(None, 0, None) == -1
    import subprocess
    deps = subprocess.run(["ldd", _lib.lib_path()], capture_output=True, text=True).stdout
    assert "rccl" not in deps


def test_packed_record_layout():
    """sh_set_record_rows(R): the NumPy mirror of the packed wire record has the library's size and keeps the field order of the
    full record around the point list (head | six trailing int32 | R rows)."""
    L = ctypes.CDLL(_lib.lib_path()) if os.path.exists(_lib.lib_path()) else _lib.load()
    L.sh_record_bytes.restype = ctypes.c_size_t
    L.sh_record_bytes.argtypes = [ctypes.c_int]
    full = _lib.LANDMARKS_DTYPE
    assert L.sh_record_bytes(0) == full.itemsize
    for R in (1, 1024, 1536, 4096):
        dt = _lib.record_dtype(R)
        assert dt.itemsize == L.sh_record_bytes(R) == 8680 + 24 * R
        for name in full.names:
            if name == "anp_points":
                continue
            off = full.fields[name][1]
            assert dt.fields[name][1] == (off if off < full.fields["anp_points"][1] else off - 4096 * 24), name
        assert dt.fields["anp_points"][1] == 8680 and dt["anp_points"].shape == (R, 3)
