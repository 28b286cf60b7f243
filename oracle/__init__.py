"""CPU oracle for the shoulder.Humerus landmark path.

TEST INFRASTRUCTURE ONLY.  This package is a NumPy/SciPy (fp64) restatement of the
reference algorithm (gregspangenberg/shoulder @ 2025-04-10, `src/shoulder/**`), one
function per reference call site, each citing the file:line it follows.  Only
`tests/`, `__graft_entry__.smoke()` and the `cpu_baseline` leg of `bench.py` may
import it, and only as the checker.  The product (`shoulder_amd`) never imports it.

Pinning status (see DESIGN.md "Oracle"):
  * functions restating the reference's own NumPy/SciPy/sklearn arithmetic
    (utils.py, slice.py resample/polar/cutoff, bicipital_groove.py) are pinned by
    golden vectors captured from the reference's own code (tests/golden/*.npz,
    generator tests/golden/make_golden.py);
  * scipy.signal / sklearn primitives are called directly (same third-party code
    the reference calls);
  * everything that lives in third-party packages absent from this image
    (trimesh, shapely, ruptures, circle_fit, lsq-ellipse, scikit-spatial,
    onnxruntime) is restated from the published algorithm of the pinned version
    (poetry.lock) -- for those pieces the reference holds no golden vectors and
    parity is UNPINNED; canonical rules for implementation-defined behaviour are
    listed in DESIGN.md.
"""
