"""CPU oracle for the beta-cores SNNLS / projection hot path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``beta_cores_amd/`` may import this
package: it is the checker that the HIP path is compared against in
``tests/``, in ``__graft_entry__.smoke()`` and in the ``cpu_baseline`` leg of
``bench.py``.  It is a NumPy restatement of the reference's algorithm, written
pass-for-pass (same number of N x S sweeps, same expression order) so that

  * element-wise formulas are bit-identical to the reference, and
  * BLAS-dependent scalars agree to ~1e-12 relative,

and it is pinned against outputs of the reference itself run in the build
container (``tests/golden/make_golden.py`` -> ``tests/golden/*.npz``; see
``tests/test_oracle_golden.py``).  The reference holds no golden vectors of
its own for this path, so those generated fixtures are
the pin.  ``scipy.optimize.nnls`` (OrthoPursuit refit, ``optimize()``) is
third-party arithmetic shared with the reference rather than restated:
the reference pins scipy 1.5.1, this image has 1.15.3 -> that boundary is
"parity unpinned" beyond the 1e-5 tolerance recorded in the goldens.
"""
from .snnls_ref import (RefGIGA, RefFrankWolfe, RefOrthoPursuit,
                        RefImportanceSampling, RefUniformSampling,
                        RefNumericalPrecisionError)
from . import models_ref, coreset_ref
