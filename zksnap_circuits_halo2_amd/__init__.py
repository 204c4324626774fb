"""zkhip: MI355X-native MSM / NTT kernels behind the halo2 prover boundary of ZkSnap's circuits.

Host-side mirror of the reference interface for this path ([DEP] halo2-axiom, reached from
/root/reference/aggregator/src/wrapper.rs:129): `arithmetic.best_multiexp`, `arithmetic.best_fft`,
`domain.EvaluationDomain`, `kzg.ParamsKZG`.  All arithmetic runs in libzkhip.so (HIP, gfx950) through the C ABI
declared in include/zkhip.h; importing this package without the built library raises ImportError on first use.
"""
from . import _lib, fields  # noqa: F401
from .arithmetic import best_fft, best_multiexp  # noqa: F401
from .domain import EvaluationDomain  # noqa: F401
from .kzg import ParamsKZG  # noqa: F401

__all__ = ["best_multiexp", "best_fft", "EvaluationDomain", "ParamsKZG", "fields"]
