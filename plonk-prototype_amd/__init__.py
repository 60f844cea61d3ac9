"""MI355X (gfx950) backend for the PLONK prover hot path of Manta-Network/Plonk-Prototype:
BLS12-381 Fr NTT (``EvaluationDomain``) and G1 MSM (``msm_variable_base`` / ``CommitKey``).

Host-side mirror of the dusk-plonk / dusk-bls12_381 interfaces the reference depends on
(ref:Cargo.toml:19-20), sitting on the C ABI of ``include/plonk_mi355x.h``.  All compute
runs in hand-written HIP kernels; importing works without a GPU, but creating a
:class:`Context` raises unless a gfx950 device and the built library are present.
"""
from ._lib import (BackendMissing, NTT_COSET, NTT_INVERSE, NTT_TRANSPOSED, SCALAR_CANONICAL,  # noqa: F401
                   SCALAR_MONTGOMERY, load, LIB_PATH)
from .host import (CommitKey, Context, DeviceVector, Error, EvaluationDomain, Polynomial,  # noqa: F401
                   msm_variable_base,
                   g1_fold, g1_to_affine, domain_info, ntt_plan)
from . import field, prover, srs, synthetic, transcript  # noqa: F401,E402
from .prover import Circuit, Proof, ProverKey, preprocess, prove  # noqa: F401,E402
from .transcript import Transcript  # noqa: F401,E402
