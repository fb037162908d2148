"""pastix_amd -- MI355X-native numerical factorization (sopalin) for PaStiX layouts.

Only the hot path named by BASELINE.json is here: the C-ABI library (csrc/, HIP for gfx950) and
the thin host-side mirror of the reference interface around it.
"""
from .solver import (COMPLEXDOUBLE, FACT_LDLT, FACT_LLT, FACT_LU, REALDOUBLE, REALSINGLE, Plan, fact_flops,  # noqa: F401
                     sopalin_tabs)
