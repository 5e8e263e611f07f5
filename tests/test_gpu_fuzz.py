"""Randomised differential test of the HIP engine against the oracle (see fuzz_cases.py)."""
import os
import sys

import pytest

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import fuzz_cases  # noqa: E402

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def env():
    import plastid_amd as pa
    from oracle import oracle
    import test_gpu_parity as T
    return pa, oracle, T


@pytest.mark.parametrize("seed", range(60))
def test_random_case_matches_oracle(env, seed):
    pa, oracle, T = env
    fuzz_cases.run_case(pa, oracle, fuzz_cases.random_case(1000 + seed, pa), T.spec_for, T.engine_for)
