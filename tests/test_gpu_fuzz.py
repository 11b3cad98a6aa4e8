"""A short run of the randomised differential campaign (tests/fuzz_gpu_parity.py): random grids, modes and code-path knobs, the
library against the checker at tolerance 0.  The long campaigns of round 4 (profiles/r04_experiments.txt item 13) used other seeds."""
import pytest

import fuzz_gpu_parity as F

pytestmark = [pytest.mark.gpu, pytest.mark.timeout(900)]


@pytest.mark.parametrize("seed", [11, 12])
def test_random_cases_against_the_checker(seed):
    assert F.campaign(40, seed, verbose=False) == []


@pytest.mark.parametrize("seed", [21])
def test_random_whole_relaxations_against_the_checker(seed):
    assert F.campaign_complete(40, seed, verbose=False) == []
