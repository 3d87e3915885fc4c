"""GPU: a fixed-seed slice of the randomized end-to-end sweep (tests/fuzz_nmf.py; scratch/fuzz_nmf2.py runs it at any length): nmf()
against the fp64 oracle over random shapes (ragged, any K), back-ends, arithmetic modes, weights, FISTA, the line search and every
operator; fp64 inputs of small problems to 1e-9.  The sweep found two defects in round 4; this keeps a slice of
it where the driver's `pytest -m gpu` sees it."""
import logging

import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def fuzz():
    import __graft_entry__ as g
    g.build()
    import fuzz_nmf
    logging.getLogger("proxmin").setLevel(logging.ERROR)
    yield fuzz_nmf
    logging.getLogger("proxmin").setLevel(logging.NOTSET)


@pytest.mark.parametrize("seed", [1, 2])
def test_random_problems_against_the_fp64_oracle(fuzz, seed):
    lines = []
    bad = fuzz.run(seed, 40, log=lines.append)
    assert bad == 0, "\n".join(l for l in lines if not l.startswith("ok"))


def test_random_small_fp64_problems_to_1e9(fuzz):
    lines = []
    bad = fuzz.run(5, 40, F64=True, log=lines.append)
    assert bad == 0, "\n".join(l for l in lines if not l.startswith("ok"))


def test_random_fp64_problems_at_size_to_1e9(fuzz):
    """[r6] fp64 inputs outside the small kernels (k_big_f64.hip: shapes up to 2200 x 3000, K up to 128, the three back-ends, every operator):
    1e-9 per entry against the fp64 oracle (+ 1e-10 of the factor's largest entry for the entries an operator holds near zero)"""
    lines = []
    bad = fuzz.run(7, 30, F64="big", log=lines.append)
    assert bad == 0, "\n".join(l for l in lines if not l.startswith("ok"))


def test_random_arguments_against_the_fp64_oracle(fuzz):
    """the ARGUMENTS of the back-ends (tests/fuzz_nmf.py: run_options): stopping tests that fire -- the run must end at the oracle's
    iteration --, b1 arrays, b2 / eps / p, capped proximal loops, warm-started moments, prox=None, bsdmm's e_abs and one-sided constraints"""
    lines = []
    bad = fuzz.run_options(1, 40, log=lines.append)
    assert bad == 0, "\n".join(l for l in lines if not l.startswith("ok"))
