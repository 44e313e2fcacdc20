"""The reference's 17 inline known-answer tests, run on the CPU oracle (pins the oracle)."""
import pytest

from kats import KATS, run_kat, run_surface_kat
from oracle_lib import oracle


@pytest.mark.parametrize("kat", KATS, ids=[k[0] for k in KATS])
@pytest.mark.parametrize("portable", [0, 1], ids=["libm", "portable"])
def test_reference_kat(kat, portable):
    o = oracle()
    o.set_trig_mode(portable)
    try:
        run_kat(o, kat)
    finally:
        o.set_trig_mode(0)


def test_surface_interaction_simple():
    run_surface_kat(oracle())  # src/interaction/surface.rs:194-200


def test_it_works():
    assert True  # src/lib.rs:180-182
