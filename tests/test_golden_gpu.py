"""libsph_hip replays every committed fixture through the C ABI: hash, sorted order, cell
populations, neighbour lists and counters bit-exact; fields within 1e-5 of field scale."""
import pytest

from sph_amd import abi
from tests.util import check_against_golden, golden_cases, load_golden, replay_golden

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("name", golden_cases())
def test_engine_replays_fixture(name):
    fx = load_golden(name)
    eng = abi.Engine(0)
    replay_golden(eng, fx)
    check_against_golden(eng, fx, 1e-5, what=name + ": ")
    eng.destroy()
