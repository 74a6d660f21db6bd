"""The C++ oracle replays every committed fixture bit for bit (pins it against regressions),
and the numpy restatement agrees with the fixtures to tolerance (a second opinion)."""
import numpy as np
import pytest

from tests.util import check_against_golden, golden_cases, load_golden, replay_golden


@pytest.mark.parametrize("name", golden_cases())
def test_oracle_replays_fixture_bitwise(name, oracle_built):
    from oracle.ora import Oracle
    fx = load_golden(name)
    ora = Oracle()
    replay_golden(ora, fx)
    check_against_golden(ora, fx, 0.0, exact_fields=True, what=name + ": ")


@pytest.mark.parametrize("name", golden_cases())
def test_openmp_oracle_equals_serial_oracle(name, oracle_built):
    """The timed CPU baseline (OpenMP build) computes the same thing as the parity target."""
    from oracle.ora import Oracle
    fx = load_golden(name)
    ora = Oracle(omp=True)
    replay_golden(ora, fx)
    check_against_golden(ora, fx, 0.0, exact_fields=True, what=name + " (omp): ")


@pytest.mark.parametrize("name", [c for c in golden_cases() if "4096" not in c])
def test_numpy_restatement_agrees_with_fixture(name):
    from oracle.sph_ref import Ref
    fx = load_golden(name)
    ref = Ref(fx["params_struct"].copy(), fx["state_in"])
    ref.step_frame(int(fx["sub_steps"]), float(fx["time_step"]))
    assert np.array_equal(ref.cell_of_particle, fx["cell_of_particle"])
    want = fx["state_out"]
    for sl in (slice(0, 3), slice(3, 6), slice(6, 9), 9, 10):
        a, b = ref.P[:, sl].astype(np.float64), want[:, sl].astype(np.float64)
        assert np.abs(a - b).max() <= 2e-5 * max(np.abs(b).max(), 1e-30)
