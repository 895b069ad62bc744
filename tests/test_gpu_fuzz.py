"""The two differential fuzzers (tests/fuzz_parity.py: bin encoder / decoder / estimator; tests/fuzz_residual.py: residual
binariser / parser) against the oracle, 20 seconds each under -m gpu.  The seed changes from run to run of the suite only when
CABAC_FUZZ_SEED says so: a failure is reproducible from the seed in its message."""
import os
import types

import pytest

pytestmark = pytest.mark.gpu

SECONDS = float(os.environ.get("CABAC_FUZZ_SECONDS", "20"))
SEED = int(os.environ.get("CABAC_FUZZ_SEED", "3"))


def _run(mod):
    try:
        line = mod.run(types.SimpleNamespace(seconds=SECONDS, seed=SEED))
    except SystemExit as e:
        pytest.fail(str(e.code))
    print(line)
    assert line.startswith("fuzz ok") and " 0 rounds" not in line


def test_fuzz_bin_codec_against_oracle():
    import fuzz_parity
    _run(fuzz_parity)


def test_fuzz_residual_binariser_and_parser_against_oracle():
    import fuzz_residual
    _run(fuzz_residual)
