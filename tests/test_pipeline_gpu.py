"""MI355X end-to-end parity: native pipeline (HIP kernels through the C ABI) vs vectors from the unmodified reference."""
import pytest

from fatezero_amd import _native

import pipeline_cases as PC

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module", autouse=True)
def hip_backend():
    _native.reset_backend()
    yield


@pytest.mark.parametrize("name", ["pipe_small_refine_reweight", "pipe_small_replace", "pipe_replace_blend",
                                  "pipe_refine_reweight_latentblend", "pipe_refine_noblend", "pipe_f4_prev_first",
                                  "pipe_f3_mid_next"])
def test_pipeline(name):
    res = PC.run_pipeline_case(name, "cuda", mixed_oracle=True)
    print(name, res)
    PC.check(res)
    assert _native.loaded_path().endswith("libfatezero_hip.so")
