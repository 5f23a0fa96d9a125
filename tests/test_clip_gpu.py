"""OpenAI CLIP on the native kernels on MI355X vs the reference's own model."""
import pytest

import clip_cases as CC
from fatezero_amd import _native

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("name", ["clip_tiny", "clip_vitb32"])
def test_clip_matches_reference_model_gpu(name):
    res = CC.run_clip_case(name, "cuda")
    print(name, res)
    CC.check(res)
    assert _native.loaded_path().endswith("libfatezero_hip.so")
