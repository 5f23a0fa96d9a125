"""Native CLIP text encoder on MI355X against transformers' CLIPTextModel."""
import pytest

from test_clip_text_emu import clip_model_case

pytestmark = pytest.mark.gpu


def test_text_encoder_matches_transformers_gpu():
    print(clip_model_case("cuda"))
