from dataclasses import dataclass
from typing import Any

from ...utils import BaseOutput


@dataclass
class StableDiffusionPipelineOutput(BaseOutput):
    images: Any
    nsfw_content_detected: Any = None
