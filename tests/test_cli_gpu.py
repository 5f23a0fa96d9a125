"""The YAML / CLI job of tests/test_cli_emu.py on MI355X: images in -> VAE -> inversion with capture -> edits -> VAE -> files."""
import pytest

from test_cli_emu import run_cli_job

pytestmark = pytest.mark.gpu


def test_cli_runs_a_yaml_job_end_to_end_gpu(tmp_path):
    out = run_cli_job(tmp_path, "cuda")
    assert len(out["samples"]) == 2
