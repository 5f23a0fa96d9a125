"""`python test_fatezero.py --config <yaml>` -- the reference's entry point (test_fatezero.py:254-276) on the MI355X build."""
from fatezero_amd.cli import run, test  # noqa: F401

if __name__ == "__main__":
    run()
