"""Test infrastructure: CPU oracle of the FateZero hot path. Never imported by the product."""
