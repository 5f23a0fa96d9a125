"""Restatement of the diffusers==0.11.1 surface imported by /root/reference (see ../README.md)."""
__version__ = "0.11.1-restated"
