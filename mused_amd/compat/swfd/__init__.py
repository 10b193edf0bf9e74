"""Drop-in package for mused's `from swfd import SeqBasedSWFD` (main.py:10)."""
from mused_amd.swfd import SeqBasedSWFD  # noqa: F401
