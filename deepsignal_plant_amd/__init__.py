"""deepsignal_plant_amd -- MI355X-native `deepsignal_plant call_mods` hot path.

Host-side mirror of the reference's interface for this path (same names, argument meaning and error
behaviour as deepsignal_plant/models.py and deepsignal_plant/call_modifications.py) over a C-ABI HIP
library (libdsp_amd.so, include/dsp_amd.h).  There is no CPU fallback: every compute entry point raises
if the HIP library is missing.
"""
from ._version import VERSION as __version__  # noqa: F401
