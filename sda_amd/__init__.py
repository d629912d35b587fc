"""sda_amd - MI355X (gfx950) secure-aggregation compute core for snipsco/sda's secret-sharing path.

The product is the C-ABI shared library ``sda_amd/lib/libsda_hip.so`` (include/sda_hip.h): hand-written
HIP kernels behind entry points that map 1:1 onto the reference's ``client::crypto`` sharing/masking
traits.  This package only loads it (``capi``) and mirrors the reference's interface on top of it
(``crypto``); ``distributed`` shards participants across GPUs.  There is no CPU fallback."""
from . import capi  # noqa: F401

__version__ = "0.6.0"          # = sda_version() of the library (tests/test_capi_cpu.py checks the two agree)
