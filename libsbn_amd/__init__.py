"""libsbn_amd -- MI355X-native phylogenetic likelihood + gradient engine behind
libsbn's Engine / FatBeagle API.  The compute path is libmi_phylo.so (hand-written
HIP for gfx950, C ABI in include/mi_phylo.h); this package is a thin ctypes mirror
of the reference's Engine interface.  No CPU fallback exists."""
from .engine import (Engine, PhyloGradient, PhyloModelSpecification,  # noqa: F401
                     site_pattern_compress_device)
from .instance import rooted_instance, unrooted_instance  # noqa: F401

__all__ = ["Engine", "PhyloGradient", "PhyloModelSpecification", "unrooted_instance",
           "rooted_instance", "site_pattern_compress_device"]
