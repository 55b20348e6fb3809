"""tfnas_amd -- MI355X-native supernet-search hot path of TF-NAS (drop-in for models/model_search.py).

Host-only helpers (geometry tables, latency LUT, elasticity scaling) import without the HIP library; the
model classes need ``libtfnas_hip.so`` (built by ``__graft_entry__.build()``) and a GPU at call time.
"""
from . import geometry, latency, elasticity                         # noqa: F401
from .geometry import get_mc_num_dddict, make_mc_mask_dddict, make_lat_lookup_key_dddict   # noqa: F401
from .latency import load_lat_lookup, get_lookup_latency            # noqa: F401


def __getattr__(name):
    if name in ('Network', 'MixedStage', 'MixedOP', 'PRIMITIVES', 'OPS'):
        from . import model_search
        return getattr(model_search, name)
    if name in ('MBInvertedResBlock', 'ConvLayer', 'LinearLayer', 'Swish'):
        from . import layers
        return getattr(layers, name)
    raise AttributeError(name)
