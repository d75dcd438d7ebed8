"""Drop-in for pcdet/ops/torch_hash/torch_hash_cuda (torch_hash_api.cpp:9-15): the four functions
the reference imports (graph_utils.py:13-18, torch_hash_modules.py:3-8), same names and argument
order, on the reference's own table layout."""
from ....ops import correspondence, hash_insert_gpu, points_in_radius_gpu, radius_graph_gpu  # noqa: F401

torch_hash_cuda = __import__(__name__, fromlist=["hash_insert_gpu"])
from .torch_hash_modules import ChamferDistance, RadiusGraph  # noqa: F401,E402
from .torch_hash_utils import HashTable  # noqa: F401,E402
