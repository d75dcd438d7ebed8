"""Drop-in for pcdet/ops/torch_hash/torch_hash_cuda (torch_hash_api.cpp:9-15): the same function
names and argument order, on the reference's own table layout."""
from ....ops import hash_insert_gpu, radius_graph_gpu  # noqa: F401

torch_hash_cuda = __import__(__name__, fromlist=["hash_insert_gpu"])
