from .spconv_blocks import SparseBasicBlock, post_act_block  # noqa: F401
