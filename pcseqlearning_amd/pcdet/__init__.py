"""Host-side mirror of the reference's plugin surface for the hot path: the same
registries, class names, constructor signatures and forward-dict contracts as `pcdet/models` and
`pcdet/ops` of xiangruhuang/PCSeqLearning, with every heavy operation dispatched to the HIP
kernels of libpcs_hip.so through pcseqlearning_amd.ops. A maintainer drops these modules over
the reference's; nothing here falls back to the CPU."""
