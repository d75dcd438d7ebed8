// Thin extern "C" wrapper around the REFERENCE's own CPU max-pool functors, compiled in place from
// /root/reference/pcdet/ops/spconv/src/maxpool.cc (SparseMaxPoolForwardFunctor :22-40,
// SparseMaxPoolBackwardFunctor :43-64) against the torch headers installed in the image. No
// reference source is copied: the file is included where it lies. Output only under oracle/_ref/
// (git-ignored). TEST INFRASTRUCTURE ONLY.
#include "/root/reference/pcdet/ops/spconv/src/maxpool.cc"

extern "C" void ref_maxpool_fwd(float* out, const float* in, const int* pairs2, int n_out, int n_in,
                                int C, int stride, int size) {
  tv::TensorView<float> tout(out, tv::Shape{n_out, C});
  tv::TensorView<const float> tin(in, tv::Shape{n_in, C});
  tv::TensorView<const int> tidx(pairs2, tv::Shape{2, stride});
  spconv::functor::SparseMaxPoolForwardFunctor<tv::CPU, float, int> f;
  f(tv::CPU(), tout, tin, tidx, size);
}

extern "C" void ref_maxpool_bwd(const float* out, const float* in, const float* dout, float* din,
                                const int* pairs2, int n_out, int n_in, int C, int stride, int size) {
  tv::TensorView<const float> tout(out, tv::Shape{n_out, C});
  tv::TensorView<const float> tin(in, tv::Shape{n_in, C});
  tv::TensorView<const float> tdo(dout, tv::Shape{n_out, C});
  tv::TensorView<float> tdi(din, tv::Shape{n_in, C});
  tv::TensorView<const int> tidx(pairs2, tv::Shape{2, stride});
  spconv::functor::SparseMaxPoolBackwardFunctor<tv::CPU, float, int> f;
  f(tv::CPU(), tout, tin, tdo, tdi, tidx, size);
}
