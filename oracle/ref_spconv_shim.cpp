// Thin extern "C" wrapper around the REFERENCE's own CPU rulebook builders, compiled in place from
// /root/reference/pcdet/ops/spconv/include/spconv/geometry.h (getIndicePairsSubM :248-297,
// getIndicePairsConv :145-194, getIndicePairsDeConv :197-245). No reference source is copied: this file only includes the
// header where it lies (oracle/Makefile passes -I). The 6 lines of parameter handling restate
// getIndicePair<3> (spconv_ops.h:73-85: subM forces stride 1 / pad k/2), which itself needs torch.
// Output only under oracle/_ref/ (git-ignored). TEST INFRASTRUCTURE ONLY.
#include <spconv/geometry.h>

#include <vector>

extern "C" int ref_get_indice_pairs(const int* indices, int V, int batch, const int* out_shape3,
                                    const int* ksize3, const int* stride3, const int* pad3,
                                    const int* dil3, int subm, int* out_ids, int* pairs,
                                    int* num) {
  int ks[3], st[3], pd[3], dl[3], os[3];
  int kv = 1;
  long vol = 1;
  for (int d = 0; d < 3; ++d) {
    ks[d] = ksize3[d];
    dl[d] = dil3[d];
    st[d] = subm ? 1 : stride3[d];
    pd[d] = subm ? ksize3[d] / 2 : pad3[d];
    os[d] = out_shape3[d];
    kv *= ks[d];
    vol *= os[d];
  }
  std::vector<int> grid((size_t)batch * vol, -1);
  for (long i = 0; i < (long)kv * 2 * V; ++i) pairs[i] = -1;
  for (int k = 0; k < kv; ++k) num[k] = 0;
  tv::TensorView<const int> tin(indices, tv::Shape{V, 4});
  tv::TensorView<int> tgrid(grid.data(), tv::Shape{(int)grid.size()});
  tv::TensorView<int> tpairs(pairs, tv::Shape{kv, 2, V});
  tv::TensorView<int> tnum(num, tv::Shape{kv});
  if (subm) {
    return spconv::getIndicePairsSubM<int, int, 3>(tin, tgrid, tpairs, tnum, ks, st, pd, dl, os);
  }
  tv::TensorView<int> tout(out_ids, tv::Shape{V * kv, 4});
  return spconv::getIndicePairsConv<int, int, 3>(tin, tout, tgrid, tpairs, tnum, ks, st, pd, dl, os);
}

// the reference's getIndicePairsDeConv (transposed conv; out_shape3 = get_deconv_output_size)
extern "C" int ref_get_indice_pairs_deconv(const int* indices, int V, int batch, const int* out_shape3,
                                           const int* ksize3, const int* stride3, const int* pad3,
                                           const int* dil3, int* out_ids, int* pairs, int* num) {
  int ks[3], st[3], pd[3], dl[3], os[3];
  int kv = 1;
  long vol = 1;
  for (int d = 0; d < 3; ++d) {
    ks[d] = ksize3[d];
    dl[d] = dil3[d];
    st[d] = stride3[d];
    pd[d] = pad3[d];
    os[d] = out_shape3[d];
    kv *= ks[d];
    vol *= os[d];
  }
  std::vector<int> grid((size_t)batch * vol, -1);
  for (long i = 0; i < (long)kv * 2 * V; ++i) pairs[i] = -1;
  for (int k = 0; k < kv; ++k) num[k] = 0;
  tv::TensorView<const int> tin(indices, tv::Shape{V, 4});
  tv::TensorView<int> tgrid(grid.data(), tv::Shape{(int)grid.size()});
  tv::TensorView<int> tpairs(pairs, tv::Shape{kv, 2, V});
  tv::TensorView<int> tnum(num, tv::Shape{kv});
  tv::TensorView<int> tout(out_ids, tv::Shape{V * kv, 4});
  return spconv::getIndicePairsDeConv<int, int, 3>(tin, tout, tgrid, tpairs, tnum, ks, st, pd, dl, os);
}
