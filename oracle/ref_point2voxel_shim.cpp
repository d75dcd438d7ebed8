// pybind11 module exposing the REFERENCE's hard voxelizer, compiled in place from
// /root/reference/pcdet/ops/spconv/include/spconv/point2voxel.h:33-92 (the header is present in
// the reference tree but not bound by its src/all.cc). No reference source is copied.
// Output only under oracle/_ref/ (git-ignored). TEST INFRASTRUCTURE ONLY.
#include <spconv/point2voxel.h>

PYBIND11_MODULE(ref_point2voxel, m) {
  m.def("points_to_voxel_3d_np", &spconv::points_to_voxel_3d_np<float, 3>);
}
