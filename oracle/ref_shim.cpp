// oracle/ref_shim.cpp -- TEST INFRASTRUCTURE.
// pybind front for the reference's own CPU RoIAlign, compiled unmodified from
// /root/reference/mmcv-1.4.7/mmcv/ops/csrc/pytorch/{roi_align.cpp,cpu/roi_align.cpp}
// by oracle/build_ref.py.  Only the two entry points that mmcv's pybind.cpp:611-620
// exports for this op are exposed, with the same keyword names, so tests can call
// it exactly like `mmcv._ext`.
#include <torch/extension.h>

using at::Tensor;

void roi_align_forward(Tensor input, Tensor rois, Tensor output, Tensor argmax_y,
                       Tensor argmax_x, int aligned_height, int aligned_width,
                       float spatial_scale, int sampling_ratio, int pool_mode,
                       bool aligned);

void roi_align_backward(Tensor grad_output, Tensor rois, Tensor argmax_y,
                        Tensor argmax_x, Tensor grad_input, int aligned_height,
                        int aligned_width, float spatial_scale, int sampling_ratio,
                        int pool_mode, bool aligned);

PYBIND11_MODULE(TORCH_EXTENSION_NAME, m) {
  m.def("roi_align_forward", &roi_align_forward, py::arg("input"), py::arg("rois"),
        py::arg("output"), py::arg("argmax_y"), py::arg("argmax_x"),
        py::arg("aligned_height"), py::arg("aligned_width"), py::arg("spatial_scale"),
        py::arg("sampling_ratio"), py::arg("pool_mode"), py::arg("aligned"));
  m.def("roi_align_backward", &roi_align_backward, py::arg("grad_output"),
        py::arg("rois"), py::arg("argmax_y"), py::arg("argmax_x"), py::arg("grad_input"),
        py::arg("aligned_height"), py::arg("aligned_width"), py::arg("spatial_scale"),
        py::arg("sampling_ratio"), py::arg("pool_mode"), py::arg("aligned"));
}
